// ku_report.hip -- clade roll-up of the report on the device.
//
// The reference's TaxReport constructor (src/taxdb.hpp:928-982) merges every counted taxon's HyperLogLog sketch into the
// sketch of each of its ancestors, one 4 KiB register merge per (taxon, ancestor) pair on the host.  Here the per-taxon
// registers never leave HBM: the host only lists, per counted clade, the slots of its members (a CSR over the slots'
// root paths) and the device
//   * ku_rollup_dense_kernel: one workgroup per clade with a dense member -- byte-wise maximum over the members'
//     registers (16 registers per lane, uint4 loads, SWAR max), reduced on the spot to the 54-bin register histogram the
//     Ertl estimator works from (src/hyperloglogplus.cpp:722-753);
//   * ku_rollup_sparse_kernel: clades whose members all kept the sparse representation merge by set union
//     (hyperloglogplus.cpp:601-604): every (slot, encoded hash) pair of the run-wide set is inserted under each of the
//     slot's all-sparse ancestor clades into a device hash set; first insertions add to that clade's histogram of
//     encoded ranks (sparseRegisterHistogram, :356-366).
// 80 bins * 4 B per clade go back to the host.
#include <hip/hip_runtime.h>

#include "ku_device.h"

namespace {

// byte-wise unsigned maximum of packed registers (values <= 64 < 128: setting bit 7 of every byte of `a` keeps the
// subtraction from borrowing across bytes, bit 7 of the difference then says a >= b)
__device__ __forceinline__ uint32_t bmax4(uint32_t a, uint32_t b) {
  const uint32_t ge = (((a | 0x80808080u) - b) & 0x80808080u) >> 7;
  const uint32_t m = ge * 0xFFu;
  return (a & m) | (b & ~m);
}

__global__ __launch_bounds__(256) void ku_rollup_dense_kernel(const uint8_t *__restrict__ registers,
                                                               const uint32_t *__restrict__ member_off,
                                                               const uint32_t *__restrict__ member_slot,
                                                               const uint8_t *__restrict__ clade_dense,
                                                               uint32_t *__restrict__ hist) {
  __shared__ uint32_t bins[KU_ROLLUP_BINS];
  const uint32_t c = blockIdx.x, t = threadIdx.x;
  if (!clade_dense[c]) return;
  if (t < KU_ROLLUP_BINS) bins[t] = 0;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint32_t lo = member_off[c], hi = member_off[c + 1];
  for (uint32_t j = lo; j < hi; ++j) {
    const uint4 v = reinterpret_cast<const uint4 *>(registers + (size_t)member_slot[j] * KU_HLL_M)[t];
    acc.x = bmax4(acc.x, v.x); acc.y = bmax4(acc.y, v.y); acc.z = bmax4(acc.z, v.z); acc.w = bmax4(acc.w, v.w);
  }
  __syncthreads();
  const uint32_t w[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t r = (w[i] >> (8 * b)) & 0xFFu;
      atomicAdd(&bins[r < 64 - KU_HLL_P + 1 ? r : 64 - KU_HLL_P + 1], 1u);
    }
  __syncthreads();
  if (t < KU_ROLLUP_BINS) hist[(size_t)c * KU_ROLLUP_BINS + t] = bins[t];
}

// rank of an encoded hash relative to p = 12 (getEncodedRank, hyperloglogplus.cpp:152-161)
__device__ __forceinline__ uint32_t encoded_rank(uint32_t e) {
  if (e & 1u) return (25 - KU_HLL_P) + ((e >> 1) & 0x3Fu);
  const uint32_t bits = e << KU_HLL_P;
  return (bits == 0 ? 32 - KU_HLL_P : (uint32_t)__clz(bits)) + 1;
}

// The clades near the root take an entry from (almost) every pair: their histogram bins are counted in the block's LDS
// and flushed once per block -- one global address per (clade, rank) would serialise the whole grid.  `clade_hot[c]` is
// the clade's row in that LDS table (0xFFFF: none), `hot_clades[h]` the reverse map.
//
// Input is the run-wide set G of the sparse-mode emulation where it lies (ku_sparse.hip: cells of (slot + 1) << 32 |
// encoding, 0 = empty; no compacted copy is made).  An entry walks up its slot's chain of all-sparse clades, leaf first:
//   * (clade, encoding) goes into the clade's set; an entry that is already there stops the walk -- whoever put it there
//     carries it further up, and the chains of two slots are the same above their first common clade.  (Rounds 2-4 gave a
//     clade with a single member no set: its entries came from G alone and were distinct already.  With two sources -- G and
//     the probe table's SEEN marks, below -- an encoding may arrive twice, and two k-mers may share one; every clade dedups.)
// So the work is one insert per DISTINCT (clade, encoding) plus one failed probe per duplicate, not entries x depth.
// The union sets are one open-addressing table PER CLADE of 4-byte cells holding the encoding alone (0 never is one: index
// 0 carries the rank flag), laid out back to back: set_off[c] / set_cells[c]; half the memory of (clade, encoding) keys --
// and the time goes into getting that memory (tens of GB for a run of many sparse taxa), not into the kernel.
//
// BIG clades (bm_of[c] != KU_BM_NONE) keep a BITMAP over the 2^25 indices instead: an encoding without the rank flag is its
// index alone (encodeHashIn32Bit, hyperloglogplus.cpp:120-150), so the union of such entries is a bitwise OR.  An entry sets
// its bit in the LOWEST big clade of its chain and stops there: the clades above take it with their children's whole
// bitmaps (ku_bitmap_or_children, level by level, no atomics), and the rank histogram is read off the finished bitmap
// (ku_bitmap_hist) -- one atomic per entry instead of one table insert per entry and level (3 G inserts for 0.6 G entries
// under a five-level taxonomy were 237 of the report's 290 ms of kernels).  The few entries that carry the flag (index
// bits p..p' all zero: 1 in 8192) walk on through small per-clade tables as before.
__device__ __forceinline__ void rollup_to_bitmap(const KuRollupPlan &p, uint32_t b, uint32_t enc);
// one entry (slot, encoding) of a sketch that stayed sparse walks up its slot's chain of all-sparse clades, leaf first
__device__ __forceinline__ void rollup_entry(const KuRollupPlan &p, uint32_t *hot, uint32_t slot, uint32_t enc) {
  uint32_t r = encoded_rank(enc);
  if (r > KU_ROLLUP_BINS - 1) r = KU_ROLLUP_BINS - 1;
  uint32_t g = enc * 0x9E3779B1u;  // (the index sits in the high bits of an encoding: spread it)
  g ^= g >> 15;
  g *= 0x85EBCA77u;
  g ^= g >> 13;
  for (uint32_t j = p.slot_off[slot]; j < p.slot_off[slot + 1]; ++j) {
    const uint32_t c = p.slot_clade[j];
    const uint32_t b = p.bm_of[c];
    if (b != KU_BM_NONE && !(enc & 1u)) {  // index = enc >> 7: word enc >> 12, bit (enc >> 7) & 31
      rollup_to_bitmap(p, b, enc);
      break;
    }
    bool fresh = true;
    const uint32_t cells = p.set_cells[c];  // 0: a clade nothing is offered to
    if (cells) {
      uint32_t *tab = p.set + p.set_off[c];
      uint32_t h = __umulhi(g, cells);
      bool done = false;
      for (uint32_t probe = 0; probe < 8192 && !done; ++probe) {
        const uint32_t old = atomicCAS(&tab[h], 0u, enc);
        if (old == 0u) done = true;
        else if (old == enc) { done = true; fresh = false; }
        h = h + 1 == cells ? 0 : h + 1;
      }
      if (!done) { atomicOr(p.err, 1u); fresh = false; }
    }
    if (!fresh) break;
    const uint32_t hi = p.clade_hot[c];
    if (hi != 0xFFFFu) atomicAdd(&hot[hi * KU_ROLLUP_BINS + r], 1u);
    else atomicAdd(&p.hist[(size_t)c * KU_ROLLUP_BINS + r], 1u);
  }
}

// source 1: the run-wide set G where it lies (cells of (slot + 1) << 32 | encoding, 0 = empty)
__global__ __launch_bounds__(256) void ku_rollup_sparse_kernel(const unsigned long long *__restrict__ g_key, uint64_t g_cells, KuRollupPlan p) {
  __shared__ uint32_t hot[KU_ROLLUP_HOT * KU_ROLLUP_BINS];
  for (uint32_t i = threadIdx.x; i < KU_ROLLUP_HOT * KU_ROLLUP_BINS; i += blockDim.x) hot[i] = 0;
  __syncthreads();
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < g_cells; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long e = g_key[i];
    if (!e) continue;
    const uint32_t slot = (uint32_t)(e >> 32) - 1;
    if (p.dense[slot]) continue;
    rollup_entry(p, hot, slot, (uint32_t)e);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < p.n_hot * KU_ROLLUP_BINS; i += blockDim.x)
    if (hot[i]) atomicAdd(&p.hist[(size_t)p.hot_clades[i / KU_ROLLUP_BINS] * KU_ROLLUP_BINS + i % KU_ROLLUP_BINS], hot[i]);
}

// an entry bound for bitmap b (its encoding carries no rank flag): index = enc >> 7 -> word enc >> 12, bit (enc >> 7) & 31.
// (Round 6 tried buckets per (bitmap, 2^18-index segment) filled through an L2-resident cursor, their bits then set in LDS and
// written once: the cursor's round trip and the scattered 4-byte stores cost what the read-modify-writes of random HBM lines
// cost -- 23 against 21 ms for 0.55 G entries -- plus 4 B of scratch per entry; not kept.)
__device__ __forceinline__ void rollup_to_bitmap(const KuRollupPlan &p, uint32_t b, uint32_t enc) {
  atomicOr(&p.bm[(size_t)b * KU_BM_WORDS + (enc >> 12)], 1u << ((enc >> 7) & 31u));
}

// source 2 (round 5): the probe table's SEEN marks (ku_device.h) -- every marked entry is a k-mer some read held, booked under
// the entry's slot by the fused kernel's fast path; its encoding is that of the k-mer's hash (hyperloglogplus.cpp:181-204).
// A thread per LINE and two lines per thread and turn: the eight SEEN bytes of each as one 8-byte load (a wave has 128 line
// requests in flight; a thread per entry -- rounds 5/6 -- had eight lanes wait for one line and every wave turn for the whole
// chain entry -> slot -> chain of clades -> bit: 44.6 ms for the 49 GB table of the bench database after a 10 M-read run; now
// 31.3 ms: 8.1 the scan, 1.9 the chain, 21 the 0.55 G read-modify-writes of random bitmap lines), then the marked entries from
// the same lines.  `slot_fast` shortens the chain for what nearly every entry is: the slot's first all-sparse clade
// keeps a bitmap (its number), the slot is dense or offers nothing (KU_FAST_SKIP), or the general walk (KU_FAST_WALK).
__global__ __launch_bounds__(256) void ku_rollup_table_kernel(const uint32_t *__restrict__ table, uint64_t n_lines, KuRollupPlan p) {
  __shared__ uint32_t hot[KU_ROLLUP_HOT * KU_ROLLUP_BINS];
  for (uint32_t i = threadIdx.x; i < KU_ROLLUP_HOT * KU_ROLLUP_BINS; i += blockDim.x) hot[i] = 0;
  __syncthreads();
  constexpr int LINES = 2;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i0 < n_lines; i0 += LINES * stride) {
    unsigned long long seen[LINES];
#pragma unroll
    for (int u = 0; u < LINES; ++u) {
      const uint64_t i = i0 + (uint64_t)u * stride;
      seen[u] = i < n_lines ? *reinterpret_cast<const unsigned long long *>(table + i * KU_LINE_DWORDS + KU_LINE_SEEN0) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < LINES; ++u) {
      const uint32_t *lp = table + (i0 + (uint64_t)u * stride) * KU_LINE_DWORDS;
      unsigned long long sm = seen[u];
      while (sm) {
        const uint32_t e = (uint32_t)__builtin_ctzll(sm) >> 3;
        sm &= ~(0xFFull << (8 * e));
        const uint32_t k_lo = lp[KU_LINE_ENTRY0 + 3 * e], k_hi = lp[KU_LINE_ENTRY0 + 3 * e + 1], slot = lp[KU_LINE_ENTRY0 + 3 * e + 2];
        const uint32_t f = p.slot_fast[slot];
        if (f == KU_FAST_SKIP) continue;
        const uint32_t enc = ks_encode(ku_fmix64(((uint64_t)k_hi << 32) | k_lo));
        if (f != KU_FAST_WALK && !(enc & 1u)) rollup_to_bitmap(p, f, enc);
        else rollup_entry(p, hot, slot, enc);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < p.n_hot * KU_ROLLUP_BINS; i += blockDim.x)
    if (hot[i]) atomicAdd(&p.hist[(size_t)p.hot_clades[i / KU_ROLLUP_BINS] * KU_ROLLUP_BINS + i % KU_ROLLUP_BINS], hot[i]);
}

// bitmap of a parent clade |= the bitmaps of its children (all finished: the launches go level by level, deepest parents
// first).  blockIdx.y = position in the list of this level's parents.
__global__ __launch_bounds__(256) void ku_bitmap_or_children_kernel(uint32_t *bm, const uint32_t *__restrict__ parents,
                                                                     const uint32_t *__restrict__ child_off,
                                                                     const uint32_t *__restrict__ child) {
  const uint32_t p = parents[blockIdx.y];
  uint4 *dst = reinterpret_cast<uint4 *>(bm + (size_t)p * KU_BM_WORDS);
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < KU_BM_WORDS / 4; w += gridDim.x * blockDim.x) {
    uint4 acc = dst[w];
    for (uint32_t j = child_off[p]; j < child_off[p + 1]; ++j) {
      const uint4 v = reinterpret_cast<const uint4 *>(bm + (size_t)child[j] * KU_BM_WORDS)[w];
      acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w;
    }
    dst[w] = acc;
  }
}

// rank histogram of a finished bitmap, added to the clade's row of hist.  The rank of an index is clz of its low 13 bits + 1
// (encoded_rank without the flag): word w holds the indices 32 w .. 32 w + 31, so unless the word's low byte is zero all
// of its bits share one rank; else the rank comes from the bit position (bit 0 there is never set: that index carries the
// flag and is not in the bitmap).
__global__ __launch_bounds__(256) void ku_bitmap_hist_kernel(const uint32_t *__restrict__ bm, const uint32_t *__restrict__ bm_clade,
                                                              uint32_t *hist) {
  __shared__ uint32_t bins[16];
  if (threadIdx.x < 16) bins[threadIdx.x] = 0;
  __syncthreads();
  const uint4 *src = reinterpret_cast<const uint4 *>(bm + (size_t)blockIdx.y * KU_BM_WORDS);
  uint32_t cnt[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) cnt[i] = 0;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < KU_BM_WORDS / 4; q += gridDim.x * blockDim.x) {
    const uint4 v4 = src[q];  // words 4 q .. 4 q + 3
    const uint32_t vs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t v = vs[u];
      if (!v) continue;
      const uint32_t lo = (4u * q + u) & 0xFFu;
      if (lo) {
        const uint32_t r = (uint32_t)__clz(lo) - 24u + 1u;  // 1 .. 8
#pragma unroll
        for (int i = 1; i <= 8; ++i) cnt[i] += r == (uint32_t)i ? (uint32_t)__popc(v) : 0u;
      } else {
        cnt[9] += (uint32_t)__popc(v & 0xFFFF0000u);
        cnt[10] += (uint32_t)__popc(v & 0x0000FF00u);
        cnt[11] += (uint32_t)__popc(v & 0x000000F0u);
        cnt[12] += (uint32_t)__popc(v & 0x0000000Cu);
        cnt[13] += (uint32_t)__popc(v & 0x00000002u);
      }
    }
  }
#pragma unroll
  for (int i = 1; i < 14; ++i)
    if (cnt[i]) atomicAdd(&bins[i], cnt[i]);
  __syncthreads();
  if (threadIdx.x >= 1 && threadIdx.x < 14 && bins[threadIdx.x])
    atomicAdd(&hist[(size_t)bm_clade[blockIdx.y] * KU_ROLLUP_BINS + threadIdx.x], bins[threadIdx.x]);
}

// classify -I: the reads of the last batch are counted under new calls (ku_ctx_replace_calls).  node_taxid is ascending:
// a binary search maps a taxid to its node; a taxid outside the node universe cannot be counted.
__device__ __forceinline__ uint32_t node_of_taxid(const uint32_t *__restrict__ node_taxid, uint32_t n_nodes, uint32_t taxid) {
  uint32_t lo = 0, hi = n_nodes;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (node_taxid[mid] < taxid) lo = mid + 1; else hi = mid;
  }
  return lo < n_nodes && node_taxid[lo] == taxid ? lo : 0xFFFFFFFFu;
}
__global__ void ku_replace_calls_kernel(const uint32_t *__restrict__ old_calls, const uint32_t *__restrict__ new_calls, uint64_t n,
                                        const uint32_t *__restrict__ node_taxid, uint32_t n_nodes, unsigned long long *n_reads,
                                        unsigned long long *n_dropped) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t o = old_calls[i], c = new_calls[i];
    if (o == c) continue;
    const uint32_t no = node_of_taxid(node_taxid, n_nodes, o), nc = node_of_taxid(node_taxid, n_nodes, c);
    if (no != 0xFFFFFFFFu) atomicAdd(&n_reads[no], ~0ull);  // - 1
    if (nc != 0xFFFFFFFFu) atomicAdd(&n_reads[nc], 1ull);
    else atomicAdd(n_dropped, 1ull);
  }
}

}  // namespace

int ku_launch_replace_calls(const uint32_t *d_old, const uint32_t *d_new, uint64_t n, const uint32_t *d_node_taxid, uint32_t n_nodes,
                            unsigned long long *d_n_reads, unsigned long long *d_dropped, hipStream_t stream) {
  if (!n) return KU_OK;
  const uint64_t want = (n + 255) / 256;
  ku_replace_calls_kernel<<<(unsigned)(want < 4096 ? want : 4096), 256, 0, stream>>>(d_old, d_new, n, d_node_taxid, n_nodes, d_n_reads, d_dropped);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_rollup_dense(const uint8_t *d_registers, const uint32_t *d_member_off, const uint32_t *d_member_slot,
                           const uint8_t *d_clade_dense, uint32_t n_clades, uint32_t *d_hist, hipStream_t stream) {
  if (!n_clades) return KU_OK;
  ku_rollup_dense_kernel<<<n_clades, 256, 0, stream>>>(d_registers, d_member_off, d_member_slot, d_clade_dense, d_hist);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_rollup_sparse(const unsigned long long *d_g_key, uint64_t g_cells, const KuRollupPlan &plan, int n_cu, hipStream_t stream) {
  if (!g_cells) return KU_OK;
  const uint64_t want = (g_cells + 255) / 256;
  const unsigned blocks = (unsigned)(want < (uint64_t)n_cu * 8 ? want : (uint64_t)n_cu * 8);
  ku_rollup_sparse_kernel<<<blocks, 256, 0, stream>>>(d_g_key, g_cells, plan);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_rollup_table(const void *d_table, uint64_t n_lines, const KuRollupPlan &plan, int n_cu, hipStream_t stream) {
  if (!n_lines) return KU_OK;
  if (!plan.slot_fast) return KU_EINVAL;
  const uint64_t want = (n_lines + 511) / 512;
  const unsigned blocks = (unsigned)(want < (uint64_t)n_cu * 8 ? want : (uint64_t)n_cu * 8);
  ku_rollup_table_kernel<<<blocks, 256, 0, stream>>>((const uint32_t *)d_table, n_lines, plan);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_bitmap_or_children(uint32_t *d_bm, const uint32_t *d_parents, uint32_t n_parents, const uint32_t *d_child_off,
                                 const uint32_t *d_child, hipStream_t stream) {
  if (!n_parents) return KU_OK;
  ku_bitmap_or_children_kernel<<<dim3(64, n_parents), 256, 0, stream>>>(d_bm, d_parents, d_child_off, d_child);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_bitmap_hist(const uint32_t *d_bm, const uint32_t *d_bm_clade, uint32_t n_bm, uint32_t *d_hist, hipStream_t stream) {
  if (!n_bm) return KU_OK;
  ku_bitmap_hist_kernel<<<dim3(64, n_bm), 256, 0, stream>>>(d_bm, d_bm_clade, d_hist);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
