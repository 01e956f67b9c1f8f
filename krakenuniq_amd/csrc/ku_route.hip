// ku_route.hip -- the owner-routed multi-GPU path (ku_mgpu.cpp) behind the scan.
//
// The database is sharded by minimizer-bin range over the GPUs (KrakenDB::prepare_chunking laid out in space,
// krakendb.cpp:430-526).  A rank scans only its own slice of the reads (ku_lookup_kernel<3,...>, ku_kernels.hip) and sends
// every run of consecutive k-mers that share their anchor occurrence to the rank that owns the run's bin as one 16-byte
// record (ku_internal.h).  This file holds
//   ku_route_prefix_*     exclusive prefix sum of the records' k-mer counts: both sides number the returned slots with it;
//   ku_route_owner_kernel what the owner does with a record: expand it into its k-mers, canonical form, locus key from the
//                         anchor the sender found, the bucket probe of kmer_query (krakendb.cpp:250-321) and
//                         ReadCounts::add_kmer (classify.cpp:939: HLL register + n_kmers, owner-computes, misses under
//                         taxon 0), one 4-byte slot per k-mer back, in record order;
//   ku_route_gather_*     the sender's side of the return: the ticket the scan left at a k-mer's position -> its slot
//                         (alone in front of the resolve kernel, or fused with resolve_tree: one wave per read).
// The k-mers of a read arrive together and in order, one lane per k-mer, so neighbouring lanes still share bucket lines.
#include <algorithm>
#include <cstdlib>

#include "ku_device.h"

#define KR_WAVES 4
#define KR_ITEMS 2
#define KR_KCT_LOG2 8
#define KR_OWN_BYTES 2048  // >= 64 records x 31 k-mers

__device__ __forceinline__ void kr_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------- prefix sum of the records' k-mer counts
#define KR_PFX_THREADS 256
#define KR_PFX_PER 4
#define KR_PFX_TILE (KR_PFX_THREADS * KR_PFX_PER)

__device__ __forceinline__ uint32_t kr_rec_n(const uint4 *__restrict__ rec, uint64_t i, uint64_t n_rec, uint64_t cap,
                                             const unsigned long long *__restrict__ cursor) {
  if (i >= n_rec) return 0;
  if (cap) {
    const uint64_t q = i / cap, within = i - q * cap;
    const unsigned long long used = cursor[q * KU_ROUTE_CURSOR_STRIDE];
    if (within >= (used < cap ? used : cap)) return 0;
  }
  return reinterpret_cast<const uint32_t *>(rec + i)[3] & 31u;
}
// cap != 0: the block's whole tile lies in the unused part of one queue and holds no queue's first record (the totals are
// read there): nothing to add, nothing anybody reads
__device__ __forceinline__ bool kr_tile_unused(uint64_t first, uint64_t n_rec, uint64_t cap, const unsigned long long *__restrict__ cursor) {
  if (!cap || first >= n_rec) return false;
  const uint64_t q = first / cap, within = first - q * cap;
  if (within == 0 || within + KR_PFX_TILE > cap) return false;
  return within >= cursor[q * KU_ROUTE_CURSOR_STRIDE];
}
// inclusive prefix sum over the block (256 threads); returns this thread's inclusive value, *total = the block's sum
__device__ __forceinline__ uint32_t kr_block_scan(uint32_t v, uint32_t *s_w, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, o);
    if (lane >= (uint32_t)o) v += t;
  }
  if (lane == 63u) s_w[wv] = v;
  __syncthreads();
  uint32_t add = 0, tot = 0;
#pragma unroll
  for (uint32_t i = 0; i < KR_PFX_THREADS / 64; ++i) {
    const uint32_t x = s_w[i];
    if (i < wv) add += x;
    tot += x;
  }
  *total = tot;
  __syncthreads();
  return v + add;
}
__global__ __launch_bounds__(KR_PFX_THREADS) void ku_route_prefix_sums_kernel(const uint4 *__restrict__ rec, uint64_t n_rec, uint64_t cap,
                                                                               const unsigned long long *__restrict__ cursor,
                                                                               uint32_t *__restrict__ bsum) {
  __shared__ uint32_t s_w[KR_PFX_THREADS / 64];
  if (kr_tile_unused((uint64_t)blockIdx.x * KR_PFX_TILE, n_rec, cap, cursor)) {  // block-uniform
    if (threadIdx.x == 0) bsum[blockIdx.x] = 0;
    return;
  }
  const uint64_t base = (uint64_t)blockIdx.x * KR_PFX_TILE + (uint64_t)threadIdx.x * KR_PFX_PER;
  uint32_t v = 0;
#pragma unroll
  for (int u = 0; u < KR_PFX_PER; ++u) v += kr_rec_n(rec, base + u, n_rec, cap, cursor);
  uint32_t tot;
  (void)kr_block_scan(v, s_w, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// one block: bsum[0 .. n_blocks) -> exclusive prefix in place, bsum[n_blocks] = the total (16 consecutive entries per thread)
#define KR_PFX_BPER 16
__global__ __launch_bounds__(KR_PFX_THREADS) void ku_route_prefix_blocks_kernel(uint32_t *bsum, uint32_t n_blocks) {
  __shared__ uint32_t s_w[KR_PFX_THREADS / 64];
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < n_blocks; b0 += KR_PFX_THREADS * KR_PFX_BPER) {  // block-uniform trip count
    const uint32_t i0 = b0 + threadIdx.x * KR_PFX_BPER;
    uint32_t x[KR_PFX_BPER], v = 0;
#pragma unroll
    for (int u = 0; u < KR_PFX_BPER; ++u) {
      x[u] = i0 + u < n_blocks ? bsum[i0 + u] : 0u;
      v += x[u];
    }
    uint32_t tot;
    uint32_t run = carry + kr_block_scan(v, s_w, &tot) - v;
#pragma unroll
    for (int u = 0; u < KR_PFX_BPER; ++u) {
      if (i0 + u < n_blocks) bsum[i0 + u] = run;
      run += x[u];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) bsum[n_blocks] = carry;
}
__global__ __launch_bounds__(KR_PFX_THREADS) void ku_route_prefix_write_kernel(const uint4 *__restrict__ rec, uint64_t n_rec, uint64_t cap,
                                                                                const unsigned long long *__restrict__ cursor,
                                                                                const uint32_t *__restrict__ bsum, uint32_t n_blocks,
                                                                                uint32_t *__restrict__ kb) {
  __shared__ uint32_t s_w[KR_PFX_THREADS / 64];
  if (blockIdx.x == 0 && threadIdx.x == 0) kb[n_rec] = bsum[n_blocks];
  if (kr_tile_unused((uint64_t)blockIdx.x * KR_PFX_TILE, n_rec, cap, cursor)) return;  // block-uniform
  const uint64_t base = (uint64_t)blockIdx.x * KR_PFX_TILE + (uint64_t)threadIdx.x * KR_PFX_PER;
  uint32_t n[KR_PFX_PER], v = 0;
#pragma unroll
  for (int u = 0; u < KR_PFX_PER; ++u) {
    n[u] = kr_rec_n(rec, base + u, n_rec, cap, cursor);
    v += n[u];
  }
  uint32_t tot;
  uint32_t run = bsum[blockIdx.x] + kr_block_scan(v, s_w, &tot) - v;
#pragma unroll
  for (int u = 0; u < KR_PFX_PER; ++u) {
    if (base + u < n_rec) kb[base + u] = run;
    run += n[u];
  }
}
// per queue of the sender's buffer: {records claimed (the cursor), k-mers}; behind them the queues' capacity
__global__ void ku_route_totals_kernel(const uint32_t *__restrict__ kb, uint64_t cap, uint32_t world,
                                       const unsigned long long *__restrict__ cursor, unsigned long long *__restrict__ tot) {
  const uint32_t q = threadIdx.x;
  if (q < world) {
    tot[2 * q] = cursor[q * KU_ROUTE_CURSOR_STRIDE];
    tot[2 * q + 1] = (unsigned long long)(kb[(uint64_t)(q + 1) * cap] - kb[(uint64_t)q * cap]);
  }
  if (q == world) tot[2 * world] = cap;
}

uint64_t ku_route_prefix_work_bytes(uint64_t n_rec) { return ((n_rec + KR_PFX_TILE - 1) / KR_PFX_TILE + 2) * 4; }

int ku_launch_route_prefix(const void *d_rec, uint64_t n_rec, uint64_t cap, const unsigned long long *d_cursor, uint32_t *d_kb,
                           unsigned long long *d_tot, void *d_work, hipStream_t stream) {
  if (!d_kb || !d_work || (n_rec && !d_rec) || (cap && (!d_cursor || n_rec % cap))) return KU_EINVAL;
  const uint64_t n_blocks = (n_rec + KR_PFX_TILE - 1) / KR_PFX_TILE;
  if (n_blocks >= (1ull << 31)) return KU_EUNSUP;
  const uint4 *rec = (const uint4 *)d_rec;
  uint32_t *bsum = (uint32_t *)d_work;
  if (n_blocks)
    hipLaunchKernelGGL(ku_route_prefix_sums_kernel, dim3((unsigned)n_blocks), dim3(KR_PFX_THREADS), 0, stream, rec, n_rec, cap, d_cursor, bsum);
  hipLaunchKernelGGL(ku_route_prefix_blocks_kernel, dim3(1), dim3(KR_PFX_THREADS), 0, stream, bsum, (uint32_t)n_blocks);
  // (n_rec = 0: one block still writes kb[0] = 0)
  hipLaunchKernelGGL(ku_route_prefix_write_kernel, dim3((unsigned)std::max<uint64_t>(n_blocks, 1)), dim3(KR_PFX_THREADS), 0, stream, rec, n_rec,
                     cap, d_cursor, bsum, (uint32_t)n_blocks, d_kb);
  if (d_tot && cap) {
    const uint32_t world = (uint32_t)(n_rec / cap);
    if (world > 64) return KU_EINVAL;
    hipLaunchKernelGGL(ku_route_totals_kernel, dim3(1), dim3(128), 0, stream, d_kb, cap, world, d_cursor, d_tot);
  }
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// ---------------------------------------------------------------------------- the owner
// code word i (16 bases) of a record; the low half of word 3 is not bases (the callers never shift it into reach)
__device__ __forceinline__ uint32_t kr_word(const uint4 &r, uint32_t i) { return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : (i == 3 ? r.w : 0u))); }

template <int LOG2>
__device__ __forceinline__ void kr_ct_flush(uint32_t *key, uint32_t *cnt, uint32_t *used, unsigned long long *global, uint32_t lane) {
  kr_wave_sync();
  for (uint32_t i = lane; i < (1u << LOG2); i += 64) {
    const uint32_t kk = key[i];
    if (kk) atomicAdd(&global[kk - 1], (unsigned long long)cnt[i]);
    key[i] = 0;
    cnt[i] = 0;
  }
  if (lane == 0) *used = 0;
  kr_wave_sync();
}

// One wave per group of 64 received records (one record per lane on the way in, one k-mer per lane and item on the way
// through the table): the lanes load their records, the group's k-mers -- kb[] numbers them -- are spread over the lanes
// through a byte map in the wave's LDS (k-mer -> record), two per lane and round, so that neighbouring lanes hold
// neighbouring k-mers of a read as in the other lookup kernels (they share bucket lines).  Per record (once): the anchor
// m-mer's key and strand bits.  Per k-mer: the k-mer from the record's bases (funnel shift), reverse complement, canonical
// form, anchor offset in the canonical frame, locus key, bucket probe, HLL + n_kmers, slot -> slots[kb[record] + j].
// No block-level barrier: the four waves of a block are independent.
// KK / MM: compile-time k-mer and minimizer lengths of the common database geometries (0 = read them from `db`): shift
// counts, the window and the flank length become literals.
template <bool DO_COUNTS, int KK, int MM>
__global__ __launch_bounds__(64 * KR_WAVES) void ku_route_owner_kernel(KuDbDev db, KuCountsDev cnt, const uint4 *__restrict__ rec,
                                                                        uint64_t n_groups, const uint32_t *__restrict__ kb,
                                                                        uint32_t *__restrict__ slots) {
  __shared__ uint4 s_rec[KR_WAVES][64];
  __shared__ uint32_t s_m0[KR_WAVES][64];   // anchor key << 6 | anchor offset in the record
  __shared__ uint32_t s_m1[KR_WAVES][64];   // k-mers of the group before the record | forward <= rc << 16 | rc <= forward << 17
  __shared__ uint8_t s_own[KR_WAVES][KR_OWN_BYTES];
  __shared__ uint32_t s_kk[KR_WAVES][1 << KR_KCT_LOG2], s_kc[KR_WAVES][1 << KR_KCT_LOG2];
  __shared__ uint32_t s_misc[KR_WAVES][2];

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint4 *w_rec = s_rec[wv];
  uint32_t *w_m0 = s_m0[wv], *w_m1 = s_m1[wv], *misc = s_misc[wv];
  uint8_t *w_own = s_own[wv];
  const uint32_t k = KK ? (uint32_t)KK : db.k, m = KK ? (uint32_t)MM : db.nt, w = k - m + 1;
  const uint32_t key_shift = ku_key_shift(m);
  const uint32_t *tab = reinterpret_cast<const uint32_t *>(db.table);
  if (DO_COUNTS) {
    for (uint32_t i = lane; i < (1u << KR_KCT_LOG2); i += 64) { s_kk[wv][i] = 0; s_kc[wv][i] = 0; }
    if (lane == 0) misc[0] = 0;
  }
  kr_wave_sync();

  const uint64_t n_waves = (uint64_t)gridDim.x * KR_WAVES;
  // the next group's records are requested while the current one is worked on
  uint4 R_n = make_uint4(0u, 0u, 0u, 0u);
  uint32_t kb_n = 0;
  {
    const uint64_t g0 = (uint64_t)blockIdx.x * KR_WAVES + wv;
    if (g0 < n_groups) { R_n = rec[g0 * 64 + lane]; kb_n = kb[g0 * 64 + lane]; }
  }
  for (uint64_t g = (uint64_t)blockIdx.x * KR_WAVES + wv; g < n_groups; g += n_waves) {
    if (DO_COUNTS && misc[0] > (1u << KR_KCT_LOG2) / 2) kr_ct_flush<KR_KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], cnt.n_kmers, lane);
    const uint4 R = R_n;
    const uint32_t kb_l = kb_n;
    if (g + n_waves < n_groups) { R_n = rec[(g + n_waves) * 64 + lane]; kb_n = kb[(g + n_waves) * 64 + lane]; }
    const uint32_t kb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)kb_l);
    const uint32_t n = R.w & 31u, a0 = (R.w >> 8) & 63u, pre = kb_l - kb0;
    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)(pre + n), 63);  // k-mers of the group
    {
      // the anchor m-mer (read strand) -> order key and strand bits, once per record
      const uint32_t wi = a0 >> 4, sh = (a0 & 15u) * 2;
      const uint64_t two = ((uint64_t)kr_word(R, wi) << 32) | kr_word(R, wi + 1);
      const uint32_t mmf = (uint32_t)((two << sh) >> (64 - 2 * m));
      const uint32_t rcm = ku_revcomp32(mmf, m);
      const uint32_t val = (mmf < rcm ? mmf : rcm) ^ db.xor_mask;
      w_m0[lane] = ((val >> key_shift) << 6) | a0;
      w_m1[lane] = pre | ((uint32_t)(mmf <= rcm) << 16) | ((uint32_t)(rcm <= mmf) << 17);
      w_rec[lane] = R;
      for (uint32_t i = 0; i < n; ++i) w_own[pre + i] = (uint8_t)lane;
    }
    kr_wave_sync();

    for (uint32_t e0 = 0; e0 < T; e0 += 64 * KR_ITEMS) {  // wave-uniform
      uint32_t v[KR_ITEMS];
      uint64_t hh[KR_ITEMS], canon[KR_ITEMS];
      bool ok[KR_ITEMS];
      const uint32_t *lp[KR_ITEMS];
      uint32_t tag[KR_ITEMS], cand[KR_ITEMS];
      bool act[KR_ITEMS], ovf[KR_ITEMS];
#pragma unroll
      for (int j = 0; j < KR_ITEMS; ++j) {
        const uint32_t e = e0 + j * 64 + lane;
        ok[j] = e < T;
        const uint32_t r = (uint32_t)w_own[ok[j] ? e : 0u];
        const uint4 Q = w_rec[r];
        const uint32_t m0 = w_m0[r], m1 = w_m1[r];
        const uint32_t pj = ok[j] ? e - (m1 & 0xFFFFu) : 0u;  // index of the k-mer in its record
        const uint32_t wi = pj >> 4, sh = (pj & 15u) * 2;
        const uint32_t c0 = wi ? Q.y : Q.x, c1 = wi ? Q.z : Q.y, c2 = wi ? Q.w : Q.z;
        // 32 bases from base pj: two 64-bit shifts, no special case for sh = 0
        const uint64_t x = ((((uint64_t)c0 << 32) | c1) << sh >> 32 << 32) | ((((uint64_t)c1 << 32) | c2) << sh >> 32);
        const uint64_t fwd = x >> (64 - 2 * k);
        const uint64_t rc = ku_revcomp64(fwd, k);
        const bool is_fwd = fwd <= rc;
        canon[j] = is_fwd ? fwd : rc;
        const uint32_t anc = m0 & 63u;
        const uint32_t t = ok[j] ? anc - pj : 0u;  // read-order offset of the anchor in the k-mer's window
        const uint32_t aoff = is_fwd ? t : w - 1 - t;
        const bool plus = ((m1 >> (is_fwd ? 16 : 17)) & 1u) != 0;
        const uint64_t locus = ku_locus_assemble(canon[j], m0 >> 6, aoff, plus, k, m);
        hh[j] = ku_fmix64(canon[j]);
        const uint64_t line = ku_locus_line(locus, db.n_lines);
        lp[j] = tab + (ok[j] ? line : 0) * KU_LINE_DWORDS;  // idle lanes share bucket 0 (no stray line fetches)
        tag[j] = ku_table_tag(hh[j]);
        act[j] = ok[j];
        v[j] = 0;
      }
      // ---- bucket probe (header round trip, entry in the same line, lockstep tail), as in the fused kernel
      uint4 h4[KR_ITEMS];
#pragma unroll
      for (int j = 0; j < KR_ITEMS; ++j) h4[j] = *reinterpret_cast<const uint4 *>(lp[j]);
      KuPair pr[KR_ITEMS];
      const uint32_t *ep[KR_ITEMS];
#pragma unroll
      for (int j = 0; j < KR_ITEMS; ++j) {
        cand[j] = ku_tag_matches(h4[j], tag[j]) & (act[j] ? 0xFFu : 0u);
        ovf[j] = act[j] && ku_line_spilled(h4[j]);
        ep[j] = lp[j] + KU_LINE_ENTRY0 + 3 * (__builtin_ctz(cand[j] | 0x100u) & 7u);  // no candidate: entry 0 (bit 8 -> 0)
      }
#pragma unroll
      for (int j = 0; j < KR_ITEMS; ++j) pr[j] = *reinterpret_cast<const KuPair *>(ep[j]);
#pragma unroll
      for (int j = 0; j < KR_ITEMS; ++j) {
        const bool hit = cand[j] != 0 && (((uint64_t)pr[j].key_hi << 32) | pr[j].key_lo) == canon[j];
        v[j] = hit ? pr[j].slot : 0u;
        cand[j] &= cand[j] - 1;  // no-op for 0
        act[j] = !hit && (cand[j] != 0 || ovf[j]);
      }
      bool any = false;
#pragma unroll
      for (int j = 0; j < KR_ITEMS; ++j) any |= act[j];
      while (any) {
        any = false;
        KuPair en[KR_ITEMS];
        uint4 a4[KR_ITEMS];
#pragma unroll
        for (int j = 0; j < KR_ITEMS; ++j) {
          if (act[j]) {
            if (cand[j]) {
              en[j] = *reinterpret_cast<const KuPair *>(lp[j] + KU_LINE_ENTRY0 + 3 * (__builtin_ctz(cand[j])));
            } else {
              lp[j] += KU_LINE_DWORDS;
              if (lp[j] == tab + db.n_lines * KU_LINE_DWORDS) lp[j] = tab;
              a4[j] = *reinterpret_cast<const uint4 *>(lp[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < KR_ITEMS; ++j) {
          if (act[j]) {
            if (cand[j]) {
              cand[j] &= cand[j] - 1;
              if ((((uint64_t)en[j].key_hi << 32) | en[j].key_lo) == canon[j]) {
                v[j] = en[j].slot;
                act[j] = false;
              }
            } else {
              cand[j] = ku_tag_matches(a4[j], tag[j]);
              ovf[j] = ku_line_spilled(a4[j]);
            }
            if (act[j]) act[j] = cand[j] != 0 || ovf[j];
            any |= act[j];
          }
        }
      }
      // ---- ReadCounts::add_kmer for every k-mer, misses included (classify.cpp:939)
      if (DO_COUNTS) {
        uint8_t *reg[KR_ITEMS];
        uint32_t rank[KR_ITEMS], seen[KR_ITEMS];
#pragma unroll
        for (int j = 0; j < KR_ITEMS; ++j) {  // every register byte is requested before the first one is looked at
          reg[j] = ku_hll_locate(cnt.registers, ok[j] ? v[j] : 0u, hh[j], rank[j]);
          seen[j] = ok[j] ? (uint32_t)*reg[j] : 0xFFu;
        }
#pragma unroll
        for (int j = 0; j < KR_ITEMS; ++j) {
          ku_hll_raise(reg[j], seen[j], rank[j]);
          // n_kmers: one counter update per distinct slot among the item's 64 k-mers (mostly one taxon and the misses) --
          // per-lane updates would queue on one or two LDS words
          unsigned long long todo = __ballot(ok[j]);
          while (todo) {  // wave-uniform
            const uint32_t lead = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t s0 = ku_wave_bcast(v[j], lead);
            const unsigned long long same = __ballot(ok[j] && v[j] == s0);
            if (lane == lead) ku_ct_add<KR_KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], s0, (uint32_t)__popcll(same), cnt.n_kmers);
            todo &= ~same;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < KR_ITEMS; ++j)
        if (ok[j]) slots[(uint64_t)kb0 + e0 + j * 64 + lane] = v[j];
    }
    kr_wave_sync();  // the next group reuses the wave's LDS arrays
  }
  if (DO_COUNTS) kr_ct_flush<KR_KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], cnt.n_kmers, lane);
}

int ku_launch_route_owner(const KuDbDev &db, const KuCountsDev &cnt, const void *d_rec, uint64_t n_rec, const uint32_t *d_kb,
                          uint32_t *d_slots, bool do_counts, int n_cu, hipStream_t stream) {
  if (n_rec == 0) return KU_OK;
  if (!db.table || !d_rec || !d_kb || !d_slots || n_rec % 64) return KU_EINVAL;
  const uint64_t n_groups = n_rec / 64, want = (n_groups + KR_WAVES - 1) / KR_WAVES;
  // persistent grid, every wave strides over the groups: six blocks per CU (22 KB of LDS each: seven would fill it) --
  // the kernel is bound by the memory system's line rate, and the scan / resolve kernels of the neighbouring rounds, bound
  // by instruction issue, are meant to run beside it on the rest of the CU (ku_mgpu.cpp)
  const char *oe = getenv("KU_ROUTE_BLOCKS_PER_CU");
  const uint64_t cap = (uint64_t)n_cu * (oe ? (uint64_t)atoi(oe) : 6ull);
  const dim3 grid((unsigned)(want < cap ? want : cap)), block(64 * KR_WAVES);
  const int geo = db.k != 31 ? 0 : (db.nt == 13 ? 13 : (db.nt == 15 ? 15 : 0));
#define KR_LAUNCH(C, K, M) \
  hipLaunchKernelGGL((ku_route_owner_kernel<C, K, M>), grid, block, 0, stream, db, cnt, (const uint4 *)d_rec, n_groups, d_kb, d_slots)
  if (do_counts) { if (geo == 13) KR_LAUNCH(true, 31, 13); else if (geo == 15) KR_LAUNCH(true, 31, 15); else KR_LAUNCH(true, 0, 0); }
  else { if (geo == 13) KR_LAUNCH(false, 31, 13); else if (geo == 15) KR_LAUNCH(false, 31, 15); else KR_LAUNCH(false, 0, 0); }
#undef KR_LAUNCH
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// ---------------------------------------------------------------------------- the sender's side of the return
// tickets -> slots, in place (the general path in front of the resolve / quick / sparse-emulation kernels)
__global__ void ku_route_gather_kernel(uint32_t *__restrict__ taxa, uint64_t n, const uint32_t *__restrict__ kb,
                                       const uint32_t *__restrict__ ret) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t t = taxa[i];
    if (t == KU_AMBIG) continue;
    taxa[i] = t == KU_ROUTE_MISS ? 0u : ret[(uint64_t)kb[t >> 5] + (t & 31u)];
  }
}

int ku_launch_route_gather(uint32_t *d_taxa, uint64_t n, const uint32_t *d_kb, const uint32_t *d_ret, hipStream_t stream) {
  if (n == 0) return KU_OK;
  if (!d_taxa || !d_kb) return KU_EINVAL;
  const uint64_t want = (n + 255) / 256;
  hipLaunchKernelGGL(ku_route_gather_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, stream, d_taxa, n, d_kb, d_ret);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
