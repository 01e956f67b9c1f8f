// ku_sparse.hip -- device side of the HyperLogLog++ SPARSE-mode emulation.
//
// The reference's sketch (HyperLogLogPlusMinus<uint64_t>, p = 12) starts sparse: a set of 32-bit encodings of the
// hashes at precision p' = 25 (encodeHashIn32Bit, hyperloglogplus.cpp:181-204).  A sketch turns dense when an insert
// finds 1024 entries in its set (the test precedes the insert, :496-498).  classify keeps one LOCAL sketch per taxon and
// work unit (-u nt of reads, classify.cpp:487-564) and merges it into the global one afterwards; a merge of two sparse
// sketches is a plain set union without any size test (:601-604).  Hence, for every taxon:
//     the global sketch is dense  <=>  in some work unit the taxon's local set reached 1024 distinct encodings and at
//                                      least one more insert (duplicate or not) followed in that unit;
//     otherwise it is sparse and holds every distinct encoding the run produced for the taxon, however many.
// The report's `kmers` column is the Ertl estimate of that state (sparse: m = 2^25, near exact).  The dense registers
// the main kernels keep are exact in both cases (sparse -> dense conversion is lossless, :559-577); this file adds what
// the sparse case needs:
//   L  per batch: set of (unit, slot, encoding) with the first position of each, to count distinct encodings per
//      (unit, slot) and to order "the insert that made the set 1024 big" against "the last insert";
//   U  per batch: (unit, slot) -> {distinct, last position, largest first position};
//   dense[slot]   sticky flag; flagged slots are skipped by every later insert (abundant taxa drop out after ~1 unit);
//   G  whole run: set of (slot, encoding) of the slots that are not dense -- what the host turns into the report.
// A unit that straddles two batches is carried over (its L / U entries re-enter the next batch as unit 0).
// One wave per read, run between the lookup and the resolve stage while taxa[] holds slot ids.
#include "ku_device.h"

#define KS_LIMIT 1024u  // m / 4 (hyperloglogplus.cpp:496)

// find or create the U cell of (unit, slot); returns its index or ~0 when the table is full
__device__ __forceinline__ uint64_t ks_u_cell(const KuSparseDev &s, uint32_t unit, uint32_t slot) {
  const unsigned long long key = ((unsigned long long)(unit + 1) << 32) | slot;
  uint64_t h = ks_mix(key) & s.u_mask;
  for (uint32_t probe = 0; probe < 4096; ++probe) {
    unsigned long long cur = __hip_atomic_load(&s.u_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) cur = atomicCAS(&s.u_key[h], 0ull, key);
    if (cur == 0 || cur == key) return h;
    h = (h + 1) & s.u_mask;
  }
  atomicOr(s.err, 2u);
  return ~0ull;
}

// one insert of the reference's local sketch of (unit, slot): encoding `enc` at position `pos`
__device__ __forceinline__ void ks_insert(const KuSparseDev &s, uint32_t unit, uint32_t slot, uint32_t enc, uint32_t pos) {
  const unsigned long long key = ((unsigned long long)(unit + 1) << 50) | ((unsigned long long)slot << 32) | enc;
  uint64_t h = ks_mix(key) & s.l_mask;
  bool fresh = false, placed = false;
  for (uint32_t probe = 0; probe < 4096; ++probe) {
    unsigned long long cur = __hip_atomic_load(&s.l_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) {
      cur = atomicCAS(&s.l_key[h], 0ull, key);
      fresh = cur == 0;
    }
    if (cur == 0 || cur == key) { placed = true; break; }
    h = (h + 1) & s.l_mask;
  }
  if (!placed) { atomicOr(s.err, 1u); return; }
  atomicMin(&s.l_first[h], pos);
  const uint64_t u = ks_u_cell(s, unit, slot);
  if (u == ~0ull) return;
  if (fresh && atomicAdd(&s.u_distinct[u], 1u) + 1 > KS_LIMIT) s.dense[slot] = 1u;  // 1025 distinct: dense whatever the order
  atomicMax(&s.u_last[u], pos);
}

__global__ __launch_bounds__(64) void ku_sparse_insert_kernel(KuSparseDev s, uint32_t k, const uint8_t *__restrict__ seqs,
                                                              const uint64_t *__restrict__ seq_off,
                                                              const uint32_t *__restrict__ seq_len,
                                                              const uint32_t *__restrict__ unit_of, uint64_t n_reads,
                                                              const uint32_t *__restrict__ taxa, uint32_t quick_min_hits) {
  const uint32_t lane = threadIdx.x;
  for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const uint32_t len = seq_len[r];
    const uint32_t n = len >= k ? len - k + 1 : 0;
    const uint64_t off = seq_off[r];
    const uint32_t unit = unit_of[r];
    uint32_t stop = n;
    if (quick_min_hits) {  // quick mode books the scanned prefix only (classify.cpp:943-944)
      uint32_t total = 0;
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t v = i < n ? taxa[off + i] : 0;
        const bool hit = v != 0 && v != KU_AMBIG;
        const unsigned long long m = __ballot(hit);
        const uint32_t c = (uint32_t)__popcll(m);
        if (total + c >= quick_min_hits) {
          const uint32_t need = quick_min_hits - total;
          const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          const unsigned long long sm = __ballot(hit && before + 1 == need);
          stop = base + (uint32_t)__ffsll((long long)sm);
          break;
        }
        total += c;
      }
    }
    for (uint32_t i = lane; i < stop; i += 64) {
      const uint32_t slot = taxa[off + i];
      if (slot == KU_AMBIG) continue;
      if (s.dense[slot]) continue;
      // canonical k-mer straight from the text (positions with a slot are unambiguous)
      uint64_t fwd = 0;
      const uint8_t *p = seqs + off + i;
      for (uint32_t j = 0; j < k; ++j) {
        const uint32_t c = p[j] & 0xDFu;
        fwd = (fwd << 2) | (((c >> 1) ^ (c >> 2)) & 3u);
      }
      const uint64_t rc = ku_revcomp64(fwd, k);
      const uint64_t h = ku_fmix64(fwd < rc ? fwd : rc);
      ks_insert(s, unit, slot, ks_encode(h), (uint32_t)(off + i) + 2u);  // positions 0 / 1 belong to carried-over state
    }
  }
}

// ---- fast path (KuSparseFast): the fused kernel counted the inserts of every (unit, slot) pair of the batch and put the
// encodings of all slots that are not dense into G itself.  What is left is the exact per-unit evaluation, and only
// where a sketch can switch at all: units in which a slot that is not dense yet received >= 1025 inserts, plus the
// units that straddle two batches (their state is carried as before).
// unit_flag[u] |= 1 for every unit of the batch with such a slot
__global__ void ku_sparse_flag_units_kernel(const uint32_t *__restrict__ u_cnt, uint64_t n_cells, uint32_t n_slots,
                                            const uint32_t *__restrict__ dense, uint8_t *unit_flag) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_cells; i += (uint64_t)gridDim.x * blockDim.x) {
    if (u_cnt[i] < KU_SPARSE_SWITCH_INSERTS) continue;
    const uint32_t slot = (uint32_t)(i % n_slots);
    if (!dense[slot]) unit_flag[i / n_slots] = 1;
  }
}

// The reads of the flagged units once more, from the run-length encoded per-k-mer codes the fused kernel left behind:
// one wave per listed read.  list_read[i] = index of the read, list_unit[i] = local unit number of this pass (bits
// 0..13), bit 31: every slot of the unit is tracked (a unit that straddles batches), else only the slots with >= 1025
// inserts; list_urow[i] = the unit's row in u_cnt.  Runs carry taxids: slot_taxid is ascending, a binary search per run
// finds the slot.
__global__ __launch_bounds__(64) void ku_sparse_insert_runs_kernel(KuSparseDev s, uint32_t k, const uint8_t *__restrict__ seqs,
                                                                   const uint64_t *__restrict__ seq_off,
                                                                   const uint32_t *__restrict__ seq_len,
                                                                   const uint32_t *__restrict__ list_read,
                                                                   const uint32_t *__restrict__ list_unit,
                                                                   const uint32_t *__restrict__ list_urow, uint64_t n_list,
                                                                   const uint2 *__restrict__ runs, const uint64_t *__restrict__ run_off,
                                                                   const uint32_t *__restrict__ run_cnt,
                                                                   const uint32_t *__restrict__ slot_taxid, uint32_t n_slots,
                                                                   const uint32_t *__restrict__ u_cnt, uint32_t pos_base) {
  // pos_base: what a unit's inserts BEFORE this buffer took of the position space (a unit whose first reads are evaluated
  // from the tail buffer, ku_api.cpp: their positions come first)
  const uint32_t lane = threadIdx.x;
  for (uint64_t li = blockIdx.x; li < n_list; li += gridDim.x) {
    const uint32_t r = list_read[li], lu = list_unit[li];
    const uint32_t unit = lu & 0x3FFFu;
    const bool all = (lu >> 31) != 0;
    const uint32_t *urow = u_cnt + (size_t)list_urow[li] * n_slots;
    const uint32_t len = seq_len[r];
    const uint32_t n = len >= k ? len - k + 1 : 0;
    const uint64_t off = seq_off[r], ro = run_off[r];
    const uint32_t nr = run_cnt[r];
    for (uint32_t j = 0; j < nr; ++j) {
      const uint2 run = runs[ro + j];
      if (run.x == KU_AMBIG) continue;
      uint32_t lo = 0, hi = n_slots;  // first slot with slot_taxid >= code (slot 0 = taxid 0 = a miss)
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (slot_taxid[mid] < run.x) lo = mid + 1; else hi = mid;
      }
      const uint32_t slot = lo;
      if (slot >= n_slots || slot_taxid[slot] != run.x) {  // a code that is no slot's taxid: never book it under a neighbour
        if (lane == 0) atomicOr(s.err, 8u);
        continue;
      }
      if (s.dense[slot]) continue;
      if (!all && urow[slot] < KU_SPARSE_SWITCH_INSERTS) continue;
      const uint32_t end = j + 1 < nr ? runs[ro + j + 1].y : n;
      for (uint32_t i = run.y + lane; i < end; i += 64) {
        uint64_t fwd = 0;
        const uint8_t *p = seqs + off + i;
        for (uint32_t b = 0; b < k; ++b) {
          const uint32_t c = p[b] & 0xDFu;
          fwd = (fwd << 2) | (((c >> 1) ^ (c >> 2)) & 3u);
        }
        const uint64_t rc = ku_revcomp64(fwd, k);
        ks_insert(s, unit, slot, ks_encode(ku_fmix64(fwd < rc ? fwd : rc)), pos_base + (uint32_t)(off + i) + 2u);
      }
    }
  }
}

// largest first position of a (unit, slot)'s encodings = the insert that brought its set to its final size
__global__ void ku_sparse_maxfirst_kernel(KuSparseDev s) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= s.l_mask; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = s.l_key[i];
    if (!key) continue;
    const uint32_t unit = (uint32_t)(key >> 50) - 1, slot = (uint32_t)(key >> 32) & 0x3FFFFu;
    const uint64_t u = ks_u_cell(s, unit, slot);
    if (u != ~0ull) atomicMax(&s.u_maxfirst[u], s.l_first[i]);
  }
}

// sum of a per-lane count over the wave, one atomic per wave (every lane of the wave must arrive)
__device__ __forceinline__ void ks_count_add(unsigned long long *counter, unsigned long long mine) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, 64);
  if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(counter, mine);
}

// closed units (unit < n_closed): did the local sketch switch to the dense representation?
__global__ void ku_sparse_eval_kernel(KuSparseDev s, uint32_t n_closed) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= s.u_mask; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = s.u_key[i];
    if (!key) continue;
    const uint32_t unit = (uint32_t)(key >> 32) - 1, slot = (uint32_t)key;
    if (unit >= n_closed) continue;
    const uint32_t d = s.u_distinct[i];
    // exactly 1024 entries: the switch happens iff an insert follows the one that added the 1024th (:496-498)
    if (d > KS_LIMIT || (d == KS_LIMIT && s.u_last[i] > s.u_maxfirst[i])) s.dense[slot] = 1u;
  }
}

// closed units of slots that stayed sparse: their encodings join the run's global set (the sparse + sparse merge)
// skip_hits: the pass belongs to the fused kernel's fast path -- the k-mers the database holds are booked by the SEEN marks of
// their table entries (ku_device.h), only the misses (slot 0) live in the set
__global__ void ku_sparse_commit_kernel(KuSparseDev s, uint32_t n_closed, uint32_t skip_hits) {
  unsigned long long n_new = 0;  // one add to the set's size per wave, not per entry (a single hot address otherwise)
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= s.l_mask; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = s.l_key[i];
    if (!key) continue;
    const uint32_t unit = (uint32_t)(key >> 50) - 1, slot = (uint32_t)(key >> 32) & 0x3FFFFu, enc = (uint32_t)key;
    if (unit >= n_closed || s.dense[slot] || (skip_hits && slot != 0)) continue;
    const unsigned long long gk = ((unsigned long long)(slot + 1) << 32) | enc;
    uint64_t h = ks_mix(gk) & s.g_mask;
    bool placed = false;
    for (uint32_t probe = 0; probe < 4096; ++probe) {
      unsigned long long cur = __hip_atomic_load(&s.g_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0) {
        cur = atomicCAS(&s.g_key[h], 0ull, gk);
        if (cur == 0) ++n_new;
      }
      if (cur == 0 || cur == gk) { placed = true; break; }
      h = (h + 1) & s.g_mask;
    }
    if (!placed) atomicOr(s.err, 4u);
  }
  ks_count_add(s.g_count, n_new);
}

// the open unit (index `unit`) moves on to the next batch: its encodings and its (unit, slot) statistics.  Places in
// the carry arrays are claimed ONCE PER BLOCK: a block owns a contiguous stretch of the tables, counts what it will take
// from it, adds that to the global counters (two adds per block), and goes over its stretch a second time to write --
// the entries of a thread in the order of the first pass, the threads' shares laid out by a scan through LDS.  (One add
// per entry, per wave, and even per block and 256 cells queued the grid on the two counters: 0.5 / 0.17 ms per batch.)
__global__ __launch_bounds__(256) void ku_sparse_carry_out_kernel(KuSparseDev s, uint32_t unit, unsigned long long *carry_l, uint32_t *carry_u,
                                                                  unsigned long long *counters, uint64_t cap_l, uint64_t cap_u) {
  __shared__ uint32_t s_l[256], s_u[256];
  __shared__ unsigned long long s_base[2];
  const uint64_t n = (s.l_mask > s.u_mask ? s.l_mask : s.u_mask) + 1;
  const uint64_t per_block = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
  auto take_l = [&](uint64_t i, unsigned long long &lk) {
    lk = i <= s.l_mask ? s.l_key[i] : 0ull;
    return lk && (uint32_t)(lk >> 50) - 1 == unit && !s.dense[(uint32_t)(lk >> 32) & 0x3FFFFu];
  };
  auto take_u = [&](uint64_t i, unsigned long long &uk) {
    uk = i <= s.u_mask ? s.u_key[i] : 0ull;
    return uk && (uint32_t)(uk >> 32) - 1 == unit && !s.dense[(uint32_t)uk];
  };
  uint32_t nl = 0, nu = 0;
  unsigned long long key;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    nl += take_l(i, key) ? 1u : 0u;
    nu += take_u(i, key) ? 1u : 0u;
  }
  s_l[threadIdx.x] = nl;
  s_u[threadIdx.x] = nu;
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive scan over the threads' counts, then the block's two claims
    uint32_t rl = 0, ru = 0;
    for (uint32_t t = 0; t < blockDim.x; ++t) {
      const uint32_t a = s_l[t], b = s_u[t];
      s_l[t] = rl; s_u[t] = ru;
      rl += a; ru += b;
    }
    s_base[0] = rl ? atomicAdd(&counters[0], (unsigned long long)rl) : 0ull;
    s_base[1] = ru ? atomicAdd(&counters[1], (unsigned long long)ru) : 0ull;
  }
  __syncthreads();
  unsigned long long el = s_base[0] + s_l[threadIdx.x], eu = s_base[1] + s_u[threadIdx.x];
  for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    if (take_l(i, key)) {
      if (el < cap_l) carry_l[el] = key & 0x3FFFFFFFFFFFFull;  // slot << 32 | encoding
      else atomicOr(s.err, 1u);
      ++el;
    }
    if (take_u(i, key)) {
      if (eu < cap_u) {
        carry_u[3 * eu] = (uint32_t)key;
        carry_u[3 * eu + 1] = s.u_distinct[i];
        carry_u[3 * eu + 2] = s.u_last[i] > s.u_maxfirst[i] ? 1u : 0u;  // all that later inserts need of the order
      } else atomicOr(s.err, 2u);
      ++eu;
    }
  }
}
// ... and re-enters as unit 0: carried encodings sit at position 0, the carried "last insert" at 0 or 1
__global__ void ku_sparse_carry_in_kernel(KuSparseDev s, const unsigned long long *carry_l, uint64_t n_l, const uint32_t *carry_u,
                                          uint64_t n_u) {
  const uint64_t n = n_l > n_u ? n_l : n_u;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (i < n_l) {
      const unsigned long long key = (1ull << 50) | carry_l[i];
      uint64_t h = ks_mix(key) & s.l_mask;
      for (uint32_t probe = 0; probe < 4096; ++probe) {
        unsigned long long cur = atomicCAS(&s.l_key[h], 0ull, key);
        if (cur == 0 || cur == key) { s.l_first[h] = 0; break; }
        h = (h + 1) & s.l_mask;
        if (probe == 4095) atomicOr(s.err, 1u);
      }
    }
    if (i < n_u) {
      const uint64_t u = ks_u_cell(s, 0, carry_u[3 * i]);
      if (u != ~0ull) {
        s.u_distinct[u] = carry_u[3 * i + 1];
        s.u_last[u] = carry_u[3 * i + 2];
      }
    }
  }
}

// the global set moves to a larger table (the entries of slots that turned dense meanwhile are dropped, g_count is recounted)
__global__ void ku_sparse_rehash_kernel(KuSparseDev s, const unsigned long long *old_keys, uint64_t old_cells) {
  unsigned long long n_new = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < old_cells; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long gk = old_keys[i];
    if (!gk || s.dense[(uint32_t)(gk >> 32) - 1]) continue;
    uint64_t h = ks_mix(gk) & s.g_mask;
    bool placed = false;
    for (uint32_t probe = 0; probe < 4096; ++probe) {
      const unsigned long long cur = atomicCAS(&s.g_key[h], 0ull, gk);
      if (cur == 0) { ++n_new; placed = true; break; }
      h = (h + 1) & s.g_mask;
    }
    if (!placed) atomicOr(s.err, 4u);
  }
  ks_count_add(s.g_count, n_new);
}

// another rank's run-wide set joins this one (several GPUs: the union of the ranks' sets; ku_ctx_sparse_absorb)
__global__ void ku_sparse_absorb_kernel(KuSparseDev s, const unsigned long long *__restrict__ keys, uint64_t n) {
  unsigned long long n_new = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long gk = keys[i];
    if (!gk) continue;
    const uint32_t slot = (uint32_t)(gk >> 32) - 1;
    if (s.dense[slot]) continue;
    if (ks_g_insert(s.g_key, s.g_mask, slot, (uint32_t)gk, s.err)) ++n_new;
  }
  ks_count_add(s.g_count, n_new);
}

// run's end: (slot, encoding) of every slot that stayed sparse
__global__ void ku_sparse_export_kernel(KuSparseDev s, unsigned long long *out, uint64_t cap, unsigned long long *counter) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= s.g_mask; i += (uint64_t)gridDim.x * blockDim.x) {
    // the table size is a power of two >= 1024: the lanes of a wave run the same number of rounds (ballots are safe);
    // one add to the output counter per wave and round
    const unsigned long long gk = s.g_key[i];
    const uint32_t slot = gk ? (uint32_t)(gk >> 32) - 1 : 0u;
    const bool emit = gk && !s.dense[slot];
    const unsigned long long bal = __ballot(emit);
    if (!bal) continue;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == (uint32_t)__ffsll((long long)bal) - 1) base = atomicAdd(counter, (unsigned long long)__popcll(bal));
    base = __shfl(base, __ffsll((long long)bal) - 1, 64);
    const unsigned long long e = base + (unsigned long long)__popcll(bal & ((1ull << lane) - 1ull));
    if (emit && e < cap) out[e] = ((unsigned long long)slot << 32) | (uint32_t)gk;
  }
}

// the per-pass tables start empty: one launch instead of six fills (a pass of the fast path is a few hundred microseconds
// of kernels; six fill launches were a tenth of a `classify -r` run's emulation time)
__global__ void ku_sparse_clear_kernel(KuSparseDev s) {
  const uint64_t nl = s.l_mask + 1, nu = s.u_mask + 1, n = nl > nu ? nl : nu;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (i < nl) { s.l_key[i] = 0ull; s.l_first[i] = 0xFFFFFFFFu; }
    if (i < nu) { s.u_key[i] = 0ull; s.u_distinct[i] = 0u; s.u_last[i] = 0u; s.u_maxfirst[i] = 0u; }
  }
}

// up to three dword ranges zeroed by one launch (the small per-batch counters of the fast path)
__global__ void ku_zero3_kernel(uint32_t *a, uint64_t na, uint32_t *b, uint64_t nb, uint32_t *c, uint64_t nc) {
  const uint64_t n = na + nb + nc;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (i < na) a[i] = 0u;
    else if (i < na + nb) b[i - na] = 0u;
    else c[i - na - nb] = 0u;
  }
}

// ---- the SEEN marks of the probe table (ku_device.h; set by the fused kernel's fast path, ku_short.hip OUT = 2)
// A marked entry stands for (slot of the entry, encoding of its k-mer's hash) in the run-wide set.  The report reads the marks
// where they lie (ku_report.hip); whoever needs the set as such -- ku_sparse_export, the union of several ranks' sets, a
// table about to be freed -- has them inserted into G first.  A thread per (line, entry).
template <int WHAT>  // 0: count the marked entries of slots that are not dense; 1: insert them into G; 2: clear all marks
__global__ __launch_bounds__(256) void ku_seen_kernel(uint32_t *table, uint64_t n_lines, KuSparseDev s, unsigned long long *count) {
  unsigned long long n = 0;
  const uint64_t n_items = n_lines * KU_LINE_SLOTS;
  // (n_items is a multiple of 8 and the stride of 64: the lanes of a wave run the same number of rounds)
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_items; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t *lp = table + (i >> 3) * KU_LINE_DWORDS;
    const uint32_t e = (uint32_t)i & 7u;
    if (WHAT == 2) {
      if (e < 2 && lp[KU_LINE_SEEN0 + e]) lp[KU_LINE_SEEN0 + e] = 0u;
      continue;
    }
    if (!reinterpret_cast<const uint8_t *>(lp + KU_LINE_SEEN0)[e]) continue;
    const uint32_t slot = lp[KU_LINE_ENTRY0 + 3 * e + 2];
    if (s.dense[slot]) continue;
    if (WHAT == 0) { ++n; continue; }
    const uint64_t key = ((uint64_t)lp[KU_LINE_ENTRY0 + 3 * e + 1] << 32) | lp[KU_LINE_ENTRY0 + 3 * e];
    if (ks_g_insert(s.g_key, s.g_mask, slot, ks_encode(ku_fmix64(key)), s.err)) ++n;
  }
  if (WHAT != 2) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(count, n);
  }
}

// ---------------------------------------------------------------------------- launch wrappers
static unsigned ks_grid(uint64_t n) {
  const uint64_t nb = (n + 255) / 256;
  return (unsigned)(nb < 16384 ? (nb ? nb : 1) : 16384);
}
int ku_launch_sparse_insert(const KuSparseDev &s, uint32_t k, const uint8_t *d_seqs, const uint64_t *d_seq_off,
                            const uint32_t *d_seq_len, const uint32_t *d_unit, uint64_t n_reads, const uint32_t *d_taxa,
                            uint32_t quick_min_hits, int n_cu, hipStream_t stream) {
  if (n_reads == 0) return KU_OK;
  const uint64_t cap = (uint64_t)n_cu * 32;
  hipLaunchKernelGGL(ku_sparse_insert_kernel, dim3((unsigned)(n_reads < cap ? n_reads : cap)), dim3(64), 0, stream, s, k, d_seqs,
                     d_seq_off, d_seq_len, d_unit, n_reads, d_taxa, quick_min_hits);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_zero3(void *a, uint64_t a_dwords, void *b, uint64_t b_dwords, void *c, uint64_t c_dwords, hipStream_t stream) {
  const uint64_t n = a_dwords + b_dwords + c_dwords;
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_zero3_kernel, dim3(ks_grid(n)), dim3(256), 0, stream, (uint32_t *)a, a_dwords, (uint32_t *)b, b_dwords, (uint32_t *)c, c_dwords);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
__global__ void ku_add_u32_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
int ku_launch_add_u32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t stream) {
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_add_u32_kernel, dim3(ks_grid(n)), dim3(256), 0, stream, dst, src, n);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_clear(const KuSparseDev &s, hipStream_t stream) {
  const uint64_t n = (s.l_mask > s.u_mask ? s.l_mask : s.u_mask) + 1;
  hipLaunchKernelGGL(ku_sparse_clear_kernel, dim3(ks_grid((n + 3) / 4)), dim3(256), 0, stream, s);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_close(const KuSparseDev &s, uint32_t n_closed, hipStream_t stream, bool skip_hits) {
  hipLaunchKernelGGL(ku_sparse_maxfirst_kernel, dim3(ks_grid(s.l_mask + 1)), dim3(256), 0, stream, s);
  hipLaunchKernelGGL(ku_sparse_eval_kernel, dim3(ks_grid(s.u_mask + 1)), dim3(256), 0, stream, s, n_closed);
  hipLaunchKernelGGL(ku_sparse_commit_kernel, dim3(ks_grid(s.l_mask + 1)), dim3(256), 0, stream, s, n_closed, skip_hits ? 1u : 0u);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_rehash(const KuSparseDev &s, const unsigned long long *d_old_keys, uint64_t old_cells, hipStream_t stream) {
  hipLaunchKernelGGL(ku_sparse_rehash_kernel, dim3(ks_grid(old_cells)), dim3(256), 0, stream, s, d_old_keys, old_cells);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_carry_out(const KuSparseDev &s, uint32_t unit, unsigned long long *d_carry_l, uint32_t *d_carry_u,
                               unsigned long long *d_counters, uint64_t cap_l, uint64_t cap_u, hipStream_t stream) {
  const uint64_t n = (s.l_mask > s.u_mask ? s.l_mask : s.u_mask) + 1;
  const uint64_t nb = (n + 4095) / 4096;  // at least 16 cells per thread: few blocks, two claims each
  hipLaunchKernelGGL(ku_sparse_carry_out_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, stream, s, unit, d_carry_l, d_carry_u,
                     d_counters, cap_l, cap_u);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_carry_in(const KuSparseDev &s, const unsigned long long *d_carry_l, uint64_t n_l, const uint32_t *d_carry_u,
                              uint64_t n_u, hipStream_t stream) {
  const uint64_t n = n_l > n_u ? n_l : n_u;
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_sparse_carry_in_kernel, dim3(ks_grid(n)), dim3(256), 0, stream, s, d_carry_l, n_l, d_carry_u, n_u);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_export(const KuSparseDev &s, unsigned long long *d_out, uint64_t cap, unsigned long long *d_counter,
                            hipStream_t stream) {
  hipLaunchKernelGGL(ku_sparse_export_kernel, dim3(ks_grid(s.g_mask + 1)), dim3(256), 0, stream, s, d_out, cap, d_counter);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_flag_units(const uint32_t *d_u_cnt, uint64_t n_cells, uint32_t n_slots, const uint32_t *d_dense, uint8_t *d_unit_flag,
                                hipStream_t stream) {
  if (n_cells == 0) return KU_OK;
  hipLaunchKernelGGL(ku_sparse_flag_units_kernel, dim3(ks_grid(n_cells)), dim3(256), 0, stream, d_u_cnt, n_cells, n_slots, d_dense, d_unit_flag);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_insert_runs(const KuSparseDev &s, uint32_t k, const uint8_t *d_seqs, const uint64_t *d_seq_off, const uint32_t *d_seq_len,
                                 const uint32_t *d_list_read, const uint32_t *d_list_unit, const uint32_t *d_list_urow, uint64_t n_list,
                                 const void *d_runs, const uint64_t *d_run_off, const uint32_t *d_run_cnt, const uint32_t *d_slot_taxid,
                                 uint32_t n_slots, const uint32_t *d_u_cnt, int n_cu, hipStream_t stream, uint32_t pos_base) {
  if (n_list == 0) return KU_OK;
  const uint64_t cap = (uint64_t)n_cu * 32;
  hipLaunchKernelGGL(ku_sparse_insert_runs_kernel, dim3((unsigned)(n_list < cap ? n_list : cap)), dim3(64), 0, stream, s, k, d_seqs, d_seq_off,
                     d_seq_len, d_list_read, d_list_unit, d_list_urow, n_list, (const uint2 *)d_runs, d_run_off, d_run_cnt, d_slot_taxid,
                     n_slots, d_u_cnt, pos_base);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_sparse_absorb(const KuSparseDev &s, const unsigned long long *d_keys, uint64_t n, hipStream_t stream) {
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_sparse_absorb_kernel, dim3(ks_grid(n)), dim3(256), 0, stream, s, d_keys, n);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
// the SEEN marks of a probe table: what = 0 counts the marked entries of slots that are not dense into *d_count, 1 inserts
// them into the run-wide set (d_count = its size counter), 2 clears every mark
int ku_launch_seen(int what, void *d_table, uint64_t n_lines, const KuSparseDev &s, unsigned long long *d_count, hipStream_t stream) {
  if (n_lines == 0) return KU_OK;
  const uint64_t nb = (n_lines * KU_LINE_SLOTS + 255) / 256;
  const dim3 grid((unsigned)(nb < 16384 ? nb : 16384)), block(256);
  if (what == 0) hipLaunchKernelGGL(ku_seen_kernel<0>, grid, block, 0, stream, (uint32_t *)d_table, n_lines, s, d_count);
  else if (what == 1) hipLaunchKernelGGL(ku_seen_kernel<1>, grid, block, 0, stream, (uint32_t *)d_table, n_lines, s, d_count);
  else hipLaunchKernelGGL(ku_seen_kernel<2>, grid, block, 0, stream, (uint32_t *)d_table, n_lines, s, d_count);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// Warm-up aid (ku_classify_batch_rle_reserve): a kernel that needs scratch memory, over the whole device.  The instances of the
// fused kernel that count spill 8-12 bytes per lane; the first dispatch on a queue that needs scratch makes the runtime allocate
// it for every wave slot of the device -- ~20 ms, which used to fall under the first batch of a run.  128 bytes per lane here, so
// that whichever instance follows finds enough.
__global__ __launch_bounds__(64) void ku_warm_scratch_kernel(uint32_t *out, uint32_t n) {
  volatile uint32_t a[32];
  for (uint32_t i = 0; i < 32; ++i) a[i] = i * n + threadIdx.x;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < 32; ++i) sum += a[(i * 7 + n) & 31];
  if (n == 0xFFFFFFFFu && out) out[0] = sum;  // (never: the caller passes n = 1)
}
int ku_launch_warm_scratch(int n_cu, hipStream_t stream) {
  hipLaunchKernelGGL(ku_warm_scratch_kernel, dim3((unsigned)(n_cu > 0 ? n_cu : 256) * 32), dim3(64), 0, stream, (uint32_t *)nullptr, 1u);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
