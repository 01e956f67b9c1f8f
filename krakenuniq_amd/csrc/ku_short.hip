// ku_short.hip -- fused classification, one wavefront per read, from ASCII to call (short reads in one pass, longer
// ones in windows).
//
// The flat kernel (ku_kernels.hip) + the resolve kernel are the general path (any read length, sharded mode,
// sorted layout).  For the headline workload -- 100-300 bp reads against the resident probe table -- this kernel
// does the whole of classify_sequence (src/classify.cpp:897-968) inside one wave64:
//   ASCII -> 2-bit codes (wave-private LDS) -> canonical k-mers, minimizer + locus key -> bucket probe
//   -> HLL / n_kmers -> hit_counts + resolve_tree + lca entirely from registers / wave-private LDS -> call,
//   per-k-mer taxids (coalesced store), n_reads.
// There is no block-level barrier anywhere: the 4 waves of a block are independent, so a wave that waits for
// HBM never holds the others back, the per-k-mer codes never make a round trip through HBM between a lookup and a
// resolve kernel, and no lane is spent on the separator bytes between reads.
// Eligibility (checked by the launcher): hash layout, whole bin range resident (not a shard), no quick mode.
// Reads of at most 64 * ITEMS k-mers (ITEMS = 2 or 3: up to 222 bp at k = 31) take one pass; longer ones (mate pairs
// 2 x 150 joined by 'N', long reads up to 65535 k-mers) the windowed instance (WIN, see the kernel's comment), beyond
// that the flat lookup + resolve kernels.
// Lane <-> position mapping: lane L owns the positions L, 64 + L (, 128 + L), NOT neighbouring ones.  Measured: a variant
// with two consecutive positions per lane (one window fetch and one 64-bit reverse complement shared by the pair, 60
// instead of 104 instructions in stage 2) ran 30.7 ms against 19.4 ms -- a bucket-header load instruction then covers
// every second k-mer, only two lanes instead of four share a bucket line per instruction, and the number of line
// requests, the scarce resource of the probe, doubles.
#include <cstdlib>
#include <type_traits>

#include "ku_device.h"

#define KS_WAVES 4  // reads in flight per 256-thread block
#ifndef KS_KCT_LOG2
#define KS_KCT_LOG2 8
#endif
#define KS_PAD 16   // sentinel elements behind a read's last m-mer (the widest doubling step reads 16 positions ahead)
#ifdef KU_ABLATION
#define KS_ABL(bit) ((ablate & (bit)) != 0)
#else
#define KS_ABL(bit) false
#endif

template <int ITEMS> struct KsGeom {
  static constexpr int MAXN = 64 * ITEMS;                      // k-mers per read
  static constexpr int NWORDS = (MAXN + 31 + 15) / 16 + 3;     // 16-base code words stage 1 fills (+ funnel slack)
  static constexpr int NCODES = (MAXN + 64) / 16 + 4;          // words allocated: lanes behind the read's end read junk here
  static constexpr int NAMB = (NWORDS + 1) / 2 + 2;            // 32-base ambiguity words
  static constexpr int NMM = MAXN + 64 + 32;                   // packed window elements: 64 * (ITEMS + 1) positions + the
                                                               // widest step (16) ahead; the last one is the dump slot
  static constexpr int TCAP_LOG2 = ITEMS <= 2 ? 8 : (ITEMS <= 4 ? 9 : 10);  // resolve table >= 2 * MAXN
  static constexpr int KCT_LOG2 = KS_KCT_LOG2;                 // n_kmers counter table (per wave)
  static constexpr int RCT_LOG2 = 6;                           // n_reads counter table (per wave)
};

// order the wave's own LDS traffic (hardware executes a wave's DS ops in order; this stops the compiler)
__device__ __forceinline__ void ks_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// `hot`: the wave's LAST flush -- the count under key 0 (the misses' counter, which every wave of a launch holds) goes to this LDS
// word instead of global memory; the block adds its waves' words up and issues one atomic (the end of the kernel).  A launch of the
// `classify` executable's size has 6 144 waves that all end within microseconds of each other: their adds to that ONE address
// queued for ~35 us of a 275 us launch (scripts/launch_shape_probe.py: 25.0 -> 22.1 ms per 10 M reads without them).
template <int LOG2>
__device__ __forceinline__ void ks_ct_flush(uint32_t *key, uint32_t *cnt, uint32_t *used, unsigned long long *global,
                                            uint32_t lane, uint32_t *hot = nullptr) {
  ks_wave_sync();
  for (uint32_t i = lane; i < (1u << LOG2); i += 64) {
    uint32_t kk = key[i];
#ifdef KS_DIAG_NOHOT  // (diagnostic builds: is it the ONE address every wave adds to -- the misses' counter -- that the flushes cost?)
    if (kk == 1) kk = 0;
#endif
    if (hot && kk == 1) { *hot = cnt[i]; kk = 0; }
    if (kk) atomicAdd(&global[kk - 1], (unsigned long long)cnt[i]);
    key[i] = 0;
    cnt[i] = 0;
  }
  if (lane == 0) *used = 0;
  ks_wave_sync();
}

// n_reads counter table of a wave: keys are taxonomy nodes, or KS_RCT_SLOT | slot for the reads that met a single
// taxon (their node is looked up once per flush instead of once per read)
#define KS_RCT_SLOT 0x40000000u
__device__ __forceinline__ uint32_t ks_rct_node(uint32_t id, const KuTaxDev &tax) {
  return (id & KS_RCT_SLOT) ? ((id & ~KS_RCT_SLOT) ? tax.slot_node[id & ~KS_RCT_SLOT] : 0u) : id;
}
template <int LOG2>
__device__ __forceinline__ void ks_rct_add(uint32_t *key, uint32_t *cnt, uint32_t *used, uint32_t id, const KuTaxDev &tax,
                                           unsigned long long *global) {
  uint32_t h = (id * 2654435761u) >> (32 - LOG2);
#pragma unroll 1
  for (int probe = 0; probe < 8; ++probe) {
    uint32_t cur = key[h];
    if (cur == 0) { key[h] = id + 1; *used += 1; cur = id + 1; }  // one lane of the wave owns the table: no atomics
    if (cur == id + 1) { cnt[h] += 1; return; }
    h = (h + 1) & ((1u << LOG2) - 1);
  }
  atomicAdd(&global[ks_rct_node(id, tax)], 1ull);
}
template <int LOG2>
__device__ __forceinline__ void ks_rct_flush(uint32_t *key, uint32_t *cnt, uint32_t *used, const KuTaxDev &tax,
                                             unsigned long long *global, uint32_t lane, uint32_t *hot = nullptr) {
  ks_wave_sync();
  for (uint32_t i = lane; i < (1u << LOG2); i += 64) {
    uint32_t kk = key[i];
#ifdef KS_DIAG_NOHOT
    if (kk - 1 == KS_RCT_SLOT) kk = 0;  // (the unclassified reads' counter)
#endif
    if (hot && kk - 1 == KS_RCT_SLOT) { *hot = cnt[i]; kk = 0; }  // (reads without a hit: node 0, see ks_ct_flush)
    if (kk) atomicAdd(&global[ks_rct_node(kk - 1, tax)], (unsigned long long)cnt[i]);
    key[i] = 0;
    cnt[i] = 0;
  }
  if (lane == 0) *used = 0;
  ks_wave_sync();
}

// ---- hit_counts of one read (classify.cpp:941) in a wave-private LDS table: open addressing, key = slot + 1, hit count
// in the low 16 bits of the value, the root-path score in the high 16 (a read has at most 65535 k-mers here)
template <int LOG2, bool TRACK>
__device__ __forceinline__ void ks_tab_add(uint32_t *t_key, uint32_t *t_cnt, uint32_t *n_distinct, uint32_t slot, uint32_t count) {
  uint32_t h = (slot * 2654435761u) >> (32 - LOG2);
  for (;;) {
    uint32_t cur = __hip_atomic_load(&t_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (cur == 0) {
      uint32_t old = atomicCAS(&t_key[h], 0u, slot + 1);
      if (TRACK && old == 0) atomicAdd(n_distinct, 1u);
      cur = old == 0 ? slot + 1 : old;
    }
    if (cur == slot + 1) {
      atomicAdd(&t_cnt[h], count);
      break;
    }
    h = (h + 1) & ((1u << LOG2) - 1);
  }
}

// resolve_tree (krakenutil.cpp:149-200) over the wave's table: score(t) = sum of the hit counts on t's root path, the
// best score wins, ties fold lca() in ascending taxid (= slot) order.  Leaves the table empty.  Returns the node.
// `urow` (sparse-sketch emulation, OUT = 2): the read's hit counts are its inserts per slot -- booked into the work unit's row of
// insert counts from here, one add per DISTINCT taxon of the read by the lane that owns the table entry (round 6; a ballot loop
// over the lanes' slots did it before: 5 of the instance's 37 ms per 10 M reads)
template <int LOG2>
__device__ __forceinline__ uint32_t ks_tab_resolve(uint32_t *t_key, uint32_t *t_cnt, uint16_t *t_list, uint32_t *list_len,
                                                   const KuTaxDev &tax, uint32_t lane, uint32_t *urow = nullptr) {
  constexpr uint32_t TCAP = 1u << LOG2;
  if (lane == 0) *list_len = 0;
  ks_wave_sync();
  for (uint32_t i = lane; i < TCAP; i += 64)
    if (t_key[i]) t_list[atomicAdd(list_len, 1u)] = (uint16_t)i;
  ks_wave_sync();
  const uint32_t n_list = *list_len;
  uint32_t my_max = 0;
  for (uint32_t e = lane; e < n_list; e += 64) {
    const uint32_t pos = t_list[e];
    const uint32_t sl = t_key[pos] - 1;
    if (urow) atomicAdd(&urow[sl], t_cnt[pos] & 0xffffu);  // (the scores are added into the high halves further down)
    uint32_t score = 0;
    for (uint32_t i = tax.slot_anc_off[sl], i_end = tax.slot_anc_off[sl + 1]; i < i_end; ++i) {
      const uint32_t s = tax.slot_anc[i];
      uint32_t h = (s * 2654435761u) >> (32 - LOG2);
      for (;;) {
        const uint32_t cur = t_key[h];
        if (cur == s + 1) { score += t_cnt[h] & 0xffffu; break; }
        if (cur == 0) break;
        h = (h + 1) & (TCAP - 1);
      }
    }
    atomicAdd(&t_cnt[pos], score << 16);
    my_max = max(my_max, score);
  }
  my_max = ku_wave_max_u32(my_max);
  ks_wave_sync();
  uint32_t last = 0, res = 0;
  bool firstt = true;
  for (;;) {
    uint32_t my_min = 0xFFFFFFFFu;
    for (uint32_t e = lane; e < n_list; e += 64) {
      const uint32_t pos = t_list[e];
      const uint32_t s = t_key[pos] - 1;
      if ((t_cnt[pos] >> 16) == my_max && s > last) my_min = min(my_min, s);
    }
    my_min = ku_wave_min_u32(my_min);
    if (my_min == 0xFFFFFFFFu) break;
    const uint32_t node = tax.slot_node[my_min];
    res = firstt ? node : ku_lca_nodes(tax.node_parent, res, node);  // uniform: every lane computes the same
    firstt = false;
    last = my_min;
  }
  ks_wave_sync();
  for (uint32_t e = lane; e < n_list; e += 64) {
    const uint32_t pos = t_list[e];
    t_key[pos] = 0;
    t_cnt[pos] = 0;
  }
  ks_wave_sync();
  return res;
}

// ---- the same table in global memory, for the rare long read that meets more distinct taxa than the LDS table holds
// (windowed variant only).  One region of `cap` keys + `cap` values per wave; device-scope atomics and atomic loads
// (the wave's own earlier atomics must be visible to its later loads past the L1).
__device__ __forceinline__ uint32_t ks_gload(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __noinline__ void ks_spill_add(uint32_t *g_key, uint32_t *g_cnt, uint32_t mask, uint32_t slot, uint32_t count) {
  uint32_t h = (slot * 2654435761u) & mask;
  for (;;) {
    uint32_t cur = ks_gload(&g_key[h]);
    if (cur == 0) {
      uint32_t old = atomicCAS(&g_key[h], 0u, slot + 1);
      cur = old == 0 ? slot + 1 : old;
    }
    if (cur == slot + 1) {
      atomicAdd(&g_cnt[h], count);
      break;
    }
    h = (h + 1) & mask;
  }
}
__device__ __noinline__ void ks_spill_migrate(uint32_t *t_key, uint32_t *t_cnt, uint32_t tcap, uint32_t *g_key, uint32_t *g_cnt,
                                              uint32_t mask, bool wipe, uint32_t lane) {
  if (wipe) {  // first use by this wave in this launch: the workspace holds whatever ran before
    for (uint32_t i = lane; i <= mask; i += 64) { g_key[i] = 0; g_cnt[i] = 0; }
    __threadfence();
  }
  ks_wave_sync();
  for (uint32_t i = lane; i < tcap; i += 64) {
    const uint32_t kk = t_key[i];
    if (kk) {
      ks_spill_add(g_key, g_cnt, mask, kk - 1, t_cnt[i] & 0xffffu);
      t_key[i] = 0;
      t_cnt[i] = 0;
    }
  }
  ks_wave_sync();
}
// (the taxonomy arrays come as plain pointers: a reference to the kernel's KuTaxDev argument would pin that struct to
// scratch memory for every user)
__device__ __noinline__ uint32_t ks_spill_resolve(uint32_t *g_key, uint32_t *g_cnt, uint32_t mask, const uint32_t *slot_anc_off,
                                                  const uint32_t *slot_anc, const uint32_t *slot_node, const uint32_t *node_parent,
                                                  uint32_t lane) {
  __threadfence();
  uint32_t my_max = 0;
  for (uint32_t pos = lane; pos <= mask; pos += 64) {
    const uint32_t kk = ks_gload(&g_key[pos]);
    if (!kk) continue;
    uint32_t score = 0;
    for (uint32_t i = slot_anc_off[kk - 1], i_end = slot_anc_off[kk]; i < i_end; ++i) {
      const uint32_t s = slot_anc[i];
      uint32_t h = (s * 2654435761u) & mask;
      for (;;) {
        const uint32_t cur = ks_gload(&g_key[h]);
        if (cur == s + 1) { score += ks_gload(&g_cnt[h]) & 0xffffu; break; }
        if (cur == 0) break;
        h = (h + 1) & mask;
      }
    }
    atomicAdd(&g_cnt[pos], score << 16);
    my_max = max(my_max, score);
  }
  my_max = ku_wave_max_u32(my_max);
  __threadfence();
  uint32_t last = 0, res = 0;
  bool firstt = true;
  for (;;) {
    uint32_t my_min = 0xFFFFFFFFu;
    for (uint32_t pos = lane; pos <= mask; pos += 64) {
      const uint32_t kk = ks_gload(&g_key[pos]);
      if (kk && (ks_gload(&g_cnt[pos]) >> 16) == my_max && kk - 1 > last) my_min = min(my_min, kk - 1);
    }
    my_min = ku_wave_min_u32(my_min);
    if (my_min == 0xFFFFFFFFu) break;
    const uint32_t node = slot_node[my_min];
    res = firstt ? node : ku_lca_nodes(node_parent, res, node);
    firstt = false;
    last = my_min;
  }
  for (uint32_t pos = lane; pos <= mask; pos += 64)
    if (ks_gload(&g_key[pos])) { g_key[pos] = 0; g_cnt[pos] = 0; }
  __threadfence();
  return res;
}

// The fused kernel's inserts into the run-wide set G (sparse fast path): the first probe of all of a lane's items goes out
// together (one compare-and-swap each, back to back: their round trips overlap), the rare collisions are finished one
// by one.  Returns the wave's number of new entries (uniform).
template <int ITEMS>
__device__ __forceinline__ uint32_t ks_g_insert_items(unsigned long long *g_key, uint64_t g_mask, const uint32_t (&slot)[ITEMS],
                                                      const uint64_t (&hash)[ITEMS], const bool (&want)[ITEMS], uint32_t *err) {
  unsigned long long key[ITEMS], old[ITEMS];
  uint64_t h[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    key[j] = ((unsigned long long)(slot[j] + 1) << 32) | ks_encode(hash[j]);
    h[j] = ks_mix(key[j]) & g_mask;
  }
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) old[j] = want[j] ? atomicCAS(&g_key[h[j]], 0ull, key[j]) : key[j];
  uint32_t fresh_n = 0;
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    bool fresh = want[j] && old[j] == 0ull;
    if (want[j] && old[j] != 0ull && old[j] != key[j]) {  // the cell holds another entry: go on from the next one
      bool placed = false;
      for (uint32_t probe = 1; probe < 4096 && !placed; ++probe) {
        h[j] = (h[j] + 1) & g_mask;
        const unsigned long long o = atomicCAS(&g_key[h[j]], 0ull, key[j]);
        fresh = o == 0ull;
        placed = o == 0ull || o == key[j];
      }
      if (!placed) atomicOr(err, 4u);
    }
    fresh_n += (uint32_t)__popcll(__ballot(fresh));
  }
  return fresh_n;
}

// waves per SIMD the variant is compiled for: 2 k-mers per lane fit 80 VGPRs (6 waves, 8 B of scratch; LDS allows 6
// blocks per CU), 3 per lane need 128 (4 waves)
#ifndef KS_OCC2
#define KS_OCC2 6
#endif
#define KS_OCC(ITEMS) ((ITEMS) == 2 ? KS_OCC2 : 4)
// KK / MM: compile-time k-mer and minimizer lengths of the common database geometries (0 = read them from `db`):
// every shift count and the window length become literals instead of loop-invariant scalars that the register
// allocator has to park in VGPR lanes and read back in the read loop.
// WIN: reads of any length up to 65535 k-mers, taken in windows of 64 * ITEMS k-mer positions -- a window is a read of
// its own to stages 1-4 (the minimizer windows of a k-mer lie inside the k-mer, so nothing crosses a window's last
// base); the hit counts accumulate over the windows (one taxon so far: two scalars; else the wave's LDS table; more
// distinct taxa than that holds: the wave's region of `spill`), the call is resolved behind the last window.  A window
// does not start on k-mers already known to be ambiguous: mate pairs joined by 'N' (2 x 150: k-mers 0-119 and
// 151-270) take two windows, not three.
// OUT: what becomes of the per-k-mer codes -- 0: the array parallel to the read buffer (taxa[]); 1: run-length encoded
// {code, start} pairs, each read's runs contiguous in a run array the waves claim in chunks (KuRunsOut; everything
// hitlist_string prints, classify.cpp:826-861, at ~20 B instead of 604 B per 150 bp read -- and no second kernel reads
// the codes back); 2: as 1, plus the sparse-mode emulation's fast path (KuSparseFast).
// ROUTE: the resolve stage of the owner-routed multi-GPU path (ku_route.hip, ku_mgpu.cpp): stages 1-4 and the k-mer
// accounting ran elsewhere (the scan on this rank, probe + HLL + n_kmers on the owners); taxa[] holds the TICKET of every
// k-mer position, the slot is returned[kb[ticket >> 5] + (ticket & 31)] (KuRouteIn).  Everything behind the probe -- hit
// counts, resolve_tree, call, n_reads (DO_COUNTS), the per-k-mer taxids in place of the tickets or as runs -- is the
// code below, unchanged: the gather of the slots and the resolve are one pass over the per-k-mer array.
template <int ITEMS, bool DO_COUNTS, int KK, int MM, bool WIN, int OUT, bool ROUTE = false>
__global__ __launch_bounds__(64 * KS_WAVES, KS_OCC(ITEMS)) void ku_classify_short_kernel(
    KuDbDev db, KuTaxDev tax, KuCountsDev cnt, const uint8_t *__restrict__ seqs, uint64_t n_bytes,
    const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len, uint64_t n_reads,
    uint32_t *__restrict__ calls, uint32_t *__restrict__ taxa, uint32_t ablate, uint32_t *__restrict__ spill,
    uint32_t spill_cap, KuRunsOut ro, KuSparseFast sf, KuRouteIn rin) {
  // KS_ABL(bit): measurement knob, compiled in only with -DKU_ABLATION (then env KU_ABLATE selects the bits; the
  // production build has no trace of it -- the flag checks cost scalar registers and branches in the read loop):
  // 1 skip probe, 2 skip HLL, 4 skip n_kmers, 8 skip taxa store,
  // 32 skip resolve (call 0), 64 skip locus/minimizer stage (bucket 0), 128 (OUT = 2) skip the SEEN marks, 256 (OUT = 2) skip
  // the misses' inserts into the run-wide set, 512 (OUT = 2) skip the insert counts per (unit, slot).  0 in production.
  using G = KsGeom<ITEMS>;
  constexpr uint32_t TCAP = 1u << G::TCAP_LOG2;
  __shared__ uint32_t s_codes[KS_WAVES][G::NCODES];
  __shared__ uint32_t s_amb[KS_WAVES][G::NAMB];
  __shared__ uint32_t s_mm[KS_WAVES][G::NMM];
  __shared__ uint32_t s_tkey[KS_WAVES][TCAP];   // resolve table: slot + 1
  __shared__ uint32_t s_tcnt[KS_WAVES][TCAP];   // hit count (low 16) | root-path score (high 16)
  __shared__ uint16_t s_tlist[KS_WAVES][WIN ? (int)TCAP : G::MAXN];
  __shared__ uint32_t s_kk[KS_WAVES][1 << G::KCT_LOG2], s_kc[KS_WAVES][1 << G::KCT_LOG2];
  __shared__ uint32_t s_rk[KS_WAVES][1 << G::RCT_LOG2], s_rc[KS_WAVES][1 << G::RCT_LOG2];
  __shared__ uint32_t s_misc[KS_WAVES][4];      // [0] n_kmers table fill, [1] n_reads table fill, [2] list length, [3] WIN: distinct taxa in the table

  const uint32_t lane = threadIdx.x & 63u;
  // the wave index is uniform: telling the compiler so moves the per-read metadata (r, len, off, n) and the wave's LDS
  // base addresses to scalar registers and scalar loads
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t *codes = s_codes[wv], *amb = s_amb[wv], *mmv = s_mm[wv];
  uint32_t *t_key = s_tkey[wv], *t_cnt = s_tcnt[wv];
  uint16_t *t_list = s_tlist[wv];
  uint16_t *amb16 = reinterpret_cast<uint16_t *>(amb);
  uint32_t *misc = s_misc[wv];
  const uint32_t k = KK ? (uint32_t)KK : db.k, m = KK ? (uint32_t)MM : db.nt, w = k - m + 1;
  const uint32_t key_shift = ku_key_shift(m);
  const uint32_t *tab = reinterpret_cast<const uint32_t *>(db.table);

  for (uint32_t i = lane; i < TCAP; i += 64) { t_key[i] = 0; t_cnt[i] = 0; }
  if (!ROUTE)  // (ROUTE: the k-mers are booked by their owners)
    for (uint32_t i = lane; i < (1u << G::KCT_LOG2); i += 64) { s_kk[wv][i] = 0; s_kc[wv][i] = 0; }
  for (uint32_t i = lane; i < (1u << G::RCT_LOG2); i += 64) { s_rk[wv][i] = 0; s_rc[wv][i] = 0; }
  if (lane < 4) misc[lane] = 0;
  ks_wave_sync();

  const uint64_t n_waves = (uint64_t)gridDim.x * KS_WAVES;
#ifdef KS_STAGGER
  // experiment of round 5 (scripts/build_variant.sh stg -DKS_STAGGER; scripts/rle_depth_probe.py): a launch of the executable's
  // batch size (61 k reads = 5 per wave) costs 3.6 us per thousand reads against 1.9 for the 10 M-read launch of the bench.
  // Were the waves of a SIMD held back by marching in step (all in the same stage, waiting for memory together), starting them
  // (ablate >> 16) * 64 cycles apart would help: measured 4 .. 256 steps, no change (110 ms per 30 M reads every time).  What is
  // left as an explanation is what a wave pays ONCE -- instruction cache and TLB misses of a kernel of this size on its first
  // read, the counter tables' flush -- spread over 5 reads instead of 800.
  {
    const uint32_t steps = ablate >> 16;
    const uint32_t slot = ((uint32_t)blockIdx.x / 256u) % 6u;
    for (uint32_t i = 0; i < slot * steps; ++i) __builtin_amdgcn_s_sleep(1);
  }
#endif
  // WIN: this wave's region of the spill workspace (keys, then values), wiped on first use
  uint32_t *g_key = WIN ? spill + ((uint64_t)blockIdx.x * KS_WAVES + wv) * 2ull * spill_cap : nullptr;
  uint32_t *g_cnt = WIN ? g_key + spill_cap : nullptr;
  bool spill_used = false;
  // OUT >= 1: the part of the run array this wave claimed and has not filled yet (wave-uniform); its first chunk may be its own
  // from the start (KuRunsOut::pre_base1)
  unsigned long long ch_pos = 0, ch_end = 0;
  if (OUT != 0 && ro.pre_base1) {
    ch_pos = ((unsigned long long)(ro.pre_base1 - 1u) + (unsigned long long)blockIdx.x * KS_WAVES + wv) * ro.chunk;
    ch_end = ch_pos + ro.chunk;
  }
  // make room for `need` more runs: a new chunk when the current one is too short.  `keep` runs of the read in progress
  // (WIN: a read's runs arrive window by window and must stay contiguous) move along, from `keep_base`.
  auto runs_room = [&](uint32_t need, uint32_t keep, unsigned long long &keep_base) {
    if (ch_pos + need <= ch_end) return;
    const uint32_t want_min = keep ? 2u * (keep + need) : need;  // a growing read doubles its room: O(runs) copies in total
    const uint32_t want = want_min > ro.chunk ? want_min : ro.chunk;
    unsigned long long b = 0;
    if (lane == 0) b = atomicAdd(ro.counter, (unsigned long long)want);
    b = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    if (keep) {
      __threadfence();  // the wave's own earlier stores
      for (uint32_t i = lane; i < keep; i += 64) {
        if (keep_base + i < ro.cap && b + i < ro.cap) {
          const uint32_t *src = reinterpret_cast<const uint32_t *>(ro.runs + keep_base + i);
          ro.runs[b + i] = make_uint2(ks_gload(src), ks_gload(src + 1));
        }
      }
      keep_base = b;
    }
    ch_pos = b + keep;
    ch_end = b + want;
  };
  uint32_t sp_fresh = 0;  // OUT == 2: entries this wave added to G
#ifdef KS_PREFETCH
  // software pipeline over the wave's reads: the text of the NEXT read is requested before the current one is worked
  // on (its latency hides behind a whole read's worth of work), its length / offset one read earlier still
  constexpr bool ONE_PASS = (G::NWORDS * 4 + 63) / 64 == 1;
  uint64_t r_first = (uint64_t)blockIdx.x * KS_WAVES + wv;
  uint32_t len_n = r_first < n_reads ? seq_len[r_first] : 0;
  uint64_t off_n = r_first < n_reads ? seq_off[r_first] : 0;
  uint32_t pre0 = 0, pre1 = 0;
  bool pre_ok = false;
  auto prefetch = [&](uint32_t plen, uint64_t poff) {
    const uint32_t b0 = 4 * lane;
    const uint64_t a = poff + b0, a0 = a & ~3ull;
    pre_ok = ONE_PASS && b0 < plen && a0 + 8 <= n_bytes;
    if (pre_ok) {
      const uint32_t *q = reinterpret_cast<const uint32_t *>(seqs + a0);
      pre0 = q[0];
      pre1 = q[1];
    }
  };
  if (r_first < n_reads) prefetch(len_n, off_n);
  uint32_t len_nn = r_first + n_waves < n_reads ? seq_len[r_first + n_waves] : 0;
  uint64_t off_nn = r_first + n_waves < n_reads ? seq_off[r_first + n_waves] : 0;
#endif
#ifndef KS_PREFETCH
  uint32_t len_n = 0;
  uint64_t off_n = 0;
  {
    const uint64_t r0 = (uint64_t)blockIdx.x * KS_WAVES + wv;
    if (r0 < n_reads) { len_n = seq_len[r0]; off_n = seq_off[r0]; }
  }
#endif
  for (uint64_t r = (uint64_t)blockIdx.x * KS_WAVES + wv; r < n_reads; r += n_waves) {
#ifdef KS_PREFETCH
    const uint32_t rlen = len_n;
    const uint64_t roff = off_n;
    const uint32_t cur0 = pre0, cur1 = pre1;
    const bool cur_ok = pre_ok;
    len_n = len_nn;
    off_n = off_nn;
    if (r + n_waves < n_reads) prefetch(len_n, off_n); else pre_ok = false;
    if (r + 2 * n_waves < n_reads) { len_nn = seq_len[r + 2 * n_waves]; off_nn = seq_off[r + 2 * n_waves]; }
#else
    // length and offset were requested one read ahead (two scalar loads off the chain of dependent round trips)
    const uint32_t rlen = len_n;
    const uint64_t roff = off_n;
    {
      const uint64_t rn = r + n_waves;
      len_n = rn < n_reads ? seq_len[rn] : 0u;
      off_n = rn < n_reads ? seq_off[rn] : 0ull;
    }
#endif
    // WIN: the read's hit counts so far -- one taxon (rd_first, rd_cnt) until a second one shows up, then the table
    const uint32_t n_all = rlen >= k ? rlen - k + 1 : 0;
    uint32_t rd_first = 0, rd_cnt = 0, win0 = 0;
    bool rd_table = false, rd_spill = false;
    uint32_t call_node = 0;
    // OUT >= 1: where the read's runs start, how many there are so far, and the code of the k-mer in front of the next
    // window (WIN)
    unsigned long long run_base = 0;
    uint32_t run_n = 0, run_prev = 0;
    bool run_has_prev = false;
    do {  // one pass per window (exactly one without WIN)
    const uint32_t len = WIN ? min(rlen - win0, (uint32_t)G::MAXN + k - 1) : rlen;
    const uint64_t off = roff + win0;
    const uint32_t n = len >= k ? len - k + 1 : 0;
    if (DO_COUNTS) {
      if (!ROUTE && misc[0] > (1u << G::KCT_LOG2) / 2) ks_ct_flush<G::KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], cnt.n_kmers, lane);
      if (!WIN || win0 == 0)
        if (misc[1] > (1u << G::RCT_LOG2) / 2) ks_rct_flush<G::RCT_LOG2>(s_rk[wv], s_rc[wv], &misc[1], tax, cnt.n_reads, lane);
    }
    uint32_t v[ITEMS];  // slot of every k-mer (0 = miss or ambiguous)
    bool amb_k[ITEMS];  // ambiguous k-mer (reported as KU_AMBIG)
    uint64_t hh[ITEMS]; // fmix64(canonical k-mer): bucket tag and HLL index / rank
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { v[j] = 0; amb_k[j] = false; hh[j] = 0; }

    if (ROUTE) {
      // the slots of this window's k-mers through their tickets: ticket (coalesced), record base, slot -- the loads of all
      // items go out together at every level
      uint32_t tk[ITEMS], kbv[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const uint32_t p = j * 64 + lane;
        tk[j] = p < n ? taxa[off + p] : KU_ROUTE_MISS;
      }
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        amb_k[j] = tk[j] == KU_AMBIG;
        kbv[j] = tk[j] < KU_ROUTE_MISS ? rin.kb[tk[j] >> 5] : 0u;
      }
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) v[j] = tk[j] < KU_ROUTE_MISS ? rin.ret[(uint64_t)kbv[j] + (tk[j] & 31u)] : 0u;
    } else if (n > 0) {
      // ---- stage 1: ASCII -> 2-bit codes + ambiguity bits in wave-private LDS: four bases per lane (one dword,
      // SWAR), the four lanes of a quad OR their bytes into one 16-base word
#pragma unroll
      for (uint32_t pl = lane; pl < (uint32_t)((G::NWORDS * 4 + 63) / 64) * 64; pl += 64) {
        const uint32_t b0 = 4 * pl;  // first base of this lane
        uint32_t c8 = 0, a4 = 0xFu;
        if (b0 < len) {
          const uint64_t a = off + b0;
          const uint64_t a0 = a & ~3ull;
          uint32_t d;
#ifdef KS_PREFETCH
          if (cur_ok && pl == lane && win0 == 0) {
            d = __builtin_amdgcn_alignbyte(cur1, cur0, (uint32_t)(a & 3ull));
          } else
#endif
          if (a0 + 8 <= n_bytes) {  // two aligned dwords cover the four bytes at any alignment
            const uint32_t *q = reinterpret_cast<const uint32_t *>(seqs + a0);
            d = __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(a & 3ull));
          } else {
            d = 0;
            for (uint32_t j = 0; j < 4; ++j) d |= (a + j < n_bytes ? (uint32_t)seqs[a + j] : (uint32_t)'N') << (8 * j);
          }
          ku_pack_dword(d, c8, a4);
          const uint32_t valid = len - b0;  // bases of this dword that belong to the read
          if (valid < 4) a4 |= (1u << (4 - valid)) - 1u;
        }
        const uint32_t q4 = pl & 3u;
        uint32_t word = c8 << (24 - 8 * q4), ab = a4 << (12 - 4 * q4);
        word |= ku_quad_xor1(word);
        ab |= ku_quad_xor1(ab);
        word |= ku_quad_xor2(word);
        ab |= ku_quad_xor2(ab);
        const uint32_t wi = pl >> 2;
        if (q4 == 0 && wi < (uint32_t)G::NWORDS) {
          codes[wi] = word;
          amb16[wi ^ 1u] = (uint16_t)ab;
        }
      }
      ks_wave_sync();

      // ---- stage 2: k-mers, ambiguity, canonical form; one packed window element (key, offset 0, strand bit) per
      // m-mer position in the wave's LDS array.  P = n + w - 1 positions carry an m-mer that belongs to a k-mer of this
      // read; the KS_PAD elements behind them are unique sentinels (the doubling steps below never read stale data)
      // and lanes whose position lies behind P compute on junk and store to a dump slot: no branch in the hot path.
      const uint32_t P = n + w - 1;
      uint64_t canon[ITEMS];
      bool is_fwd[ITEMS], ok[ITEMS];
      uint32_t pk[ITEMS + 1], wr[ITEMS + 1];
      uint32_t tacc = 0xFFFFFFFFu;  // min over the steps of (own ^ neighbour) - 2: < 62 <=> equal keys at two positions
      bool tie = false;             // ... somewhere in the read -> exact scan below
      if (lane < KS_PAD) mmv[P + lane] = (0x03FFFFFFu - lane) << KU_PK_KEYSHIFT;
      // The k-mers take ITEMS rounds of 64 lanes, the m-mer positions w - 1 more than that: a round of their own when they do not
      // fit ITEMS rounds (a 150 bp read: 138 positions) -- and none when they do (reads up to 140 bp at ITEMS = 2, up to 204 bp at
      // ITEMS = 3: round 6; measured on 100 bp reads: 14.4 -> 14.1 ms per 10 M).
      const bool tail_round = P > (uint32_t)ITEMS * 64u;  // (wave-uniform)
      pk[ITEMS] = 0;
      wr[ITEMS] = (uint32_t)G::NMM - 1u;
#pragma unroll
      for (int j = 0; j <= ITEMS; ++j) {
        if (j == ITEMS && !tail_round) break;
        const uint32_t p = j * 64 + lane;
        wr[j] = p < P ? p : (uint32_t)G::NMM - 1u;
        const uint32_t wi = p >> 4, sh = (p & 15u) * 2;
        const uint32_t c0 = codes[wi], c1 = codes[wi + 1], c2 = codes[wi + 2];
        // 32 bases from position p: two 64-bit shifts, no special case for sh = 0
        const uint64_t x = ((((uint64_t)c0 << 32) | c1) << sh >> 32 << 32) | ((((uint64_t)c1 << 32) | c2) << sh >> 32);
        const uint32_t mm = (uint32_t)(x >> (64 - 2 * m));
        const uint32_t mrc = ku_revcomp32(mm, m);
        pk[j] = ku_pk_make((mm < mrc ? mm : mrc) ^ db.xor_mask, key_shift, mm <= mrc);
        if ((m & 1u) == 0) tie |= p < P && mm == mrc;  // palindromic m-mer: its strand bit is not enough
        mmv[wr[j]] = pk[j];
        if (j < ITEMS) {
          const uint64_t fwd = x >> (64 - 2 * k);
          const uint64_t rc = ku_revcomp64(fwd, k);
          is_fwd[j] = fwd <= rc;
          canon[j] = is_fwd[j] ? fwd : rc;
          const uint32_t ai = p >> 5, as = p & 31u;
          const uint64_t a = (((uint64_t)amb[ai] << 32) | amb[ai + 1]) << as;
          amb_k[j] = p < n && (a >> (64 - k)) != 0;
          ok[j] = p < n && !amb_k[j];
        }
      }
      ks_wave_sync();

      // ---- stage 3: anchor = sliding-window minimum of the packed elements (ku_device.h): log2(w) doubling steps
      // through the wave's own LDS array (block minima of 2, 4, 8, 16 positions), then one overlapping step for the
      // window length itself -- five dword minima per k-mer where a scan needs w
      uint32_t key[ITEMS], aoff[ITEMS];
      bool plus[ITEMS];
      if (KS_ABL(64u)) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { key[j] = 0; aoff[j] = 0; plus[j] = true; }
      } else {
        uint32_t blk = 1;
        uint32_t tj[ITEMS + 1];
#pragma unroll
        for (int j = 0; j <= ITEMS; ++j) tj[j] = 0xFFFFFFFFu;
        // the doubling steps over NR rounds of positions (ITEMS + 1 with the tail round, else ITEMS: the positions behind the
        // last round that takes part are sentinels nobody updates)
        auto doubling = [&](auto nr_c) {
          constexpr int NR = decltype(nr_c)::value;
#pragma unroll
          for (uint32_t sft = 0; sft < 5; ++sft) {
            const uint32_t st = 1u << sft;
            if (2 * st > w) break;
            uint32_t nb[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) nb[j] = mmv[j * 64 + lane + st] + (st << 1);
            ks_wave_sync();  // every read of this step before any write (in-place update)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
              tj[j] = min(tj[j], (pk[j] ^ nb[j]) - 2u);
              pk[j] = min(pk[j], nb[j]);
              mmv[wr[j]] = pk[j];
            }
            ks_wave_sync();
            blk = 2 * st;
          }
        };
        if (tail_round) doubling(std::integral_constant<int, ITEMS + 1>{});
        else doubling(std::integral_constant<int, ITEMS>{});
        if (w > blk) {
#pragma unroll
          for (int j = 0; j < ITEMS; ++j) {
            const uint32_t nbj = mmv[j * 64 + lane + (w - blk)] + ((w - blk) << 1);
            tj[j] = min(tj[j], (pk[j] ^ nbj) - 2u);  // the two blocks overlap: the same element in both is no tie
            pk[j] = min(pk[j], nbj);
          }
        }
#pragma unroll
        for (int j = 0; j <= ITEMS; ++j) tacc = min(tacc, j * 64 + lane < P ? tj[j] : 0xFFFFFFFFu);
        tie |= tacc < (1u << KU_PK_KEYSHIFT) - 2u;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const uint32_t t = (pk[j] >> 1) & 31u;  // read-order offset of the first minimal key
          key[j] = pk[j] >> KU_PK_KEYSHIFT;
          aoff[j] = is_fwd[j] ? t : w - 1 - t;
          plus[j] = ((pk[j] & 1u) != 0) == is_fwd[j];
        }
        if (__any(tie)) {
          // rare (low-complexity sequence): the raw values go back into the array and every lane scans its window in
          // the canonical k-mer's frame
          ks_wave_sync();
#pragma unroll
          for (int j = 0; j <= ITEMS; ++j) {
            const uint32_t p = j * 64 + lane;
            if (p < P) {
              const uint32_t wi = p >> 4, sh = (p & 15u) * 2;
              const uint64_t two = ((uint64_t)codes[wi] << 32) | codes[wi + 1];
              const uint32_t mm = (uint32_t)((two << sh) >> (64 - 2 * m));
              const uint32_t mrc = ku_revcomp32(mm, m);
              mmv[p] = (mm < mrc ? mm : mrc) ^ db.xor_mask;
            }
          }
          ks_wave_sync();
#pragma unroll
          for (int j = 0; j < ITEMS; ++j) {
            const uint32_t p = j * 64 + lane;
            if (ok[j]) {
              uint32_t bin;
              key[j] = ku_anchor_exact(mmv + p, w, key_shift, is_fwd[j], aoff[j], bin);
              const uint32_t q = p + (is_fwd[j] ? aoff[j] : w - 1 - aoff[j]);
              const uint32_t wi = q >> 4, sh = (q & 15u) * 2;
              const uint64_t two = ((uint64_t)codes[wi] << 32) | codes[wi + 1];
              const uint32_t mmf = (uint32_t)((two << sh) >> (64 - 2 * m));
              const uint32_t rcm = ku_revcomp32(mmf, m);
              plus[j] = is_fwd[j] ? (mmf <= rcm) : (rcm <= mmf);
            }
          }
        }
      }
      const uint32_t *lp[ITEMS];
      uint32_t tag[ITEMS], cand[ITEMS];
      bool act[ITEMS], ovf[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const uint64_t locus = ku_locus_assemble(canon[j], key[j], aoff[j], plus[j], k, m);
        hh[j] = ku_fmix64(canon[j]);
        const uint64_t line = ku_locus_line(locus, db.n_lines);
        lp[j] = tab + (ok[j] ? line : 0) * KU_LINE_DWORDS;  // idle lanes share bucket 0 (no stray line fetches)
        tag[j] = ku_table_tag(hh[j]);
        act[j] = ok[j] && !KS_ABL(1u);
        if (KS_ABL(64u)) lp[j] = tab;
      }

      // ---- stage 4: bucket probe (header round trip, entry in the same line, lockstep tail)
      // header loads and the first-candidate entry load are issued for every lane (inactive lanes point at
      // bucket 0): the instructions are issued per wave anyway, predicating them only costs exec-mask juggling
      uint4 h4[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) h4[j] = *reinterpret_cast<const uint4 *>(lp[j]);
      KuPair pr[ITEMS];
      const uint32_t *ep[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {  // no branch between the header wait and the entry loads of all items
        cand[j] = ku_tag_matches(h4[j], tag[j]) & (act[j] ? 0xFFu : 0u);
        ovf[j] = act[j] && ku_line_spilled(h4[j]);
        ep[j] = lp[j] + KU_LINE_ENTRY0 + 3 * (__builtin_ctz(cand[j] | 0x100u) & 7u);  // no candidate: entry 0 (bit 8 -> 0)
      }
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) pr[j] = *reinterpret_cast<const KuPair *>(ep[j]);
      // OUT == 2 (sparse-sketch emulation): the line's SEEN bytes, one per entry (ku_device.h), come with the entry -- same
      // line, same round trip.  A k-mer the database holds is "inserted into its taxon's sparse sketch" by setting the byte of
      // its entry: a plain byte store, and only the first time (a stale 0 from this CU's L1 costs a second store, never a
      // lost one).  The report reads the marks back (ku_report.hip); the run-wide set only takes the misses.
      uint32_t sb[ITEMS];
      if (OUT == 2 && DO_COUNTS) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
          sb[j] = reinterpret_cast<const uint8_t *>(lp[j] + KU_LINE_SEEN0)[__builtin_ctz(cand[j] | 0x100u) & 7u];
      }
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const bool hit = cand[j] != 0 && (((uint64_t)pr[j].key_hi << 32) | pr[j].key_lo) == canon[j];
        v[j] = hit ? pr[j].slot : 0u;
        if (OUT == 2 && DO_COUNTS) {
          if (hit && !sb[j] && !KS_ABL(128u)) ku_seen_mark(lp[j], (uint32_t)__builtin_ctz(cand[j]));
        }
        cand[j] &= cand[j] - 1;  // no-op for 0
        act[j] = !hit && (cand[j] != 0 || ovf[j]);
      }
      bool any = false;
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) any |= act[j];
      while (any) {
        any = false;
        KuPair e[ITEMS];
        uint4 a4[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          if (act[j]) {
            if (cand[j]) {
              e[j] = *reinterpret_cast<const KuPair *>(lp[j] + KU_LINE_ENTRY0 + 3 * (__builtin_ctz(cand[j])));
            } else {
              lp[j] += KU_LINE_DWORDS;
              if (lp[j] == tab + db.n_lines * KU_LINE_DWORDS) lp[j] = tab;
              a4[j] = *reinterpret_cast<const uint4 *>(lp[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          if (act[j]) {
            if (cand[j]) {
              const uint32_t ei = (uint32_t)__builtin_ctz(cand[j]);
              cand[j] &= cand[j] - 1;
              if ((((uint64_t)e[j].key_hi << 32) | e[j].key_lo) == canon[j]) {
                v[j] = e[j].slot;
                act[j] = false;
                if (OUT == 2 && DO_COUNTS && !KS_ABL(128u)) ku_seen_mark(lp[j], ei);  // (rare path: marked without a look)
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

    }

    // ---- resolve_tree (krakenutil.cpp:149-200) from registers
    bool uni = false;        // at most one distinct hit taxon in the read / window (the common case)
    uint32_t uni_slot = 0;   // that taxon's slot (0 = no hit at all)
    if (!KS_ABL(32u)) {
      uint32_t mine = 0;
#pragma unroll
      for (int j = 0; j < ITEMS; ++j)
        if (mine == 0) mine = v[j];
      const unsigned long long bal = __ballot(mine != 0);
      const uint32_t first = bal ? ku_wave_bcast(mine, (uint32_t)__ffsll((long long)bal) - 1) : 0u;
      bool diff = false;
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) diff |= (v[j] != 0 && v[j] != first);
      uni = !__any(diff);
      uni_slot = uni ? first : 0u;
      if (!WIN) {
        // at most one distinct hit taxon: the call is that taxon (its taxid comes from the slot table below, the read
        // counter is keyed by the slot); else hit_counts in the wave's LDS table
        if (!uni) {
#pragma unroll
          for (int j = 0; j < ITEMS; ++j)
            if (v[j] != 0) ks_tab_add<G::TCAP_LOG2, false>(t_key, t_cnt, nullptr, v[j], 1u);
          uint32_t *urow_hits = nullptr;  // OUT = 2: the read's hits are booked per (work unit, slot) from the table
          if (OUT == 2 && DO_COUNTS && !KS_ABL(512u)) urow_hits = sf.u_cnt + (size_t)(sf.unit_of[r] - sf.unit_base) * sf.n_slots;
          call_node = ks_tab_resolve<G::TCAP_LOG2>(t_key, t_cnt, t_list, &misc[2], tax, lane, urow_hits);
        }
      } else if (bal) {
        // the window's hits join the read's: (rd_first, rd_cnt) while a single taxon has been met, the table from the
        // second one on
        uint32_t w_hits = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) w_hits += (uint32_t)__popcll(__ballot(v[j] != 0));
        if (!rd_table && !(uni && (rd_first == 0 || rd_first == first))) {
          if (rd_first && lane == 0) ks_tab_add<G::TCAP_LOG2, true>(t_key, t_cnt, &misc[3], rd_first, rd_cnt);
          rd_table = true;
        }
        if (!rd_table) {
          rd_first = first;
          rd_cnt += w_hits;
        } else {
          ks_wave_sync();
          // a window adds at most MAXN distinct taxa: beyond this fill the table moves to the wave's spill region
          if (!rd_spill && misc[3] > TCAP - (uint32_t)G::MAXN - 32u) {
            ks_spill_migrate(t_key, t_cnt, TCAP, g_key, g_cnt, spill_cap - 1, !spill_used, lane);
            spill_used = true;
            rd_spill = true;
          }
          if (!rd_spill) {
            if (uni) {
              if (lane == 0) ks_tab_add<G::TCAP_LOG2, true>(t_key, t_cnt, &misc[3], first, w_hits);
            } else {
#pragma unroll
              for (int j = 0; j < ITEMS; ++j)
                if (v[j] != 0) ks_tab_add<G::TCAP_LOG2, true>(t_key, t_cnt, &misc[3], v[j], 1u);
            }
          } else {
            if (uni) {
              if (lane == 0) ks_spill_add(g_key, g_cnt, spill_cap - 1, first, w_hits);
            } else {
#pragma unroll
              for (int j = 0; j < ITEMS; ++j)
                if (v[j] != 0) ks_spill_add(g_key, g_cnt, spill_cap - 1, v[j], 1u);
            }
          }
        }
      }
    }

    // the taxid of the single taxon is requested here: its round trip runs under the HLL loads below
    const uint32_t uni_code = uni && uni_slot ? tax.slot_taxid[uni_slot] : 0u;

    // ---- ReadCounts::add_kmer for every unambiguous k-mer, misses included (classify.cpp:939): HLL register per
    // k-mer; n_kmers per read when the read met one taxon at most (two counter updates instead of one per lane)
    uint32_t n_hit = 0, n_miss = 0;
    if (DO_COUNTS && !ROUTE && n > 0) {
      uint8_t *reg[ITEMS];
      uint32_t rank[ITEMS], seen[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {  // every register byte is requested before the first one is looked at
        const bool okc = j * 64 + lane < n && !amb_k[j];
        reg[j] = ku_hll_locate(cnt.registers, okc ? v[j] : 0u, hh[j], rank[j]);
        seen[j] = (okc && !KS_ABL(2u)) ? (uint32_t)*reg[j] : 0xFFu;
      }
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const bool okc = j * 64 + lane < n && !amb_k[j];
        ku_hll_raise(reg[j], seen[j], rank[j]);
        if (uni) {
          n_hit += (uint32_t)__popcll(__ballot(okc && v[j] != 0));
          n_miss += (uint32_t)__popcll(__ballot(okc && v[j] == 0));
        } else if (okc && !KS_ABL(4u)) {
          ku_ct_add<G::KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], v[j], 1, cnt.n_kmers);
        }
      }
      if (uni && lane == 0 && !KS_ABL(4u)) {
        if (n_hit) ku_ct_add<G::KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], uni_slot, n_hit, cnt.n_kmers);
        if (n_miss) ku_ct_add<G::KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], 0u, n_miss, cnt.n_kmers);
      }
    }

    // ---- sparse-mode emulation, fast path (KuSparseFast): inserts per (work unit, slot), and the encoded hash of every
    // k-mer whose slot is not known to be dense into the run-wide set
    // (round 5: the k-mers the database holds are booked by the SEEN byte of their table entry, stage 4 -- 1.16 G
    // compare-and-swaps on a 16 GB set per 10 M reads became byte stores into lines the probe had fetched anyway; what is left
    // for the set are the misses, under taxon 0, until that sketch turns dense, i.e. for about one work unit)
    if (OUT == 2 && DO_COUNTS && n > 0) {
      uint32_t *urow = sf.u_cnt + (size_t)(sf.unit_of[r] - sf.unit_base) * sf.n_slots;
      bool any_miss;  // uniform
      if (KS_ABL(512u)) {  // (measurement: without the insert counts per (unit, slot))
        any_miss = false;
      } else if (uni) {
        if (lane == 0) {
          if (n_hit) atomicAdd(&urow[uni_slot], n_hit);
          if (n_miss) atomicAdd(&urow[0], n_miss);
        }
        any_miss = n_miss != 0;
      } else if (!WIN && !KS_ABL(32u)) {
        // several taxa, one pass: the hits went into the unit's row with the resolve table (ks_tab_resolve); the misses here
        uint32_t nm = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) nm += (uint32_t)__popcll(__ballot(j * 64 + lane < n && !amb_k[j] && v[j] == 0));
        if (lane == 0 && nm) atomicAdd(&urow[0], nm);
        any_miss = nm != 0;
      } else {
        unsigned long long miss_any = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const bool okc = j * 64 + lane < n && !amb_k[j];
          // inserts per (unit, slot): one add per DISTINCT slot among the item's k-mers.  A read of two or three taxa had every
          // lane add 1 on its own -- 64 lanes queueing on two or three words -- which was a third of this instance's time
          // (98.5 -> 69.3 ms per 10 M reads of `classify -r`, profiles/r04_cli_report_kernel_stats.csv)
          unsigned long long todo = __ballot(okc);
          while (todo) {  // wave-uniform
            const uint32_t lead = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t s0 = ku_wave_bcast(v[j], lead);
            const unsigned long long same = __ballot(okc && v[j] == s0);
            if (lane == lead) atomicAdd(&urow[s0], (uint32_t)__popcll(same));
            todo &= ~same;
          }
          miss_any |= __ballot(okc && v[j] == 0);
        }
        any_miss = miss_any != 0;
      }
      if (any_miss && sf.dense[0] == 0 && !KS_ABL(256u)) {
        bool want[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) want[j] = j * 64 + lane < n && !amb_k[j] && v[j] == 0;
        sp_fresh += ks_g_insert_items<ITEMS>(sf.g_key, sf.g_mask, v, hh, want, sf.err);
      }
    }

    // ---- outputs
    uint32_t tcode[ITEMS];  // taxid per k-mer: slot 0 (miss) is taxid 0; the lookups of all items go out together
    if (uni) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) tcode[j] = v[j] ? uni_code : 0u;
    } else {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) tcode[j] = tax.slot_taxid[v[j]];
    }
    if (OUT == 0) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const uint32_t p = j * 64 + lane;
        if (p < n && !KS_ABL(8u)) taxa[off + p] = amb_k[j] ? KU_AMBIG : tcode[j];
      }
    } else {
      // run starts by neighbour compare (the lane below; lane 0: the last position of the item / window before), their
      // number per item by ballot; the wave's chunk of the run array takes them in position order
      unsigned long long sm[ITEMS];
      uint32_t carry = run_prev, w_runs = 0;
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        tcode[j] = amb_k[j] ? KU_AMBIG : tcode[j];
        uint32_t up = ku_wave_up1(tcode[j]);
        if (lane == 0) up = carry;
        const bool first = j == 0 && lane == 0 && !run_has_prev;
        sm[j] = __ballot(j * 64 + lane < n && (tcode[j] != up || first));
        w_runs += (uint32_t)__popcll(sm[j]);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)tcode[j], 63);
      }
      if (!WIN || run_n == 0) {
        unsigned long long none = 0;
        runs_room(w_runs, 0u, none);
        run_base = ch_pos;
      } else {
        runs_room(w_runs, run_n, run_base);
      }
      uint32_t before = 0;
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const unsigned long long at = ch_pos + before + (uint32_t)__popcll(sm[j] & ((1ull << lane) - 1ull));
        if (((sm[j] >> lane) & 1ull) && at < ro.cap) ro.runs[at] = make_uint2(tcode[j], win0 + j * 64 + lane);
        before += (uint32_t)__popcll(sm[j]);
      }
      ch_pos += w_runs;
      run_n += w_runs;
      run_prev = carry;  // (a window that is not the read's last one ends on lane 63 of the last item)
      run_has_prev = true;
    }
    if (!WIN) {
      if (OUT != 0 && lane == 0) {
        ro.run_off[r] = run_base;
        ro.run_cnt[r] = run_n;
      }
      if (lane == 0) {
        calls[r] = uni ? uni_code : tax.node_taxid[call_node];
        // incrementReadCount (classify.cpp:968): per node; single-taxon reads are booked under their slot and become
        // nodes when the wave's table is flushed
        if (DO_COUNTS) ks_rct_add<G::RCT_LOG2>(s_rk[wv], s_rc[wv], &misc[1], uni ? (KS_RCT_SLOT | uni_slot) : call_node, tax, cnt.n_reads);
      }
      ks_wave_sync();  // the next read reuses the wave's LDS arrays
      break;
    }
    // next window: behind this one, and behind every k-mer that holds the last ambiguous base seen so far (those
    // k-mers are ambiguous whatever follows: they get their code here and no lane of a window)
    uint32_t nxt = win0 + (uint32_t)G::MAXN;
    if (!ROUTE && nxt < n_all) {  // (ROUTE: the ambiguity words of stage 1 do not exist; every position has its ticket)
      uint32_t last1 = 0;  // 1 + window index of the last ambiguous base among the window's `len` bases
      if (lane < (uint32_t)G::NAMB) {
        const uint32_t lo = 32u * lane;
        uint32_t wd = lo < len ? amb[lane] : 0u;
        if (len - lo < 32u && lo < len) wd &= ~0u << (32u - (len - lo));
        if (wd) last1 = lo + 32u - (uint32_t)__builtin_ctz(wd);
      }
      last1 = ku_wave_max_u32(last1);
      if (last1 > (uint32_t)G::MAXN) {  // base MAXN or later: the k-mers at window positions MAXN .. last1 - 1 hold it
        const uint32_t skip_to = min(win0 + last1, n_all);
        if (OUT == 0) {
          if (nxt + lane < skip_to) taxa[roff + nxt + lane] = KU_AMBIG;  // at most k - 1 positions
        } else if (skip_to > nxt && run_prev != KU_AMBIG) {  // the skipped k-mers open a run of their own
          runs_room(1u, run_n, run_base);
          if (lane == 0 && ch_pos < ro.cap) ro.runs[ch_pos] = make_uint2(KU_AMBIG, nxt);
          ch_pos += 1;
          run_n += 1;
          run_prev = KU_AMBIG;
        }
        nxt = skip_to;
      }
    }
    ks_wave_sync();  // the next window reuses the wave's LDS arrays
    win0 = nxt;
    } while (win0 < n_all);
    if (WIN) {
      // behind the last window: the call (a read without k-mers or hits: 0)
      uint32_t code;
      if (!rd_table) code = rd_first ? tax.slot_taxid[rd_first] : 0u;
      else {
        call_node = rd_spill ? ks_spill_resolve(g_key, g_cnt, spill_cap - 1, tax.slot_anc_off, tax.slot_anc, tax.slot_node, tax.node_parent, lane)
                             : ks_tab_resolve<G::TCAP_LOG2>(t_key, t_cnt, t_list, &misc[2], tax, lane);
        if (lane == 0) misc[3] = 0;
        code = tax.node_taxid[call_node];
      }
      if (lane == 0) {
        calls[r] = code;
        if (OUT != 0) {
          ro.run_off[r] = run_base;
          ro.run_cnt[r] = run_n;
        }
        if (DO_COUNTS) ks_rct_add<G::RCT_LOG2>(s_rk[wv], s_rc[wv], &misc[1], rd_table ? call_node : (KS_RCT_SLOT | rd_first), tax, cnt.n_reads);
      }
      ks_wave_sync();
    }
  }
#ifndef KS_DIAG_NOFLUSH  // (diagnostic builds: what do the waves' last flushes cost a small launch?  scripts/launch_shape_probe.py)
  if (DO_COUNTS) {
    // the counters every wave holds -- misses (slot 0), reads without a hit (node 0) -- leave the block as ONE add each
    __shared__ uint32_t s_hot[KS_WAVES][2];
    if (lane < 2) s_hot[wv][lane] = 0;
    if (!ROUTE) ks_ct_flush<G::KCT_LOG2>(s_kk[wv], s_kc[wv], &misc[0], cnt.n_kmers, lane, &s_hot[wv][0]);
    ks_rct_flush<G::RCT_LOG2>(s_rk[wv], s_rc[wv], &misc[1], tax, cnt.n_reads, lane, &s_hot[wv][1]);
    __syncthreads();  // (every wave of the block reaches the end of the kernel: the only block barrier in it)
    if (wv == 0 && lane < 2) {
      unsigned long long sum = 0;
#pragma unroll
      for (int q = 0; q < KS_WAVES; ++q) sum += s_hot[q][lane];
      if (sum) atomicAdd(lane == 0 ? &cnt.n_kmers[0] : &cnt.n_reads[0], sum);
    }
  }
#endif
  if (OUT == 2 && lane == 0 && sp_fresh) atomicAdd(sf.g_count, (unsigned long long)sp_fresh);
}

// k-mers per read the fused kernel can take (0 = not eligible): in one pass / in windows
static uint32_t ks_one_pass_max() { return getenv("KU_SHORT_ONE_PASS_MAX") ? (uint32_t)atoi(getenv("KU_SHORT_ONE_PASS_MAX")) : 192u; }
uint32_t ku_short_max_kmers(const KuDbDev &db) {
  const bool whole = db.bin_lo == 0 && db.bin_hi == (1ull << (2 * db.nt));
  return (db.table && whole) ? ks_one_pass_max() : 0;  // longer reads: the windowed variant, or the flat lookup + resolve kernels
}
uint32_t ku_short_max_kmers_windowed(const KuDbDev &db) {
  return ku_short_max_kmers(db) ? 65535u : 0u;  // 16-bit hit counts and scores per read
}

static unsigned ks_grid(uint64_t n_reads, int items, int n_cu) {
  // persistent grid: two rounds of the blocks a CU holds at once (KS_OCC blocks of KS_WAVES = 4 waves per CU) -- one round for the
  // launches of the `classify` executable's size, where a second round's waves would start up (LDS tables, first chunk) and flush
  // their counters for a handful of reads each (scripts/launch_shape_probe.py, launches of 120 k reads: 30.3 -> 24.9 ms per 10 M reads
  // on one stream; from 1 M reads per launch on two rounds are the better: 19.95 against 20.32)
  const char *oe = getenv("KU_SHORT_BLOCKS_PER_CU");
  const uint64_t per_cu = oe ? (uint64_t)atoi(oe) : (n_reads < 500000 ? 1ull : 2ull) * KS_OCC(items);
  const uint64_t want = (n_reads + KS_WAVES - 1) / KS_WAVES, cap = (uint64_t)n_cu * per_cu;
  return (unsigned)(want < cap ? want : cap);
}
// entries of a wave's spill table: twice the distinct taxa a read can meet
static uint32_t ks_spill_cap(uint32_t max_kmers, uint32_t n_slots) {
  const uint32_t distinct = max_kmers < n_slots ? max_kmers : n_slots;
  uint32_t cap = 1024;
  while (cap < 2 * distinct) cap <<= 1;
  return cap;
}
// workspace of the windowed variant (0 when the reads fit one pass)
uint64_t ku_short_workspace_bytes(uint32_t max_kmers, uint32_t n_slots, uint64_t n_reads, int n_cu) {
  if (max_kmers <= ks_one_pass_max()) return 0;
  return (uint64_t)ks_grid(n_reads, 2, n_cu) * KS_WAVES * 2ull * ks_spill_cap(max_kmers, n_slots) * 4ull;
}

uint64_t ku_short_grid_waves(uint64_t n_reads, uint32_t max_kmers, int n_cu) {
  const bool windowed = max_kmers > ks_one_pass_max();
  return (uint64_t)ks_grid(n_reads, max_kmers <= 128 || windowed ? 2 : 3, n_cu) * KS_WAVES;
}

int ku_launch_classify_short(const KuDbDev &db, const KuTaxDev &tax, const KuCountsDev &cnt, const uint8_t *d_seqs,
                             uint64_t n_bytes, const uint64_t *d_seq_off, const uint32_t *d_seq_len, uint64_t n_reads,
                             uint32_t max_kmers, uint32_t flags, uint32_t *d_calls, uint32_t *d_taxa, uint32_t *d_hits,
                             void *d_workspace, uint64_t workspace_bytes, int n_cu, hipStream_t stream,
                             const KuRunsOut *runs_out, const KuSparseFast *sparse) {
  if (n_reads == 0) return KU_OK;
  const bool counts = !(flags & KU_F_NO_COUNTS);
  if (flags & KU_F_KEEP_SLOTS) return KU_EINVAL;  // slot ids are for the sharded path, which does not come here
  if (sparse && (!runs_out || !counts)) return KU_EINVAL;
  if (!runs_out && !d_taxa) return KU_EINVAL;
  if (d_hits && hipMemsetAsync(d_hits, 0, n_reads * 4, stream) != hipSuccess) return KU_EHIP;  // "Q:n" is quick mode only
  const char *ab = getenv("KU_ABLATE");
  const uint32_t ablate = ab ? (uint32_t)atoi(ab) : 0u;
  const bool windowed = max_kmers > ks_one_pass_max();
  const int items = max_kmers <= 128 || windowed ? 2 : 3;
  const dim3 grid(ks_grid(n_reads, items, n_cu)), block(64 * KS_WAVES);
  uint32_t spill_cap = 0;
  if (windowed) {
    if (max_kmers > 65535u) return KU_EINVAL;
    spill_cap = ks_spill_cap(max_kmers, tax.n_slots);
    if (!d_workspace || workspace_bytes < (uint64_t)grid.x * KS_WAVES * 2ull * spill_cap * 4ull) return KU_EINVAL;
  }
  const KuRunsOut ro = runs_out ? *runs_out : KuRunsOut{};
  const KuSparseFast sf = sparse ? *sparse : KuSparseFast{};
  const int out = sparse ? 2 : (runs_out ? 1 : 0);
#define KS_LAUNCH(I, C, K, M, W, O)                                                                                          \
  hipLaunchKernelGGL((ku_classify_short_kernel<I, C, K, M, W, O>), grid, block, 0, stream, db, tax, cnt, d_seqs, n_bytes,    \
                     d_seq_off, d_seq_len, n_reads, d_calls, d_taxa, ablate, (uint32_t *)d_workspace, spill_cap, ro, sf, KuRouteIn{})
  // specialised geometries (accounting runs only): k = 31 with nt = 13 (MiniKraken-size databases) or 15 (standard)
  const int geo = !counts || db.k != 31 ? 0 : (db.nt == 13 ? 13 : (db.nt == 15 ? 15 : 0));
#define KS_GEO(I, W, O)                                      \
  do {                                                       \
    if (!counts) KS_LAUNCH(I, false, 0, 0, W, (O) == 2 ? 1 : (O)); \
    else if (geo == 13) KS_LAUNCH(I, true, 31, 13, W, O);    \
    else if (geo == 15) KS_LAUNCH(I, true, 31, 15, W, O);    \
    else KS_LAUNCH(I, true, 0, 0, W, O);                     \
  } while (0)
#define KS_OUT(I, W)                  \
  do {                                \
    if (out == 0) KS_GEO(I, W, 0);    \
    else if (out == 1) KS_GEO(I, W, 1); \
    else KS_GEO(I, W, 2);             \
  } while (0)
  if (windowed) KS_OUT(2, true);
  else if (max_kmers <= 128) KS_OUT(2, false);
  else KS_OUT(3, false);
#undef KS_OUT
#undef KS_GEO
#undef KS_LAUNCH
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// The resolve stage of the owner-routed path, fused with the gather of the returned slots (ROUTE instances): d_taxa holds
// tickets on entry, taxids (or nothing: runs_out) afterwards.  Reads of up to 128 k-mers in one pass, up to 65535 in windows
// (workspace as for the windowed fused kernel); longer ones: ku_launch_route_gather + the resolve kernel.
uint32_t ku_route_resolve_max_kmers() { return 65535u; }
int ku_launch_route_resolve(const KuDbDev &db, const KuTaxDev &tax, const KuCountsDev &cnt, const uint64_t *d_seq_off,
                            const uint32_t *d_seq_len, uint64_t n_reads, uint32_t max_kmers, uint32_t flags, uint32_t *d_calls,
                            uint32_t *d_taxa, uint32_t *d_hits, const uint32_t *d_kb, const uint32_t *d_ret, void *d_workspace,
                            uint64_t workspace_bytes, int n_cu, hipStream_t stream, const KuRunsOut *runs_out) {
  if (n_reads == 0) return KU_OK;
  if (!d_taxa || !d_kb || !d_seq_off || !d_seq_len || !d_calls || max_kmers > 65535u) return KU_EINVAL;
  if (flags & (KU_F_KEEP_SLOTS | KU_F_QUICK)) return KU_EINVAL;
  if (d_hits && hipMemsetAsync(d_hits, 0, n_reads * 4, stream) != hipSuccess) return KU_EHIP;  // "Q:n" is quick mode only
  const bool counts = !(flags & KU_F_NO_COUNTS), windowed = max_kmers > 128;
  const dim3 grid(ks_grid(n_reads, 2, n_cu)), block(64 * KS_WAVES);
  uint32_t spill_cap = 0;
  if (windowed) {
    spill_cap = ks_spill_cap(max_kmers, tax.n_slots);
    if (!d_workspace || workspace_bytes < (uint64_t)grid.x * KS_WAVES * 2ull * spill_cap * 4ull) return KU_EINVAL;
  }
  const KuRunsOut ro = runs_out ? *runs_out : KuRunsOut{};
  const KuRouteIn rin{d_kb, d_ret};
#define KS_RLAUNCH(C, W, O)                                                                                                     \
  hipLaunchKernelGGL((ku_classify_short_kernel<2, C, 0, 0, W, O, true>), grid, block, 0, stream, db, tax, cnt, (const uint8_t *)nullptr, \
                     (uint64_t)0, d_seq_off, d_seq_len, n_reads, d_calls, d_taxa, 0u, (uint32_t *)d_workspace, spill_cap, ro,    \
                     KuSparseFast{}, rin)
#define KS_RSEL(W, O)                  \
  do {                                 \
    if (counts) KS_RLAUNCH(true, W, O); \
    else KS_RLAUNCH(false, W, O);      \
  } while (0)
  if (windowed) { if (runs_out) KS_RSEL(true, 1); else KS_RSEL(true, 0); }
  else { if (runs_out) KS_RSEL(false, 1); else KS_RSEL(false, 0); }
#undef KS_RSEL
#undef KS_RLAUNCH
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
