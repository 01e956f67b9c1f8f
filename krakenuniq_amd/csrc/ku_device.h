// ku_device.h -- device helpers shared by the gfx950 kernels (ku_kernels.hip, ku_short.hip): hashes, reverse
// complement, LDS counter tables, HLL register update, the bucketised probe table and the locus key.
#pragma once
#include "ku_internal.h"

#define KU_CT_LOG2 9                      // per-block LDS counter table (n_kmers / n_reads aggregation)
#define KU_CT_CAP (1 << KU_CT_LOG2)

// ----------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ku_fmix64(uint64_t key) {  // hyperloglogplus.cpp:830-838
  key += 1;
  key ^= key >> 33;
  key *= 0xff51afd7ed558ccdULL;
  key ^= key >> 33;
  key *= 0xc4ceb9fe1a85ec53ULL;
  key ^= key >> 33;
  return key;
}

// reverse complement of the n-mer held in the low 2n bits (krakendb.cpp:218-225):
// full bit reversal (v_bfrev_b32 x2) + swap inside every 2-bit group == reversal
// of the 2-bit groups; complement is bitwise NOT in this encoding.
__device__ __forceinline__ uint64_t ku_revcomp64(uint64_t x, uint32_t n) {
  uint64_t r = __builtin_bitreverse64(x);
  r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
  return (~r) >> (64 - 2 * n);
}
__device__ __forceinline__ uint32_t ku_revcomp32(uint32_t x, uint32_t n) {
  uint32_t r = __builtin_bitreverse32(x);
  r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
  return (~r) >> (32 - 2 * n);
}

// ---- cross-lane helpers on the VALU's data-parallel-primitive path (no LDS crossbar trip as __shfl takes)
// value of the quad neighbour lane ^ 1 / lane ^ 2
__device__ __forceinline__ uint32_t ku_quad_xor1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t ku_quad_xor2(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false); }
// maximum over the wave, the same value in every lane (row shifts, then the two row broadcasts, lane 63 holds it)
__device__ __forceinline__ uint32_t ku_wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t ku_wave_min_u32(uint32_t v) { return ~ku_wave_max_u32(~v); }
// value of the lane below (lane 0 gets 0): wave_shr:1
__device__ __forceinline__ uint32_t ku_wave_up1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, false); }
// value of lane `src` (the same in every lane: e.g. from a ballot) in every lane
__device__ __forceinline__ uint32_t ku_wave_bcast(uint32_t x, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)x, __builtin_amdgcn_readfirstlane((int)src));
}

// LDS-aggregated counters: id -> count, flushed to a global uint64 array.  `used` counts occupied
// entries; callers flush + clear at a block-uniform point once the table is half full
// (ku_ct_maybe_flush), so the 8-probe fallback to a global atomic stays rare whatever the
// number of distinct taxa a block meets.
template <int LOG2 = KU_CT_LOG2>
__device__ __forceinline__ void ku_ct_add(uint32_t *ct_key, uint32_t *ct_cnt, uint32_t *used, uint32_t id, uint32_t n,
                                          unsigned long long *global) {
  uint32_t h = (id * 2654435761u) >> (32 - LOG2);
#pragma unroll 1
  for (int probe = 0; probe < 8; ++probe) {
    uint32_t cur = __hip_atomic_load(&ct_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (cur == 0) {
      uint32_t old = atomicCAS(&ct_key[h], 0u, id + 1);
      if (old == 0) atomicAdd(used, 1u);
      cur = old == 0 ? id + 1 : old;
    }
    if (cur == id + 1) {
      atomicAdd(&ct_cnt[h], n);
      return;
    }
    h = (h + 1) & ((1u << LOG2) - 1);
  }
  atomicAdd(&global[id], (unsigned long long)n);
}
template <int LOG2 = KU_CT_LOG2>
__device__ __forceinline__ void ku_ct_clear(uint32_t *ct_key, uint32_t *ct_cnt, uint32_t *used) {
  for (int i = threadIdx.x; i < (1 << LOG2); i += blockDim.x) {
    ct_key[i] = 0;
    ct_cnt[i] = 0;
  }
  if (threadIdx.x == 0) *used = 0;
}
template <int LOG2 = KU_CT_LOG2>
__device__ __forceinline__ void ku_ct_flush(uint32_t *ct_key, uint32_t *ct_cnt, unsigned long long *global) {
  for (int i = threadIdx.x; i < (1 << LOG2); i += blockDim.x) {
    uint32_t kk = ct_key[i];
    if (kk) atomicAdd(&global[kk - 1], (unsigned long long)ct_cnt[i]);
  }
}
// call at a point every thread of the block reaches, after a __syncthreads()
template <int LOG2 = KU_CT_LOG2>
__device__ __forceinline__ void ku_ct_maybe_flush(uint32_t *ct_key, uint32_t *ct_cnt, uint32_t *used,
                                                  unsigned long long *global) {
  if (*used > (1u << LOG2) / 2) {  // block-uniform (read after the barrier)
    __syncthreads();
    ku_ct_flush<LOG2>(ct_key, ct_cnt, global);
    __syncthreads();
    ku_ct_clear<LOG2>(ct_key, ct_cnt, used);
    __syncthreads();
  }
}

// HLL register update: M[slot][idx] = max(M, rank) (hyperloglogplus.cpp:508-522, p = 12).
// The plain pre-check load may be stale (other CUs' updates are not visible in
// this CU's L1) -- stale values are only ever too small, so the worst case is a
// redundant CAS, never a lost update.
// two steps so that a caller with several k-mers in flight can request all their register bytes before it looks at
// the first one (one memory round trip instead of one per k-mer)
__device__ __forceinline__ uint8_t *ku_hll_locate(uint8_t *registers, uint32_t slot, uint64_t h, uint32_t &rank) {
  // h = ku_fmix64(canonical k-mer)
  const uint32_t idx = (uint32_t)(h >> (64 - KU_HLL_P));
  // (clz of the bits behind the index, + 1; all of them zero: 64 - p + 1 -- bit p - 1, set below the hash's bits, stops the
  // count there without a compare and select)
  const uint64_t rest = (h << KU_HLL_P) | (1ull << (KU_HLL_P - 1));
  rank = (uint32_t)__builtin_clzll(rest) + 1;
  return registers + (size_t)slot * KU_HLL_M + idx;
}
__device__ __forceinline__ void ku_hll_raise(uint8_t *r, uint32_t seen, uint32_t rank) {
  if (seen < rank) {
    uint32_t *w = (uint32_t *)((uintptr_t)r & ~(uintptr_t)3);
    uint32_t sh = ((uint32_t)(uintptr_t)r & 3u) * 8;
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (((old >> sh) & 0xffu) < rank) {
      uint32_t nw = (old & ~(0xffu << sh)) | (rank << sh);
      uint32_t prev = atomicCAS(w, old, nw);
      if (prev == old) break;
      old = prev;
    }
  }
}
__device__ __forceinline__ void ku_hll_update(uint8_t *registers, uint32_t slot, uint64_t h) {
  uint32_t rank;
  uint8_t *r = ku_hll_locate(registers, slot, h, rank);
  ku_hll_raise(r, *r, rank);
}

// ---- HyperLogLog++ sparse representation (ku_sparse.hip, and the fused kernel's sparse fast path)
#define KS_PPRIME 25
// encodeHashIn32Bit (hyperloglogplus.cpp:181-204), p = 12, p' = 25
__device__ __forceinline__ uint32_t ks_encode(uint64_t h) {
  const uint32_t idx = (uint32_t)(h >> (64 - KS_PPRIME)) << (32 - KS_PPRIME);
  if ((uint32_t)(idx << KU_HLL_P) == 0) {
    const uint64_t rest = h << KS_PPRIME;
    const uint32_t add = rest ? (uint32_t)__builtin_clzll(rest) + 1 : (64 - KS_PPRIME + 1);
    return idx | (add << 1) | 1u;
  }
  return idx;
}

__device__ __forceinline__ uint64_t ks_mix(unsigned long long k) {
  k ^= k >> 31;
  k *= 0x9E3779B97F4A7C15ULL;
  k ^= k >> 29;
  return k;
}

// insert (slot, encoding) into the run-wide set G of the sparse-mode emulation: open addressing over 8-byte cells, one
// compare-and-swap per probe (an empty cell and "already there" both end the search).  true: the entry is new.
__device__ __forceinline__ bool ks_g_insert(unsigned long long *g_key, uint64_t g_mask, uint32_t slot, uint32_t enc, uint32_t *err) {
  const unsigned long long gk = ((unsigned long long)(slot + 1) << 32) | enc;
  uint64_t h = ks_mix(gk) & g_mask;
  for (uint32_t probe = 0; probe < 4096; ++probe) {
    const unsigned long long old = atomicCAS(&g_key[h], 0ull, gk);
    if (old == 0ull) return true;
    if (old == gk) return false;
    h = (h + 1) & g_mask;
  }
  atomicOr(err, 4u);
  return false;
}

// ----------------------------------------------------------------------------
// ku_lookup_kernel
// ----------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) KuPair {  // the on-disk record (krakendb.cpp:176)
  uint32_t key_lo, key_hi, slot;
};

// Four ASCII bases in one dword (first base in the low byte) -> 8 bits of 2-bit codes (first base in bits 7..6) and
// 4 ambiguity bits (first base in bit 3): the SWAR form of four ku_pack_byte calls.  The per-byte fields are
// gathered with one multiply each (disjoint partial products, no carries).
__device__ __forceinline__ void ku_pack_dword(uint32_t d, uint32_t &codes8, uint32_t &amb4) {
  const uint32_t c = d & 0xDFDFDFDFu;                              // fold case
  const uint32_t x = ((c >> 1) ^ (c >> 2)) & 0x03030303u;          // A=0 C=1 G=2 T=3 per byte
  codes8 = (x * 0x40100401u) >> 24;
  const uint32_t b0 = x & 0x01010101u, b1 = (x >> 1) & 0x01010101u, b01 = b0 & b1;
  // the letter each code stands for: 'A' + {0, 2, 6, 19}; any other byte differs from it
  const uint32_t expect = 0x41414141u + (b0 << 1) + (b1 << 2) + (b1 << 1) + (b01 << 3) + (b01 << 1) + b01;
  const uint32_t diff = c ^ expect;
  const uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;
  amb4 = (((nz >> 7) * 0x08040201u) >> 24) & 0xFu;
}

// ASCII -> (2-bit code, valid): A/a=0 C/c=1 G/g=2 T/t=3 (krakenutil.cpp:252-263)
__device__ __forceinline__ void ku_pack_byte(uint32_t b, uint32_t j, uint32_t &word, uint32_t &amb) {
  uint32_t c = b & 0xDFu;
  uint32_t code = ((c >> 1) ^ (c >> 2)) & 3u;
  uint32_t valid = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
  word |= code << (30 - 2 * j);
  amb |= (valid ^ 1u) << (15 - j);
}

// Hash layout (LAYOUT 1, default): at upload time the shard's pairs are re-laid out as a bucketised hash table,
// one bucket per 128-byte line:
//     dwords 0..3   eight 16-bit tag fields, the field of entry i in the (i >> 2 ? high : low) half of dword i & 3 (round 6: entries
//                   0..3 in the low halves, 4..7 in the high ones, so that the packed compare below yields the candidate
//                   mask with three shifts): low 15 bits = tag of entry i (1 .. 0x7FFF; 0 = entry unused); bit 15 of entry 0's
//                   field (bit 15 of dword 0) = the bucket received more than eight keys and spilled into the following line(s)
//     dwords 4..27  eight 12-byte entries {key_lo, key_hi, slot};  dwords 28..29 eight SEEN bytes (below);  30..31 unused
// MI355X moves 128 B per L2 miss and sustains ~48 G random line fetches/s whatever the access width
// (scripts/calib_gather.hip), so a lookup costs the number of distinct lines it touches -- and, per wave, the
// number of *dependent* round trips of its slowest lane.  The sorted-bin binary search touches ~2.5 lines in ~8
// dependent probes; here the 16-byte header -- ONE dwordx4 load per k-mer; the nine-entry line of rounds 1-2 had a
// 20-byte header whose fifth dword was a load instruction of its own and cost 4 % of the fused kernel -- answers
// "which entry, if any" in one round trip (15-bit tags, false positive rate 8 * 2^-15), the entry itself is then an
// L1/L2 hit in the same line.  A bucket that received more than 8 keys spills into the following line(s).
#define KU_LINE_DWORDS 32
#define KU_LINE_SLOTS 8
#define KU_LINE_ENTRY0 4  // first entry dword
// dwords 28..29: one SEEN byte per entry (byte 112 + i of the line), 0 when the table is built.  The HyperLogLog++ sparse-mode
// emulation's fast path (ku_short.hip, OUT = 2) sets the byte of every entry a read's k-mer found: the set of (taxon, k-mer)
// pairs the reference keeps in its sparse sketches (hyperloglogplus.cpp:485-523) is then the set of marked entries --
// recorded with a plain byte store into a line the probe had fetched anyway instead of a compare-and-swap on a second, random
// line of a run-wide hash set.  Bytes only ever go 0 -> 1 within a run (monotone: stale reads cost a repeated store, never a
// lost mark); ku_ctx_reset_counts clears them; the report (ku_report.hip) and ku_sparse_export read them back.
#define KU_LINE_SEEN0 28
__device__ __forceinline__ void ku_seen_mark(const uint32_t *line, uint32_t entry) {
  reinterpret_cast<uint8_t *>(const_cast<uint32_t *>(line) + KU_LINE_SEEN0)[entry] = 1;
}
// 15-bit non-zero entry tag from h = fmix64(kmer + 1) (the HLL hash, reused; the HLL consumes bits 63..52 and the
// leading zeros below them, the tag takes bits 42..28)
__device__ __forceinline__ uint32_t ku_table_tag(uint64_t h) {
  const uint32_t t = (uint32_t)(h >> 28) & 0x7FFFu;
  return t ? t : 1u;
}
#ifdef KU_TAG_FIELDWISE  // (A/B builds: round 5's header -- field i in the (i & 1 ? high : low) half of dword i >> 1, compared field by field)
__device__ __forceinline__ uint32_t ku_tag_dword(uint32_t i) { return i >> 1; }
__device__ __forceinline__ uint32_t ku_tag_shift(uint32_t i) { return (i & 1u) * 16u; }
__device__ __forceinline__ uint32_t ku_tag_matches(uint4 h4, uint32_t tag) {
  uint32_t m = 0;
  m |= ((h4.x & 0x7FFFu) == tag) << 0;
  m |= ((h4.x >> 16) == tag) << 1;
  m |= ((h4.y & 0xFFFFu) == tag) << 2;
  m |= ((h4.y >> 16) == tag) << 3;
  m |= ((h4.z & 0xFFFFu) == tag) << 4;
  m |= ((h4.z >> 16) == tag) << 5;
  m |= ((h4.w & 0xFFFFu) == tag) << 6;
  m |= ((h4.w >> 16) == tag) << 7;
  return m;
}
#else
// where the tag field of entry i lies: dword, shift
__device__ __forceinline__ uint32_t ku_tag_dword(uint32_t i) { return i & 3u; }
__device__ __forceinline__ uint32_t ku_tag_shift(uint32_t i) { return (i >> 2) * 16u; }
// both 16-bit halves of a dword at once: min(half, 1) -- 0 where the half is 0, else 1 (v_pk_min_u16; the compiler turns the
// portable form into two compares, two selects and a permute)
__device__ __forceinline__ uint32_t ku_pk_min_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// bit i set <=> the tag field of entry i equals `tag` (unused fields are 0, tags are not).  Packed 16-bit arithmetic on the four
// header dwords: XOR with the tag in both halves, min(half, 1) leaves 0 exactly where a field matched, the four dwords' flags are
// interleaved into bits 0..3 (low halves: entries 0..3) and 16..19 (entries 4..7).  16 vector instructions per header where the
// field-by-field compare took 43 (round 6; the kernel is bound by instruction issue).
__device__ __forceinline__ uint32_t ku_tag_matches(uint4 h4, uint32_t tag) {
  const uint32_t t2 = tag | (tag << 16), one = 0x00010001u;
  const uint32_t e0 = ku_pk_min_u16((h4.x & 0xFFFF7FFFu) ^ t2, one);  // (entry 0's field carries the spill flag in bit 15)
  const uint32_t e1 = ku_pk_min_u16(h4.y ^ t2, one), e2 = ku_pk_min_u16(h4.z ^ t2, one), e3 = ku_pk_min_u16(h4.w ^ t2, one);
  const uint32_t differ = ((((e3 << 1) | e2) << 1 | e1) << 1) | e0;  // bit i / 16 + i: entry i / 4 + i does NOT match
  const uint32_t m = ~differ & 0x000F000Fu;
  return (m | (m >> 12)) & 0xFFu;
}
#endif
__device__ __forceinline__ bool ku_line_spilled(uint4 h4) { return (h4.x & 0x8000u) != 0; }

// Locality-aware bucket choice.  Consecutive k-mers of a read share their minimizer *occurrence* ~(k-nt+1)/2
// times in a row.  The bucket of a k-mer is therefore derived not from the k-mer itself but from its "locus
// key": (minimizer value, the KU_FLANK bases next to the minimizer occurrence on the longer side, a coarse
// offset class) -- all taken on the strand where the minimizer m-mer is canonical, so both strands of a locus
// agree.  The ~5 overlapping k-mers that share a locus key share one 128-byte bucket: a wave's 64 consecutive
// k-mers touch ~20 lines instead of 64.  The key is a pure function of the canonical k-mer (first minimum in
// the canonical k-mer's frame on ties), so build and lookup agree by construction; which k-mers share a bucket
// only affects speed, never results.
#define KU_FLANK 8
#ifndef KU_OFFCLASS
#define KU_OFFCLASS 4  // offsets per locus class: 4 measured best (3: 25.7, 4: 25.3, 5: 26.8, 8: 28.1 ms at load 0.3)
#endif

// ---- the anchor of a k-mer: WHICH m-mer occurrence its bucket is derived from.
// Every m-mer position j of the canonical k-mer (j = offset from its left / most significant end) has the value
// v_j = scrambled canonical m-mer -- the terms KrakenDB::bin_key minimises (krakendb.cpp:200-215); the minimizer bin
// is min_j v_j.  The anchor is the FIRST j (canonical frame) that minimises the 26-bit order key
//     key_j = v_j >> ku_key_shift(m)            (= v_j itself for m <= 13)
// 26 bits because key, window offset (5 bits: w = k - m + 1 <= 31) and the strand bit of the occurrence then fit one
// dword, which makes the sliding-window argmin a log-step min of packed dwords for every minimizer length up to
// 15.  With a unique minimal key the anchor is the minimizer occurrence itself (key_a < key_j  =>  v_a < v_j).  The rule is a
// pure function of the canonical k-mer, so table build and lookup agree by construction.
__device__ __forceinline__ uint32_t ku_key_shift(uint32_t m) { return m > 13 ? 2 * m - 26 : 0; }
// packed window element: key << 6 | offset << 1 | strand bit (1: the read-strand m-mer is <= its reverse complement)
#define KU_PK_KEYSHIFT 6
__device__ __forceinline__ uint32_t ku_pk_make(uint32_t value, uint32_t key_shift, bool fwd_le) {
  return ((value >> key_shift) << KU_PK_KEYSHIFT) | (uint32_t)fwd_le;
}
// One doubling step: `own` covers the positions [p, p + s), `nb` the block [p + s, p + 2s) as stored (offsets relative
// to p + s).  Ties between DIFFERENT positions with equal keys are reported through `tie`: the packed minimum then
// holds the smallest read position, which is the canonical-frame rule only for k-mers whose read strand is the canonical
// one -- callers fall back to the exact scan (ku_anchor_exact) when any lane saw a tie.
__device__ __forceinline__ uint32_t ku_pk_combine(uint32_t own, uint32_t nb, uint32_t s, bool &tie) {
  nb += s << 1;
  tie |= (own ^ nb) < (1u << KU_PK_KEYSHIFT);
  return min(own, nb);
}
// last step of a window that is not a power of two: the two blocks overlap, the same element may win in both
__device__ __forceinline__ uint32_t ku_pk_combine_overlap(uint32_t own, uint32_t nb, uint32_t s, bool &tie) {
  nb += s << 1;
  tie |= ((own ^ nb) - 2u) < (1u << KU_PK_KEYSHIFT) - 2u;  // equal keys, different offsets
  return min(own, nb);
}
// Exact anchor by a sequential scan in the canonical k-mer's frame over the raw values raw[0 .. w) of the k-mer's
// m-mer positions in READ order (is_fwd: the read strand is the canonical one).  Returns the minimal key, sets a =
// its first canonical-frame offset and bin = the minimizer (min raw value).
__device__ __forceinline__ uint32_t ku_anchor_exact(const uint32_t *raw, uint32_t w, uint32_t key_shift, bool is_fwd,
                                                    uint32_t &a, uint32_t &bin) {
  const int32_t j0 = is_fwd ? 0 : (int32_t)w - 1, dj = is_fwd ? 1 : -1;
  uint32_t best = 0xFFFFFFFFu, mn = 0xFFFFFFFFu;
  a = 0;
  for (uint32_t t = 0; t < w; ++t) {
    const uint32_t vv = raw[j0 + dj * (int32_t)t], kk = vv >> key_shift;
    const bool lt = kk < best;
    best = lt ? kk : best;
    a = lt ? t : a;
    mn = min(mn, vv);
  }
  bin = mn;
  return best;
}
// Locus key from the anchor: (key, the KU_FLANK bases next to the anchor occurrence on the longer side, offset class),
// all taken on the strand where the anchor m-mer is canonical (plus: that is the canonical k-mer's own strand).
__device__ __forceinline__ uint64_t ku_locus_assemble(uint64_t canon, uint32_t key, uint32_t a, bool plus, uint32_t k,
                                                      uint32_t m) {
  const uint32_t w = k - m + 1;
  const uint32_t ap = plus ? a : w - 1 - a;  // offset of the occurrence on that strand
  const uint32_t left = ap, right = w - 1 - ap;
  const bool use_r = right >= left;
  const uint32_t side = use_r ? right : left;
  // the longer side is at least (w - 1) / 2 positions long: a constant flank length for the usual windows
  const uint32_t flen = (w - 1) / 2 >= KU_FLANK ? (uint32_t)KU_FLANK : (side < KU_FLANK ? side : KU_FLANK);
  // right flank = offsets [ap+m, ap+m+flen), left flank = [ap-flen, ap); one past the flank's last base:
  const uint32_t end = use_r ? ap + m + flen : ap;
  // bases [end - flen, end) of that strand; on the other strand they are the reverse complement of bases
  // [k - end, k - end + flen) of the canonical k-mer (no second 64-bit copy of the k-mer needed)
  const uint32_t seg = (uint32_t)(canon >> (2 * (plus ? k - end : end - flen))) & ((1u << (2 * flen)) - 1u);
  const uint32_t flank = flen ? (plus ? seg : ku_revcomp32(seg, flen)) : 0u;
  return ((uint64_t)key << 32) | ((uint64_t)flank << 12) | (flen << 8) | ((side / KU_OFFCLASS) << 1) | (uint32_t)use_r;
}
// table build: locus key of a stored canonical k-mer c (+ its minimizer bin)
__device__ __forceinline__ uint64_t ku_locus_key(uint64_t c, uint32_t k, uint32_t m, uint32_t xor_mask, uint32_t &bin) {
  const uint32_t w = k - m + 1, sh = ku_key_shift(m);
  const uint32_t mask = (1u << (2 * m)) - 1u;  // m <= 15
  uint32_t best = 0xFFFFFFFFu, mn = 0xFFFFFFFFu, a = 0;
  bool plus = true;
  for (uint32_t j = 0; j < w; ++j) {  // j = offset of the m-mer from the left (most significant) end of c
    const uint32_t mm = (uint32_t)(c >> (2 * (k - m - j))) & mask;
    const uint32_t rcmm = ku_revcomp32(mm, m);
    const uint32_t v = (mm < rcmm ? mm : rcmm) ^ xor_mask, kk = v >> sh;
    const bool lt = kk < best;
    best = lt ? kk : best;
    a = lt ? j : a;
    plus = lt ? (mm <= rcmm) : plus;
    mn = min(mn, v);
  }
  bin = mn;
  return ku_locus_assemble(c, best, a, plus, k, m);
}
// 32-bit mix of a locus key: everything of the bucket choice that does not depend on the table's size (a k-mer routed to
// the GPU that owns its bin travels with this word; the owner scales it to its own table)
__device__ __forceinline__ uint32_t ku_locus_prehash(uint64_t locus) {
  // 32-bit mixing (v_mul_lo/hi_u32 are quarter-rate on CDNA; 64-bit multiplies cost four of them each)
  uint32_t g = (uint32_t)locus * 0x9E3779B1u ^ __builtin_rotateleft32((uint32_t)(locus >> 32) * 0x85EBCA77u, 15);
  g ^= g >> 15;
  g *= 0x2C1B3C6Du;
  g ^= g >> 13;
  return g;
}
__device__ __forceinline__ uint64_t ku_locus_line(uint64_t locus, uint64_t n_lines) {
  // n_lines < 2^32 is enforced at table construction (512 GiB of table per shard)
  return __umulhi(ku_locus_prehash(locus), (uint32_t)n_lines);  // (a 2-multiply variant filled the buckets unevenly: 43 ms instead of 20)
}

// lca() in node space (krakenutil.cpp:90-118).  Nodes are ranks of taxids in a
// sorted universe that always contains 0 and 1, so "taxid > 1" == "node > 1".
__device__ __forceinline__ uint32_t ku_lca_nodes(const uint32_t *__restrict__ parent, uint32_t a, uint32_t b) {
  if (a == 0 || b == 0) return a ? a : b;
  // path(a) is walked once per candidate of b's path: O(depth^2), depth <= ~40, ties only
  for (uint32_t guard_b = 0; b > 1 && guard_b < 4096; ++guard_b) {
    uint32_t x = a;
    for (uint32_t guard_a = 0; x > 1 && guard_a < 4096; ++guard_a) {
      if (x == b) return b;
      x = parent[x];
    }
    b = parent[b];
  }
  return 1;
}

