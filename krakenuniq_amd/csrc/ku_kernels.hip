// ku_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the KrakenUniq
// classify hot path.  Integer / byte work bounded by HBM gathers; no MFMA.
//
//   ku_lookup_kernel    rows A1-A6 + A12/A13 of SURVEY.md 8(a):
//                       KmerScanner (krakenutil.cpp:237-282), canonical form
//                       (krakendb.cpp:218-246), minimizer bin key (:200-215),
//                       idx pair fetch (:268-269), in-bin search (:280-299),
//                       ReadCounts::add_kmer -> HLL register max + n_kmers
//                       (readcounts.hpp:71-74, hyperloglogplus.cpp:485-523)
//   ku_resolve_kernel   rows A7-A9: hit_counts, resolve_tree, lca
//                       (classify.cpp:941-968, krakenutil.cpp:90-118,149-200)
//   ku_quick_kernel     quick mode (-q/-m, classify.cpp:943-944,962-963)
//   DB preparation      repack / distinct values / slot remap / count_taxons
//                       (krakendb.cpp:90-113)
//
// Work decomposition of the lookup kernel is FLAT: one lane per k-mer start
// position of the concatenated read buffer (reads are separated by >= 1
// non-ACGT byte, so every k-mer that would span two reads is "ambiguous" by
// construction and costs nothing).  That keeps all 64 lanes busy for 150 bp,
// paired 301 bp and 10 kbp reads alike, makes the taxa[] stores perfectly
// coalesced, and puts neighbouring lanes on neighbouring k-mers -- which share
// their minimizer bin ~(k-nt+1)/2 times in a row, so the idx and pair gathers
// of a wave collapse onto a handful of cache lines.
#include <algorithm>
#include <cstdlib>

#include "ku_device.h"

#define KU_THREADS 256
// One k-mer position per lane and 8 waves per SIMD (64 VGPRs): measured best on MI355X once the kernel became
// instruction-issue bound -- (items, waves, lookup ms per 5 M pairs 2x150): (3, 4) 26.7, (2, 5) 23.6, (1, 7) 22.2,
// (1, 8) 21.3.  More resident waves hide the LDS / ALU latencies of the scan stages better than more probes in
// flight per lane hide HBM latency.
#ifndef KU_MIN_WAVES
#define KU_MIN_WAVES 8  // waves per SIMD the lookup kernel is compiled for
#endif
#ifndef KU_ITEMS
#define KU_ITEMS 1
#endif
#define KU_CTL_MERGE 16u                  // lookup kernel control bit: do not store positions another chunk owns
#define KU_TILE (KU_THREADS * KU_ITEMS)   // k-mer start positions per block iteration
#define KU_PACKW ((KU_TILE + 64) / 16)    // 16-base words staged per tile (covers TILE + 63 bases)

// MODE 0: lookup only; MODE 1: lookup + per-taxon accounting; MODE 2: measurement
// only -- no search, accumulates {queries, sum ceil(log2(n_b+1)), queries into
// non-empty bins, sum n_b} into stats[4] (the algorithmic-bytes model of
// SURVEY.md 8(d), DESIGN.md "Roofline").  MODE 3: the scan of the owner-routed multi-GPU path (KuRouteDev): stages 1-3
// only; every maximal run of consecutive unambiguous k-mers that share their anchor occurrence goes as one 16-byte record
// (its bases + length + anchor offset, ku_internal.h) into the queue of the rank that owns the run's minimizer bin;
// taxa[] gets the k-mers' tickets (record index, index within the record), KU_AMBIG or KU_ROUTE_MISS.
// LAYOUT 0: sorted bins + binary search (the on-disk order); LAYOUT 1: hash table.
// SHARDED: the context owns a strict sub-range of the minimizer bins, so the minimizer of
// every k-mer is needed for the ownership test (always needed by LAYOUT 0 and MODE 2).
// PRIOR: a later database of a hierarchical run (classify.cpp:928-936): taxa[] holds the slots found in the earlier
// databases; positions that already have one are not searched again, and the accounting (MODE 1, last database
// only) books the k-mer under whichever slot it ends up with.
// ITEMS: k-mer positions per lane and block iteration (ITEMS = 1 for the probing instances, see above; the scan of the
// owner-routed path has no probes in flight and amortises its barriers and LDS traffic over two: 42.9 -> 40.6 ms per 10 M reads
// over eight ranks)
template <int MODE, int LAYOUT, bool SHARDED, bool PRIOR, int ITEMS = KU_ITEMS>
__global__ __launch_bounds__(KU_THREADS, KU_MIN_WAVES) void ku_lookup_kernel(KuDbDev db, KuCountsDev cnt,
                                                               const uint8_t *__restrict__ seqs,
                                                               uint64_t n_bytes, uint32_t *__restrict__ taxa,
                                                               unsigned long long *stats, uint32_t ablate, KuRouteDev rt) {
  // bit 4 of `ablate` (KU_CTL_MERGE) is a production flag: chunk pass of an out-of-core run -- positions owned by
  // another chunk keep what that chunk's pass wrote ("non-zero wins" merge, classify.cpp:445-452).  The other
  // bits are a measurement knob (env KU_ABLATE, scripts/ablate_lookup.py): bit0 skip the table/bin probe,
  // bit1 skip the HLL update, bit2 skip the n_kmers counter, bit3 skip the taxa store.  0 in production.
  constexpr int TILE = KU_THREADS * ITEMS;        // k-mer start positions per block iteration
  constexpr int PACKW = (TILE + 64) / 16;         // 16-base words staged per tile (covers TILE + 63 bases)
  constexpr bool DO_COUNTS = MODE == 1;
  constexpr bool NEED_MIN = true;  // every variant uses the LDS sliding-window minimizer (bin, and for LAYOUT 1 its position)
  unsigned long long st_q = 0, st_lg = 0, st_ne = 0, st_nb = 0;
  // 16 bases per word, MSB first (base 16w in bits 31..30): a k-mer is a
  // funnel shift over three consecutive words.
  __shared__ uint32_t s_codes[PACKW + 4];
  __shared__ uint32_t s_amb[(PACKW + 4) / 2 + 2];       // 32 bases per word, MSB first
  __shared__ uint32_t s_mm[NEED_MIN ? TILE + 64 : 1];   // scrambled canonical m-mer per start position (raw value)
  // hash layout: packed window elements (key, offset, strand bit; ku_device.h) of the anchor search, two buffers so
  // that a doubling step reads one and writes the other (one barrier per step)
  constexpr bool PK = LAYOUT == 1 && MODE != 2;
  constexpr bool ROUTE = MODE == 3;
  // ROUTE, per owner: k-mers of the tile; where they go (up to the split index: the rest of the block's current chunk of
  // the owner's queue, beyond it: the chunk claimed for this tile); the current chunk and how much of it is used; the
  // owners' minimizer ranges
  __shared__ uint32_t s_rcnt[ROUTE ? 64 : 1], s_rsplit[ROUTE ? 64 : 1], s_cused[ROUTE ? 64 : 1];
  __shared__ unsigned long long s_rbase[ROUTE ? 64 : 1], s_rbase1[ROUTE ? 64 : 1], s_cbase[ROUTE ? 64 : 1];
  __shared__ unsigned long long s_olo[ROUTE ? 64 : 1], s_ohi[ROUTE ? 64 : 1];
  if (ROUTE) {
    if (threadIdx.x < 64) {
      s_rcnt[threadIdx.x] = 0;
      s_cused[threadIdx.x] = rt.chunk;  // no chunk yet: the first tile claims one
      s_cbase[threadIdx.x] = KU_ROUTE_NONE;
      s_olo[threadIdx.x] = threadIdx.x < rt.world ? rt.own_lo[threadIdx.x] : 0ull;
      s_ohi[threadIdx.x] = threadIdx.x < rt.world ? rt.own_hi[threadIdx.x] : 0ull;
    }
  }
  __shared__ uint32_t s_pa[PK ? TILE + 64 : 1];
  __shared__ uint32_t s_pb[PK ? TILE + 64 : 1];
  __shared__ uint32_t s_tie;
  __shared__ uint32_t s_ctk[KU_CT_CAP];
  __shared__ uint32_t s_ctc[KU_CT_CAP];
  __shared__ uint32_t s_ctu;

  const uint32_t tid = threadIdx.x;
  const uint32_t k = db.k, m = db.nt;
  const uint32_t w = k - m + 1;  // m-mers per k-mer (krakendb.cpp:208)
  const uint64_t n_tiles = (n_bytes + TILE - 1) / TILE;
  const uint32_t key_shift = ku_key_shift(m);
  const uint32_t n_pos = TILE + w - 1;  // m-mer positions a tile needs
  uint16_t *s_amb16 = reinterpret_cast<uint16_t *>(s_amb);

  if (PK) {  // unique sentinels behind the last position: the widest doubling step reads 16 elements ahead
    if (tid < 16) s_pa[n_pos + tid] = s_pb[n_pos + tid] = (0x03FFFFFFu - tid) << KU_PK_KEYSHIFT;
    if (tid == 0) s_tie = 0;
  }
  if (DO_COUNTS) ku_ct_clear(s_ctk, s_ctc, &s_ctu);

  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t tile0 = tile * TILE;
    __syncthreads();  // previous iteration's LDS readers are done
    if (DO_COUNTS) ku_ct_maybe_flush(s_ctk, s_ctc, &s_ctu, cnt.n_kmers);
    // ---- stage 1: ASCII -> packed 2-bit codes + ambiguity bits: four bases per thread (one dword, SWAR), the four
    // lanes of a quad OR their bytes into one 16-base word
    static_assert(4 * (PACKW + 4) <= KU_THREADS, "one packing pass");
    if (tid < 4 * (PACKW + 4)) {
      const uint32_t wi = tid >> 2, q4 = tid & 3u;
      const uint64_t b0 = tile0 + 4ull * tid;
      uint32_t c8 = 0, a4 = 0xFu;
      if (wi < PACKW && b0 < n_bytes) {
        const uint8_t *src = seqs + b0;
        const uintptr_t mis = (uintptr_t)src & 3u;
        uint32_t d;
        if (b0 + 8 - mis <= n_bytes) {  // two aligned dwords cover the four bytes at any alignment
          const uint32_t *q = reinterpret_cast<const uint32_t *>(src - mis);
          d = mis ? __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)mis) : q[0];
        } else {
          d = 0;
          for (uint32_t j = 0; j < 4; ++j) d |= (b0 + j < n_bytes ? (uint32_t)src[j] : (uint32_t)'N') << (8 * j);
        }
        ku_pack_dword(d, c8, a4);
      }
      uint32_t word = c8 << (24 - 8 * q4), amb = a4 << (12 - 4 * q4);
      word |= ku_quad_xor1(word);
      amb |= ku_quad_xor1(amb);
      word |= ku_quad_xor2(word);
      amb |= ku_quad_xor2(amb);
      if (q4 == 0) {
        s_codes[wi] = word;
        s_amb16[wi ^ 1u] = (uint16_t)amb;  // even word -> high half (little-endian uint16 view)
      }
    }
    __syncthreads();

    // ---- stage 2: forward k-mer, ambiguity, canonical form (+ m-mer value) per position
    uint64_t canon[ITEMS];
    uint64_t kfwd[ROUTE ? ITEMS : 1];  // ROUTE: the read-strand k-mer
    bool is_fwd[ITEMS];   // the read-strand k-mer is the canonical one
    bool ok[ITEMS];       // k-mer is unambiguous, inside the buffer and (after stage 3) owned by this shard
    bool foreign[ITEMS];  // unambiguous but its bin belongs to another shard
    uint32_t pk[ITEMS + 1];  // packed window element of the position, then of growing blocks starting there
    bool tie = false;
#pragma unroll
    for (int j = 0; j <= ITEMS; ++j) {
      uint32_t p = j * KU_THREADS + tid;
      pk[j] = 0;
      if (j == ITEMS && (!NEED_MIN || p >= n_pos)) break;
      uint32_t wi = p >> 4, sh = (p & 15u) * 2;
      const uint32_t c0 = s_codes[wi], c1 = s_codes[wi + 1], c2 = s_codes[wi + 2];
      // 32 bases from p: two 64-bit shifts, no special case for sh = 0
      uint64_t x = ((((uint64_t)c0 << 32) | c1) << sh >> 32 << 32) | ((((uint64_t)c1 << 32) | c2) << sh >> 32);
      if (NEED_MIN) {  // m-mer starting at p, canonical, scrambled (krakendb.cpp:209)
        uint32_t mm = (uint32_t)(x >> (64 - 2 * m));
        uint32_t mrc = ku_revcomp32(mm, m);
        const uint32_t val = (mm < mrc ? mm : mrc) ^ db.xor_mask;
        s_mm[p] = val;
        if (PK) {
          pk[j] = ku_pk_make(val, key_shift, mm <= mrc);
          s_pa[p] = pk[j];
          if ((m & 1u) == 0) tie |= mm == mrc;  // palindromic m-mer: its strand bit is not enough
        }
      }
      if (j < ITEMS) {
        uint64_t fwd = x >> (64 - 2 * k);
        if (ROUTE) {  // the k-mer itself is the owner's business; its strand is needed where ties are resolved only
          kfwd[j] = fwd;
          canon[j] = 0;
          is_fwd[j] = true;
        } else {
          uint64_t rc = ku_revcomp64(fwd, k);
          canon[j] = fwd <= rc ? fwd : rc;
          is_fwd[j] = fwd <= rc;
        }
        uint32_t ai = p >> 5, as = p & 31u;
        uint64_t a = (((uint64_t)s_amb[ai] << 32) | s_amb[ai + 1]) << as;
        ok[j] = (a >> (64 - k)) == 0 && (tile0 + p + k <= n_bytes);
        foreign[j] = false;
      }
    }

    uint32_t prior[ITEMS];  // PRIOR: slot from an earlier database (0 = not found yet)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) prior[j] = (PRIOR && ok[j]) ? taxa[tile0 + (uint64_t)j * KU_THREADS + tid] : 0u;

    // ---- stage 3: minimizer = sliding-window minimum of the m-mer values; ownership; idx fetch
    uint32_t n_b[ITEMS];          // bin size (LAYOUT 0 / MODE 2); ROUTE: the owner
    uint32_t tro[ITEMS];          // ROUTE: offset of the anchor occurrence from the k-mer's first base, read order
    const uint32_t *bp[ITEMS];    // first pair of the bin (LAYOUT 0)
    uint64_t locus[ITEMS];        // locus key (LAYOUT 1), see ku_locus_key()
    if (NEED_MIN) {
      __syncthreads();
      uint32_t key[ITEMS], aoff[ITEMS], bin_v[ITEMS];
      bool plus[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) tro[j] = 0;
      if (PK) {
        // anchor search (ku_device.h): block minima of 2, 4, 8, 16 positions by doubling, then one overlapping step
        // for the window length -- five packed dword minima per position for any minimizer length
        uint32_t *src = s_pa, *dst = s_pb;
        uint32_t blk = 1;
        for (uint32_t st = 1; 2 * st <= w; st <<= 1) {
#pragma unroll
          for (int j = 0; j <= ITEMS; ++j) {
            const uint32_t p = j * KU_THREADS + tid;
            if (p < n_pos) {
              pk[j] = ku_pk_combine(pk[j], src[p + st], st, tie);
              dst[p] = pk[j];
            }
          }
          __syncthreads();
          uint32_t *t = src; src = dst; dst = t;
          blk = 2 * st;
        }
        if (w > blk) {
#pragma unroll
          for (int j = 0; j < ITEMS; ++j) {
            const uint32_t p = j * KU_THREADS + tid;
            pk[j] = ku_pk_combine_overlap(pk[j], src[p + (w - blk)], w - blk, tie);
          }
        }
        if (tie) s_tie = 1;
        __syncthreads();
        const bool exact = s_tie != 0;  // block-uniform
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const uint32_t p = j * KU_THREADS + tid;
          if (!exact) {
            const uint32_t t = (pk[j] >> 1) & 31u;  // read-order offset of the first minimal key
            key[j] = pk[j] >> KU_PK_KEYSHIFT;
            aoff[j] = is_fwd[j] ? t : w - 1 - t;
            plus[j] = ((pk[j] & 1u) != 0) == is_fwd[j];
            tro[j] = t;
            // a unique minimal key is the minimizer occurrence: its raw value is the bin
            bin_v[j] = (SHARDED && key_shift) ? s_mm[p + t] : key[j];
          } else if (ok[j]) {
            // rare (low-complexity sequence): scan the raw values in the canonical k-mer's frame
            if (ROUTE) is_fwd[j] = kfwd[j] <= ku_revcomp64(kfwd[j], k);
            key[j] = ku_anchor_exact(s_mm + p, w, key_shift, is_fwd[j], aoff[j], bin_v[j]);
            tro[j] = is_fwd[j] ? aoff[j] : w - 1 - aoff[j];
            const uint32_t q = p + tro[j];
            const uint32_t wi = q >> 4, sh = (q & 15u) * 2;
            const uint64_t two = ((uint64_t)s_codes[wi] << 32) | s_codes[wi + 1];
            const uint32_t mmf = (uint32_t)((two << sh) >> (64 - 2 * m));
            const uint32_t rcm = ku_revcomp32(mmf, m);
            plus[j] = is_fwd[j] ? (mmf <= rcm) : (rcm <= mmf);
          } else {
            key[j] = 0; aoff[j] = 0; plus[j] = true; bin_v[j] = 0;
          }
        }
        if (exact) {
          __syncthreads();  // every thread has read the flag
          if (tid == 0) s_tie = 0;
        }
      }
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        uint32_t p = j * KU_THREADS + tid;
        n_b[j] = 0;
        if (ok[j]) {
          uint32_t mn = 0xFFFFFFFFu;
          if (PK) {
            locus[j] = ku_locus_assemble(canon[j], key[j], aoff[j], plus[j], k, m);
            mn = bin_v[j];
          } else {
            for (uint32_t i = 0; i < w; ++i) mn = min(mn, s_mm[p + i]);
          }
          uint64_t bin = mn;
          if (ROUTE) {
            // the owner: the rank whose minimizer range holds the bin (none: the k-mer is nobody's, a miss).  The ranges
            // ascend with the rank (ku_mgpu checks): the last one that starts at or below the bin, if it reaches beyond it
            uint32_t q = 0;
            for (uint32_t nq = rt.world; nq > 1;) {  // uniform trip count
              const uint32_t half = nq >> 1;
              if (s_olo[q + half] <= bin) q += half;
              nq -= half;
            }
            n_b[j] = (bin >= s_olo[q] && bin < s_ohi[q]) ? q : 0xFFu;
          } else
          if (!SHARDED || (bin >= db.bin_lo && bin < db.bin_hi)) {  // is_minimizer_in_chunk (krakendb.cpp:524-526)
            if (LAYOUT == 0 || MODE == 2) {
              const uint64_t *o = db.offsets + (bin - db.bin_lo);
              uint64_t lo = o[0], hi = o[1];
              n_b[j] = (uint32_t)(hi - lo);
              bp[j] = db.pairs + 3 * (lo - db.pair_base);
            }
          } else {
            ok[j] = false;  // another shard owns this k-mer (and accounts its miss)
            foreign[j] = true;
          }
        }
      }
    }

    if (ROUTE) {
      // (n_b[j] = owner, 0xFF none.)  Runs: a k-mer continues the run of the lane below when both are unambiguous and
      // share the anchor POSITION (then they share bin and owner); a wave's lane 0 always starts one, and a run is cut at
      // KU_ROUTE_MAXN k-mers (the record holds 56 bases).  The run's first lane writes the record.  The tile's records get
      // consecutive places in their owners' queues (a local index per start lane from an LDS counter); the BLOCK owns a chunk of rt.chunk records in every queue and claims the next one(s) from
      // the queue's global cursor only when a tile does not fit the rest of it (device-scope adds on a handful of
      // addresses from every block and tile were the whole cost of this kernel once).  The records of a read stay
      // together and in read order, which keeps the owner's bucket probes on few lines per wave.  What a block leaves
      // unused of its last chunks is filled with null records (n = 0).  A queue has room for rt.cap records; chunks
      // beyond it are counted, not written, so the cursors always end as the true per-owner totals and the host can size a
      // second pass.
      uint32_t idx[ITEMS], rs[ITEMS], rn[ITEMS];
      bool st[ITEMS];
      const uint32_t lane = tid & 63u;
      const unsigned long long below = (1ull << lane) - 1ull, incl = below | (1ull << lane);
      const uint32_t nmax = ku_route_maxn(k, m);
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const uint32_t p = j * KU_THREADS + tid;
        const bool go = ok[j] && n_b[j] != 0xFFu;
        const uint32_t qa = p + tro[j];
        const uint32_t up_q = ku_wave_up1(qa), up_go = ku_wave_up1((uint32_t)go);
        bool start = go && (lane == 0 || up_go == 0 || up_q != qa);
        unsigned long long sm = __ballot(start);
        uint32_t r0 = (sm & incl) ? 63u - (uint32_t)__builtin_clzll(sm & incl) : lane;
        start = start || (go && lane - r0 == nmax);  // (a run has at most w <= 31 < 2 * nmax k-mers: one cut is enough)
        sm = __ballot(start);
        r0 = (sm & incl) ? 63u - (uint32_t)__builtin_clzll(sm & incl) : lane;
        rs[j] = r0;
        const unsigned long long bound = sm | ~__ballot(go);   // where a run ends: the next start, or a lane that queues nothing
        const unsigned long long rest = lane == 63u ? 0ull : (bound >> (lane + 1u));
        rn[j] = rest ? (uint32_t)__builtin_ctzll(rest) + 1u : 64u - lane;
        st[j] = start;
        // the record's place among the tile's records of its owner: one LDS add per start lane (a handful per wave; a
        // wave-level loop over the distinct owners cost ten times the instructions).  The order within (tile, owner) is
        // whatever the adds make it: a ticket names its record, nothing depends on the order.
        idx[j] = start ? atomicAdd(&s_rcnt[n_b[j]], 1u) : 0u;
      }
      __syncthreads();
      if (tid < rt.world) {
        const uint32_t n = s_rcnt[tid], used = s_cused[tid];
        const unsigned long long base = s_cbase[tid];
        s_rbase[tid] = base == KU_ROUTE_NONE ? KU_ROUTE_NONE : base + used;
        if (used + n <= rt.chunk) {
          s_rsplit[tid] = n;
          s_cused[tid] = used + n;
        } else {
          // the chunk's rest is used up by the first records of the tile, the others open the next chunk(s)
          const uint32_t need = n - (rt.chunk - used), nch = (need + rt.chunk - 1u) / rt.chunk;
          unsigned long long nb = atomicAdd(&rt.cursor[tid * KU_ROUTE_CURSOR_STRIDE], (unsigned long long)nch * rt.chunk);
          if (nb + (unsigned long long)nch * rt.chunk > rt.cap) nb = KU_ROUTE_NONE;
          s_rsplit[tid] = rt.chunk - used;
          s_rbase1[tid] = nb;
          s_cbase[tid] = nb == KU_ROUTE_NONE ? KU_ROUTE_NONE : nb + (unsigned long long)(nch - 1u) * rt.chunk;
          s_cused[tid] = need - (nch - 1u) * rt.chunk;
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const uint32_t p = j * KU_THREADS + tid;
        uint32_t e32 = 0xFFFFFFFFu;  // the record's index in the queue buffer (none: the queue was full, the pass is redone)
        if (st[j]) {
          const uint32_t o = n_b[j], split = s_rsplit[o];
          const unsigned long long at = idx[j] < split ? s_rbase[o] : s_rbase1[o];
          if (at != KU_ROUTE_NONE) {
            const unsigned long long e = (unsigned long long)o * rt.cap + at + (idx[j] < split ? idx[j] : idx[j] - split);
            // 56 bases from position p: four funnel shifts over five code words
            const uint32_t wi = p >> 4, sh = (p & 15u) * 2;
            const uint32_t c0 = s_codes[wi], c1 = s_codes[wi + 1], c2 = s_codes[wi + 2], c3 = s_codes[wi + 3], c4 = s_codes[wi + 4];
            uint4 rec;
            rec.x = (uint32_t)(((((uint64_t)c0 << 32) | c1) << sh) >> 32);
            rec.y = (uint32_t)(((((uint64_t)c1 << 32) | c2) << sh) >> 32);
            rec.z = (uint32_t)(((((uint64_t)c2 << 32) | c3) << sh) >> 32);
            rec.w = ((uint32_t)(((((uint64_t)c3 << 32) | c4) << sh) >> 32) & 0xFFFF0000u) | (tro[j] << 8) | rn[j];
            rt.q_rec[e] = rec;
            e32 = (uint32_t)e;
          }
        }
        const uint32_t eb = (uint32_t)__shfl((int)e32, (int)rs[j]);  // the record of the run this k-mer belongs to
        const uint64_t pos = tile0 + p;
        if (pos < n_bytes)
          taxa[pos] = !ok[j] ? KU_AMBIG : ((n_b[j] == 0xFFu || eb == 0xFFFFFFFFu) ? KU_ROUTE_MISS : ((eb << 5) | (lane - rs[j])));
      }
      __syncthreads();
      if (tid < 64) s_rcnt[tid] = 0;
      continue;
    }
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j)
        if (ok[j]) {
          st_q += 1;
          st_lg += n_b[j] ? 32 - __builtin_clz(n_b[j]) : 0;
          st_ne += n_b[j] != 0;
          st_nb += n_b[j];
        }
      continue;
    }

    // ---- stage 4: the lookup proper, ITEMS independent probes in flight per lane
    uint32_t slot[ITEMS];
    uint64_t hh[ITEMS];  // fmix64(kmer + 1): table position and HLL index/rank
    if (LAYOUT == 1) {
      const uint32_t *lp[ITEMS];  // the bucket (line) being examined
      uint32_t tag[ITEMS], cand[ITEMS];
      bool act[ITEMS], ovf[ITEMS];
      uint4 h4[ITEMS];
      const uint32_t *tab = reinterpret_cast<const uint32_t *>(db.table);
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        hh[j] = ku_fmix64(canon[j]);
        lp[j] = tab + (ok[j] ? ku_locus_line(locus[j], db.n_lines) : 0) * KU_LINE_DWORDS;
        tag[j] = ku_table_tag(hh[j]);
        slot[j] = 0;
        act[j] = ok[j] && !(ablate & 1u) && !(PRIOR && prior[j]);
      }
      // round trip 1: the 16-byte bucket headers of all items
#pragma unroll
      for (int j = 0; j < ITEMS; ++j)
        if (act[j]) h4[j] = *reinterpret_cast<const uint4 *>(lp[j]);
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        cand[j] = 0;
        ovf[j] = false;
        if (act[j]) {
          cand[j] = ku_tag_matches(h4[j], tag[j]);
          ovf[j] = ku_line_spilled(h4[j]);
          act[j] = cand[j] != 0 || ovf[j];
        }
      }
      // round trip 2 (same line: L1/L2 hit): the first candidate entry of every item
      KuPair pr[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j)
        if (cand[j]) pr[j] = *reinterpret_cast<const KuPair *>(lp[j] + KU_LINE_ENTRY0 + 3 * (__builtin_ctz(cand[j])));
#pragma unroll
      for (int j = 0; j < ITEMS; ++j)
        if (cand[j]) {
          if ((((uint64_t)pr[j].key_hi << 32) | pr[j].key_lo) == canon[j]) {
            slot[j] = pr[j].slot;
            act[j] = false;
          } else {
            cand[j] &= cand[j] - 1;
            act[j] = cand[j] != 0 || ovf[j];
          }
        }
      // rare tail: further tag matches in the bucket (false positives) and spilled buckets; the items of a
      // lane advance in lockstep so their round trips overlap
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
            } else {  // ovf[j]: continue in the next line
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
              cand[j] &= cand[j] - 1;
              if ((((uint64_t)e[j].key_hi << 32) | e[j].key_lo) == canon[j]) {
                slot[j] = e[j].slot;
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
    } else {
      uint32_t lo[ITEMS], hi[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        lo[j] = 0;
        hi[j] = ((ablate & 1u) || (PRIOR && prior[j])) ? 0 : n_b[j];
        slot[j] = 0;
        if (DO_COUNTS) hh[j] = ku_fmix64(canon[j]);
      }
      bool any = true;
      while (any) {
        any = false;
        KuPair pr[ITEMS];
        uint32_t mid[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          mid[j] = (lo[j] + hi[j]) >> 1;
          if (lo[j] < hi[j]) pr[j] = reinterpret_cast<const KuPair *>(bp[j])[mid[j]];  // one 12-byte load: key + value
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          if (lo[j] < hi[j]) {
            uint64_t key = ((uint64_t)pr[j].key_hi << 32) | pr[j].key_lo;
            if (key == canon[j]) {
              slot[j] = pr[j].slot;
              lo[j] = hi[j];
            } else if (key < canon[j]) {
              lo[j] = mid[j] + 1;
            } else {
              hi[j] = mid[j];
            }
            any |= lo[j] < hi[j];
          }
        }
      }
    }

    // ---- stage 5: per-taxon accounting (classify.cpp:939) + coalesced store
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      uint64_t pos = tile0 + (uint64_t)j * KU_THREADS + tid;
      if (PRIOR && prior[j]) slot[j] = prior[j];
      if (DO_COUNTS) {
        if (ok[j] && !(ablate & 2u)) ku_hll_update(cnt.registers, slot[j], hh[j]);
        // n_kmers: neighbouring k-mers mostly carry one slot -- when the whole wave agrees, one lane books them all
        // (one LDS update instead of 64 on the same word)
        const bool okc = ok[j] && !(ablate & 4u);
        const unsigned long long booked = __ballot(okc);
        if (booked) {
          const uint32_t lead = (uint32_t)__ffsll((long long)booked) - 1;
          const uint32_t s0 = ku_wave_bcast(slot[j], lead);
          if (__ballot(okc && slot[j] == s0) == booked) {
            if ((tid & 63u) == lead) ku_ct_add(s_ctk, s_ctc, &s_ctu, s0, (uint32_t)__popcll(booked), cnt.n_kmers);
          } else if (okc) {
            ku_ct_add(s_ctk, s_ctc, &s_ctu, slot[j], 1, cnt.n_kmers);
          }
        }
      }
      if (pos < n_bytes && !(ablate & 8u) && !((ablate & KU_CTL_MERGE) && foreign[j])) {
        // ambiguous -> KU_AMBIG on every shard; not owned -> 0 (the owner's value wins the max-reduce)
        taxa[pos] = ok[j] ? slot[j] : (foreign[j] ? 0u : KU_AMBIG);
      }
    }
  }
  if (DO_COUNTS) {
    __syncthreads();
    ku_ct_flush(s_ctk, s_ctc, cnt.n_kmers);
  }
  if (ROUTE) {  // the unused rest of the block's chunks: null records
    __syncthreads();
    for (uint32_t o = 0; o < rt.world; ++o) {
      const unsigned long long base = s_cbase[o];
      if (base == KU_ROUTE_NONE) continue;
      for (uint32_t i = s_cused[o] + tid; i < rt.chunk; i += KU_THREADS)
        rt.q_rec[(unsigned long long)o * rt.cap + base + i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if (MODE == 2) {
    unsigned long long v[4] = {st_q, st_lg, st_ne, st_nb};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
      if ((tid & 63) == 0 && v[i]) atomicAdd(&stats[i], v[i]);
    }
  }
}

static unsigned ku_lookup_grid(uint64_t n_bytes, int n_cu) {
  uint64_t n_tiles = (n_bytes + KU_TILE - 1) / KU_TILE;
  uint64_t max_blocks = (uint64_t)n_cu * 2 * KU_MIN_WAVES;  // two rounds of the blocks a CU holds at once
  return (unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks);
}

int ku_launch_lookup(const KuDbDev &db, const KuCountsDev &cnt, const uint8_t *d_seqs, uint64_t n_bytes,
                     uint32_t *d_taxa, bool do_counts, bool prior, bool merge_chunk, int n_cu, hipStream_t stream) {
  if (n_bytes == 0) return KU_OK;
  const dim3 grid(ku_lookup_grid(n_bytes, n_cu)), block(KU_THREADS);
  unsigned long long *ns = nullptr;
  const bool sharded = !(db.bin_lo == 0 && db.bin_hi == (1ull << (2 * db.nt)));
  const char *ab = getenv("KU_ABLATE");
  const uint32_t ablate = ((ab ? (uint32_t)atoi(ab) : 0u) & ~KU_CTL_MERGE) | (merge_chunk ? KU_CTL_MERGE : 0u);
  if (prior && sharded) return KU_EUNSUP;
#define KU_LAUNCH(M, L, S, P) hipLaunchKernelGGL((ku_lookup_kernel<M, L, S, P>), grid, block, 0, stream, db, cnt, d_seqs, n_bytes, d_taxa, ns, ablate, KuRouteDev{})
  if (prior) {
    if (db.table) { if (do_counts) KU_LAUNCH(1, 1, false, true); else KU_LAUNCH(0, 1, false, true); }
    else { if (do_counts) KU_LAUNCH(1, 0, true, true); else KU_LAUNCH(0, 0, true, true); }
  } else if (db.table) {
    if (do_counts) { if (sharded) KU_LAUNCH(1, 1, true, false); else KU_LAUNCH(1, 1, false, false); }
    else { if (sharded) KU_LAUNCH(0, 1, true, false); else KU_LAUNCH(0, 1, false, false); }
  } else {
    if (do_counts) KU_LAUNCH(1, 0, true, false); else KU_LAUNCH(0, 0, true, false);
  }
#undef KU_LAUNCH
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_lookup_stats(const KuDbDev &db, const uint8_t *d_seqs, uint64_t n_bytes, unsigned long long *d_stats,
                           int n_cu, hipStream_t stream) {
  if (n_bytes == 0) return KU_OK;
  hipLaunchKernelGGL((ku_lookup_kernel<2, 0, true, false>), dim3(ku_lookup_grid(n_bytes, n_cu)), dim3(KU_THREADS), 0, stream,
                     db, KuCountsDev{}, d_seqs, n_bytes, (uint32_t *)nullptr, d_stats, 0u, KuRouteDev{});
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// the scan of the owner-routed path over one rank's slice of the reads
#ifndef KU_ROUTE_ITEMS
#define KU_ROUTE_ITEMS 2
#endif
unsigned ku_route_scan_grid(uint64_t n_bytes, int n_cu) {
  // (a block pads its last chunk of every queue: at least 8 tiles per block keep that a small share of what it queues)
  const uint64_t tile = (uint64_t)KU_THREADS * KU_ROUTE_ITEMS, n_tiles = (n_bytes + tile - 1) / tile;
  const uint64_t max_blocks = (uint64_t)n_cu * 2 * KU_MIN_WAVES;  // two rounds of the blocks a CU holds at once
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(n_tiles, max_blocks), (n_tiles + 7) / 8));
}
int ku_launch_route_scan(const KuDbDev &db, const uint8_t *d_seqs, uint64_t n_bytes, uint32_t *d_taxa, const KuRouteDev &rt, int n_cu,
                         hipStream_t stream) {
  if (n_bytes == 0) return KU_OK;
  if (!db.table || rt.world == 0 || rt.world > 64 || rt.cap == 0 || rt.chunk == 0 || rt.chunk % KU_ROUTE_CHUNK || rt.cap % rt.chunk) return KU_EINVAL;
  if ((uint64_t)rt.world * rt.cap > KU_ROUTE_MAX_RECORDS) return KU_EUNSUP;
  hipLaunchKernelGGL((ku_lookup_kernel<3, 1, true, false, KU_ROUTE_ITEMS>), dim3(ku_route_scan_grid(n_bytes, n_cu)), dim3(KU_THREADS), 0, stream, db,
                     KuCountsDev{}, d_seqs, n_bytes, d_taxa, (unsigned long long *)nullptr, 0u, rt);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// hash-table construction from the (already slot-remapped) 12-byte pairs
__global__ void ku_build_table_kernel(const uint32_t *__restrict__ pairs, uint64_t n, uint32_t *table, uint64_t n_lines,
                                      uint32_t k, uint32_t m, uint32_t xor_mask, unsigned long long *n_spilled) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t klo = pairs[3 * i], khi = pairs[3 * i + 1], val = pairs[3 * i + 2];
    const uint64_t key = ((uint64_t)khi << 32) | klo;
    const uint32_t tag = ku_table_tag(ku_fmix64(key));
    uint32_t bin;
    uint64_t line = ku_locus_line(ku_locus_key(key, k, m, xor_mask, bin), n_lines);
    for (uint32_t hops = 0;; ++hops) {
      uint32_t *lp = table + line * KU_LINE_DWORDS;
      // claim the first unused entry of the bucket: its tag field goes from 0 to the (non-zero) tag with a CAS on the
      // dword that holds it (the neighbour field and the spill flag may change under it: retry on the fresh value)
      int got = -1;
      for (int i = 0; i < KU_LINE_SLOTS && got < 0; ++i) {
        uint32_t *w = lp + ku_tag_dword((uint32_t)i);
        const uint32_t sh = ku_tag_shift((uint32_t)i);
        uint32_t cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (((cur >> sh) & 0x7FFFu) == 0) {
          const uint32_t prev = atomicCAS(w, cur, cur | (tag << sh));
          if (prev == cur) { got = i; break; }
          cur = prev;
        }
      }
      if (got >= 0) {
        uint32_t *e = lp + KU_LINE_ENTRY0 + 3 * got;
        e[0] = klo; e[1] = khi; e[2] = val;
        if (hops) atomicAdd(n_spilled, 1ull);
        break;
      }
      atomicOr(lp, 0x8000u);  // full: the bucket is marked "spilled" before the key moves on
      line = line + 1 == n_lines ? 0 : line + 1;
    }
  }
}
int ku_launch_build_table(const uint32_t *d_pairs, uint64_t n_pairs, void *d_table, uint64_t n_lines, uint32_t k,
                          uint32_t m, uint32_t xor_mask, unsigned long long *d_spilled, hipStream_t stream) {
  if (hipMemsetAsync(d_table, 0, n_lines * 128, stream) != hipSuccess) return KU_EHIP;
  if (n_pairs == 0) return KU_OK;
  uint64_t nb = (n_pairs + 255) / 256;
  hipLaunchKernelGGL(ku_build_table_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, stream, d_pairs,
                     n_pairs, (uint32_t *)d_table, n_lines, k, m, xor_mask, d_spilled);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// count_taxons over the table entries (hash layout)
__global__ __launch_bounds__(256) void ku_count_table_kernel(const uint32_t *__restrict__ table, uint64_t n_lines,
                                                             unsigned long long *counts) {
  __shared__ uint32_t s_ctk[KU_CT_CAP];
  __shared__ uint32_t s_ctc[KU_CT_CAP];
  __shared__ uint32_t s_ctu;
  ku_ct_clear(s_ctk, s_ctc, &s_ctu);
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = blockIdx.x * (uint64_t)blockDim.x; base < n_lines; base += stride) {
    uint64_t l = base + threadIdx.x;
    if (l < n_lines) {
      const uint32_t *lp = table + l * KU_LINE_DWORDS;
      for (uint32_t i = 0; i < KU_LINE_SLOTS; ++i)
        if ((lp[ku_tag_dword(i)] >> ku_tag_shift(i)) & 0x7FFFu) ku_ct_add(s_ctk, s_ctc, &s_ctu, lp[KU_LINE_ENTRY0 + 3 * i + 2], 1, counts);
    }
    __syncthreads();
    ku_ct_maybe_flush(s_ctk, s_ctc, &s_ctu, counts);
  }
  __syncthreads();
  ku_ct_flush(s_ctk, s_ctc, counts);
}
int ku_launch_count_table(const void *d_table, uint64_t n_lines, unsigned long long *d_counts, hipStream_t stream) {
  hipLaunchKernelGGL(ku_count_table_kernel, dim3(2048), dim3(256), 0, stream, (const uint32_t *)d_table, n_lines,
                     d_counts);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// ----------------------------------------------------------------------------
// ku_resolve_kernel: one group (wave or block) per read
// ----------------------------------------------------------------------------
// MODE 0: GROUP 64,  LDS table of 512 entries    -> reads with n_kmers <= 384
// MODE 1: GROUP 256, LDS table of 16384 entries  -> n_kmers <= 12288
// MODE 2: GROUP 256, table in global workspace   -> anything longer
// Table entry: key = slot + 1 (0 = empty); cnt word = hit count (MODE 0/1: low 16
// bits, the root-path score is accumulated in the high 16 bits; MODE 2: separate
// score array).
template <int MODE> struct KuResolveCfg;
template <> struct KuResolveCfg<0> { static constexpr int GROUP = 64, CAP_LOG2 = 9, MAX_N = 384; };
// (the 16384-cell table fills most of a CU's LDS, so one block per CU: 1024 threads keep its 4 SIMDs at 4 waves each)
template <> struct KuResolveCfg<1> { static constexpr int GROUP = 1024, CAP_LOG2 = 14, MAX_N = 12288; };
template <> struct KuResolveCfg<2> { static constexpr int GROUP = 256, CAP_LOG2 = 0, MAX_N = 0x7fffffff; };

template <int GROUP> __device__ __forceinline__ uint32_t ku_group_max(uint32_t v, uint32_t *s_red) {
  v = ku_wave_max_u32(v);
  if (GROUP > 64) {
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    v = s_red[0];
#pragma unroll
    for (int i = 1; i < GROUP / 64; ++i) v = max(v, s_red[i]);
  }
  return v;
}
template <int GROUP> __device__ __forceinline__ uint32_t ku_group_min(uint32_t v, uint32_t *s_red) {
  return ~ku_group_max<GROUP>(~v, s_red);
}
template <int GROUP> __device__ __forceinline__ uint32_t ku_group_sum(uint32_t v, uint32_t *s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  if (GROUP > 64) {
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    v = 0;
#pragma unroll
    for (int i = 0; i < GROUP / 64; ++i) v += s_red[i];
  }
  return v;
}

template <int MODE>
__global__ __launch_bounds__(KuResolveCfg<MODE>::GROUP) void ku_resolve_kernel(
    KuTaxDev tax, KuCountsDev cnt, uint32_t k, const uint64_t *__restrict__ seq_off,
    const uint32_t *__restrict__ seq_len, uint64_t n_reads, uint32_t min_n, uint32_t flags,
    uint32_t *__restrict__ calls, uint32_t *__restrict__ taxa, uint32_t *__restrict__ hits_out, uint32_t *ws,
    uint32_t ws_cap_log2) {
  using Cfg = KuResolveCfg<MODE>;
  constexpr int GROUP = Cfg::GROUP;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  __shared__ uint32_t s_red[16];  // one word per wave of the group
  __shared__ uint32_t s_n_list;
  __shared__ uint32_t s_bcast;
  constexpr int RCT = 7;  // n_reads counter table: 128 entries keep the block's LDS at ~6 KB (occupancy)
  __shared__ uint32_t s_ctk[1 << RCT];
  __shared__ uint32_t s_ctc[1 << RCT];
  __shared__ uint32_t s_ctu;

  const uint32_t tid = threadIdx.x;
  // a one-wave group orders its own LDS traffic with a fence (the hardware runs a wave's DS operations in order); a
  // block barrier would also drain every outstanding global load and store of the read pipeline below
  auto gsync = [&]() {
    if (GROUP == 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
  };
  const uint32_t cap_log2 = MODE == 2 ? ws_cap_log2 : (uint32_t)Cfg::CAP_LOG2;
  const uint32_t cap = 1u << cap_log2;
  uint32_t *t_key, *t_cnt, *t_score, *t_list;
  if (MODE == 2) {
    uint32_t *base = ws + (size_t)blockIdx.x * 4 * cap;
    t_key = base;
    t_cnt = base + cap;
    t_score = base + 2 * (size_t)cap;
    t_list = base + 3 * (size_t)cap;
  } else {
    t_key = smem;
    t_cnt = smem + cap;
    t_score = nullptr;
    t_list = smem + 2 * cap;  // uint16 entries
  }
  uint16_t *t_list16 = reinterpret_cast<uint16_t *>(t_list);
  const bool do_counts = !(flags & KU_F_NO_COUNTS);
  const bool keep_slots = (flags & KU_F_KEEP_SLOTS) != 0;

  ku_ct_clear<RCT>(s_ctk, s_ctc, &s_ctu);
  for (uint32_t i = tid; i < cap; i += GROUP) {
    t_key[i] = 0;
    t_cnt[i] = 0;
    if (MODE == 2) t_score[i] = 0;
  }
  __syncthreads();

  // MODE 0 keeps the read's codes in registers (<= 6 per lane) and short-cuts the common case of at most
  // one distinct hit taxon: resolve_tree() then returns that taxon (or 0) without any tree walk.
  constexpr int NV = MODE == 0 ? (Cfg::MAX_N + 63) / 64 : 1;
  // MODE 0 is a chain of dependent memory round trips per read (length / offset -> codes -> slot table -> stores) with
  // almost no arithmetic in between, so the wave runs a software pipeline over its reads: the codes of the NEXT read
  // are requested before the current one is resolved, its length and offset one read earlier still.
  const uint64_t stride = gridDim.x;
  uint32_t p_len = 0, pp_len = 0, pv[NV];
  uint64_t p_off = 0, pp_off = 0;
  auto request_codes = [&](uint32_t plen, uint64_t poff) {
    const uint32_t pn = plen >= k ? plen - k + 1 : 0;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const uint32_t i = c * 64 + tid;
      pv[c] = (i < pn && pn <= (uint32_t)Cfg::MAX_N) ? taxa[poff + i] : 0u;
    }
  };
  if (MODE == 0) {
    const uint64_t r0 = blockIdx.x;
    if (r0 < n_reads) { p_len = seq_len[r0]; p_off = seq_off[r0]; request_codes(p_len, p_off); }
    if (r0 + stride < n_reads) { pp_len = seq_len[r0 + stride]; pp_off = seq_off[r0 + stride]; }
  }

  for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const uint32_t len = MODE == 0 ? p_len : seq_len[r];
    const uint64_t off = MODE == 0 ? p_off : seq_off[r];
    uint32_t v[NV];
    if (MODE == 0) {  // take what was requested for this read, request the next read's
#pragma unroll
      for (int c = 0; c < NV; ++c) v[c] = pv[c];
      p_len = pp_len;
      p_off = pp_off;
      if (r + stride < n_reads) request_codes(p_len, p_off);
      if (r + 2 * stride < n_reads) { pp_len = seq_len[r + 2 * stride]; pp_off = seq_off[r + 2 * stride]; }
    }
    const uint32_t n = len >= k ? len - k + 1 : 0;
    if (n < min_n || n > (uint32_t)Cfg::MAX_N) continue;  // another MODE's launch handles it
    if (GROUP > 64 || s_ctu > (1u << RCT) / 2) ku_ct_maybe_flush<RCT>(s_ctk, s_ctc, &s_ctu, cnt.n_reads);
    uint32_t call_node = 0, uni_taxid = 0;
    bool resolved = false;
    if (MODE == 0) {
      uint32_t mine = 0;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        if (v[c] == KU_AMBIG) v[c] = 0;  // ambiguous k-mers carry no hit
        if (mine == 0) mine = v[c];
      }
      unsigned long long bal = __ballot(mine != 0);
      uint32_t first = bal ? ku_wave_bcast(mine, (uint32_t)__ffsll((long long)bal) - 1) : 0u;
      bool diff = false;
#pragma unroll
      for (int c = 0; c < NV; ++c) diff |= (v[c] != 0 && v[c] != first);
      if (!__any(diff)) {
        resolved = true;
        // the two table reads depend on `first` only: issued together (one round trip instead of a chain)
        call_node = first ? tax.slot_node[first] : 0u;
        uni_taxid = first ? tax.slot_taxid[first] : 0u;
      }
    }
    if (!resolved) {
    if (tid == 0) s_n_list = 0;
    gsync();  // the list is appended to by whichever lane claims a table cell
    // ---- hit_counts[taxon]++ (classify.cpp:941-942)
    auto table_insert = [&](uint32_t s, uint32_t count) {
      uint32_t h = (s * 2654435761u) >> (32 - cap_log2);
      for (;;) {
        uint32_t cur = __hip_atomic_load(&t_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0) {
          uint32_t old = atomicCAS(&t_key[h], 0u, s + 1);
          if (old == 0) {  // first sight of this taxon in the read: it gets a place in the list of distinct taxa
            const uint32_t e = atomicAdd(&s_n_list, 1u);
            if (MODE == 2) t_list[e] = h; else t_list16[e] = (uint16_t)h;
          }
          cur = old == 0 ? s + 1 : old;
        }
        if (cur == s + 1) {
          atomicAdd(&t_cnt[h], count);
          break;
        }
        h = (h + 1) & (cap - 1);
      }
    };
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < NV; ++c)
        if (v[c] != 0) table_insert(v[c], 1u);
    } else {
      // long reads: neighbouring k-mers mostly carry the same slot, so a wave first groups its 64 codes (one
      // ballot per distinct slot) and a single lane books the whole group -- instead of 64 atomics on one LDS word
      for (uint32_t base = 0; base < n; base += GROUP) {
        const uint32_t i = base + tid;
        const uint32_t s = i < n ? taxa[off + i] : 0u;
        const bool active = s != 0 && s != KU_AMBIG;
        unsigned long long todo = __ballot(active);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t ls = ku_wave_bcast(s, (uint32_t)leader);
          const unsigned long long same = __ballot(active && s == ls);
          if ((int)(tid & 63u) == leader) table_insert(ls, (uint32_t)__popcll(same));
          todo &= ~same;
        }
      }
    }
    gsync();
    // every distinct taxon of the read is in the list now (appended by the lane that claimed its table cell -- no
    // scan over the table, which for the 16384-cell table of 10 kbp reads cost as much as the lookups)
    const uint32_t n_list = s_n_list;
    if (n_list > 0) {
      // ---- score(t) = sum of hit counts on t's root path (krakenutil.cpp:157-177)
      uint32_t my_max = 0;
      for (uint32_t e = tid; e < n_list; e += GROUP) {
        uint32_t pos = MODE == 2 ? t_list[e] : (uint32_t)t_list16[e];
        const uint32_t sl = t_key[pos] - 1;
        uint32_t score = 0;
        for (uint32_t i = tax.slot_anc_off[sl], i_end = tax.slot_anc_off[sl + 1]; i < i_end; ++i) {
          const uint32_t s = tax.slot_anc[i];
          uint32_t h = (s * 2654435761u) >> (32 - cap_log2);
          for (;;) {
            uint32_t cur = t_key[h];
            if (cur == s + 1) {
              score += MODE == 2 ? t_cnt[h] : (t_cnt[h] & 0xffffu);
              break;
            }
            if (cur == 0) break;
            h = (h + 1) & (cap - 1);
          }
        }
        if (MODE == 2) t_score[pos] = score; else atomicAdd(&t_cnt[pos], score << 16);
        my_max = max(my_max, score);
      }
      const uint32_t max_score = ku_group_max<GROUP>(my_max, s_red);
      gsync();
      // ---- winner; ties -> fold lca() over the tied taxa in ascending taxid (= slot) order
      uint32_t last = 0;  // slots are >= 1
      bool first = true;
      for (;;) {
        uint32_t my_min = 0xFFFFFFFFu;
        for (uint32_t e = tid; e < n_list; e += GROUP) {
          uint32_t pos = MODE == 2 ? t_list[e] : (uint32_t)t_list16[e];
          uint32_t sc = MODE == 2 ? t_score[pos] : (t_cnt[pos] >> 16);
          uint32_t s = t_key[pos] - 1;
          if (sc == max_score && s > last) my_min = min(my_min, s);
        }
        uint32_t next = ku_group_min<GROUP>(my_min, s_red);
        if (next == 0xFFFFFFFFu) break;
        if (tid == 0) {
          uint32_t node = tax.slot_node[next];
          s_bcast = first ? node : ku_lca_nodes(tax.node_parent, s_bcast, node);
        }
        first = false;
        last = next;
        gsync();
      }
      gsync();
      call_node = s_bcast;
      gsync();
      // ---- reset the used entries for the next read
      for (uint32_t e = tid; e < n_list; e += GROUP) {
        uint32_t pos = MODE == 2 ? t_list[e] : (uint32_t)t_list16[e];
        t_key[pos] = 0;
        t_cnt[pos] = 0;
        if (MODE == 2) t_score[pos] = 0;
      }
    }
    }  // !resolved
    if (tid == 0) {
      calls[r] = (MODE == 0 && resolved) ? uni_taxid : tax.node_taxid[call_node];
      if (hits_out) hits_out[r] = 0;
      if (do_counts) ku_ct_add<RCT>(s_ctk, s_ctc, &s_ctu, call_node, 1, cnt.n_reads);  // incrementReadCount (classify.cpp:968)
    }
    // ---- slot -> taxid for the hit string
    if (!keep_slots) {
      if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          uint32_t i = c * 64 + tid;
          if (i < n && v[c] != 0) taxa[off + i] = resolved ? uni_taxid : tax.slot_taxid[v[c]];  // one taxon: no gather
        }
      } else {
        for (uint32_t i = tid; i < n; i += GROUP) {
          uint32_t s = taxa[off + i];
          if (s != 0 && s != KU_AMBIG) taxa[off + i] = tax.slot_taxid[s];
        }
      }
    }
    gsync();
  }
  __syncthreads();
  ku_ct_flush<RCT>(s_ctk, s_ctc, cnt.n_reads);
}

// canonical k-mer straight from ASCII (quick mode only; positions known non-ambiguous)
__device__ __forceinline__ uint64_t ku_canon_from_ascii(const uint8_t *p, uint32_t k) {
  uint64_t fwd = 0;
  for (uint32_t i = 0; i < k; ++i) {
    uint32_t c = p[i] & 0xDFu;
    fwd = (fwd << 2) | (((c >> 1) ^ (c >> 2)) & 3u);
  }
  uint64_t rc = ku_revcomp64(fwd, k);
  return fwd < rc ? fwd : rc;
}

// Exact distinct k-mer counting (classifyExact = classify built with EXACT_COUNTING, classify.cpp:46-53: the per-taxon
// container is a hash set of k-mers instead of a HyperLogLog sketch).  A canonical k-mer has exactly one database
// value, so one global open-addressing set of k-mers + a "first insertion" counter per slot is the same thing as one
// set per taxon.  Runs between the lookup and the resolve stage (taxa[] still holds slot ids), one wave per read.
__global__ __launch_bounds__(64) void ku_exact_kernel(uint32_t k, const uint8_t *__restrict__ seqs,
                                                      const uint64_t *__restrict__ seq_off,
                                                      const uint32_t *__restrict__ seq_len, uint64_t n_reads,
                                                      const uint32_t *__restrict__ taxa, unsigned long long *set,
                                                      uint64_t mask, unsigned long long *unique, uint32_t *overflow,
                                                      uint32_t quick_min_hits) {
  const uint32_t tid = threadIdx.x;
  for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const uint32_t len = seq_len[r];
    const uint32_t n = len >= k ? len - k + 1 : 0;
    const uint64_t off = seq_off[r];
    uint32_t stop = n;
    if (quick_min_hits) {  // quick mode counts the k-mers scanned up to and including the min_hits-th hit (ku_quick_kernel)
      uint32_t total = 0;
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + tid;
        const uint32_t s = i < n ? taxa[off + i] : 0;
        const bool hit = s != 0 && s != KU_AMBIG;
        const unsigned long long hm = __ballot(hit);
        const uint32_t c = (uint32_t)__popcll(hm);
        if (total + c >= quick_min_hits) {
          const uint32_t need = quick_min_hits - total;
          const bool is_stop = hit && (uint32_t)__popcll(hm & ((1ull << tid) - 1ull)) + 1 == need;
          stop = base + (uint32_t)__ffsll((long long)__ballot(is_stop));
          break;
        }
        total += c;
      }
    }
    for (uint32_t i = tid; i < stop; i += 64) {
      const uint32_t s = taxa[off + i];
      if (s == KU_AMBIG || s == KU_FOREIGN_MARK) continue;  // (several GPUs: a k-mer another rank owns is that rank's to count)
      const uint64_t canon = ku_canon_from_ascii(seqs + off + i, k);
      const unsigned long long key = canon + 1;  // 0 marks an empty cell
      uint64_t h = ku_fmix64(canon) & mask;
      bool done = false;
      for (uint32_t probe = 0; probe < 4096 && !done; ++probe) {
        unsigned long long cur = __hip_atomic_load(&set[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
          cur = atomicCAS(&set[h], 0ull, key);
          if (cur == 0) { atomicAdd(&unique[s], 1ull); done = true; }
        }
        if (cur == key) done = true;
        h = (h + 1) & mask;
      }
      if (!done) atomicExch(overflow, 1u);
    }
  }
}
int ku_launch_exact(uint32_t k, const uint8_t *d_seqs, const uint64_t *d_seq_off, const uint32_t *d_seq_len, uint64_t n_reads,
                    const uint32_t *d_taxa, unsigned long long *d_set, uint64_t mask, unsigned long long *d_unique,
                    uint32_t *d_overflow, int n_cu, hipStream_t stream, uint32_t quick_min_hits) {
  if (n_reads == 0) return KU_OK;
  const uint64_t cap = (uint64_t)n_cu * 32;
  hipLaunchKernelGGL(ku_exact_kernel, dim3((unsigned)(n_reads < cap ? n_reads : cap)), dim3(64), 0, stream, k, d_seqs,
                     d_seq_off, d_seq_len, n_reads, d_taxa, d_set, mask, d_unique, d_overflow, quick_min_hits);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// Quick mode (classify.cpp:943-944,962-963): stop at the min_hits-th hit, call =
// its taxon; only the k-mers scanned up to and including that one are counted.
__global__ __launch_bounds__(64) void ku_quick_kernel(KuTaxDev tax, KuCountsDev cnt, uint32_t k,
                                                      const uint8_t *__restrict__ seqs,
                                                      const uint64_t *__restrict__ seq_off,
                                                      const uint32_t *__restrict__ seq_len, uint64_t n_reads,
                                                      uint32_t flags, uint32_t min_hits,
                                                      uint32_t *__restrict__ calls, uint32_t *__restrict__ taxa,
                                                      uint32_t *__restrict__ hits_out) {
  __shared__ uint32_t s_ctk[KU_CT_CAP];
  __shared__ uint32_t s_ctc[KU_CT_CAP];
  __shared__ uint32_t s_ckk[KU_CT_CAP];
  __shared__ uint32_t s_ckc[KU_CT_CAP];
  __shared__ uint32_t s_ctu, s_cku;
  const uint32_t tid = threadIdx.x;
  const bool do_counts = !(flags & KU_F_NO_COUNTS);
  const bool keep_slots = (flags & KU_F_KEEP_SLOTS) != 0;
  ku_ct_clear(s_ctk, s_ctc, &s_ctu);
  ku_ct_clear(s_ckk, s_ckc, &s_cku);
  __syncthreads();
  for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const uint32_t len = seq_len[r];
    const uint32_t n = len >= k ? len - k + 1 : 0;
    const uint64_t off = seq_off[r];
    __syncthreads();
    ku_ct_maybe_flush(s_ctk, s_ctc, &s_ctu, cnt.n_reads);
    ku_ct_maybe_flush(s_ckk, s_ckc, &s_cku, cnt.n_kmers);
    uint32_t total = 0, call_slot = 0;
    uint32_t stop = n;  // exclusive end of the scanned prefix
    for (uint32_t base = 0; base < n; base += 64) {
      uint32_t i = base + tid;
      uint32_t s = i < n ? taxa[off + i] : 0;
      bool hit = s != 0 && s != KU_AMBIG;
      unsigned long long mask = __ballot(hit);
      uint32_t c = (uint32_t)__popcll(mask);
      if (total + c >= min_hits) {
        uint32_t need = min_hits - total;  // >= 1
        uint32_t before = (uint32_t)__popcll(mask & ((1ull << tid) - 1ull));
        bool is_stop = hit && before + 1 == need;
        unsigned long long sm = __ballot(is_stop);
        uint32_t lane = (uint32_t)__ffsll((long long)sm) - 1;
        call_slot = ku_wave_bcast(s, lane);
        stop = base + lane + 1;
        total = min_hits;
        break;
      }
      total += c;
    }
    if (do_counts) {
      for (uint32_t i = tid; i < stop; i += 64) {
        uint32_t s = taxa[off + i];
        if (s != KU_AMBIG) {
          ku_hll_update(cnt.registers, s, ku_fmix64(ku_canon_from_ascii(seqs + off + i, k)));
          ku_ct_add(s_ckk, s_ckc, &s_cku, s, 1, cnt.n_kmers);
        }
      }
    }
    uint32_t call_node = (total >= min_hits) ? tax.slot_node[call_slot] : 0;
    if (tid == 0) {
      calls[r] = tax.node_taxid[call_node];
      if (hits_out) hits_out[r] = total;
      if (do_counts) ku_ct_add(s_ctk, s_ctc, &s_ctu, call_node, 1, cnt.n_reads);
    }
    if (!keep_slots) {
      for (uint32_t i = tid; i < n; i += 64) {
        uint32_t s = taxa[off + i];
        if (s != 0 && s != KU_AMBIG) taxa[off + i] = tax.slot_taxid[s];
      }
    }
  }
  __syncthreads();
  ku_ct_flush(s_ctk, s_ctc, cnt.n_reads);
  ku_ct_flush(s_ckk, s_ckc, cnt.n_kmers);
}

// Quick mode of a CHUNKED run (classify.cpp:686-737): the hits are counted over the merged per-k-mer taxa up to
// min_hits, every unambiguous k-mer of the read was booked by the chunk passes, and the call is the taxon of the
// read's LAST unambiguous k-mer when min_hits was reached (0 otherwise) -- the reference's loop leaves that value behind.
__global__ __launch_bounds__(64) void ku_quick_chunked_kernel(KuTaxDev tax, KuCountsDev cnt, uint32_t k,
                                                              const uint64_t *__restrict__ seq_off,
                                                              const uint32_t *__restrict__ seq_len, uint64_t n_reads,
                                                              uint32_t flags, uint32_t min_hits, uint32_t *__restrict__ calls,
                                                              uint32_t *__restrict__ taxa, uint32_t *__restrict__ hits_out) {
  __shared__ uint32_t s_ctk[KU_CT_CAP];
  __shared__ uint32_t s_ctc[KU_CT_CAP];
  __shared__ uint32_t s_ctu;
  const uint32_t tid = threadIdx.x;
  const bool do_counts = !(flags & KU_F_NO_COUNTS);
  const bool keep_slots = (flags & KU_F_KEEP_SLOTS) != 0;
  ku_ct_clear(s_ctk, s_ctc, &s_ctu);
  __syncthreads();
  for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const uint32_t len = seq_len[r];
    const uint32_t n = len >= k ? len - k + 1 : 0;
    const uint64_t off = seq_off[r];
    __syncthreads();
    ku_ct_maybe_flush(s_ctk, s_ctc, &s_ctu, cnt.n_reads);
    uint32_t total = 0, last_slot = 0;
    for (uint32_t base = 0; base < n; base += 64) {
      const uint32_t i = base + tid;
      const uint32_t s = i < n ? taxa[off + i] : KU_AMBIG;
      total += (uint32_t)__popcll(__ballot(s != 0 && s != KU_AMBIG));
      const unsigned long long clean = __ballot(s != KU_AMBIG);
      if (clean) last_slot = ku_wave_bcast(s, 63u - (uint32_t)__builtin_clzll(clean));
    }
    const uint32_t hits = total < min_hits ? total : min_hits;
    const uint32_t call_node = hits >= min_hits && last_slot ? tax.slot_node[last_slot] : 0u;
    if (tid == 0) {
      calls[r] = tax.node_taxid[call_node];
      if (hits_out) hits_out[r] = hits;
      if (do_counts) ku_ct_add(s_ctk, s_ctc, &s_ctu, call_node, 1, cnt.n_reads);
    }
    if (!keep_slots)
      for (uint32_t i = tid; i < n; i += 64) {
        const uint32_t s = taxa[off + i];
        if (s != 0 && s != KU_AMBIG) taxa[off + i] = tax.slot_taxid[s];
      }
  }
  __syncthreads();
  ku_ct_flush(s_ctk, s_ctc, cnt.n_reads);
}
int ku_launch_quick_chunked(const KuTaxDev &tax, const KuCountsDev &cnt, uint32_t k, const uint64_t *d_seq_off,
                            const uint32_t *d_seq_len, uint64_t n_reads, uint32_t flags, uint32_t min_hits, uint32_t *d_calls,
                            uint32_t *d_taxa, uint32_t *d_hits, int n_cu, hipStream_t stream) {
  if (n_reads == 0) return KU_OK;
  const uint64_t mb = (uint64_t)n_cu * 16;
  hipLaunchKernelGGL(ku_quick_chunked_kernel, dim3((unsigned)(n_reads < mb ? n_reads : mb)), dim3(64), 0, stream, tax, cnt, k,
                     d_seq_off, d_seq_len, n_reads, flags, min_hits ? min_hits : 1u, d_calls, d_taxa, d_hits);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

static inline uint32_t ku_ceil_log2(uint64_t v) {
  uint32_t l = 0;
  while ((1ull << l) < v) ++l;
  return l;
}

uint64_t ku_resolve_workspace_bytes(uint32_t max_read_len, uint32_t k, int n_cu) {
  uint32_t n = max_read_len >= k ? max_read_len - k + 1 : 0;
  if (n <= (uint32_t)KuResolveCfg<1>::MAX_N) return 0;
  uint32_t cap_log2 = ku_ceil_log2(2ull * n);
  uint64_t blocks = (uint64_t)(n_cu < 64 ? n_cu : 64);
  return blocks * 4ull * (1ull << cap_log2) * sizeof(uint32_t);
}

int ku_launch_resolve(const KuDbDev &db, const KuTaxDev &tax, const KuCountsDev &cnt, const uint8_t *d_seqs,
                      const uint64_t *d_seq_off, const uint32_t *d_seq_len, uint64_t n_reads, uint32_t flags,
                      uint32_t min_hits, uint32_t max_read_len, uint32_t *d_calls, uint32_t *d_taxa,
                      uint32_t *d_hits, void *d_workspace, uint64_t workspace_bytes, int n_cu,
                      hipStream_t stream) {
  if (n_reads == 0) return KU_OK;
  const uint32_t k = db.k;
  const uint32_t max_n = max_read_len >= k ? max_read_len - k + 1 : 0;
  if (flags & KU_F_QUICK) {
    uint64_t mb = (uint64_t)n_cu * 16;
    unsigned grid = (unsigned)(n_reads < mb ? n_reads : mb);
    hipLaunchKernelGGL(ku_quick_kernel, dim3(grid), dim3(64), 0, stream, tax, cnt, k, d_seqs, d_seq_off, d_seq_len,
                       n_reads, flags, min_hits ? min_hits : 1u, d_calls, d_taxa, d_hits);
    return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
  }
  {  // MODE 0: every read with n <= 384 (incl. reads shorter than k).  24 one-wave blocks per CU all fit at once
     // (6 KB of LDS each); 32 would run as 26 + a straggling 6 and measured slower
    uint64_t mb = (uint64_t)n_cu * 24;
    unsigned grid = (unsigned)(n_reads < mb ? n_reads : mb);
    size_t lds = (2u << KuResolveCfg<0>::CAP_LOG2) * 4 + KuResolveCfg<0>::MAX_N * 2 + 64;
    hipLaunchKernelGGL(ku_resolve_kernel<0>, dim3(grid), dim3(64), lds, stream, tax, cnt, k, d_seq_off, d_seq_len,
                       n_reads, 0u, flags, d_calls, d_taxa, d_hits, (uint32_t *)nullptr, 0u);
  }
  if (max_n > (uint32_t)KuResolveCfg<0>::MAX_N) {
    uint64_t mb = (uint64_t)n_cu;
    unsigned grid = (unsigned)(n_reads < mb ? n_reads : mb);
    size_t lds = (2u << KuResolveCfg<1>::CAP_LOG2) * 4 + KuResolveCfg<1>::MAX_N * 2 + 64;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ku_resolve_kernel<1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(ku_resolve_kernel<1>, dim3(grid), dim3(KuResolveCfg<1>::GROUP), lds, stream, tax, cnt, k, d_seq_off, d_seq_len,
                       n_reads, (uint32_t)KuResolveCfg<0>::MAX_N + 1, flags, d_calls, d_taxa, d_hits,
                       (uint32_t *)nullptr, 0u);
  }
  if (max_n > (uint32_t)KuResolveCfg<1>::MAX_N) {
    uint32_t cap_log2 = ku_ceil_log2(2ull * max_n);
    uint64_t blocks = (uint64_t)(n_cu < 64 ? n_cu : 64);
    if (workspace_bytes < blocks * 4ull * (1ull << cap_log2) * sizeof(uint32_t) || !d_workspace) return KU_ENOMEM;
    unsigned grid = (unsigned)(n_reads < blocks ? n_reads : blocks);
    hipLaunchKernelGGL(ku_resolve_kernel<2>, dim3(grid), dim3(256), 0, stream, tax, cnt, k, d_seq_off, d_seq_len,
                       n_reads, (uint32_t)KuResolveCfg<1>::MAX_N + 1, flags, d_calls, d_taxa, d_hits,
                       (uint32_t *)d_workspace, cap_log2);
  }
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// ----------------------------------------------------------------------------
// ku_rle_kernel: run-length encoding of the per-k-mer codes (hitlist_string, classify.cpp:826-861)
// ----------------------------------------------------------------------------
// One wave per group of `chunk` consecutive reads (64 for short reads, fewer for long ones).  A run is stored as {code,
// start index}; its length is the next run's start (or the read's k-mer count) minus its own -- the host formatter
// recovers it.  Pass 1 counts the runs of the group's reads (lane j keeps read j's count), a wave prefix sum turns the
// counts into places and ONE add to the global bump counter claims the group's space (one add per read had the whole
// grid queue on a single address: 12 ns per read, several times the classification itself); pass 2 writes.  `runs` is
// dense, in read order within a group; (run_off, run_cnt) say where.
__global__ __launch_bounds__(256) void ku_rle_kernel(const uint32_t *__restrict__ taxa, uint32_t k,
                                                     const uint64_t *__restrict__ seq_off,
                                                     const uint32_t *__restrict__ seq_len, uint64_t n_reads, uint32_t chunk,
                                                     uint2 *runs, unsigned long long runs_cap, unsigned long long *counter,
                                                     uint64_t *run_off, uint32_t *run_cnt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t n_groups = (n_reads + chunk - 1) / chunk;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
  for (uint64_t g = wave; g < n_groups; g += n_waves) {
    const uint64_t r0 = g * chunk;
    const uint32_t n_here = (uint32_t)(n_reads - r0 < chunk ? n_reads - r0 : chunk);
    const uint64_t my_r = r0 + lane;
    const uint32_t my_len = lane < n_here ? seq_len[my_r] : 0u;
    const uint64_t my_off = lane < n_here ? seq_off[my_r] : 0ull;
    // pass 1: count the run starts of every read of the group
    uint32_t my_total = 0;
    for (uint32_t j = 0; j < n_here; ++j) {
      const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)my_len, (int)j);
      const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_off >> 32), (int)j) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_off, (int)j);
      const uint32_t n = len >= k ? len - k + 1 : 0;
      uint32_t total = 0, carry = 0;
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t v = i < n ? taxa[off + i] : 0u;
        uint32_t prev = ku_wave_up1(v);
        if (lane == 0) prev = carry;
        const bool start = i < n && (i == 0 || v != prev);
        total += (uint32_t)__popcll(__ballot(start));
        carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
      }
      if (lane == j) my_total = total;
    }
    // places: exclusive prefix sum of the counts over the lanes
    uint32_t incl = my_total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= (uint32_t)d) incl += up;
    }
    const uint32_t group_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    unsigned long long basep = 0;
    if (lane == 0 && group_total) basep = atomicAdd(counter, (unsigned long long)group_total);
    basep = __shfl(basep, 0, 64);
    const bool fits = basep + group_total <= runs_cap;
    if (lane < n_here) {
      run_off[my_r] = basep + (incl - my_total);
      run_cnt[my_r] = fits ? my_total : 0xFFFFFFFFu;  // overflow marker: the caller retries with a larger array
    }
    if (!fits) continue;
    // pass 2: write {code, start}
    const uint32_t my_excl = incl - my_total;
    for (uint32_t j = 0; j < n_here; ++j) {
      const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)my_len, (int)j);
      const uint64_t off = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_off >> 32), (int)j) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_off, (int)j);
      const uint64_t at = basep + (uint32_t)__builtin_amdgcn_readlane((int)my_excl, (int)j);
      const uint32_t n = len >= k ? len - k + 1 : 0;
      uint32_t done = 0, carry = 0;
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t v = i < n ? taxa[off + i] : 0u;
        uint32_t prev = ku_wave_up1(v);
        if (lane == 0) prev = carry;
        const bool start = i < n && (i == 0 || v != prev);
        const unsigned long long m = __ballot(start);
        if (start) runs[at + done + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(v, i);
        done += (uint32_t)__popcll(m);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
      }
    }
  }
}
int ku_launch_rle(const uint32_t *d_taxa, uint32_t k, const uint64_t *d_seq_off, const uint32_t *d_seq_len,
                  uint64_t n_reads, uint64_t n_bytes, void *d_runs, uint64_t runs_cap, unsigned long long *d_counter,
                  uint64_t *d_run_off, uint32_t *d_run_cnt, int n_cu, hipStream_t stream) {
  if (hipMemsetAsync(d_counter, 0, 8, stream) != hipSuccess) return KU_EHIP;
  if (n_reads == 0) return KU_OK;
  // reads per wave: 64 short ones, fewer long ones (about 8 K positions per group, and enough groups to fill the chip)
  uint32_t chunk = 64;
  const uint64_t avg = n_bytes / n_reads + 1;
  while (chunk > 1 && ((uint64_t)chunk * avg > 8192 || (n_reads + chunk - 1) / chunk < (uint64_t)n_cu * 8)) chunk >>= 1;
  const uint64_t n_groups = (n_reads + chunk - 1) / chunk, want = (n_groups + 3) / 4, cap = (uint64_t)n_cu * 16;
  hipLaunchKernelGGL(ku_rle_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, stream, d_taxa, k, d_seq_off,
                     d_seq_len, n_reads, chunk, (uint2 *)d_runs, (unsigned long long)runs_cap, d_counter, d_run_off, d_run_cnt);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// ----------------------------------------------------------------------------
// small utility kernels
// ----------------------------------------------------------------------------
__global__ void ku_max_len_kernel(const uint32_t *__restrict__ len, uint64_t n, uint32_t *out) {
  uint32_t m = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    m = max(m, len[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
int ku_launch_max_len(const uint32_t *d_seq_len, uint64_t n_reads, uint32_t *d_out, hipStream_t stream) {
  if (hipMemsetAsync(d_out, 0, 4, stream) != hipSuccess) return KU_EHIP;
  if (n_reads == 0) return KU_OK;
  uint64_t nb = (n_reads + 255) / 256;
  hipLaunchKernelGGL(ku_max_len_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, stream, d_seq_len,
                     n_reads, d_out);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// element-wise merges (multi-GPU driver, same-process exchange): "non-zero wins" on per-k-mer slots (classify.cpp:445-452)
// is a max because exactly one shard is non-zero and KU_AMBIG is the same on all; HLL registers merge by max, counters add
__global__ void ku_merge_max_u32_kernel(uint32_t *dst, const uint32_t *__restrict__ src, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = max(dst[i], src[i]);
}
__global__ void ku_merge_max_u8_kernel(uint8_t *dst, const uint8_t *__restrict__ src, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = dst[i] > src[i] ? dst[i] : src[i];
}
__global__ void ku_merge_add_u64_kernel(unsigned long long *dst, const unsigned long long *__restrict__ src, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] += src[i];
}
static unsigned ku_merge_grid(uint64_t n) {
  uint64_t nb = (n + 255) / 256;
  return (unsigned)(nb < 8192 ? (nb ? nb : 1) : 8192);
}
int ku_launch_merge_max_u32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t stream) {
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_merge_max_u32_kernel, dim3(ku_merge_grid(n)), dim3(256), 0, stream, dst, src, n);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_merge_max_u8(uint8_t *dst, const uint8_t *src, uint64_t n, hipStream_t stream) {
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_merge_max_u8_kernel, dim3(ku_merge_grid(n)), dim3(256), 0, stream, dst, src, n);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
int ku_launch_merge_add_u64(unsigned long long *dst, const unsigned long long *src, uint64_t n, hipStream_t stream) {
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_merge_add_u64_kernel, dim3(ku_merge_grid(n)), dim3(256), 0, stream, dst, src, n);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// every `from` in a uint32 array becomes `to` (the marks of positions other ranks own, before the exchange)
__global__ void ku_replace_u32_kernel(uint32_t *p, uint64_t n, uint32_t from, uint32_t to) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (p[i] == from) p[i] = to;
}
int ku_launch_replace_u32(uint32_t *p, uint64_t n, uint32_t from, uint32_t to, hipStream_t stream) {
  if (n == 0) return KU_OK;
  hipLaunchKernelGGL(ku_replace_u32_kernel, dim3(ku_merge_grid(n)), dim3(256), 0, stream, p, n, from, to);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// on-disk pairs with key_len < 8 (k < 29) -> the fixed 12-byte record the kernels use
__global__ void ku_repack_kernel(const uint8_t *__restrict__ raw, uint64_t n, uint32_t key_len,
                                 uint32_t *__restrict__ out) {
  const uint32_t ps = key_len + 4;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t *p = raw + i * ps;
    uint64_t key = 0;
    for (uint32_t b = 0; b < key_len; ++b) key |= (uint64_t)p[b] << (8 * b);
    uint32_t val = 0;
    for (uint32_t b = 0; b < 4; ++b) val |= (uint32_t)p[key_len + b] << (8 * b);
    out[3 * i] = (uint32_t)key;
    out[3 * i + 1] = (uint32_t)(key >> 32);
    out[3 * i + 2] = val;
  }
}
int ku_launch_repack(const uint8_t *d_raw, uint64_t n_pairs, uint32_t key_len, uint32_t *d_pairs,
                     hipStream_t stream) {
  if (n_pairs == 0) return KU_OK;
  uint64_t nb = (n_pairs + 255) / 256;
  hipLaunchKernelGGL(ku_repack_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream, d_raw, n_pairs,
                     key_len, d_pairs);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// distinct DB values: 2^32-bit bitmap (512 MiB scratch), then compaction
__global__ void ku_mark_values_kernel(const uint32_t *__restrict__ pairs, uint64_t n, uint32_t *bitmap) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t v = pairs[3 * i + 2];
    uint32_t bit = 1u << (v & 31u);
    if (!(bitmap[v >> 5] & bit)) atomicOr(&bitmap[v >> 5], bit);
  }
}
int ku_launch_mark_values(const uint32_t *d_pairs, uint64_t n_pairs, uint32_t *d_bitmap, hipStream_t stream) {
  if (n_pairs == 0) return KU_OK;
  uint64_t nb = (n_pairs + 255) / 256;
  hipLaunchKernelGGL(ku_mark_values_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream, d_pairs,
                     n_pairs, d_bitmap);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
__global__ void ku_collect_values_kernel(const uint32_t *__restrict__ bitmap, uint32_t *out, uint32_t cap,
                                         uint32_t *count) {
  const uint64_t n_words = 1ull << 27;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words;
       i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t wv = bitmap[i];
    while (wv) {
      uint32_t b = (uint32_t)__ffs((int)wv) - 1;
      wv &= wv - 1;
      uint32_t e = atomicAdd(count, 1u);
      if (e < cap) out[e] = (uint32_t)(i << 5) | b;
    }
  }
}
int ku_launch_collect_values(const uint32_t *d_bitmap, uint32_t *d_out, uint32_t cap, uint32_t *d_count,
                             hipStream_t stream) {
  hipLaunchKernelGGL(ku_collect_values_kernel, dim3(4096), dim3(256), 0, stream, d_bitmap, d_out, cap, d_count);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// raw taxid -> slot id (rank in the ascending slot table; 0 stays 0)
__global__ void ku_remap_values_kernel(uint32_t *pairs, uint64_t n, const uint32_t *__restrict__ slot_taxid,
                                       uint32_t n_slots, uint32_t *err) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t v = pairs[3 * i + 2];
    if (v == 0) continue;
    uint32_t lo = 1, hi = n_slots;  // slot_taxid[0] = 0
    while (lo < hi) {
      uint32_t mid = (lo + hi) >> 1;
      if (slot_taxid[mid] < v) lo = mid + 1; else hi = mid;
    }
    if (lo < n_slots && slot_taxid[lo] == v) pairs[3 * i + 2] = lo;
    else { pairs[3 * i + 2] = 0; atomicAdd(err, 1u); }
  }
}
int ku_launch_remap_values(uint32_t *d_pairs, uint64_t n_pairs, const uint32_t *d_slot_taxid, uint32_t n_slots,
                           uint32_t *d_err, hipStream_t stream) {
  if (n_pairs == 0) return KU_OK;
  uint64_t nb = (n_pairs + 255) / 256;
  hipLaunchKernelGGL(ku_remap_values_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream, d_pairs,
                     n_pairs, d_slot_taxid, n_slots, d_err);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

// count_taxons (krakendb.cpp:90-113) over slot ids
__global__ __launch_bounds__(256) void ku_count_slots_kernel(const uint32_t *__restrict__ pairs, uint64_t n,
                                                             unsigned long long *counts) {
  __shared__ uint32_t s_ctk[KU_CT_CAP];
  __shared__ uint32_t s_ctc[KU_CT_CAP];
  __shared__ uint32_t s_ctu;
  ku_ct_clear(s_ctk, s_ctc, &s_ctu);
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = blockIdx.x * (uint64_t)blockDim.x; base < n; base += stride) {  // block-uniform trip count
    uint64_t i = base + threadIdx.x;
    if (i < n) ku_ct_add(s_ctk, s_ctc, &s_ctu, pairs[3 * i + 2], 1, counts);
    __syncthreads();
    ku_ct_maybe_flush(s_ctk, s_ctc, &s_ctu, counts);
  }
  __syncthreads();
  ku_ct_flush(s_ctk, s_ctc, counts);
}
int ku_launch_count_slots(const uint32_t *d_pairs, uint64_t n_pairs, unsigned long long *d_counts, uint32_t,
                          hipStream_t stream) {
  if (n_pairs == 0) return KU_OK;
  uint64_t nb = (n_pairs + 255) / 256;
  hipLaunchKernelGGL(ku_count_slots_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, stream, d_pairs,
                     n_pairs, d_counts);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
