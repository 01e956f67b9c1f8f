// ku_route.hip -- the owner's side of the owner-routed multi-GPU path (ku_mgpu.cpp).
//
// The database is sharded by minimizer-bin range over the GPUs (KrakenDB::prepare_chunking laid out in space,
// krakendb.cpp:430-526).  Round 2 had every rank scan every read and exchanged 4-byte slots per base position.  Here a
// rank scans only its own slice of the reads (ku_lookup_kernel<3,...>, ku_kernels.hip) and sends each unambiguous
// canonical k-mer to the rank that owns its bin, together with the size-independent half of its bucket hash (12 bytes per
// k-mer); this file holds what the owner does with them -- the bucket probe of kmer_query (krakendb.cpp:250-321) and
// ReadCounts::add_kmer (classify.cpp:939: HLL register + n_kmers, owner-computes, misses under taxon 0) -- and the
// scatter of the returned slots into the sender's per-k-mer array.  The k-mers of a read arrive together and in order
// (the scan fills the queues tile by tile), so neighbouring lanes still share bucket lines.
#include "ku_device.h"

// one lane per routed k-mer: ent[3 i .. 3 i + 2] = {k-mer low, k-mer high, bucket prehash} -> slots[i]
template <bool DO_COUNTS>
__global__ __launch_bounds__(256) void ku_route_probe_kernel(KuDbDev db, KuCountsDev cnt, const uint32_t *__restrict__ ent, uint64_t n,
                                                             uint32_t *__restrict__ slots) {
  __shared__ uint32_t s_ctk[KU_CT_CAP];
  __shared__ uint32_t s_ctc[KU_CT_CAP];
  __shared__ uint32_t s_ctu;
  if (DO_COUNTS) ku_ct_clear(s_ctk, s_ctc, &s_ctu);
  __syncthreads();
  const uint32_t *tab = reinterpret_cast<const uint32_t *>(db.table);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = blockIdx.x * (uint64_t)blockDim.x; base < n; base += stride) {  // block-uniform trip count
    const uint64_t i = base + threadIdx.x;
    bool ok = i < n;
    uint32_t slot = 0;
    uint64_t canon = 0, hh = 0;
    uint32_t lo = 0, hi = 0, g = 0;
    if (ok) {
      lo = ent[3 * i]; hi = ent[3 * i + 1]; g = ent[3 * i + 2];
      if (lo == KU_ROUTE_NULL && hi == KU_ROUTE_NULL) {  // padding of a sender's chunk: nothing to look up or to book
        slots[i] = 0;
        ok = false;
      }
    }
    if (ok) {
      canon = ((uint64_t)hi << 32) | lo;
      hh = ku_fmix64(canon);
      const uint32_t tag = ku_table_tag(hh);
      const uint32_t *lp = tab + (uint64_t)__umulhi(g, (uint32_t)db.n_lines) * KU_LINE_DWORDS;
      for (;;) {  // header round trip, candidate entries of the line, spilled buckets go on in the next line
        const uint4 h4 = *reinterpret_cast<const uint4 *>(lp);
        uint32_t cand = ku_tag_matches(h4, tag);
        bool found = false;
        while (cand) {
          const KuPair pr = *reinterpret_cast<const KuPair *>(lp + KU_LINE_ENTRY0 + 3 * __builtin_ctz(cand));
          if ((((uint64_t)pr.key_hi << 32) | pr.key_lo) == canon) { slot = pr.slot; found = true; break; }
          cand &= cand - 1;
        }
        if (found || !ku_line_spilled(h4)) break;
        lp += KU_LINE_DWORDS;
        if (lp == tab + db.n_lines * KU_LINE_DWORDS) lp = tab;
      }
      slots[i] = slot;
    }
    if (DO_COUNTS) {
      if (ok) ku_hll_update(cnt.registers, slot, hh);
      // neighbouring k-mers mostly carry one slot: when the whole wave agrees one lane books them all
      const unsigned long long booked = __ballot(ok);
      if (booked) {
        const uint32_t lead = (uint32_t)__ffsll((long long)booked) - 1;
        const uint32_t s0 = ku_wave_bcast(slot, lead);
        if (__ballot(ok && slot == s0) == booked) {
          if ((threadIdx.x & 63u) == lead) ku_ct_add(s_ctk, s_ctc, &s_ctu, s0, (uint32_t)__popcll(booked), cnt.n_kmers);
        } else if (ok) {
          ku_ct_add(s_ctk, s_ctc, &s_ctu, slot, 1, cnt.n_kmers);
        }
      }
      __syncthreads();
      ku_ct_maybe_flush(s_ctk, s_ctc, &s_ctu, cnt.n_kmers);
    }
  }
  if (DO_COUNTS) {
    __syncthreads();
    ku_ct_flush(s_ctk, s_ctc, cnt.n_kmers);
  }
}

// the slots the owners sent back, into the per-k-mer array at the positions the scan recorded
__global__ void ku_route_scatter_kernel(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ slots, uint64_t n, uint32_t *taxa) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t p = pos[i];
    if (p != KU_ROUTE_NULL) taxa[p] = slots[i];
  }
}

int ku_launch_route_probe(const KuDbDev &db, const KuCountsDev &cnt, const uint32_t *d_ent, uint64_t n, uint32_t *d_slots, bool do_counts,
                          int n_cu, hipStream_t stream) {
  if (n == 0) return KU_OK;
  if (!db.table) return KU_EINVAL;
  const uint64_t want = (n + 255) / 256, cap = (uint64_t)n_cu * 16;
  const dim3 grid((unsigned)(want < cap ? want : cap)), block(256);
  if (do_counts) hipLaunchKernelGGL(ku_route_probe_kernel<true>, grid, block, 0, stream, db, cnt, d_ent, n, d_slots);
  else hipLaunchKernelGGL(ku_route_probe_kernel<false>, grid, block, 0, stream, db, cnt, d_ent, n, d_slots);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_route_scatter(const uint32_t *d_pos, const uint32_t *d_slots, uint64_t n, uint32_t *d_taxa, hipStream_t stream) {
  if (n == 0) return KU_OK;
  const uint64_t want = (n + 255) / 256;
  hipLaunchKernelGGL(ku_route_scatter_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, stream, d_pos, d_slots, n, d_taxa);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
