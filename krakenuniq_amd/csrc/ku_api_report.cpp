// ku_api_report.cpp -- C ABI: counts export and the report from the resident state (kernels: ku_report.hip)
#include "ku_ctx.h"

// ---------------------------------------------------------------------------- counts
extern "C" int ku_counts_dims_get(ku_ctx *ctx, ku_counts_dims *out) {
  if (!ctx || !out) return fail(KU_EINVAL, "ku_counts_dims_get: null argument");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  out->n_slots = ctx->tax.n_slots;
  out->n_nodes = ctx->tax.n_nodes;
  return KU_OK;
}

extern "C" int ku_counts_export(ku_ctx *ctx, uint32_t *slot_taxid, uint64_t *n_kmers, uint8_t *registers,
                                uint32_t *node_taxid, uint64_t *n_reads) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  KU_TRY(ctx_activate(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const size_t ns = ctx->tax.n_slots, nn = ctx->tax.n_nodes;
  if (slot_taxid) memcpy(slot_taxid, ctx->h_slot_taxid.data(), ns * 4);
  if (node_taxid) memcpy(node_taxid, ctx->h_node_taxid.data(), nn * 4);
  if (n_kmers) HIP_TRY(hipMemcpy(n_kmers, ctx->cnt.n_kmers, ns * 8, hipMemcpyDeviceToHost));
  if (registers) HIP_TRY(hipMemcpy(registers, ctx->cnt.registers, ns * KU_HLL_M, hipMemcpyDeviceToHost));
  if (n_reads) HIP_TRY(hipMemcpy(n_reads, ctx->cnt.n_reads, nn * 8, hipMemcpyDeviceToHost));
  return KU_OK;
}

// ---------------------------------------------------------------------------- report from the resident state
namespace {
struct DevTmp {  // device scratch of one ku_ctx_report call
  std::vector<void *> ptrs;
  ~DevTmp() { for (void *p : ptrs) (void)hipFree(p); }
  template <typename T> int put(T **dst, const std::vector<T> &src) {
    if (hipMalloc((void **)dst, std::max<size_t>(src.size(), 1) * sizeof(T)) != hipSuccess) { *dst = nullptr; return KU_ENOMEM; }
    ptrs.push_back(*dst);
    if (!src.empty() && hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return KU_EHIP;
    return KU_OK;
  }
  // zeroed ON THE STREAM the kernels run on: a plain hipMemset goes to the null stream, which a non-blocking stream does not
  // wait for -- a large table could still be being cleared when the first kernel had already put entries into it (the
  // union sets of the sparse roll-up lost a few entries that way and counted their duplicates again)
  hipStream_t stream = nullptr;
  template <typename T> int zeros(T **dst, size_t n) {
    if (hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { *dst = nullptr; return KU_ENOMEM; }
    ptrs.push_back(*dst);
    return hipMemsetAsync(*dst, 0, std::max<size_t>(n, 1) * sizeof(T), stream) == hipSuccess ? KU_OK : KU_EHIP;
  }
};
}  // namespace

extern "C" int ku_ctx_report(ku_ctx *ctx, const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, char **out,
                             size_t *out_len) {
  return ku_ctx_report_cols(ctx, tax, counts_paths, n_paths, 0u, out, out_len);
}

extern "C" int ku_ctx_report_cols(ku_ctx *ctx, const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, uint32_t flags,
                                  char **out, size_t *out_len) {
  if (!ctx || !tax || !out || !out_len) return fail(KU_EINVAL, "ku_ctx_report: null argument");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  KU_TRY(rle_idle(ctx, "ku_ctx_report"));
  KU_TRY(ctx_activate(ctx));
  const size_t ns = ctx->tax.n_slots, nn = ctx->tax.n_nodes, nt = tax->ids.size();
  const bool six = (flags & KU_R_NO_KMER_COLS) != 0;  // `classify -p 0`: no k-mer columns, so no sketch is looked at
  const bool exact = ctx->d_exact_unique != nullptr || six, sparse = ctx->sp.on && !exact;
  // KU_REPORT_TIMES=1: where the call spends its time, on stderr
  const bool times = getenv("KU_REPORT_TIMES") != nullptr;
  auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
  double t_last = now();
  auto lap = [&](const char *what) {
    if (!times) return;
    (void)hipStreamSynchronize(ctx->stream);
    const double t = now();
    fprintf(stderr, "ku_ctx_report: %-28s %8.1f ms\n", what, (t - t_last) * 1e3);
    t_last = t;
  };
  // the run-wide (slot, encoding) set of the sparse sketches is read where it lies (no compacted copy): the end of the run
  // closes the last, partial work unit (classify.cpp:522-523)
  uint64_t n_pairs = 0;
  std::vector<uint8_t> slot_sparse(ns, 0);
  if (sparse) {
    KU_TRY(sparse_close_open_unit(ctx));
    std::vector<uint32_t> dense(ns);
    unsigned long long total = 0;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&total, ctx->sp.dev.g_count, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&err, ctx->sp.dev.err, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dense.data(), ctx->sp.dev.dense, ns * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table overflowed");
    for (size_t i = 0; i < ns; ++i) slot_sparse[i] = dense[i] ? 0 : 1;
    n_pairs = total;  // entries of the set (an upper bound of the sparse slots' entries)
    lap("close the last work unit");
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  std::vector<uint64_t> nk(ns), nr(nn), uq;
  HIP_TRY(hipMemcpy(nk.data(), ctx->cnt.n_kmers, ns * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(nr.data(), ctx->cnt.n_reads, nn * 8, hipMemcpyDeviceToHost));
  if (exact) {
    uq.assign(ns, 0);
    if (!six) HIP_TRY(hipMemcpy(uq.data(), ctx->d_exact_unique, ns * 8, hipMemcpyDeviceToHost));
  }
  // counted taxa (taxon_counts entries, classify.cpp:939,968) -> every entry of their root paths is a counted clade
  // (taxdb.hpp:928-973); taxa without a taxDB entry are dropped ("No entry for X in database!")
  std::vector<uint8_t> present(nt, 0);
  std::vector<uint64_t> c_reads(nt, 0), t_reads(nt, 0), c_kmers(nt, 0), c_uniq(nt, 0);
  std::vector<int32_t> clade_of(nt, -1);
  std::vector<uint32_t> clade_row;
  std::vector<std::pair<uint32_t, uint32_t>> memb;  // (clade, slot) over the root paths of the slots with k-mers
  std::vector<uint8_t> clade_dense;
  auto clade_id = [&](size_t row) {
    if (clade_of[row] < 0) { clade_of[row] = (int32_t)clade_row.size(); clade_row.push_back((uint32_t)row); clade_dense.push_back(0); present[row] = 1; }
    return (uint32_t)clade_of[row];
  };
  auto walk = [&](uint32_t taxid, auto &&visit) {
    auto it = tax->row.find(taxid);
    if (it == tax->row.end()) return;
    int64_t q = it->second;
    for (uint32_t guard = 0; q >= 0 && guard < 4096; ++guard, q = tax->parent_row((size_t)q)) visit((size_t)q);
  };
  for (size_t s = 0; s < ns; ++s) {
    if (!nk[s]) continue;
    const bool dense = !exact && !(sparse && slot_sparse[s]);
    walk(ctx->h_slot_taxid[s], [&](size_t row) {
      const uint32_t c = clade_id(row);
      c_kmers[row] += nk[s];
      if (exact) c_uniq[row] += uq[s];
      else memb.emplace_back(c, (uint32_t)s);
      if (dense) clade_dense[c] = 1;
    });
  }
  for (size_t i = 0; i < nn; ++i) {
    if (!nr[i]) continue;
    bool first = true;
    walk(ctx->h_node_taxid[i], [&](size_t row) {
      clade_id(row);
      c_reads[row] += nr[i];
      if (first) { t_reads[row] = nr[i]; first = false; }
    });
  }
  const uint32_t n_clades = (uint32_t)clade_row.size();
  if (!exact && n_clades) {
    DevTmp tmp;
    tmp.stream = ctx->stream;
    // members per clade (CSR)
    std::sort(memb.begin(), memb.end());
    std::vector<uint32_t> m_off(n_clades + 1, 0), m_slot(memb.size());
    for (size_t j = 0; j < memb.size(); ++j) { ++m_off[memb[j].first + 1]; m_slot[j] = memb[j].second; }
    for (uint32_t c = 0; c < n_clades; ++c) m_off[c + 1] += m_off[c];
    uint32_t *d_moff = nullptr, *d_mslot = nullptr, *d_hist = nullptr;
    uint8_t *d_dense = nullptr;
    int st = tmp.put(&d_moff, m_off);
    if (st == KU_OK) st = tmp.put(&d_mslot, m_slot);
    if (st == KU_OK) st = tmp.put(&d_dense, clade_dense);
    if (st == KU_OK) st = tmp.zeros(&d_hist, (size_t)n_clades * KU_ROLLUP_BINS);
    if (st != KU_OK) return fail(st, "ku_ctx_report: device memory for the clade roll-up");
    lap("clade lists (host)");
    KU_TRY(ku_launch_rollup_dense(ctx->cnt.registers, d_moff, d_mslot, d_dense, n_clades, d_hist, ctx->stream));
    lap("dense roll-up");
    // (entries of sparse sketches lie in the run-wide set G -- n_pairs of them -- and, since round 5, as SEEN marks in the probe
    // table: what the fused kernel's fast path booked, ku_device.h)
    if (sparse && (n_pairs || ctx->m.seen_dirty)) {
      // all-sparse clades per slot (its root path up to the first clade with a dense member: density is inherited upwards)
      std::vector<uint32_t> s_off(ns + 1, 0), s_clade;
      for (size_t s = 0; s < ns; ++s) {
        s_off[s] = (uint32_t)s_clade.size();
        if (!nk[s] || !slot_sparse[s]) continue;
        walk(ctx->h_slot_taxid[s], [&](size_t row) { if (!clade_dense[clade_of[row]]) s_clade.push_back((uint32_t)clade_of[row]); });
      }
      s_off[ns] = (uint32_t)s_clade.size();
      uint32_t *d_soff = nullptr, *d_sclade = nullptr, *d_err = nullptr, *d_set = nullptr, *d_setcells = nullptr;
      unsigned long long *d_setoff = nullptr;
      const KuSparseDev &sd = ctx->sp.dev;
      // What a slot may offer its clades: at most one entry per k-mer booked under it.  (Rounds 2-4 counted the set's entries
      // per slot first -- a pass over all of G through LDS tables, 25 ms of the report's 83 per 10 M reads; the bound sizes the
      // union sets generously instead, and the big clades take bitmaps of a fixed size anyway.)
      std::vector<unsigned long long> per_slot(ns, 0);
      for (size_t s = 0; s < ns; ++s)
        if (slot_sparse[s]) per_slot[s] = nk[s];
      std::vector<uint64_t> clade_pairs(n_clades, 0);  // entries each clade's histogram may receive
      for (size_t s = 0; s < ns; ++s)
        for (uint32_t j = s_off[s]; j < s_off[s + 1]; ++j) clade_pairs[s_clade[j]] += per_slot[s];
      // union sets, one table of 4-byte cells per clade (also for a clade with one member: the two sources may hold an
      // encoding twice): room for what its members offer -- at most every encoding there is (2^25 indices; the 2^12 of
      // them whose low 13 bits are zero come with up to 40 ranks) -- at a load of 2/3
      const uint64_t enc_space = (1ull << 25) + (1ull << 12) * 40;
      const uint64_t flag_space = (1ull << 12) * 40;  // encodings that carry the rank flag
      // BIG clades keep a bitmap over the 2^25 indices instead (4 MiB each; ku_report.hip): every clade that may receive
      // at least KU_ROLLUP_BITMAP_MIN entries (default 2^17; a test hook), most entries first and, among equals, nearest
      // the root first -- a clade's parent is offered at least as much as the clade, so whatever prefix of that order fits
      // the memory budget (a quarter of the free device memory) is closed upwards: above a bitmap there are only bitmaps
      std::vector<uint32_t> depth(n_clades, 0);
      for (uint32_t c = 0; c < n_clades; ++c) {
        int64_t q = tax->parent_row(clade_row[c]);
        for (uint32_t guard = 0; q >= 0 && guard < 4096; ++guard, q = tax->parent_row((size_t)q)) ++depth[c];
      }
      uint64_t bm_min = 1ull << 17;
      if (const char *e = getenv("KU_ROLLUP_BITMAP_MIN")) bm_min = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
      std::vector<uint32_t> cand;
      for (uint32_t c = 0; c < n_clades; ++c)
        if (clade_pairs[c] >= bm_min && !clade_dense[c]) cand.push_back(c);
      std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) {
        return clade_pairs[a] != clade_pairs[b] ? clade_pairs[a] > clade_pairs[b] : (depth[a] != depth[b] ? depth[a] < depth[b] : a < b);
      });
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const size_t bm_budget = free_b / 4 / ((size_t)KU_BM_WORDS * 4);
      if (cand.size() > bm_budget) cand.resize(bm_budget);
      std::vector<uint32_t> bm_of(n_clades, KU_BM_NONE), bm_clade(cand);
      for (uint32_t b = 0; b < cand.size(); ++b) bm_of[cand[b]] = b;
      const uint32_t n_bm = (uint32_t)cand.size();
      // parents among the bitmap clades (the next clade up a slot's chain), children lists, parents by level
      std::vector<uint32_t> bm_parent(n_bm, KU_BM_NONE);
      for (size_t s = 0; s < ns; ++s)
        for (uint32_t j = s_off[s]; j + 1 < s_off[s + 1]; ++j)
          if (bm_of[s_clade[j]] != KU_BM_NONE) bm_parent[bm_of[s_clade[j]]] = bm_of[s_clade[j + 1]];
      std::vector<uint32_t> ch_off(n_bm + 1, 0), ch;
      for (uint32_t b = 0; b < n_bm; ++b)
        if (bm_parent[b] != KU_BM_NONE) ++ch_off[bm_parent[b] + 1];
      for (uint32_t b = 0; b < n_bm; ++b) ch_off[b + 1] += ch_off[b];
      ch.resize(ch_off[n_bm]);
      {
        std::vector<uint32_t> at(ch_off.begin(), ch_off.end() - 1);
        for (uint32_t b = 0; b < n_bm; ++b)
          if (bm_parent[b] != KU_BM_NONE) ch[at[bm_parent[b]]++] = b;
      }
      std::vector<uint32_t> bm_parents_by_level;  // parents with children, deepest level first
      std::vector<std::pair<uint32_t, uint32_t>> level_ranges;
      {
        std::vector<uint32_t> ps;
        for (uint32_t b = 0; b < n_bm; ++b)
          if (ch_off[b + 1] > ch_off[b]) ps.push_back(b);
        std::sort(ps.begin(), ps.end(), [&](uint32_t a, uint32_t b) { return depth[bm_clade[a]] != depth[bm_clade[b]] ? depth[bm_clade[a]] > depth[bm_clade[b]] : a < b; });
        for (size_t i = 0; i < ps.size();) {
          size_t j = i;
          while (j < ps.size() && depth[bm_clade[ps[j]]] == depth[bm_clade[ps[i]]]) ++j;
          level_ranges.emplace_back((uint32_t)i, (uint32_t)j);
          i = j;
        }
        bm_parents_by_level = ps;
      }
      std::vector<unsigned long long> set_off(n_clades, 0);
      std::vector<uint32_t> set_cells(n_clades, 0);
      uint64_t cells = 0;
      for (uint32_t c = 0; c < n_clades; ++c) {
        if (!clade_pairs[c] || clade_dense[c]) continue;
        // a bitmap clade's table only takes the entries with the rank flag (1 in 8192 of what hashes offer)
        const uint64_t bound = bm_of[c] != KU_BM_NONE ? std::min(clade_pairs[c] / 512 + 4096, flag_space) : std::min(clade_pairs[c], enc_space);
        set_off[c] = cells;
        set_cells[c] = (uint32_t)(bound + bound / 2 + 16);
        cells += set_cells[c];
      }
      // the busiest clades (the ones near the root) count in LDS
      std::vector<uint32_t> hot_clades(n_clades);
      for (uint32_t c = 0; c < n_clades; ++c) hot_clades[c] = c;
      const uint32_t n_hot = std::min<uint32_t>(KU_ROLLUP_HOT, n_clades);
      std::partial_sort(hot_clades.begin(), hot_clades.begin() + n_hot, hot_clades.end(),
                        [&](uint32_t a, uint32_t b) { return clade_pairs[a] != clade_pairs[b] ? clade_pairs[a] > clade_pairs[b] : a < b; });
      hot_clades.resize(n_hot);
      std::vector<uint16_t> clade_hot(n_clades, 0xFFFFu);
      for (uint32_t h = 0; h < n_hot; ++h) clade_hot[hot_clades[h]] = (uint16_t)h;
      uint16_t *d_chot = nullptr;
      uint32_t *d_hotc = nullptr;
      st = tmp.put(&d_soff, s_off);
      if (st == KU_OK) st = tmp.put(&d_sclade, s_clade);
      if (st == KU_OK) st = tmp.zeros(&d_err, 1);
      if (st == KU_OK) st = tmp.put(&d_chot, clade_hot);
      if (st == KU_OK) st = tmp.put(&d_hotc, hot_clades);
      if (st == KU_OK) st = tmp.put(&d_setoff, set_off);
      if (st == KU_OK) st = tmp.put(&d_setcells, set_cells);
      uint32_t *d_bmof = nullptr, *d_bm = nullptr, *d_bmclade = nullptr, *d_choff = nullptr, *d_ch = nullptr, *d_bmpar = nullptr;
      if (st == KU_OK) st = tmp.put(&d_bmof, bm_of);
      if (st == KU_OK) st = tmp.put(&d_bmclade, bm_clade);
      if (st == KU_OK) st = tmp.put(&d_choff, ch_off);
      if (st == KU_OK) st = tmp.put(&d_ch, ch);
      if (st == KU_OK) st = tmp.put(&d_bmpar, bm_parents_by_level);
      std::vector<uint32_t> slot_fast(ns, KU_FAST_SKIP);
      for (size_t s = 0; s < ns; ++s)
        if (s_off[s + 1] > s_off[s]) slot_fast[s] = bm_of[s_clade[s_off[s]]] != KU_BM_NONE ? bm_of[s_clade[s_off[s]]] : KU_FAST_WALK;
      uint32_t *d_sfast = nullptr;
      if (st == KU_OK) st = tmp.put(&d_sfast, slot_fast);
      lap("union plan (host)");
      if (st == KU_OK) st = tmp.zeros(&d_set, cells);
      if (st == KU_OK) st = tmp.zeros(&d_bm, (size_t)std::max<uint32_t>(n_bm, 1) * (n_bm ? KU_BM_WORDS : 1));
      if (st != KU_OK) return fail(st, "ku_ctx_report: device memory for the union of the sparse sketches");
      lap("union set allocated + cleared");
      KuRollupPlan plan{};
      plan.dense = sd.dense; plan.slot_off = d_soff; plan.slot_clade = d_sclade; plan.set_off = d_setoff; plan.set_cells = d_setcells;
      plan.clade_hot = d_chot; plan.hot_clades = d_hotc; plan.n_hot = n_hot; plan.set = d_set; plan.hist = d_hist; plan.err = d_err;
      plan.bm_of = d_bmof; plan.bm = d_bm; plan.slot_fast = d_sfast;
      if (n_pairs) KU_TRY(ku_launch_rollup_sparse(sd.g_key, sd.g_mask + 1, plan, ctx->n_cu, ctx->stream));
      if (ctx->m.seen_dirty && ctx->m.d_table) KU_TRY(ku_launch_rollup_table(ctx->m.d_table, ctx->m.db.n_lines, plan, ctx->n_cu, ctx->stream));
      for (const auto &lv : level_ranges)  // children into parents, deepest parents first
        KU_TRY(ku_launch_bitmap_or_children(d_bm, d_bmpar + lv.first, lv.second - lv.first, d_choff, d_ch, ctx->stream));
      KU_TRY(ku_launch_bitmap_hist(d_bm, d_bmclade, n_bm, d_hist, ctx->stream));
      uint32_t err = 0;
      HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (err) return fail(KU_EHIP, "ku_ctx_report: the sparse-union set overflowed");
      lap("sparse roll-up");
    }
    std::vector<uint32_t> hist((size_t)n_clades * KU_ROLLUP_BINS);
    HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t c = 0; c < n_clades; ++c) {
      const size_t row = clade_row[c];
      if (c_reads[row] == 0) continue;  // not printed
      const bool has_members = m_off[c + 1] > m_off[c];
      // a clade counted through reads only has an empty sketch
      c_uniq[row] = has_members ? ku_hll_estimate_hist(hist.data() + (size_t)c * KU_ROLLUP_BINS, sparse && !clade_dense[c], c_kmers[row]) : 0;
    }
  }
  lap("estimates (host)");
  const int rst = ku_report_rows_cols(tax, counts_paths, n_paths, present.data(), c_reads.data(), t_reads.data(), c_kmers.data(), c_uniq.data(), nt,
                                      flags, out, out_len);
  lap("report text");
  return rst;
}

extern "C" int ku_counts_device_ptrs(ku_ctx *ctx, uint8_t **d_registers, uint64_t *n_register_bytes,
                                     uint64_t **d_n_kmers, uint64_t *n_slots, uint64_t **d_n_reads, uint64_t *n_nodes) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  if (d_registers) *d_registers = ctx->cnt.registers;
  if (n_register_bytes) *n_register_bytes = (uint64_t)ctx->tax.n_slots * KU_HLL_M;
  if (d_n_kmers) *d_n_kmers = (uint64_t *)ctx->cnt.n_kmers;
  if (n_slots) *n_slots = ctx->tax.n_slots;
  if (d_n_reads) *d_n_reads = (uint64_t *)ctx->cnt.n_reads;
  if (n_nodes) *n_nodes = ctx->tax.n_nodes;
  return KU_OK;
}
