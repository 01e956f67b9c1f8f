// ku_api_sparse.cpp -- C ABI: the HyperLogLog++ sparse-mode emulation on a context (kernels: ku_sparse.hip)
#include "ku_ctx.h"

// ---------------------------------------------------------------------------- HLL sparse-mode emulation
extern "C" int ku_ctx_enable_sparse(ku_ctx *ctx, uint64_t work_unit_nt, uint32_t global_log2) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_ctx_enable_sparse"));
  if (ctx->tax.n_slots > KU_SPARSE_MAX_SLOTS) return fail(KU_EUNSUP, "sparse-mode emulation handles up to 2^18 distinct database taxids");
  if (global_log2 == 0) global_log2 = 26;
  if (global_log2 < 10 || global_log2 > 34) return fail(KU_EINVAL, "ku_ctx_enable_sparse: global_log2 out of range (10..34)");
  if (ctx->sp.on) {  // a second call starts afresh with the new work-unit size
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx_free_sparse(ctx);
  }
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  // One pass of the emulation covers at most 2^25 bases (sparse_pass), the carried open unit at most 2^25 entries more: L
  // (distinct (unit, slot, encoding) of a pass) at 2^27 cells and U ((unit, slot) pairs: at most one per k-mer, and at
  // most n_slots per unit) sized from the slot count never run above half full, whatever the sample looks like
  const uint32_t l_log2 = 27;
  uint32_t u_log2 = 22;
  while (u_log2 < 27 && (1ull << u_log2) < 128ull * ctx->tax.n_slots) ++u_log2;
  d.l_mask = (1ull << l_log2) - 1;
  d.u_mask = (1ull << u_log2) - 1;
  d.g_mask = (1ull << global_log2) - 1;
  sp.cap_carry_l = std::min<uint64_t>(1024ull * ctx->tax.n_slots, 1ull << 25);
  sp.cap_carry_u = ctx->tax.n_slots;
  bool ok = hipMalloc((void **)&d.l_key, (d.l_mask + 1) * 8) == hipSuccess && hipMalloc((void **)&d.l_first, (d.l_mask + 1) * 4) == hipSuccess &&
            hipMalloc((void **)&d.u_key, (d.u_mask + 1) * 8) == hipSuccess && hipMalloc((void **)&d.u_distinct, (d.u_mask + 1) * 4) == hipSuccess &&
            hipMalloc((void **)&d.u_last, (d.u_mask + 1) * 4) == hipSuccess && hipMalloc((void **)&d.u_maxfirst, (d.u_mask + 1) * 4) == hipSuccess &&
            hipMalloc((void **)&d.g_key, (d.g_mask + 1) * 8) == hipSuccess && hipMalloc((void **)&d.dense, (size_t)ctx->tax.n_slots * 4) == hipSuccess &&
            hipMalloc((void **)&d.err, 4) == hipSuccess && hipMalloc((void **)&sp.d_counters, 32) == hipSuccess &&
            sp.carry_l.reserve(sp.cap_carry_l * 8) == KU_OK && sp.carry_u.reserve(sp.cap_carry_u * 12) == KU_OK;
  if (!ok) {
    for (void *p : {(void *)d.l_key, (void *)d.l_first, (void *)d.u_key, (void *)d.u_distinct, (void *)d.u_last, (void *)d.u_maxfirst,
                    (void *)d.g_key, (void *)d.dense, (void *)d.err, (void *)sp.d_counters})
      if (p) (void)hipFree(p);
    sp.carry_l.release();
    sp.carry_u.release();
    sp = ku_ctx::Sparse{};
    return fail(KU_ENOMEM, "device memory for the sparse-mode emulation");
  }
  d.g_count = sp.d_counters;
  sp.unit_nt = work_unit_nt;
  sp.on = true;
  return ku_ctx_reset_counts(ctx);
}

// room in the run-wide (slot, encoding) set for `incoming` more entries at load <= 1/2: a larger table takes over when
// the current one could fill (the set only grows with the distinct k-mers of the taxa that stay sparse -- on a run of
// many taxa that is most of what the reads hold)
// the kernels of every batch in flight are through (they run on streams of their own: what is about to replace a table they
// write to -- the run-wide set growing -- waits for them on the host; rare)

int sparse_reserve_global(ku_ctx *ctx, uint64_t incoming, hipStream_t s) {
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  const uint64_t need = 2 * (sp.g_count + incoming);
  if (need <= d.g_mask + 1) return KU_OK;
  KU_TRY(rle_drain_kernels(ctx));
  uint64_t cells = (d.g_mask + 1) * 2;
  while (cells < need) cells *= 2;
  const char *cap_env = getenv("KU_SPARSE_MAX_LOG2");  // test hook: a small ceiling stands in for a full device
  if (cells > (1ull << (cap_env ? atoi(cap_env) : 36))) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set outgrew its ceiling");
  unsigned long long *nk = nullptr;
  if (hipMalloc((void **)&nk, cells * 8) != hipSuccess) {
    (void)hipGetLastError();
    return fail(KU_ENOMEM, "sparse-mode emulation: device memory for the run-wide set of encoded hashes (" + std::to_string(cells >> 17) + " MiB)");
  }
  unsigned long long *old = d.g_key;
  const uint64_t old_cells = d.g_mask + 1;
  HIP_TRY(hipMemsetAsync(nk, 0, cells * 8, s));
  HIP_TRY(hipMemsetAsync(d.g_count, 0, 8, s));
  d.g_key = nk;
  d.g_mask = cells - 1;
  KU_TRY(ku_launch_sparse_rehash(d, old, old_cells, s));
  unsigned long long n = 0;
  HIP_TRY(hipMemcpyAsync(&n, d.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(old);
  sp.g_count = n;
  return KU_OK;
}

// The per-pass tables L / U of the exact evaluation, sized for what the pass holds: a prefix of the allocated arrays (the
// kernels that close a unit scan whole tables -- with 2^26 cells for a few hundred thousand entries those scans and the
// memsets were most of the emulation's cost in a `classify -r` run).  Returns the device view to hand to the kernels.
int sparse_pass_tables(ku_ctx *ctx, uint64_t n_entries, KuSparseDev *view, hipStream_t s) {
  const KuSparseDev &d = ctx->sp.dev;
  uint64_t l_cells = 1ull << 14, u_cells = 1ull << 12;
  while (l_cells < 4 * n_entries && l_cells < d.l_mask + 1) l_cells <<= 1;
  while (u_cells < 2 * n_entries && u_cells < d.u_mask + 1) u_cells <<= 1;
  *view = d;
  view->l_mask = l_cells - 1;
  view->u_mask = u_cells - 1;
  return ku_launch_sparse_clear(*view, s);
}

// the reads [r0, r1) of a batch whose taxa[] holds slot ids: one pass of the emulation (at most KU_SPARSE_MAX_UNITS
// work units and 2^25 bases at a time; a unit that is still open at the end is carried into the next pass)
int sparse_pass(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off,
                       const uint32_t *h_len, uint64_t n_reads, uint64_t n_bytes, const uint32_t *d_taxa, uint32_t quick_min_hits,
                       hipStream_t s) {
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  KU_TRY(sparse_tail_to_carry(ctx));  // (an open unit the fast path left: into the form these passes carry along)
  if (n_bytes + 2 >= (1ull << 32)) return fail(KU_EUNSUP, "sparse-mode emulation: batches of at most 4 G bases");
  for (uint64_t i = 1; i < n_reads; ++i)
    if (h_off[i] < h_off[i - 1]) return fail(KU_EINVAL, "sparse-mode emulation: the reads of a batch must be in buffer order");
  std::vector<uint32_t> unit(n_reads);
  if (sp.unit.reserve(std::max<uint64_t>(n_reads, 1) * 4) != KU_OK) return fail(KU_ENOMEM, "device memory for the work-unit ids");
  uint64_t r0 = 0;
  while (r0 < n_reads) {
    // cut: units and bases of this pass
    uint32_t cur = 0;
    uint64_t acc = sp.acc_nt, bases = 0, r1 = r0;
    while (r1 < n_reads && cur < KU_SPARSE_MAX_UNITS && bases < (1ull << 25)) {
      unit[r1] = cur;
      acc += h_len[r1];
      bases += h_len[r1];
      ++r1;
      if (sp.unit_nt && acc >= sp.unit_nt) { ++cur; acc = 0; }  // the unit closes behind the read that fills it (classify.cpp:510-521)
    }
    const bool open_after = acc > 0 || (sp.unit_nt == 0 && (sp.open || r1 > r0));
    const uint32_t n_closed = cur;  // units 0 .. cur-1 are complete; unit `cur` (if any read fell into it) stays open
    KU_TRY(sparse_reserve_global(ctx, bases + sp.n_carry_l, s));
    KU_TRY(ku_launch_sparse_clear(d, s));
    if (sp.open) KU_TRY(ku_launch_sparse_carry_in(d, (const unsigned long long *)sp.carry_l.p, sp.n_carry_l, (const uint32_t *)sp.carry_u.p,
                                                   sp.n_carry_u, s));
    HIP_TRY(hipMemcpyAsync((uint32_t *)sp.unit.p + r0, unit.data() + r0, (r1 - r0) * 4, hipMemcpyHostToDevice, s));
    KU_TRY(ku_launch_sparse_insert(d, ctx->m.db.k, (const uint8_t *)d_seqs, d_off + r0, d_len + r0, (const uint32_t *)sp.unit.p + r0, r1 - r0,
                                   d_taxa, quick_min_hits, ctx->n_cu, s));
    KU_TRY(ku_launch_sparse_close(d, n_closed, s));
    sp.n_carry_l = sp.n_carry_u = 0;
    if (open_after) {
      HIP_TRY(hipMemsetAsync(sp.d_counters + 1, 0, 16, s));
      KU_TRY(ku_launch_sparse_carry_out(d, cur, (unsigned long long *)sp.carry_l.p, (uint32_t *)sp.carry_u.p, sp.d_counters + 1, sp.cap_carry_l,
                                        sp.cap_carry_u, s));
      unsigned long long c[3] = {0, 0, 0};
      HIP_TRY(hipMemcpyAsync(c, sp.d_counters, 24, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      sp.g_count = c[0];
      sp.n_carry_l = std::min<uint64_t>(c[1], sp.cap_carry_l);
      sp.n_carry_u = std::min<uint64_t>(c[2], sp.cap_carry_u);
    } else {
      unsigned long long c = 0;
      HIP_TRY(hipMemcpyAsync(&c, sp.d_counters, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));  // `unit` is reused by the next pass
      sp.g_count = c;
    }
    sp.open = open_after;
    sp.acc_nt = acc;
    r0 = r1;
  }
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&err, d.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (err) return fail(KU_ENOMEM, std::string("sparse-mode emulation: a device table is full (") + ((err & 1) ? "L " : "") + ((err & 2) ? "U " : "") +
                                      ((err & 4) ? "G" : "") + ")");
  return KU_OK;
}

// the unit that is still open ends here (end of an input file / of the run): evaluate and commit what was carried
int sparse_close_open_unit(ku_ctx *ctx) {
  ku_ctx::Sparse &sp = ctx->sp;
  hipStream_t s = ctx->stream;
  KU_TRY(sparse_tail_close(ctx));  // (a unit the fast path kept as its reads + insert counts)
  if (sp.open) {
    KU_TRY(sparse_reserve_global(ctx, sp.n_carry_l, s));
    KuSparseDev d;
    KU_TRY(sparse_pass_tables(ctx, sp.n_carry_l + sp.n_carry_u, &d, s));
    KU_TRY(ku_launch_sparse_carry_in(d, (const unsigned long long *)sp.carry_l.p, sp.n_carry_l, (const uint32_t *)sp.carry_u.p, sp.n_carry_u, s));
    KU_TRY(ku_launch_sparse_close(d, 1, s));
    unsigned long long c = 0;
    HIP_TRY(hipMemcpyAsync(&c, sp.d_counters, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    sp.g_count = c;
  }
  sp.open = false;
  sp.acc_nt = 0;
  sp.n_carry_l = sp.n_carry_u = 0;
  return KU_OK;
}

// The entries the fused kernel's fast path marked in the probe table (ku_device.h: SEEN bytes) join the run-wide set: for whoever
// needs the set as such (ku_sparse_export, the union of several ranks' sets, a table that is about to go).  The marks stay.
int ctx_seen_harvest(ku_ctx *ctx) {
  if (!ctx->sp.on || !ctx->m.seen_dirty || !ctx->m.d_table) return KU_OK;
  hipStream_t s = ctx->stream;
  unsigned long long *d_n = ctx->sp.d_counters + 3, n = 0;
  HIP_TRY(hipMemsetAsync(d_n, 0, 8, s));
  KU_TRY(ku_launch_seen(0, ctx->m.d_table, ctx->m.db.n_lines, ctx->sp.dev, d_n, s));
  HIP_TRY(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n == 0) return KU_OK;
  KU_TRY(sparse_reserve_global(ctx, n, s));
  KU_TRY(ku_launch_seen(1, ctx->m.d_table, ctx->m.db.n_lines, ctx->sp.dev, ctx->sp.dev.g_count, s));
  unsigned long long c = 0;
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&c, ctx->sp.dev.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, ctx->sp.dev.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  ctx->sp.g_count = c;
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set is full");
  return KU_OK;
}

extern "C" int ku_ctx_disable_sparse(ku_ctx *ctx) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  KU_TRY(rle_idle(ctx, "ku_ctx_disable_sparse"));
  KU_TRY(ctx_activate(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx_free_sparse(ctx);
  return KU_OK;
}

extern "C" int ku_ctx_sparse_state(const ku_ctx *ctx) { return !ctx ? 0 : (ctx->sp.on ? 1 : (ctx->sp.gave_up ? 2 : 0)); }

extern "C" int ku_sparse_close_unit(ku_ctx *ctx) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_sparse_close_unit"));
  if (!ctx->sp.on) return KU_OK;
  if (ctx->sp.unit_nt == 0) return KU_OK;  // one unit for the whole run
  return sparse_close_open_unit(ctx);
}

extern "C" int ku_sparse_export(ku_ctx *ctx, uint8_t *slot_is_sparse, uint64_t *pairs, uint64_t *n_pairs) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_sparse_export"));
  if (!ctx->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled (ku_ctx_enable_sparse)");
  if (!n_pairs) return fail(KU_EINVAL, "ku_sparse_export: null argument");
  KU_TRY(sparse_close_open_unit(ctx));
  KU_TRY(ctx_seen_harvest(ctx));
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  hipStream_t s = ctx->stream;  // end of the run: the last, partial work unit closes (classify.cpp:522-523)
  unsigned long long total = 0;
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&total, d.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, d.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table overflowed");
  if (sp.out.reserve(std::max<uint64_t>(total, 1) * 8) != KU_OK) return fail(KU_ENOMEM, "device memory for the sparse export");
  HIP_TRY(hipMemsetAsync(sp.d_counters + 3, 0, 8, s));
  KU_TRY(ku_launch_sparse_export(d, (unsigned long long *)sp.out.p, total, sp.d_counters + 3, s));
  unsigned long long n = 0;
  HIP_TRY(hipMemcpyAsync(&n, sp.d_counters + 3, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (slot_is_sparse) {
    std::vector<uint32_t> dense(ctx->tax.n_slots);
    HIP_TRY(hipMemcpy(dense.data(), d.dense, (size_t)ctx->tax.n_slots * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < ctx->tax.n_slots; ++i) slot_is_sparse[i] = dense[i] ? 0 : 1;
  }
  if (pairs) {
    if (*n_pairs < n) return fail(KU_EINVAL, "ku_sparse_export: output array too small");
    if (n) HIP_TRY(hipMemcpy(pairs, sp.out.p, n * 8, hipMemcpyDeviceToHost));
  }
  *n_pairs = n;
  return KU_OK;
}

// ---- the emulation over several GPUs (ku_mgpu.cpp): every rank runs it on whole work units of the read stream, the open
// unit travels to the rank that classifies the next reads, and the ranks' states are folded into one at the end of the run
// (a taxon's global sketch is dense iff some unit made it dense, on whichever rank; else it holds every encoding of the
// run: the union of the ranks' sets)
int ku_ctx_sparse_on(const ku_ctx *ctx) { return ctx && ctx->sp.on ? 1 : 0; }
uint64_t ku_ctx_sparse_unit_nt(const ku_ctx *ctx) { return ctx ? ctx->sp.unit_nt : 0; }
// one pass of the emulation over reads whose per-k-mer array holds slot ids (the sharded path, between the exchange and
// the resolve stage); KU_ENOMEM switches the emulation off on this context like the single-GPU path does
int ku_ctx_sparse_pass_slots(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off,
                             const uint32_t *h_len, uint64_t n_reads, uint64_t n_bytes, const uint32_t *d_taxa, uint32_t quick_min_hits,
                             hipStream_t s) {
  KU_TRY(check_ready(ctx));
  if (!ctx->sp.on || n_reads == 0) return KU_OK;
  int st = sparse_pass(ctx, d_seqs, d_off, d_len, h_off, h_len, n_reads, n_bytes, d_taxa, quick_min_hits, s);
  if (st == KU_ENOMEM) {
    (void)hipStreamSynchronize(s);
    (void)hipGetLastError();
    ctx_free_sparse(ctx);
    ctx->sp.gave_up = true;
    return KU_OK;
  }
  return st;
}
// the unit that is still open on `src` continues on `dst` (same process; the contexts may sit on different devices)
int ku_ctx_sparse_move_open_unit(ku_ctx *src, ku_ctx *dst) {
  if (!src || !dst || !src->sp.on || !dst->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled on both contexts");
  if (src == dst || !(src->sp.open || src->sp.tail_open)) return KU_OK;
  if (dst->sp.open || dst->sp.tail_open) return fail(KU_ESTATE, "the destination context holds an open work unit of its own");
  KU_TRY(ctx_activate(src));
  KU_TRY(sparse_tail_to_carry(src));  // (what travels is the staged form: L / U entries)
  HIP_TRY(hipStreamSynchronize(src->stream));
  KU_TRY(ctx_activate(dst));
  HIP_TRY(hipStreamSynchronize(dst->stream));
  if (src->sp.n_carry_l > dst->sp.cap_carry_l || src->sp.n_carry_u > dst->sp.cap_carry_u) return fail(KU_ESTATE, "carry buffers differ between the contexts");
  if (src->sp.n_carry_l) HIP_TRY(hipMemcpy(dst->sp.carry_l.p, src->sp.carry_l.p, src->sp.n_carry_l * 8, hipMemcpyDefault));
  if (src->sp.n_carry_u) HIP_TRY(hipMemcpy(dst->sp.carry_u.p, src->sp.carry_u.p, src->sp.n_carry_u * 12, hipMemcpyDefault));
  dst->sp.n_carry_l = src->sp.n_carry_l;
  dst->sp.n_carry_u = src->sp.n_carry_u;
  dst->sp.open = true;
  dst->sp.acc_nt = src->sp.acc_nt;
  src->sp.open = false;
  src->sp.acc_nt = 0;
  src->sp.n_carry_l = src->sp.n_carry_u = 0;
  return KU_OK;
}
// end of the run on this rank: the last, partial unit closes; dense flags out (host, one per slot)
int ku_ctx_sparse_finish(ku_ctx *ctx, uint32_t *h_dense) {
  KU_TRY(check_ready(ctx));
  if (!ctx->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled");
  KU_TRY(sparse_close_open_unit(ctx));
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&err, ctx->sp.dev.err, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(h_dense, ctx->sp.dev.dense, (size_t)ctx->tax.n_slots * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table overflowed");
  return KU_OK;
}
int ku_ctx_sparse_set_dense(ku_ctx *ctx, const uint32_t *h_dense) {
  KU_TRY(check_ready(ctx));
  if (!ctx->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled");
  HIP_TRY(hipMemcpyAsync(ctx->sp.dev.dense, h_dense, (size_t)ctx->tax.n_slots * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return KU_OK;
}
// the run-wide set of `src` joins that of `dst` (entries of slots that are dense by now are dropped on the way)
int ku_ctx_sparse_absorb(ku_ctx *dst, ku_ctx *src) {
  if (!dst || !src || !dst->sp.on || !src->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled on both contexts");
  KU_TRY(ctx_activate(src));
  KU_TRY(ctx_seen_harvest(src));  // (what src's fused kernel marked in its probe table; after the group's dense flags were set)
  HIP_TRY(hipStreamSynchronize(src->stream));
  unsigned long long n_src = 0;
  HIP_TRY(hipMemcpy(&n_src, src->sp.dev.g_count, 8, hipMemcpyDeviceToHost));
  KU_TRY(ctx_activate(dst));
  hipStream_t s = dst->stream;
  KU_TRY(sparse_reserve_global(dst, n_src, s));
  const uint64_t cells = src->sp.dev.g_mask + 1, step = 1ull << 23;  // 64 MB of cells at a time
  if (dst->sp.out.reserve(std::min(cells, step) * 8) != KU_OK) return fail(KU_ENOMEM, "device memory for the merge of the sparse sets");
  for (uint64_t c0 = 0; c0 < cells; c0 += step) {
    const uint64_t n = std::min(step, cells - c0);
    HIP_TRY(hipMemcpyAsync(dst->sp.out.p, src->sp.dev.g_key + c0, n * 8, hipMemcpyDefault, s));
    KU_TRY(ku_launch_sparse_absorb(dst->sp.dev, (const unsigned long long *)dst->sp.out.p, n, s));
  }
  unsigned long long c = 0;
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&c, dst->sp.dev.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, dst->sp.dev.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  dst->sp.g_count = c;
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set is full");
  return KU_OK;
}
