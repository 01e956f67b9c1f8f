// ku_api_classify.cpp -- C ABI: lookup / resolve / classify on device and host buffers, the owner-routing entry points of the
// multi-GPU driver (kernels: ku_kernels.hip, ku_short.hip, ku_route.hip)
#include "ku_ctx.h"

// ---------------------------------------------------------------------------- classification
int check_ready(ku_ctx *ctx) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  if (!ctx->db_loaded) return fail(KU_ESTATE, "no database loaded");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  return ctx_activate(ctx);
}

extern "C" int ku_lookup_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const ku_opts *opts,
                                uint32_t *d_taxa, void *stream) {
  KU_TRY(check_ready(ctx));
  if (n_bytes && (!d_seqs || !d_taxa)) return fail(KU_EINVAL, "ku_lookup_device: null buffer");
  const uint32_t flags = opts ? opts->flags : 0;
  // quick mode counts only the scanned prefix of each read -> accounted in the resolve stage
  const bool counts = !(flags & (KU_F_NO_COUNTS | KU_F_QUICK));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  // hierarchical run: one pass per database in command-line order; later passes only search the positions that
  // are still 0, the last one does the per-taxon accounting (classify.cpp:928-939)
  const size_t nd = 1 + ctx->extra.size();
  for (size_t i = 0; i < nd; ++i) {
    const DbStore &d = i ? ctx->extra[i - 1] : ctx->m;
    int st = ku_launch_lookup(d.db, ctx->cnt, (const uint8_t *)d_seqs, n_bytes, d_taxa, counts && i + 1 == nd, i > 0,
                              (flags & KU_F_MERGE_CHUNK) != 0, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "lookup kernel launch failed");
  }
  return KU_OK;
}

// ---- owner routing (ku_mgpu.cpp): the context's database / counters behind the three kernels
int ku_ctx_route_info(const ku_ctx *ctx, uint64_t *bin_lo, uint64_t *bin_hi, int *is_hash, int *single_db) {
  if (!ctx || !ctx->db_loaded || !ctx->tax_set) return fail(KU_ESTATE, "no database / taxonomy on this context");
  if (bin_lo) *bin_lo = ctx->m.db.bin_lo;
  if (bin_hi) *bin_hi = ctx->m.db.bin_hi;
  if (is_hash) *is_hash = ctx->m.db.table != nullptr;
  if (single_db) *single_db = ctx->extra.empty();
  return KU_OK;
}
int ku_ctx_route_scan(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, uint32_t *d_taxa, const KuRouteDev &rt, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  int st = ku_launch_route_scan(ctx->m.db, (const uint8_t *)d_seqs, n_bytes, d_taxa, rt, ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "route scan kernel launch failed");
}
// whether the resolve stage of a routed step can run as the fused kernel's ROUTE instance (KU_EUNSUP: no -- quick mode, reads
// beyond 65535 k-mers, unknown read length); reserves the windowed instance's spill workspace for any number of reads, so
// that the per-round calls below never reallocate it under a kernel of the other stream
int ku_ctx_route_resolve_prepare(ku_ctx *ctx, const ku_opts *opts, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  const uint32_t flags = opts ? opts->flags : 0;
  const uint32_t max_len = opts ? opts->max_read_len : 0;
  if (max_len == 0 || (flags & (KU_F_QUICK | KU_F_KEEP_SLOTS)) || getenv("KU_NO_FUSED")) return KU_EUNSUP;  // (no message: the caller has another path)
  const uint32_t max_n = max_len >= ctx->m.db.k ? max_len - ctx->m.db.k + 1 : 0;
  if (max_n > ku_route_resolve_max_kmers()) return KU_EUNSUP;
  if (max_n > 128) {
    const uint64_t ws = ku_short_workspace_bytes(std::max(max_n, 193u), ctx->tax.n_slots, ~0ull >> 8, ctx->n_cu);
    if (ws > ctx->b_ws.cap) {
      HIP_TRY(hipStreamSynchronize(s));
      if (ctx->b_ws.reserve(ws) != KU_OK) { (void)hipGetLastError(); return KU_EUNSUP; }
    }
  }
  return KU_OK;
}
int ku_ctx_route_resolve(ku_ctx *ctx, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                         uint32_t *d_taxa, uint32_t *d_hits, const uint32_t *d_kb, const uint32_t *d_ret, hipStream_t s) {
  const uint32_t flags = opts ? opts->flags : 0;
  const uint32_t max_len = opts ? opts->max_read_len : 0;
  const uint32_t max_n = max_len >= ctx->m.db.k ? max_len - ctx->m.db.k + 1 : 0;
  int st = ku_launch_route_resolve(ctx->m.db, ctx->tax, ctx->cnt, d_off, d_len, n_reads, max_n, flags, d_calls, d_taxa, d_hits, d_kb, d_ret,
                                   ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "routed resolve kernel launch failed");
}
int ku_ctx_route_owner(ku_ctx *ctx, const void *d_rec, uint64_t n_rec, const uint32_t *d_kb, uint32_t *d_slots, bool do_counts, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  int st = ku_launch_route_owner(ctx->m.db, ctx->cnt, d_rec, n_rec, d_kb, d_slots, do_counts, ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "route owner kernel launch failed");
}

int ku_exact_owned_step(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads, uint64_t n_bytes,
                        const ku_opts *opts, uint32_t *d_taxa, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  if (!ctx->d_exact_set) return fail(KU_ESTATE, "exact counting is not enabled on this context");
  if (!ctx->extra.empty()) return fail(KU_EUNSUP, "exact counting on a shard goes with one database");
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  if (o.flags & (KU_F_QUICK | KU_F_NO_COUNTS)) return fail(KU_EUNSUP, "exact counting goes with the plain classification only");
  o.flags |= KU_F_KEEP_SLOTS | KU_F_MERGE_CHUNK;
  HIP_TRY(hipMemsetAsync(d_taxa, 0xFE, n_bytes * 4, s));
  KU_TRY(ku_lookup_device(ctx, d_seqs, n_bytes, &o, d_taxa, s));
  int st = ku_launch_exact(ctx->m.db.k, (const uint8_t *)d_seqs, d_off, d_len, n_reads, d_taxa, ctx->d_exact_set, ctx->exact_mask,
                           ctx->d_exact_unique, ctx->d_scalar + 6, ctx->n_cu, s);
  if (st != KU_OK) return fail(st, "exact counting kernel launch failed");
  return ku_launch_replace_u32(d_taxa, n_bytes, KU_FOREIGN_MARK, 0u, s);
}

extern "C" int ku_lookup_stats_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, uint64_t *stats_out,
                                      void *stream) {
  KU_TRY(check_ready(ctx));
  if (!stats_out || (n_bytes && !d_seqs)) return fail(KU_EINVAL, "ku_lookup_stats_device: null argument");
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  unsigned long long *d_stats = (unsigned long long *)(ctx->d_scalar + 8);  // 32 bytes at offset 32
  HIP_TRY(hipMemsetAsync(d_stats, 0, 32, s));
  int st = ku_launch_lookup_stats(ctx->m.db, (const uint8_t *)d_seqs, n_bytes, d_stats, ctx->n_cu, s);
  if (st != KU_OK) return fail(st, "stats kernel launch failed");
  HIP_TRY(hipMemcpyAsync(stats_out, d_stats, 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return KU_OK;
}

extern "C" int ku_resolve_device(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_seq_off, const uint32_t *d_seq_len,
                                 uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls, uint32_t *d_taxa,
                                 uint32_t *d_hits, void *stream) {
  KU_TRY(check_ready(ctx));
  if (n_reads && (!d_seq_off || !d_seq_len || !d_calls || !d_taxa)) return fail(KU_EINVAL, "ku_resolve_device: null buffer");
  const uint32_t flags = opts ? opts->flags : 0;
  if ((flags & KU_F_QUICK) && !d_seqs) return fail(KU_EINVAL, "quick mode needs the sequence buffer");
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  uint32_t max_len = opts ? opts->max_read_len : 0;
  if (max_len == 0 && n_reads) {
    KU_TRY(ku_launch_max_len(d_seq_len, n_reads, ctx->d_scalar + 4, s));
    HIP_TRY(hipMemcpyAsync(&max_len, ctx->d_scalar + 4, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  uint64_t ws = (flags & KU_F_QUICK) ? 0 : ku_resolve_workspace_bytes(max_len, ctx->m.db.k, ctx->n_cu);
  if (ws > ctx->b_ws.cap) {
    HIP_TRY(hipStreamSynchronize(s));
    if (ctx->b_ws.reserve(ws) != KU_OK) return fail(KU_ENOMEM, "resolve workspace allocation failed");
  }
  if (ws) HIP_TRY(hipMemsetAsync(ctx->b_ws.p, 0, ws, s));
  int st = ku_launch_resolve(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)d_seqs, d_seq_off, d_seq_len, n_reads, flags,
                             opts ? opts->min_hits : 1, max_len, d_calls, d_taxa, d_hits, ctx->b_ws.p, ctx->b_ws.cap,
                             ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "resolve kernel launch failed");
}

// h_off / h_len: host copies of the read offsets / lengths when the caller has them (the host-buffer entry points);
// the sparse-mode emulation needs them for the work-unit plan
int classify_device_impl(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                                const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                                uint32_t *d_taxa, uint32_t *d_hits, void *stream, const uint64_t *h_off, const uint32_t *h_len) {
  KU_TRY(check_ready(ctx));
  const uint32_t flags = opts ? opts->flags : 0;
  // short reads against the resident probe table: one fused kernel, a wave per read (ku_short.hip)
  const bool exact = ctx->d_exact_set != nullptr;
  if (exact && !store_whole(ctx->m)) return fail(KU_EUNSUP, "exact counting on a shard runs through the multi-GPU driver (ku_mgpu_enable_exact)");
  if (exact && (flags & (KU_F_KEEP_SLOTS | KU_F_NO_COUNTS)))
    return fail(KU_EUNSUP, "exact counting goes with a whole classification (no slot output / count-less runs)");
  const bool sparse = ctx->sp.on && !(flags & KU_F_NO_COUNTS);
  if (sparse && !h_len) return fail(KU_EUNSUP, "the sparse-mode emulation runs through the host-buffer entry points (it needs the read lengths on the host)");
  if (sparse && (flags & KU_F_KEEP_SLOTS)) return fail(KU_EUNSUP, "the sparse-mode emulation does not combine with slot output");
  const uint32_t short_max = (getenv("KU_NO_FUSED") || !ctx->extra.empty() || exact || sparse) ? 0 : ku_short_max_kmers(ctx->m.db);
  if (short_max && !(flags & (KU_F_QUICK | KU_F_KEEP_SLOTS)) && n_reads) {
    if (!d_seqs || !d_seq_off || !d_seq_len || !d_calls || !d_taxa) return fail(KU_EINVAL, "ku_classify_batch_device: null buffer");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    uint32_t max_len = opts ? opts->max_read_len : 0;
    if (max_len == 0) {
      KU_TRY(ku_launch_max_len(d_seq_len, n_reads, ctx->d_scalar + 4, s));
      HIP_TRY(hipMemcpyAsync(&max_len, ctx->d_scalar + 4, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    const uint32_t max_n = max_len >= ctx->m.db.k ? max_len - ctx->m.db.k + 1 : 0;
    // longer reads (mate pairs, long reads up to 65535 k-mers): the same kernel in windows of 128 k-mers -- when its
    // spill workspace can be had; KU_NO_WINDOWED=1 keeps them on the flat lookup + resolve kernels
    bool fused = max_n <= short_max;
    if (!fused && max_n <= ku_short_max_kmers_windowed(ctx->m.db) && !getenv("KU_NO_WINDOWED")) {
      const uint64_t ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, n_reads, ctx->n_cu);
      if (ws > ctx->b_ws.cap) HIP_TRY(hipStreamSynchronize(s));
      fused = ctx->b_ws.reserve(ws) == KU_OK;
      if (!fused) (void)hipGetLastError();
    }
    if (fused) {
      int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)d_seqs, n_bytes, d_seq_off, d_seq_len,
                                        n_reads, max_n, flags, d_calls, d_taxa, d_hits, ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s);
      return st == KU_OK ? KU_OK : fail(st, "fused short-read kernel launch failed");
    }
    ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
    o.max_read_len = max_len;
    KU_TRY(ku_lookup_device(ctx, d_seqs, n_bytes, &o, d_taxa, stream));
    return ku_resolve_device(ctx, d_seqs, d_seq_off, d_seq_len, n_reads, &o, d_calls, d_taxa, d_hits, stream);
  }
  KU_TRY(ku_lookup_device(ctx, d_seqs, n_bytes, opts, d_taxa, stream));
  if (sparse && n_reads) {  // between the stages: d_taxa holds slot ids
    int st = sparse_pass(ctx, d_seqs, d_seq_off, d_seq_len, h_off, h_len, n_reads, n_bytes, d_taxa,
                         (flags & KU_F_QUICK) ? std::max(1u, opts ? opts->min_hits : 1u) : 0u, stream ? (hipStream_t)stream : ctx->stream);
    if (st == KU_ENOMEM) {
      // no room for the emulation's tables: the classification itself does not depend on them -- the run goes on with
      // the dense registers alone and says so (ku_ctx_sparse_state; the reports then carry their estimates)
      (void)hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream);
      (void)hipGetLastError();
      ctx_free_sparse(ctx);
      ctx->sp.gave_up = true;
    } else if (st != KU_OK) return st;
  }
  if (exact) {  // between the stages: d_taxa holds slot ids
    if (n_reads && (!d_seqs || !d_seq_off || !d_seq_len || !d_taxa)) return fail(KU_EINVAL, "ku_classify_batch_device: null buffer");
    int st = ku_launch_exact(ctx->m.db.k, (const uint8_t *)d_seqs, d_seq_off, d_seq_len, n_reads, d_taxa, ctx->d_exact_set,
                             ctx->exact_mask, ctx->d_exact_unique, ctx->d_scalar + 6, ctx->n_cu,
                             stream ? (hipStream_t)stream : ctx->stream,
                             (flags & KU_F_QUICK) ? std::max(1u, opts ? opts->min_hits : 1u) : 0u);
    if (st != KU_OK) return fail(st, "exact counting kernel launch failed");
  }
  return ku_resolve_device(ctx, d_seqs, d_seq_off, d_seq_len, n_reads, opts, d_calls, d_taxa, d_hits, stream);
}

extern "C" int ku_classify_batch_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                                        const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts,
                                        uint32_t *d_calls, uint32_t *d_taxa, uint32_t *d_hits, void *stream) {
  return classify_device_impl(ctx, d_seqs, n_bytes, d_seq_off, d_seq_len, n_reads, opts, d_calls, d_taxa, d_hits, stream, nullptr, nullptr);
}

extern "C" int ku_classify_batch(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                 const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                 uint32_t *taxa, uint32_t *hits) {
  KU_TRY(check_ready(ctx));
  if ((n_bytes && !seqs) || (n_reads && (!seq_off || !seq_len || !calls))) return fail(KU_EINVAL, "ku_classify_batch: null buffer");
  if (n_reads == 0) return KU_OK;
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  if (o.max_read_len == 0) for (uint64_t i = 0; i < n_reads; ++i) o.max_read_len = std::max(o.max_read_len, seq_len[i]);
  for (uint64_t i = 0; i < n_reads; ++i)
    if (seq_off[i] + seq_len[i] > n_bytes) return fail(KU_EINVAL, "read " + std::to_string(i) + " exceeds the sequence buffer");
  if (ctx->b_seqs.reserve(n_bytes + 16) || ctx->b_off.reserve(n_reads * 8) || ctx->b_len.reserve(n_reads * 4) ||
      ctx->b_calls.reserve(n_reads * 4) || ctx->b_taxa.reserve((n_bytes + 16) * 4) || ctx->b_hits.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->b_seqs.p, seqs, n_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_off.p, seq_off, n_reads * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_len.p, seq_len, n_reads * 4, hipMemcpyHostToDevice, s));
  KU_TRY(classify_device_impl(ctx, ctx->b_seqs.p, n_bytes, (const uint64_t *)ctx->b_off.p, (const uint32_t *)ctx->b_len.p,
                              n_reads, &o, (uint32_t *)ctx->b_calls.p, (uint32_t *)ctx->b_taxa.p,
                              (uint32_t *)ctx->b_hits.p, s, seq_off, seq_len));
  HIP_TRY(hipMemcpyAsync(calls, ctx->b_calls.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  if (taxa) HIP_TRY(hipMemcpyAsync(taxa, ctx->b_taxa.p, n_bytes * 4, hipMemcpyDeviceToHost, s));
  if (hits) HIP_TRY(hipMemcpyAsync(hits, ctx->b_hits.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return KU_OK;
}

// run-length encode d_taxa into ctx->b_runs and bring calls / hits / (run_off, run_cnt) / the run total to the host
int rle_and_fetch(ku_ctx *ctx, const uint32_t *d_taxa, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads,
                         uint64_t runs_cap, bool quick, uint32_t *calls, uint32_t *hits, uint64_t *run_off,
                         uint32_t *run_cnt, uint64_t *n_runs) {
  hipStream_t s = ctx->stream;
  unsigned long long *d_counter = (unsigned long long *)(ctx->d_scalar + 2);
  if (quick) {  // quick mode stops at the first hits: no per-k-mer codes, no runs
    HIP_TRY(hipMemsetAsync(d_counter, 0, 8, s));
    HIP_TRY(hipMemsetAsync(ctx->b_roff.p, 0, n_reads * 8, s));
    HIP_TRY(hipMemsetAsync(ctx->b_rcnt.p, 0, n_reads * 4, s));
  } else {
    KU_TRY(ku_launch_rle(d_taxa, ctx->m.db.k, d_off, d_len, n_reads, runs_cap /* ~ bases of the batch */, ctx->b_runs.p, runs_cap, d_counter,
                         (uint64_t *)ctx->b_roff.p, (uint32_t *)ctx->b_rcnt.p, ctx->n_cu, s));
  }
  unsigned long long total = 0;
  HIP_TRY(hipMemcpyAsync(&total, d_counter, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(calls, ctx->b_calls.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  if (hits) HIP_TRY(hipMemcpyAsync(hits, ctx->b_hits.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(run_off, ctx->b_roff.p, n_reads * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(run_cnt, ctx->b_rcnt.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (total > runs_cap) return fail(KU_EHIP, "run-length encoder overflowed its bound");
  *n_runs = ctx->n_runs = total;
  ctx->fetch_runs_src = ctx->b_runs.p;
  ctx->last_calls_dev = ctx->b_calls.p;
  return KU_OK;
}
