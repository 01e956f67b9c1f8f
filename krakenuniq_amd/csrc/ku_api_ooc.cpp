// ku_api_ooc.cpp -- C ABI: out-of-core runs (-x): chunk swap / prefetch, device-resident batches, merging contexts
#include "ku_ctx.h"

// ---------------------------------------------------------------------------- out-of-core run
extern "C" int ku_ctx_prefetch_shard(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi) {
  // Safe to call from a second host thread while the first one runs lookups on the resident shard: it works on its own
  // stream, its own scratch and its own store and only reads the (frozen) slot table of the context.
  if (!ctx || !db) return fail(KU_EINVAL, "ku_ctx_prefetch_shard: null argument");
  if (!ctx->db_loaded || !ctx->tax_set) return fail(KU_ESTATE, "ku_ctx_prefetch_shard: load a shard and the taxonomy first");
  if (bin_lo > bin_hi || bin_hi > db->info.n_bins) return fail(KU_EINVAL, "bin range out of bounds");
  if (!ctx->extra.empty()) return fail(KU_EUNSUP, "chunked runs use one database (as the reference's: classify.cpp:639)");
  if (db->info.k != ctx->m.db.k) return fail(KU_EINVAL, "ku_ctx_prefetch_shard: k differs from the resident shard's");
  KU_TRY(ctx_activate(ctx));
  ku_ctx::Prefetch &pf = ctx->pf;
  if (pf.valid) { store_free(pf.store); pf.valid = false; }
  if (!pf.stream) HIP_TRY(hipStreamCreateWithFlags(&pf.stream, hipStreamNonBlocking));
  if (!pf.d_scalar) HIP_TRY(hipMalloc((void **)&pf.d_scalar, 64));
  int st = store_upload(ctx, pf.store, db, bin_lo, bin_hi, /*scan_values=*/false, pf.stream);
  if (st == KU_OK) {
    pf.store.hash_layout = ctx->hash_layout;
    st = store_finalize(ctx, pf.store, pf.stream, pf.d_scalar);
    if (st == KU_EDATA) st = fail(KU_EINVAL, "ku_ctx_prefetch_shard: the slot table does not cover this shard's values "
                                             "(pass ku_db_values() of the whole database to ku_ctx_set_taxonomy)");
  }
  if (st != KU_OK) { store_free(pf.store); return st; }
  pf.db = db;
  pf.bin_lo = bin_lo;
  pf.bin_hi = bin_hi;
  pf.valid = true;
  return KU_OK;
}

extern "C" int ku_ctx_swap_shard(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi) {
  KU_TRY(check_ready(ctx));
  if (!db) return fail(KU_EINVAL, "ku_ctx_swap_shard: null argument");
  if (bin_lo > bin_hi || bin_hi > db->info.n_bins) return fail(KU_EINVAL, "bin range out of bounds");
  if (!ctx->extra.empty()) return fail(KU_EUNSUP, "chunked runs use one database (as the reference's: classify.cpp:639)");
  if (db->info.k != ctx->m.db.k) return fail(KU_EINVAL, "ku_ctx_swap_shard: k differs from the resident shard's");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  KU_TRY(ctx_seen_harvest(ctx));  // (marks of a fast-path run on the table that is about to go)
  ctx_drop_count_cache(ctx);
  if (ctx->pf.valid && ctx->pf.db == db && ctx->pf.bin_lo == bin_lo && ctx->pf.bin_hi == bin_hi) {
    // the chunk was prefetched (ku_ctx_prefetch_shard): it only has to change places with the resident one
    store_free(ctx->m);
    ctx->m = ctx->pf.store;
    ctx->pf.store = DbStore{};
    ctx->pf.valid = false;
    return KU_OK;
  }
  store_free(ctx->m);
  ctx->db_loaded = false;
  KU_TRY(store_upload(ctx, ctx->m, db, bin_lo, bin_hi, /*scan_values=*/false));
  int st = store_finalize(ctx, ctx->m);
  if (st == KU_EDATA) return fail(KU_EINVAL, "ku_ctx_swap_shard: the slot table does not cover this shard's values "
                                             "(pass ku_db_values() of the whole database to ku_ctx_set_taxonomy)");
  KU_TRY(st);
  ctx->db_loaded = true;
  return KU_OK;
}

extern "C" int ku_ctx_mem_info(ku_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  KU_TRY(ctx_activate(ctx));
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return KU_OK;
}

extern "C" void ku_batch_destroy(ku_batch *b) {
  if (!b) return;
  if (b->ctx) (void)hipSetDevice(b->ctx->device);
  for (void *p : {b->d_seqs, (void *)b->d_off, (void *)b->d_len, (void *)b->d_taxa})
    if (p) (void)hipFree(p);
  delete b;
}

extern "C" int ku_batch_create(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                               const uint32_t *seq_len, uint64_t n_reads, ku_batch **out) {
  if (!ctx || !out || (n_bytes && !seqs) || (n_reads && (!seq_off || !seq_len))) return fail(KU_EINVAL, "ku_batch_create: null argument");
  *out = nullptr;
  KU_TRY(ctx_activate(ctx));
  uint32_t max_len = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    if (seq_off[i] + seq_len[i] > n_bytes) return fail(KU_EINVAL, "read " + std::to_string(i) + " exceeds the sequence buffer");
    max_len = std::max(max_len, seq_len[i]);
  }
  ku_batch *b = new ku_batch();
  b->ctx = ctx; b->n_bytes = n_bytes; b->n_reads = n_reads; b->max_len = max_len;
  if (ctx->sp.on) { b->h_off.assign(seq_off, seq_off + n_reads); b->h_len.assign(seq_len, seq_len + n_reads); }
  hipStream_t s = ctx->stream;
  bool ok = hipMalloc(&b->d_seqs, n_bytes + 16) == hipSuccess && hipMalloc((void **)&b->d_off, std::max<uint64_t>(n_reads, 1) * 8) == hipSuccess &&
            hipMalloc((void **)&b->d_len, std::max<uint64_t>(n_reads, 1) * 4) == hipSuccess &&
            hipMalloc((void **)&b->d_taxa, (n_bytes + 16) * 4) == hipSuccess;
  if (!ok) { ku_batch_destroy(b); return fail(KU_ENOMEM, "device memory for a resident read batch"); }
  ok = (!n_bytes || hipMemcpyAsync(b->d_seqs, seqs, n_bytes, hipMemcpyHostToDevice, s) == hipSuccess) &&
       (!n_reads || (hipMemcpyAsync(b->d_off, seq_off, n_reads * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
                     hipMemcpyAsync(b->d_len, seq_len, n_reads * 4, hipMemcpyHostToDevice, s) == hipSuccess)) &&
       hipMemsetAsync(b->d_taxa, 0, (n_bytes + 16) * 4, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  if (!ok) { ku_batch_destroy(b); return fail(KU_EHIP, "upload of a resident read batch failed"); }
  *out = b;
  return KU_OK;
}

extern "C" int ku_batch_absorb(ku_ctx *ctx, ku_batch *dst, const ku_batch *src) {
  KU_TRY(check_ready(ctx));
  if (!dst || !src || dst->ctx != ctx || !src->ctx) return fail(KU_EINVAL, "ku_batch_absorb: null argument / batch of another context");
  if (dst->n_bytes != src->n_bytes || dst->n_reads != src->n_reads) return fail(KU_EINVAL, "ku_batch_absorb: the batches hold different reads");
  if (dst->finished || src->finished) return fail(KU_ESTATE, "ku_batch_absorb: a batch was already finished");
  if (dst->n_bytes == 0 || dst == src) return KU_OK;
  // the other copy's passes must be complete; then its slots come over (staged on this device when it lives on another)
  if (hipSetDevice(src->ctx->device) != hipSuccess || hipStreamSynchronize(src->ctx->stream) != hipSuccess) return fail(KU_EHIP, "ku_batch_absorb: the source context's stream failed");
  KU_TRY(ctx_activate(ctx));
  hipStream_t s = ctx->stream;
  const uint32_t *from = src->d_taxa;
  if (src->ctx->device != ctx->device) {
    HIP_TRY(hipStreamSynchronize(s));
    if (ctx->b_taxa.reserve(dst->n_bytes * 4) != KU_OK) return fail(KU_ENOMEM, "device memory for the slots of another GPU's batch");
    HIP_TRY(hipMemcpyAsync(ctx->b_taxa.p, src->d_taxa, dst->n_bytes * 4, hipMemcpyDefault, s));
    from = (const uint32_t *)ctx->b_taxa.p;
  }
  if (ku_launch_merge_max_u32(dst->d_taxa, from, dst->n_bytes, s) != KU_OK) return fail(KU_EHIP, "slot merge kernel launch failed");
  HIP_TRY(hipStreamSynchronize(s));
  return KU_OK;
}

extern "C" int ku_ctx_merge_state(ku_ctx *dst, ku_ctx *src) {
  KU_TRY(check_ready(dst));
  KU_TRY(check_ready(src));
  if (dst == src) return KU_OK;
  if (dst->tax.n_slots != src->tax.n_slots || dst->tax.n_nodes != src->tax.n_nodes) return fail(KU_EINVAL, "ku_ctx_merge_state: the contexts number their taxa differently");
  if (hipSetDevice(src->device) != hipSuccess || hipStreamSynchronize(src->stream) != hipSuccess) return fail(KU_EHIP, "ku_ctx_merge_state: the source context's stream failed");
  KU_TRY(ctx_activate(dst));
  hipStream_t s = dst->stream;
  const uint64_t n_regs = (uint64_t)dst->tax.n_slots * KU_HLL_M, n_slots = dst->tax.n_slots, n_nodes = dst->tax.n_nodes;
  const uint8_t *regs = src->cnt.registers;
  const unsigned long long *nk = src->cnt.n_kmers, *nr = src->cnt.n_reads;
  DevBuf stage;
  if (src->device != dst->device) {
    if (stage.reserve(n_regs + (n_slots + n_nodes) * 8) != KU_OK) return fail(KU_ENOMEM, "device memory for another GPU's per-taxon state");
    uint8_t *sp = (uint8_t *)stage.p;
    HIP_TRY(hipMemcpyAsync(sp, regs, n_regs, hipMemcpyDefault, s));
    HIP_TRY(hipMemcpyAsync(sp + n_regs, nk, n_slots * 8, hipMemcpyDefault, s));
    HIP_TRY(hipMemcpyAsync(sp + n_regs + n_slots * 8, nr, n_nodes * 8, hipMemcpyDefault, s));
    regs = sp;
    nk = (const unsigned long long *)(sp + n_regs);
    nr = (const unsigned long long *)(sp + n_regs + n_slots * 8);
  }
  if (ku_launch_merge_max_u8(dst->cnt.registers, regs, n_regs, s) != KU_OK || ku_launch_merge_add_u64(dst->cnt.n_kmers, nk, n_slots, s) != KU_OK ||
      ku_launch_merge_add_u64(dst->cnt.n_reads, nr, n_nodes, s) != KU_OK)
    return fail(KU_EHIP, "state merge kernel launch failed");
  HIP_TRY(hipStreamSynchronize(s));
  stage.release();
  return KU_OK;
}

extern "C" int ku_batch_lookup(ku_ctx *ctx, ku_batch *b, const ku_opts *opts) {
  KU_TRY(check_ready(ctx));
  if (!b || b->ctx != ctx) return fail(KU_EINVAL, "ku_batch_lookup: batch of another context");
  if (b->finished) return fail(KU_ESTATE, "ku_batch_lookup: the batch was already finished");
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  // quick mode does not shorten a chunk pass: the reference's chunked run books every k-mer of every read and only
  // derives the call differently at the end (classify.cpp:686-737)
  o.flags = (o.flags & ~KU_F_QUICK) | KU_F_MERGE_CHUNK | KU_F_KEEP_SLOTS;
  return ku_lookup_device(ctx, b->d_seqs, b->n_bytes, &o, b->d_taxa, nullptr);
}

extern "C" int ku_batch_finish(ku_ctx *ctx, ku_batch *b, const ku_opts *opts, uint32_t *calls, uint32_t *hits,
                               uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs) {
  KU_TRY(check_ready(ctx));
  if (!b || b->ctx != ctx) return fail(KU_EINVAL, "ku_batch_finish: batch of another context");
  if (!n_runs || (b->n_reads && (!calls || !run_off || !run_cnt))) return fail(KU_EINVAL, "ku_batch_finish: null buffer");
  if (b->finished) return fail(KU_ESTATE, "ku_batch_finish: the batch was already finished");
  *n_runs = 0;
  ctx->n_runs = 0;
  const uint64_t n_reads = b->n_reads;
  if (n_reads == 0) return KU_OK;
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  o.flags &= ~(KU_F_KEEP_SLOTS | KU_F_MERGE_CHUNK);
  o.max_read_len = b->max_len;
  const uint64_t runs_cap = b->n_bytes + 1;
  if (ctx->b_calls.reserve(n_reads * 4) || ctx->b_hits.reserve(n_reads * 4) || ctx->b_runs.reserve(runs_cap * 8) ||
      ctx->b_roff.reserve(n_reads * 8) || ctx->b_rcnt.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  if (ctx->sp.on && !(o.flags & KU_F_NO_COUNTS)) {  // the merged slots of all chunks are in place: the emulation's pass
    if (b->h_len.size() != n_reads) return fail(KU_ESTATE, "ku_batch_finish: enable the sparse-mode emulation before the batches are created");
    int sst = sparse_pass(ctx, b->d_seqs, b->d_off, b->d_len, b->h_off.data(), b->h_len.data(), n_reads, b->n_bytes, b->d_taxa, 0u, s);
    if (sst == KU_ENOMEM) {  // as in classify_device_impl: the run goes on without the emulation
      (void)hipStreamSynchronize(s);
      (void)hipGetLastError();
      ctx_free_sparse(ctx);
      ctx->sp.gave_up = true;
    } else if (sst != KU_OK) return sst;
  }
  if (ctx->d_exact_set && !(o.flags & KU_F_NO_COUNTS)) {
    // exact counting of a chunked run: the merged slots of all chunks are in place, and a chunked run books every k-mer of
    // every read whatever the mode (classify.cpp:686-737)
    int st = ku_launch_exact(ctx->m.db.k, (const uint8_t *)b->d_seqs, b->d_off, b->d_len, n_reads, b->d_taxa, ctx->d_exact_set, ctx->exact_mask,
                             ctx->d_exact_unique, ctx->d_scalar + 6, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "exact counting kernel launch failed");
  }
  if (o.flags & KU_F_QUICK) {  // the chunked run's quick mode: hits up to min_hits, call = the last k-mer's taxon
    int st = ku_launch_quick_chunked(ctx->tax, ctx->cnt, ctx->m.db.k, b->d_off, b->d_len, n_reads, o.flags, o.min_hits,
                                     (uint32_t *)ctx->b_calls.p, b->d_taxa, (uint32_t *)ctx->b_hits.p, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "quick-mode kernel launch failed");
  } else {
    KU_TRY(ku_resolve_device(ctx, b->d_seqs, b->d_off, b->d_len, n_reads, &o, (uint32_t *)ctx->b_calls.p, b->d_taxa,
                             (uint32_t *)ctx->b_hits.p, s));
  }
  b->finished = true;
  return rle_and_fetch(ctx, b->d_taxa, b->d_off, b->d_len, n_reads, runs_cap, (o.flags & KU_F_QUICK) != 0, calls, hits,
                       run_off, run_cnt, n_runs);
}

extern "C" int ku_ctx_synchronize(ku_ctx *ctx) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  KU_TRY(ctx_activate(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return KU_OK;
}
