// ku_api_rle.cpp -- C ABI: host batches through the fused kernel with run-length encoded output, in one step and in two
// (ku_classify_batch_rle_enqueue / _finish: up to KU_RLE_MAX_IN_FLIGHT batches in flight)
#include "ku_ctx.h"

// ---- ku_classify_batch_rle through the fused kernel with run-length encoded output (ku_short.hip, OUT >= 1): no
// per-k-mer array, no second kernel; with the sparse-mode emulation on, its fast path.

// The open unit's reads in tail form (bases of each read followed by '\n') go up to the device and through the exact
// evaluation as local unit `unit` of the pass `d`: a count-less lookup gives their slots (the flat kernel; these reads were
// classified, booked and marked when their batch went through the fused kernel), ku_sparse_insert_kernel feeds L / U as
// for any staged batch.  Positions start at 2; *pos_end = the first position the reads behind the tail may use.
static int sparse_tail_insert(ku_ctx *ctx, const KuSparseDev &d, const std::vector<char> &text, const std::vector<uint32_t> &lens,
                              uint32_t unit, hipStream_t s, uint32_t *pos_end) {
  ku_ctx::Sparse &sp = ctx->sp;
  const uint64_t n_reads = lens.size(), n_bytes = text.size();
  if (pos_end) *pos_end = (uint32_t)n_bytes;
  if (n_reads == 0) return KU_OK;
  std::vector<uint64_t> off(n_reads);
  uint64_t at = 0;
  for (uint64_t r = 0; r < n_reads; ++r) { off[r] = at; at += (uint64_t)lens[r] + 1; }
  if (at != n_bytes) return fail(KU_ESTATE, "sparse-mode emulation: the open unit's reads are inconsistent");
  if (sp.t_seqs.reserve(n_bytes + 16) || sp.t_off.reserve(n_reads * 8) || sp.t_len.reserve(n_reads * 4) || sp.t_taxa.reserve((n_bytes + 16) * 4) ||
      sp.t_unit.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device memory for the open work unit's reads");
  HIP_TRY(hipMemcpyAsync(sp.t_seqs.p, text.data(), n_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(sp.t_off.p, off.data(), n_reads * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(sp.t_len.p, lens.data(), n_reads * 4, hipMemcpyHostToDevice, s));
  std::vector<uint32_t> units(n_reads, unit);
  HIP_TRY(hipMemcpyAsync(sp.t_unit.p, units.data(), n_reads * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // (`off`, `units` go out of scope; this path is rare)
  int st = ku_launch_lookup(ctx->m.db, ctx->cnt, (const uint8_t *)sp.t_seqs.p, n_bytes, (uint32_t *)sp.t_taxa.p, /*do_counts=*/false, false, false,
                            ctx->n_cu, s);
  if (st != KU_OK) return fail(st, "lookup kernel launch failed");
  return ku_launch_sparse_insert(d, ctx->m.db.k, (const uint8_t *)sp.t_seqs.p, (const uint64_t *)sp.t_off.p, (const uint32_t *)sp.t_len.p,
                                 (const uint32_t *)sp.t_unit.p, n_reads, (const uint32_t *)sp.t_taxa.p, 0u, ctx->n_cu, s);
}

// a read of the caller's batch joins the open unit's tail
static void sparse_tail_append(ku_ctx::Sparse &sp, const char *seqs, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t r0, uint64_t r1) {
  for (uint64_t r = r0; r < r1; ++r) {
    sp.tail_text.insert(sp.tail_text.end(), seqs + seq_off[r], seqs + seq_off[r] + seq_len[r]);
    sp.tail_text.push_back('\n');
    sp.tail_len.push_back(seq_len[r]);
  }
}

// The open unit changes from tail form into the staged form (its L / U entries in the carry buffers): what a staged batch and
// ku_ctx_sparse_move_open_unit expect.  Every slot that is not dense is tracked, as the staged passes do.
int sparse_tail_to_carry(ku_ctx *ctx) {
  ku_ctx::Sparse &sp = ctx->sp;
  if (!sp.tail_open) return KU_OK;
  hipStream_t s = ctx->stream;
  KU_TRY(sparse_reserve_global(ctx, sp.tail_text.size(), s));
  KuSparseDev d;
  KU_TRY(sparse_pass_tables(ctx, sp.tail_text.size(), &d, s));
  KU_TRY(sparse_tail_insert(ctx, d, sp.tail_text, sp.tail_len, 0u, s, nullptr));
  KU_TRY(ku_launch_sparse_close(d, 0u, s));  // nothing closes: the largest first positions for the carry
  HIP_TRY(hipMemsetAsync(sp.d_counters + 1, 0, 16, s));
  KU_TRY(ku_launch_sparse_carry_out(d, 0u, (unsigned long long *)sp.carry_l.p, (uint32_t *)sp.carry_u.p, sp.d_counters + 1, sp.cap_carry_l,
                                    sp.cap_carry_u, s));
  unsigned long long c[3] = {0, 0, 0};
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(c, sp.d_counters, 24, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, sp.dev.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table is full");
  sp.g_count = c[0];
  sp.n_carry_l = std::min<uint64_t>(c[1], sp.cap_carry_l);
  sp.n_carry_u = std::min<uint64_t>(c[2], sp.cap_carry_u);
  sp.open = true;
  sp.tail_open = false;
  sp.tail_text.clear();
  sp.tail_len.clear();
  return KU_OK;
}

// The open unit in tail form ends here (end of an input file / of the run): it can only have turned a sketch dense if it gave
// it >= 1025 inserts -- then, and only then, the exact evaluation runs over its reads.
int sparse_tail_close(ku_ctx *ctx) {
  ku_ctx::Sparse &sp = ctx->sp;
  if (!sp.tail_open) return KU_OK;
  hipStream_t s = ctx->stream;
  if (sp.u_flag.reserve(4)) return fail(KU_ENOMEM, "device memory for the work-unit counters");
  HIP_TRY(hipMemsetAsync(sp.u_flag.p, 0, 4, s));
  KU_TRY(ku_launch_sparse_flag_units((const uint32_t *)sp.tail_row.p, ctx->tax.n_slots, ctx->tax.n_slots, sp.dev.dense, (uint8_t *)sp.u_flag.p, s));
  uint32_t flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, sp.u_flag.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (flag & 0xFFu) {
    KU_TRY(sparse_reserve_global(ctx, sp.tail_text.size(), s));
    KuSparseDev d;
    KU_TRY(sparse_pass_tables(ctx, sp.tail_text.size(), &d, s));
    KU_TRY(sparse_tail_insert(ctx, d, sp.tail_text, sp.tail_len, 0u, s, nullptr));
    KU_TRY(ku_launch_sparse_close(d, 1u, s, /*skip_hits=*/true));
    unsigned long long c = 0;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&c, sp.d_counters, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&err, sp.dev.err, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table is full");
    sp.g_count = c;
  }
  sp.tail_open = false;
  sp.tail_text.clear();
  sp.tail_len.clear();
  return KU_OK;
}

// KU_RLE_TIMES=1: where the batch calls spend their time on the host, summed over the run, printed when the context goes
static double g_rle_t[10];  // checks, plan + enqueue, waiting for the device in _finish, behind the wait, calls; of the enqueue: [5] buffers + plan ([9]: sparse_reserve_global in it), [6] uploads, [7] launches, [8] copies back + events
static double g_rle_x[6];  // of 'behind the wait': [0] flagging again, [1] exact passes, [2] their number, [3] units they evaluated, [4] reads in them
static double g_rle_kernel_ms = 0;  // HIP events around every batch's kernels (fused kernel + the emulation's flag kernel): the time
                                    // covered by the batches' intervals -- they overlap since the batches' kernels run on two streams --
static double g_rle_kernel_sum_ms = 0, g_rle_cover_end = 0;  // ... their plain sum, and where the covered time ends (ms behind g_rle_ref)
static hipEvent_t g_rle_ref = nullptr;
static unsigned long long g_rle_reads = 0;
static const bool g_rle_times = getenv("KU_RLE_TIMES") != nullptr;
static double rle_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
void rle_times_print() {
  if (g_rle_times && g_rle_t[4] > 0)
    fprintf(stderr, "ku_classify_batch_rle over %.0f batches: checks %.3f s, plan + enqueue %.3f s, waiting for the device %.3f s, behind the wait %.3f s; "
                    "kernels %.3f ms for %llu reads (HIP events on their streams: the time the batches' intervals cover; their sum is %.3f ms); of the enqueue: buffers + plan %.3f s, uploads %.3f s, launches %.3f s, "
                    "copies back + events %.3f s; of buffers + plan: room in the emulation's run-wide set %.3f s\n",
            g_rle_t[4], g_rle_t[0], g_rle_t[1], g_rle_t[2], g_rle_t[3], g_rle_kernel_ms, g_rle_reads, g_rle_kernel_sum_ms, g_rle_t[5], g_rle_t[6], g_rle_t[7], g_rle_t[8], g_rle_t[9]);
  if (g_rle_times && g_rle_x[2] > 0)
    fprintf(stderr, "ku_classify_batch_rle, behind the wait: flagging again %.3f s, %.0f exact passes over %.0f work units (%.0f reads) %.3f s, of it %.3f s waiting for their kernels\n", g_rle_x[0], g_rle_x[2],
            g_rle_x[3], g_rle_x[4], g_rle_x[1], g_rle_x[5]);
}

// The exact per-unit evaluation of the emulation for the units the fused kernel could not settle by counting: `flagged`
// (ascending unit numbers of the job's batch; flag_all: every slot of the unit is tracked -- a unit in the staged form).
// last_is_open: the last flagged unit stays open behind the batch (staged form: its entries go into the carry buffers).
static int sparse_fast_exact(ku_ctx *ctx, RleJob &j, const std::vector<uint32_t> &flagged, const std::vector<uint8_t> &flag_all,
                             bool last_is_open, hipStream_t s) {
  ku_ctx::Sparse &sp = ctx->sp;
  const uint32_t *h_len = j.h_len;
  const std::vector<uint64_t> &unit_first_read = j.unit_first_read;
  const uint32_t *d_u_cnt = (const uint32_t *)j.u_cnt.p;
  // the batch's runs: its own run array -- or, when that overflowed, the context's buffers, where rle_job_finish's redo (per-k-mer
  // codes + ku_rle_kernel, whose bound cannot overflow) left them; counts, SEEN marks and u_cnt are the fused kernel's either way
  const void *d_runs = j.runs_in_ctx ? ctx->b_runs.p : j.runs.p;
  const uint64_t *d_roff = (const uint64_t *)(j.runs_in_ctx ? ctx->b_roff.p : j.roff.p);
  const uint32_t *d_rcnt = (const uint32_t *)(j.runs_in_ctx ? ctx->b_rcnt.p : j.rcnt.p);
  // unit 0 continues a unit in tail form: its earlier reads are evaluated with it, and come first in the position space
  const bool with_tail = j.cont_tail && !flagged.empty() && flagged[0] == 0 && !j.tail_len.empty();
  const uint64_t tail_bytes = with_tail ? j.tail_text.size() : 0;
  size_t at = 0;
  bool first_pass = true;
  while (at < flagged.size()) {
    // units of this pass: at most 2^25 bases and KU_SPARSE_MAX_UNITS units (the tables of ku_sparse.hip)
    size_t end = at;
    uint64_t bases = first_pass ? tail_bytes : 0, n_list = 0;
    while (end < flagged.size() && end - at < KU_SPARSE_MAX_UNITS) {
      const uint32_t u = flagged[end];
      uint64_t ub = 0;
      for (uint64_t r = unit_first_read[u]; r < unit_first_read[u + 1]; ++r) ub += h_len[r];
      if (end > at && bases + ub > (1ull << 25)) break;
      bases += ub;
      n_list += unit_first_read[u + 1] - unit_first_read[u];
      ++end;
    }
    const bool has_open = last_is_open && end == flagged.size();
    std::vector<uint32_t> list(3 * n_list);
    uint64_t li = 0;
    for (size_t f = at; f < end; ++f) {
      const uint32_t u = flagged[f];
      for (uint64_t r = unit_first_read[u]; r < unit_first_read[u + 1]; ++r, ++li) {
        list[li] = (uint32_t)r;
        list[n_list + li] = (uint32_t)(f - at) | (flag_all[f] ? 0x80000000u : 0u);
        list[2 * n_list + li] = u;
      }
    }
    if (sp.list.reserve(std::max<uint64_t>(n_list, 1) * 12) != KU_OK) return fail(KU_ENOMEM, "device memory for the flagged work units' reads");
    KU_TRY(sparse_reserve_global(ctx, bases + sp.n_carry_l, s));
    KuSparseDev d;  // this pass's view: the run-wide set as it is now, L / U sized for the pass
    KU_TRY(sparse_pass_tables(ctx, bases + sp.n_carry_l + sp.n_carry_u, &d, s));
    if (first_pass && j.cont_carry)  // the unit carried over from the batch before is local unit 0 of the first pass
      KU_TRY(ku_launch_sparse_carry_in(d, (const unsigned long long *)sp.carry_l.p, sp.n_carry_l, (const uint32_t *)sp.carry_u.p, sp.n_carry_u, s));
    uint32_t pos_base = 0;
    if (first_pass && with_tail) KU_TRY(sparse_tail_insert(ctx, d, j.tail_text, j.tail_len, 0u, s, &pos_base));
    if (n_list) HIP_TRY(hipMemcpyAsync(sp.list.p, list.data(), n_list * 12, hipMemcpyHostToDevice, s));
    const uint32_t *dl = (const uint32_t *)sp.list.p;
    KU_TRY(ku_launch_sparse_insert_runs(d, ctx->m.db.k, (const uint8_t *)j.seqs.p, (const uint64_t *)j.off.p, (const uint32_t *)j.len.p,
                                        dl, dl + n_list, dl + 2 * n_list, n_list, d_runs, d_roff, d_rcnt, ctx->d_slot_taxid, ctx->tax.n_slots,
                                        d_u_cnt, ctx->n_cu, s, pos_base));
    const uint32_t n_local = (uint32_t)(end - at);
    // (a unit in the staged form may hold k-mers of a staged batch, which marks nothing in the probe table: its entries all
    // go into the set; the fast path's own units only contribute their misses)
    KU_TRY(ku_launch_sparse_close(d, has_open ? n_local - 1 : n_local, s, /*skip_hits=*/!j.cont_carry));
    if (first_pass && j.cont_carry) sp.n_carry_l = sp.n_carry_u = 0;
    unsigned long long c[3] = {0, 0, 0};
    if (has_open) {
      HIP_TRY(hipMemsetAsync(sp.d_counters + 1, 0, 16, s));
      KU_TRY(ku_launch_sparse_carry_out(d, n_local - 1, (unsigned long long *)sp.carry_l.p, (uint32_t *)sp.carry_u.p, sp.d_counters + 1,
                                        sp.cap_carry_l, sp.cap_carry_u, s));
    }
    HIP_TRY(hipMemcpyAsync(c, sp.d_counters, 24, hipMemcpyDeviceToHost, s));
    const double t_sy0 = g_rle_times ? rle_now() : 0.0;
    HIP_TRY(hipStreamSynchronize(s));  // `list` goes out of scope
    if (g_rle_times) g_rle_x[5] += rle_now() - t_sy0;
    sp.g_count = std::max<uint64_t>(sp.g_count, c[0]);
    if (has_open) {
      sp.n_carry_l = std::min<uint64_t>(c[1], sp.cap_carry_l);
      sp.n_carry_u = std::min<uint64_t>(c[2], sp.cap_carry_u);
    }
    first_pass = false;
    at = end;
  }
  return KU_OK;
}

// may the batch take the fused kernel with run-length encoded output?  (the same conditions as the fused path of
// classify_device_impl, plus what the emulation's fast path needs)
static bool rle_fused_eligible(ku_ctx *ctx, uint32_t flags, uint32_t max_n, uint64_t n_bytes, uint64_t n_reads, bool monotonic) {
  if (getenv("KU_NO_FUSED") || getenv("KU_NO_FUSED_RLE") || !ctx->extra.empty() || ctx->d_exact_set) return false;
  if (flags & (KU_F_QUICK | KU_F_KEEP_SLOTS)) return false;
  const uint32_t short_max = ku_short_max_kmers(ctx->m.db);
  if (!short_max) return false;
  if (max_n > short_max && (max_n > ku_short_max_kmers_windowed(ctx->m.db) || getenv("KU_NO_WINDOWED"))) return false;
  if (n_reads >= (1ull << 32)) return false;
  const bool sparse = ctx->sp.on && !(flags & KU_F_NO_COUNTS);
  if (sparse) {
    const ku_ctx::Sparse &sp = ctx->sp;
    // (positions of the exact evaluation are 32-bit: the batch, behind the reads of an open unit of at most 2^24 nt + one read)
    if (getenv("KU_NO_SPARSE_FAST") || !monotonic || sp.unit_nt == 0 || sp.unit_nt > (1ull << 24) || n_bytes + (1ull << 26) >= (1ull << 32)) return false;
    const uint64_t max_units = n_bytes / sp.unit_nt + 2;
    if (max_units * ctx->tax.n_slots > (1ull << 29)) return false;  // the (unit, slot) counters: at most 2 GiB
  }
  return true;
}

// chunk of the run array a wave claims at a time: large enough for few claims, small enough that the unused tails of the
// last chunks do not dominate a small batch
static uint32_t rle_chunk(uint64_t n_reads, uint64_t total_waves) {
  const uint64_t reads_per_wave = n_reads / std::max<uint64_t>(total_waves, 1);
  return reads_per_wave >= 64 ? 256u : (reads_per_wave >= 16 ? 64u : 16u);
}

extern "C" uint64_t ku_device_rle_runs_cap(const ku_ctx *ctx, uint64_t n_bytes, uint64_t n_reads, uint32_t max_read_len) {
  if (!ctx || !ctx->tax_set) return 0;
  const uint32_t max_n = max_read_len >= ctx->m.db.k ? max_read_len - ctx->m.db.k + 1 : 0;
  const uint64_t waves = ku_short_grid_waves(n_reads, max_n, ctx->n_cu);
  return n_bytes / 6 + 4 * n_reads + waves * rle_chunk(n_reads, waves) + 4096;
}

extern "C" int ku_classify_batch_device_rle(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                                            const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                                            ku_run *d_runs, uint64_t runs_cap, uint64_t *d_run_off, uint32_t *d_run_cnt,
                                            uint64_t *d_n_runs, void *stream) {
  KU_TRY(check_ready(ctx));
  if (!d_n_runs || (n_reads && (!d_seqs || !d_seq_off || !d_seq_len || !d_calls || !d_runs || !d_run_off || !d_run_cnt)))
    return fail(KU_EINVAL, "ku_classify_batch_device_rle: null buffer");
  const ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  HIP_TRY(hipMemsetAsync(d_n_runs, 0, 8, s));
  if (n_reads == 0) return KU_OK;
  if (o.max_read_len == 0) return fail(KU_EINVAL, "ku_classify_batch_device_rle: opts->max_read_len (the longest read of the batch) is required");
  const uint32_t max_n = o.max_read_len >= ctx->m.db.k ? o.max_read_len - ctx->m.db.k + 1 : 0;
  if (ctx->sp.on && !(o.flags & KU_F_NO_COUNTS))
    return fail(KU_EUNSUP, "the sparse-mode emulation runs through the host-buffer entry points (it needs the read lengths on the host)");
  if (!store_whole(ctx->m) || !rle_fused_eligible(ctx, o.flags, max_n, n_bytes, n_reads, true))
    return fail(KU_EUNSUP, "ku_classify_batch_device_rle: the fused kernel does not apply to this context / these options (ku_classify_batch_device does)");
  uint64_t ws = 0;
  if (max_n > ku_short_max_kmers(ctx->m.db)) {  // windowed instance: its spill workspace
    ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, n_reads, ctx->n_cu);
    if (ws > ctx->b_ws.cap) HIP_TRY(hipStreamSynchronize(s));
    if (ctx->b_ws.reserve(ws)) return fail(KU_ENOMEM, "device memory for the windowed kernel's workspace");
  }
  const uint64_t waves = ku_short_grid_waves(n_reads, max_n, ctx->n_cu);
  KuRunsOut ro{};
  ro.runs = (uint2 *)d_runs;
  ro.counter = (unsigned long long *)d_n_runs;
  ro.cap = runs_cap;
  ro.chunk = rle_chunk(n_reads, waves);
  ro.run_off = d_run_off;
  ro.run_cnt = d_run_cnt;
  // every wave owns its first chunk, the counter starts behind those (as in rle_job_enqueue: no claim storm at the launch's start)
  if (waves * ro.chunk <= runs_cap && waves * ro.chunk < (1ull << 31)) {
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_n_runs, (int)(waves * ro.chunk), 1, s));
    ro.pre_base1 = 1;
  }
  int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)d_seqs, n_bytes, d_seq_off, d_seq_len, n_reads, max_n, o.flags,
                                    d_calls, nullptr, nullptr, ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s, &ro, nullptr);
  return st == KU_OK ? KU_OK : fail(st, "fused kernel launch failed");
}

// no batch may be in flight (entry points that read or change what the batches in flight work on)
int rle_idle(const ku_ctx *ctx, const char *who) {
  if (ctx->rle_in_flight) return fail(KU_ESTATE, std::string(who) + ": batches are in flight (ku_classify_batch_rle_finish first)");
  return KU_OK;
}

// ---- step one: plan the batch, start its upload (in segments, on the copy stream), its kernels and the copies back.
// Nothing here waits for the device.
static int rle_job_enqueue(ku_ctx *ctx, RleJob &j, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off, const uint32_t *seq_len,
                           uint64_t n_reads, const ku_opts &o, uint32_t max_n, bool monotonic, uint32_t *calls, uint32_t *hits,
                           uint64_t *run_off, uint32_t *run_cnt, ku_run *h_runs, uint64_t h_runs_cap) {
  // the batch's kernels: consecutive batches take the two kernel streams in turn (KU_RLE_KERNEL_STREAMS=1: the context's one stream)
  static const bool one_kernel_stream = getenv("KU_RLE_KERNEL_STREAMS") && atoi(getenv("KU_RLE_KERNEL_STREAMS")) == 1;
  hipStream_t s = ctx->stream;
  if (!one_kernel_stream) {
    hipStream_t &ks = ctx->k_streams[(&j - &ctx->rle[0]) & 1];
    if (!ks) HIP_TRY(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
    if (!ctx->main_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->main_ev, hipEventDisableTiming));
    s = ks;
  }
  const double t_in = g_rle_times ? rle_now() : 0.0;
  const bool counts = !(o.flags & KU_F_NO_COUNTS);
  const bool sparse = ctx->sp.on && counts;
  ku_ctx::Sparse &sp = ctx->sp;
  // ---- segments of the batch: cut at read boundaries, uploaded one after the other on the copy stream while the
  // compute stream classifies the ones before
  uint64_t n_seg = 1;
  // (16 MiB per segment since round 5 -- was 8: with several batches in flight the overlap of upload and kernels comes from the
  // OTHER batches, and a launch of 120 k reads costs 3.0 us per thousand reads where two of 60 k cost 3.8, ku_short.hip)
  static const uint64_t seg_bytes = (uint64_t)std::max(1, getenv("KU_RLE_SEG_MB") ? atoi(getenv("KU_RLE_SEG_MB")) : 16) << 20;
  if (monotonic && !getenv("KU_NO_H2D_OVERLAP")) n_seg = std::min<uint64_t>(8, std::max<uint64_t>(1, n_bytes / seg_bytes));
  std::vector<uint64_t> seg(n_seg + 1, 0);
  for (uint64_t g = 1; g < n_seg; ++g) {
    const uint64_t target = n_bytes / n_seg * g;
    seg[g] = std::max<uint64_t>(seg[g - 1], (uint64_t)(std::lower_bound(seq_off, seq_off + n_reads, target) - seq_off));
  }
  seg[n_seg] = n_reads;
  uint64_t total_waves = 0, max_seg_reads = 0;
  for (uint64_t g = 0; g < n_seg; ++g) {
    total_waves += ku_short_grid_waves(seg[g + 1] - seg[g], max_n, ctx->n_cu);
    max_seg_reads = std::max(max_seg_reads, seg[g + 1] - seg[g]);
  }
  // a wave claims `chunk` run entries at a time: large enough for few claims, small enough that the unused tails of the
  // last chunks do not dominate a small batch
  const uint32_t chunk = rle_chunk(n_reads, total_waves);
  // room for ~ one run per 6 bases + the chunk tails; a batch that needs more (many taxa per read) is redone through
  // the per-k-mer array (in _finish), whose run-length encoder cannot overflow
  uint64_t runs_cap = n_bytes / 6 + 4 * n_reads + total_waves * chunk + 4096;
  if (const char *e = getenv("KU_RUNS_CAP")) runs_cap = std::max<uint64_t>(1, (uint64_t)atoll(e));  // test hook
  uint64_t ws = 0;
  if (max_n > ku_short_max_kmers(ctx->m.db)) ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, max_seg_reads, ctx->n_cu);
  // (the job's buffers are its own and its previous batch is through: growing them needs no synchronisation of ours)
  if (j.seqs.reserve(n_bytes + 16) || j.off.reserve(n_reads * 8) || j.len.reserve(n_reads * 4) || j.calls.reserve(n_reads * 4) ||
      j.runs.reserve(runs_cap * 8) || j.roff.reserve(n_reads * 8) || j.rcnt.reserve(n_reads * 4) || j.ws.reserve(ws))
    return fail(KU_ENOMEM, "device batch buffers");
  if (!ctx->h2d_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
  if (!ctx->d2h_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
  if (!j.done) HIP_TRY(hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
  if (!j.kernels_done) HIP_TRY(hipEventCreateWithFlags(&j.kernels_done, hipEventDisableTiming));
  while (j.seg_events.size() < n_seg) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    j.seg_events.push_back(e);
  }
  // ---- sparse fast path: work-unit plan (the unit closes behind the read that fills it, classify.cpp:510-521)
  j.sparse = sparse;
  j.cont_carry = j.cont_tail = j.open_after = false;
  j.n_units = 0;
  j.kmers = 0;
  j.unit_first_read.clear();
  j.tail_text.clear();
  j.tail_len.clear();
  KuSparseFast sf{};
  uint32_t *h_unit = nullptr;
  uint8_t *h_flag = nullptr;
  if (sparse) {
    if (j.pin_unit.reserve(n_reads * 4)) return fail(KU_ENOMEM, "page-locked memory for the work-unit plan");
    h_unit = (uint32_t *)j.pin_unit.p;
    uint64_t acc = sp.acc_nt;
    uint32_t cur = 0;
    j.cont_carry = sp.open;
    j.cont_tail = sp.tail_open && !sp.open;
    j.unit_first_read.push_back(0);
    for (uint64_t r = 0; r < n_reads; ++r) {
      h_unit[r] = cur;
      acc += seq_len[r];
      if (acc >= sp.unit_nt) { ++cur; acc = 0; j.unit_first_read.push_back(r + 1); }
    }
    j.open_after = acc > 0;
    j.acc_after = acc;
    j.n_units = cur + (j.unit_first_read.back() < n_reads ? 1u : 0u);
    if (j.unit_first_read.back() < n_reads) j.unit_first_read.push_back(n_reads);
    const uint64_t cells = (uint64_t)j.n_units * ctx->tax.n_slots;
    if (j.unit.reserve(n_reads * 4) || j.u_cnt.reserve(std::max<uint64_t>(cells, 1) * 4) || j.u_flag.reserve(((uint64_t)std::max<uint32_t>(j.n_units, 1) + 3) & ~3ull) ||
        sp.tail_row.reserve((size_t)ctx->tax.n_slots * 4))
      return fail(KU_ENOMEM, "device memory for the work-unit counters");
    for (uint64_t r = 0; r < n_reads; ++r) j.kmers += seq_len[r] >= ctx->m.db.k ? seq_len[r] - ctx->m.db.k + 1 : 0;
    uint64_t in_flight_kmers = 0;  // what the batches in flight may still add: the host's count of the set lags behind them
    for (const RleJob &q : ctx->rle) if (q.busy && &q != &j && q.sparse) in_flight_kmers += q.kmers;
    const double t_g0 = g_rle_times ? rle_now() : 0.0;
    KU_TRY(sparse_reserve_global(ctx, j.kmers + in_flight_kmers + sp.n_carry_l, ctx->stream));  // (drains the batches in flight when it grows the set)
    if (g_rle_times) g_rle_t[9] += rle_now() - t_g0;
    sf.g_key = sp.dev.g_key;
    sf.g_mask = sp.dev.g_mask;
    sf.g_count = sp.dev.g_count;
    sf.dense = sp.dev.dense;
    sf.u_cnt = (uint32_t *)j.u_cnt.p;
    sf.err = sp.dev.err;
    sf.n_slots = ctx->tax.n_slots;
    sf.unit_base = 0;
    ctx->m.seen_dirty = true;  // the kernel books the k-mers the database holds by marking their table entries
  }
  if (j.pin.reserve(64 + (size_t)std::max<uint32_t>(j.n_units, 1) + 8)) return fail(KU_ENOMEM, "page-locked memory for the batch totals");
  unsigned long long *h_tot = (unsigned long long *)j.pin.p;
  h_flag = (uint8_t *)j.pin.p + 64;
  if (g_rle_times) g_rle_t[5] += rle_now() - t_in;
  if (s != ctx->stream) {  // whatever was queued on the context's own stream before this batch comes first
    HIP_TRY(hipEventRecord(ctx->main_ev, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(s, ctx->main_ev, 0));
  }
  unsigned long long *d_counter = j.d_counter;
  // (emulation) the per-(unit, slot) insert counts and the unit flags start at zero: one launch
  if (sparse) {
    if (ku_launch_zero3(d_counter, 2, j.u_cnt.p, std::max<uint64_t>((uint64_t)j.n_units * ctx->tax.n_slots, 1), j.u_flag.p,
                        ((uint64_t)std::max<uint32_t>(j.n_units, 1) + 3) / 4, s) != KU_OK)
      return fail(KU_EHIP, "clearing the batch counters failed");
    // (unit 0 continues the open unit: the inserts that unit has had so far join its row BEHIND the kernels, below)
  }
  // the run counter starts behind the chunks the waves own from the start (one per wave of every segment's launch)
  h_tot[3] = total_waves * (unsigned long long)chunk;
  HIP_TRY(hipMemcpyAsync(d_counter, &h_tot[3], 8, hipMemcpyHostToDevice, s));
  KuRunsOut ro{};
  ro.runs = (uint2 *)j.runs.p;
  ro.counter = d_counter;
  ro.cap = runs_cap;
  ro.chunk = chunk;
  if (g_rle_times && !j.t_k0) { HIP_TRY(hipEventCreate(&j.t_k0)); HIP_TRY(hipEventCreate(&j.t_k1)); }
  bool clock_started = false;
  uint64_t waves_before = 0;
  for (uint64_t g = 0; g < n_seg; ++g) {
    const uint64_t a = seg[g], b = seg[g + 1];
    const uint64_t lo = g == 0 ? 0 : seq_off[a], hi = g + 1 == n_seg ? n_bytes : seq_off[b];
    // (always the copy stream: with a batch in flight, this one's upload runs under that one's kernels)
    hipStream_t cs = ctx->h2d_stream;
    const double t_u0 = g_rle_times ? rle_now() : 0.0;
    if (hi > lo) HIP_TRY(hipMemcpyAsync((char *)j.seqs.p + lo, seqs + lo, hi - lo, hipMemcpyHostToDevice, cs));
    if (b > a) {
      HIP_TRY(hipMemcpyAsync((uint64_t *)j.off.p + a, seq_off + a, (b - a) * 8, hipMemcpyHostToDevice, cs));
      HIP_TRY(hipMemcpyAsync((uint32_t *)j.len.p + a, seq_len + a, (b - a) * 4, hipMemcpyHostToDevice, cs));
      if (sparse) HIP_TRY(hipMemcpyAsync((uint32_t *)j.unit.p + a, h_unit + a, (b - a) * 4, hipMemcpyHostToDevice, cs));
    }
    HIP_TRY(hipEventRecord(j.seg_events[g], cs));
    HIP_TRY(hipStreamWaitEvent(s, j.seg_events[g], 0));
    const double t_u1 = g_rle_times ? rle_now() : 0.0;
    if (g_rle_times) g_rle_t[6] += t_u1 - t_u0;
    if (b == a) continue;
    ro.run_off = (uint64_t *)j.roff.p + a;
    ro.run_cnt = (uint32_t *)j.rcnt.p + a;
    ro.pre_base1 = (uint32_t)(1 + waves_before);
    waves_before += ku_short_grid_waves(b - a, max_n, ctx->n_cu);
    sf.unit_of = sparse ? (const uint32_t *)j.unit.p + a : nullptr;
    if (g_rle_times && !clock_started) {  // (behind the first segment's upload)
      if (!g_rle_ref) { HIP_TRY(hipEventCreate(&g_rle_ref)); HIP_TRY(hipEventRecord(g_rle_ref, s)); }
      HIP_TRY(hipEventRecord(j.t_k0, s));
      clock_started = true;
    }
    int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)j.seqs.p, n_bytes, (const uint64_t *)j.off.p + a,
                                      (const uint32_t *)j.len.p + a, b - a, max_n, o.flags, (uint32_t *)j.calls.p + a, nullptr, nullptr,
                                      j.ws.p, j.ws.cap, ctx->n_cu, s, &ro, sparse ? &sf : nullptr);
    if (st != KU_OK) { (void)hipStreamSynchronize(ctx->h2d_stream); (void)hipStreamSynchronize(s); return fail(st, "fused kernel launch failed"); }
    if (g_rle_times) g_rle_t[7] += rle_now() - t_u1;
  }
  const double t_c0 = g_rle_times ? rle_now() : 0.0;
  if (sparse) {
    // The open unit's insert counts (sp.tail_row) travel from batch to batch through this small step behind the kernels: it
    // reads what the batch before wrote behind ITS kernels on the other stream, and what it writes must not land before that
    // batch has read -- so the step (not the kernels) waits for the whole of the batch before, in enqueue order.
    if (ctx->prev_kernels_done && s != ctx->stream) HIP_TRY(hipStreamWaitEvent(s, ctx->prev_kernels_done, 0));
    // unit 0 continues the open unit: the inserts that unit had before this batch join its row
    if (j.cont_tail && j.n_units)
      KU_TRY(ku_launch_add_u32((uint32_t *)j.u_cnt.p, (const uint32_t *)sp.tail_row.p, ctx->tax.n_slots, s));
    KU_TRY(ku_launch_sparse_flag_units((const uint32_t *)j.u_cnt.p, (uint64_t)j.n_units * ctx->tax.n_slots, ctx->tax.n_slots, sp.dev.dense,
                                       (uint8_t *)j.u_flag.p, s));
    // the unit that stays open (tail form): its insert counts so far
    if (j.open_after && !(j.cont_carry && j.n_units == 1))
      HIP_TRY(hipMemcpyAsync(sp.tail_row.p, (const uint32_t *)j.u_cnt.p + (size_t)(j.n_units - 1) * ctx->tax.n_slots, (size_t)ctx->tax.n_slots * 4,
                             hipMemcpyDeviceToDevice, s));
  }
  if (g_rle_times && clock_started) HIP_TRY(hipEventRecord(j.t_k1, s));
  // ---- the copies back run on a stream of their own, behind this batch's kernels -- not in front of the next batch's
  static const bool own_d2h_stream = !(getenv("KU_RLE_D2H_STREAM") && atoi(getenv("KU_RLE_D2H_STREAM")) == 0);
  hipStream_t ds = own_d2h_stream ? ctx->d2h_stream : s;
  HIP_TRY(hipEventRecord(j.kernels_done, s));
  ctx->prev_kernels_done = j.kernels_done;  // (the jobs' events live as long as the context)
  if (own_d2h_stream) {
    HIP_TRY(hipStreamWaitEvent(ds, j.kernels_done, 0));
  }
  HIP_TRY(hipMemcpyAsync(calls, j.calls.p, n_reads * 4, hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipMemcpyAsync(run_off, j.roff.p, n_reads * 8, hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipMemcpyAsync(run_cnt, j.rcnt.p, n_reads * 4, hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipMemcpyAsync(&h_tot[0], d_counter, 8, hipMemcpyDeviceToHost, ds));
  // the runs themselves, when the caller said where they go: as much of the run array as its buffer holds (the extent in use
  // is only known behind the kernel; _finish tells whether it fitted -- else ku_fetch_runs, into a larger buffer)
  j.runs_copied = h_runs ? std::min<uint64_t>(h_runs_cap, runs_cap) : 0;
  if (j.runs_copied) HIP_TRY(hipMemcpyAsync(h_runs, j.runs.p, j.runs_copied * 8, hipMemcpyDeviceToHost, ds));
  if (sparse) {
    HIP_TRY(hipMemcpyAsync(h_flag, j.u_flag.p, std::max<uint32_t>(j.n_units, 1), hipMemcpyDeviceToHost, ds));
    HIP_TRY(hipMemcpyAsync(&h_tot[1], sp.dev.g_count, 8, hipMemcpyDeviceToHost, ds));
    HIP_TRY(hipMemcpyAsync(&h_tot[2], sp.dev.err, 4, hipMemcpyDeviceToHost, ds));
  }
  HIP_TRY(hipEventRecord(j.done, ds));
  if (g_rle_times) g_rle_t[8] += rle_now() - t_c0;
  // ---- the emulation's state behind this batch (what the next batch's plan starts from)
  if (sparse && j.n_units) {
    const bool whole_batch_one_open_unit = j.n_units == 1 && j.open_after;
    if (j.cont_tail && !whole_batch_one_open_unit) {  // unit 0 closes in this batch: its earlier reads go with the job
      j.tail_text.swap(sp.tail_text);
      j.tail_len.swap(sp.tail_len);
      sp.tail_text.clear();
      sp.tail_len.clear();
    }
    if (j.cont_carry && whole_batch_one_open_unit) {
      // (staged form, still open: stays in the carry buffers -- _finish writes them)
    } else if (j.open_after) {
      if (!(j.cont_tail && whole_batch_one_open_unit)) { sp.tail_text.clear(); sp.tail_len.clear(); }
      sparse_tail_append(sp, seqs, seq_off, seq_len, j.unit_first_read[j.n_units - 1], n_reads);
      sp.tail_open = true;
      sp.open = false;
    } else {
      sp.tail_open = false;
      sp.open = false;
      sp.tail_text.clear();
      sp.tail_len.clear();
    }
    sp.acc_nt = j.acc_after;
  }
  j.n_bytes = n_bytes;
  j.n_reads = n_reads;
  j.runs_cap = runs_cap;
  j.max_n = max_n;
  j.o = o;
  j.h_len = seq_len;
  j.h_calls = calls;
  j.h_hits = hits;
  j.h_roff = run_off;
  j.h_rcnt = run_cnt;
  j.settled = false;
  j.runs_in_ctx = false;
  j.busy = true;
  if (g_rle_times) g_rle_t[1] += rle_now() - t_in;
  return KU_OK;
}

// ---- step two: wait for the batch (one event), settle what the emulation has to settle for it
static int rle_job_finish(ku_ctx *ctx, RleJob &j, uint64_t *n_runs, bool *classified) {
  hipStream_t s = ctx->stream;
  *classified = false;
  ctx->last_runs_copied = 0;
  if (j.settled) {
    j.busy = false;
    *n_runs = ctx->n_runs = j.n_runs;
    *classified = true;
    return KU_OK;
  }
  const double t_w0 = g_rle_times ? rle_now() : 0.0;
  HIP_TRY(hipEventSynchronize(j.done));
  const double t_w1 = g_rle_times ? rle_now() : 0.0;
  if (g_rle_times && j.t_k0 && j.n_reads) {
    float ms = 0;
    float a = 0, b = 0;
    if (g_rle_ref && hipEventElapsedTime(&ms, j.t_k0, j.t_k1) == hipSuccess && hipEventElapsedTime(&a, g_rle_ref, j.t_k0) == hipSuccess &&
        hipEventElapsedTime(&b, g_rle_ref, j.t_k1) == hipSuccess) {
      g_rle_kernel_sum_ms += ms;
      g_rle_kernel_ms += std::max(0.0, (double)b - std::max((double)a, g_rle_cover_end));  // (the batches come in the order of their starts)
      g_rle_cover_end = std::max(g_rle_cover_end, (double)b);
      g_rle_reads += j.n_reads;
    } else (void)hipGetLastError();
  }
  struct Lap { double a, b; ~Lap() { if (g_rle_times) { g_rle_t[2] += b - a; g_rle_t[3] += rle_now() - b; g_rle_t[4] += 1; } } } lap_{t_w0, t_w1};
  j.busy = false;
  ku_ctx::Sparse &sp = ctx->sp;
  const unsigned long long *h_tot = (const unsigned long long *)j.pin.p;
  const uint8_t *h_flag = (const uint8_t *)j.pin.p + 64;
  const unsigned long long total = h_tot[0];
  if (j.h_hits) memset(j.h_hits, 0, j.n_reads * 4);  // "Q:n" is quick mode only
  ctx->last_calls_dev = j.calls.p;
  if (total > j.runs_cap) {
    // the run array was too small for this batch (reads that change taxon every few k-mers): the per-k-mer codes once more
    // without any accounting, through the array parallel to the reads and its own run-length encoder (the context's buffers)
    if (ctx->b_taxa.reserve((j.n_bytes + 16) * 4) || ctx->b_runs.reserve((j.n_bytes + 1) * 8) || ctx->b_roff.reserve(j.n_reads * 8) ||
        ctx->b_rcnt.reserve(j.n_reads * 4) || ctx->b_calls.reserve(j.n_reads * 4))
      return fail(KU_ENOMEM, "device batch buffers");
    uint64_t ws2 = 0;
    if (j.max_n > ku_short_max_kmers(ctx->m.db)) ws2 = ku_short_workspace_bytes(j.max_n, ctx->tax.n_slots, j.n_reads, ctx->n_cu);
    if (ctx->b_ws.reserve(ws2)) return fail(KU_ENOMEM, "device batch buffers");
    int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)j.seqs.p, j.n_bytes, (const uint64_t *)j.off.p,
                                      (const uint32_t *)j.len.p, j.n_reads, j.max_n, j.o.flags | KU_F_NO_COUNTS, (uint32_t *)ctx->b_calls.p,
                                      (uint32_t *)ctx->b_taxa.p, nullptr, ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "fused kernel launch failed");
    KU_TRY(rle_and_fetch(ctx, (const uint32_t *)ctx->b_taxa.p, (const uint64_t *)j.off.p, (const uint32_t *)j.len.p, j.n_reads, j.n_bytes + 1,
                         false, j.h_calls, nullptr, j.h_roff, j.h_rcnt, n_runs));
    j.runs_in_ctx = true;
  } else {
    *n_runs = ctx->n_runs = total;
    ctx->fetch_runs_src = j.runs.p;
    ctx->last_runs_copied = j.runs_copied;
  }
  *classified = true;  // what follows only concerns the emulation's state
  if (j.sparse && sp.on) {
    if ((uint32_t)h_tot[2]) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set is full");
    sp.g_count = std::max<uint64_t>(sp.g_count, h_tot[1]);
    // Units the counting could not settle.  A unit that is still open behind the batch waits (tail form: it is looked at
    // when it closes, with everything it got); a unit in the staged form (it came from a staged batch) is always tracked.
    std::vector<uint32_t> flagged;
    std::vector<uint8_t> flag_all;
    const bool carry_stays_open = j.cont_carry && j.n_units == 1 && j.open_after;
    // The flags are from when the batch's kernels ran -- with several batches in flight, before the exact pass of a batch AHEAD
    // of this one turned dense the very sketch that flags these units (the first units of a run: taxon 0's; without this, every
    // unit of the two batches behind went through the exact evaluation for nothing, 25-75 ms per 10 M reads).  Sketches only ever
    // turn dense, so flagging once more with the state as it is now can only take flags away; the counts are complete (the
    // batch's event), the state is at rest (exact passes end synchronised), and a stream of its own does not queue behind the
    // kernels of the batches in flight.
    bool any_flag = false;
    for (uint32_t u = 0; u < j.n_units; ++u) any_flag |= !(u + 1 == j.n_units && j.open_after) && h_flag[u];
    const double t_rf0 = g_rle_times ? rle_now() : 0.0;
    if (any_flag && !getenv("KU_NO_REFLAG")) {  // (test hook: the flags as the kernels left them)
      // (its own stream: d2h_stream holds the waits for the kernels and the copies back of the batches in flight)
      if (!ctx->fetch_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
      hipStream_t fs = ctx->fetch_stream;
      const size_t fb = ((size_t)j.n_units + 3) & ~(size_t)3;
      HIP_TRY(hipMemsetAsync(j.u_flag.p, 0, fb, fs));
      KU_TRY(ku_launch_sparse_flag_units((const uint32_t *)j.u_cnt.p, (uint64_t)j.n_units * ctx->tax.n_slots, ctx->tax.n_slots, sp.dev.dense,
                                         (uint8_t *)j.u_flag.p, fs));
      HIP_TRY(hipMemcpyAsync((uint8_t *)j.pin.p + 64, j.u_flag.p, j.n_units, hipMemcpyDeviceToHost, fs));
      HIP_TRY(hipStreamSynchronize(fs));
    }
    for (uint32_t u = 0; u < j.n_units; ++u) {
      const bool open = u + 1 == j.n_units && j.open_after;
      if (u == 0 && j.cont_carry) { flagged.push_back(u); flag_all.push_back(1); }
      else if (!open && h_flag[u]) { flagged.push_back(u); flag_all.push_back(0); }
    }
    const double t_ex0 = g_rle_times ? rle_now() : 0.0;
    if (g_rle_times) g_rle_x[0] += t_ex0 - t_rf0;
    if (!flagged.empty()) {
      // as on one stream, where they were queued in front of it: the kernels of the batches in flight come before the exact pass
      // (it closes work units -- sketches turn dense, entries join the run-wide set -- and those kernels read both)
      for (const RleJob &q : ctx->rle)
        if (q.busy && q.kernels_done) HIP_TRY(hipStreamWaitEvent(s, q.kernels_done, 0));
      KU_TRY(sparse_fast_exact(ctx, j, flagged, flag_all, carry_stays_open, s));
    }
    if (g_rle_times && !flagged.empty()) {
      g_rle_x[1] += rle_now() - t_ex0;
      g_rle_x[2] += 1;
      g_rle_x[3] += (double)flagged.size();
      for (uint32_t u : flagged) g_rle_x[4] += (double)(j.unit_first_read[u + 1] - j.unit_first_read[u]);
    }
  }
  return KU_OK;
}

int rle_drain_kernels(ku_ctx *ctx) {
  for (RleJob &q : ctx->rle)
    if (q.busy && q.kernels_done) HIP_TRY(hipEventSynchronize(q.kernels_done));
  return KU_OK;
}

// which of the two jobs takes the next batch / is the oldest in flight
static RleJob &rle_next_job(ku_ctx *ctx) {
  RleJob &j = ctx->rle[(ctx->rle_head + ctx->rle_in_flight) % KU_RLE_MAX_IN_FLIGHT];
  static const int counter_at[KU_RLE_MAX_IN_FLIGHT] = {26, 20, 22, 24};  // (dwords of the context's 32 scalars nobody else uses: dword 2 is
                                                                          // rle_and_fetch's counter, which an overflow redo launches on while batches are in flight)
  if (!j.d_counter) j.d_counter = (unsigned long long *)(ctx->d_scalar + counter_at[&j - &ctx->rle[0]]);
  return j;
}

static int rle_check_batch(const char *seqs, uint64_t n_bytes, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                           uint32_t *calls, uint64_t *run_off, uint32_t *run_cnt, ku_opts &o, bool &monotonic) {
  if ((n_bytes && !seqs) || (n_reads && (!seq_off || !seq_len || !calls || !run_off || !run_cnt)))
    return fail(KU_EINVAL, "ku_classify_batch_rle: null buffer");
  o.flags &= ~KU_F_KEEP_SLOTS;
  const double t_chk = g_rle_times ? rle_now() : 0.0;
  if (o.max_read_len == 0) for (uint64_t i = 0; i < n_reads; ++i) o.max_read_len = std::max(o.max_read_len, seq_len[i]);
  monotonic = true;
  for (uint64_t i = 0; i < n_reads; ++i) {
    if (seq_off[i] + seq_len[i] > n_bytes) return fail(KU_EINVAL, "read " + std::to_string(i) + " exceeds the sequence buffer");
    if (i && seq_off[i] < seq_off[i - 1] + seq_len[i - 1]) monotonic = false;
  }
  if (g_rle_times) g_rle_t[0] += rle_now() - t_chk;
  return KU_OK;
}

// the one-step paths (quick mode, several databases, sorted layout, shards, reads beyond 65535 k-mers, ...): through the
// context's own buffers, synchronously
static int rle_staged_batch(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                            const ku_opts &o, uint32_t *calls, uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs) {
  // a run needs at least one k-mer, so n_bytes bounds the number of runs: the device side cannot overflow
  const uint64_t runs_cap = n_bytes + 1;
  if (ctx->b_seqs.reserve(n_bytes + 16) || ctx->b_off.reserve(n_reads * 8) || ctx->b_len.reserve(n_reads * 4) ||
      ctx->b_calls.reserve(n_reads * 4) || ctx->b_taxa.reserve((n_bytes + 16) * 4) || ctx->b_hits.reserve(n_reads * 4) ||
      ctx->b_runs.reserve(runs_cap * 8) || ctx->b_roff.reserve(n_reads * 8) || ctx->b_rcnt.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->b_seqs.p, seqs, n_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_off.p, seq_off, n_reads * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_len.p, seq_len, n_reads * 4, hipMemcpyHostToDevice, s));
  KU_TRY(classify_device_impl(ctx, ctx->b_seqs.p, n_bytes, (const uint64_t *)ctx->b_off.p, (const uint32_t *)ctx->b_len.p,
                              n_reads, &o, (uint32_t *)ctx->b_calls.p, (uint32_t *)ctx->b_taxa.p,
                              (uint32_t *)ctx->b_hits.p, s, seq_off, seq_len));
  return rle_and_fetch(ctx, (const uint32_t *)ctx->b_taxa.p, (const uint64_t *)ctx->b_off.p, (const uint32_t *)ctx->b_len.p, n_reads,
                       runs_cap, (o.flags & KU_F_QUICK) != 0, calls, hits, run_off, run_cnt, n_runs);
}

// The buffers of `n_jobs` batches of up to n_bytes / n_reads ahead of the first batch (device memory, page-locked scratch,
// streams, events): what _enqueue would otherwise set up inside the caller's timing window, a few milliseconds per job.
extern "C" int ku_classify_batch_rle_reserve(ku_ctx *ctx, uint64_t n_bytes, uint64_t n_reads, uint32_t max_read_len, uint32_t n_jobs) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_classify_batch_rle_reserve"));
  const uint32_t k = ctx->m.db.k;
  const uint32_t max_n = max_read_len >= k ? max_read_len - k + 1 : 0;
  if (!ku_short_max_kmers(ctx->m.db)) return KU_OK;  // (the fused kernel does not apply: the one-step paths use the context's own buffers)
  if (!ctx->h2d_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
  if (!ctx->d2h_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
  if (!ctx->fetch_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
  const uint64_t waves = ku_short_grid_waves(n_reads, max_n, ctx->n_cu);
  const uint64_t runs_cap = n_bytes / 6 + 4 * n_reads + waves * rle_chunk(n_reads, waves) + 4096;
  uint64_t ws = 0;
  if (max_n > ku_short_max_kmers(ctx->m.db) && max_n <= ku_short_max_kmers_windowed(ctx->m.db)) ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, n_reads, ctx->n_cu);
  const bool sparse = ctx->sp.on && ctx->sp.unit_nt;
  const uint64_t n_units = sparse ? n_bytes / ctx->sp.unit_nt + 2 : 0;
  for (uint32_t q = 0; q < std::min<uint32_t>(n_jobs, KU_RLE_MAX_IN_FLIGHT); ++q) {
    RleJob &j = ctx->rle[q];
    if (j.seqs.reserve(n_bytes + 16) || j.off.reserve(n_reads * 8) || j.len.reserve(n_reads * 4) || j.calls.reserve(n_reads * 4) ||
        j.runs.reserve(runs_cap * 8) || j.roff.reserve(n_reads * 8) || j.rcnt.reserve(n_reads * 4) || j.ws.reserve(ws) ||
        j.pin.reserve(64 + (size_t)n_units + 64))
      return fail(KU_ENOMEM, "device batch buffers");
    if (sparse && (j.pin_unit.reserve(n_reads * 4) || j.unit.reserve(n_reads * 4) || j.u_cnt.reserve(std::max<uint64_t>(n_units * ctx->tax.n_slots, 1) * 4) ||
                   j.u_flag.reserve((n_units + 3) & ~3ull)))
      return fail(KU_ENOMEM, "device memory for the work-unit counters");
    if (!j.done) HIP_TRY(hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
    if (!j.kernels_done) HIP_TRY(hipEventCreateWithFlags(&j.kernels_done, hipEventDisableTiming));
    while (j.seg_events.size() < 2) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      j.seg_events.push_back(e);
    }
  }
  if (sparse && ctx->sp.tail_row.reserve((size_t)ctx->tax.n_slots * 4)) return fail(KU_ENOMEM, "device memory for the work-unit counters");
  // One read through the whole path, count-less (no state changes): the fused kernel's code object is loaded, the copy streams get
  // their queues, the events exist -- here rather than under the caller's first batch, whose enqueue took 22 ms of a 10 M-read
  // `classify` run's 0.19 s window (profiles/r05_e2e_sweep.log, the pipeline trace).  Failure is harmless.
  if (!getenv("KU_NO_WARMUP")) {
    // A dozen batches of the caller's size class through the whole path, count-less (no state changes), three in flight, from and
    // into page-locked memory like the executable's.  What a run's first batches otherwise pay for, as the pipeline trace of
    // `classify` showed it one by one (profiles/r05_e2e_sweep.log): the fused kernel's code object; the copy streams' queues (the
    // first copy of some hundred KB from the device into page-locked memory on a stream: 23.5 ms -- 4-byte copies go another
    // way); each job's events and scratch; scratch memory for the counting instances (they spill a few bytes per lane); and a
    // 13 ms ioctl of the runtime's under the first enqueue that follows a finished batch.  Failure is harmless.
    (void)ku_launch_warm_scratch(ctx->n_cu, ctx->stream);
    double keep_t[10], keep_x[6];  // (KU_RLE_TIMES: the warm-up's batches are none of the caller's)
    memcpy(keep_t, g_rle_t, sizeof keep_t);
    memcpy(keep_x, g_rle_x, sizeof keep_x);
    const double keep_ms = g_rle_kernel_ms, keep_sum = g_rle_kernel_sum_ms;
    const unsigned long long keep_reads = g_rle_reads;
    const uint64_t wn = (std::min<uint64_t>(std::max<uint64_t>(n_reads, 1), 65536) + 1) & ~1ull, stride = 101;  // (even: the arrays behind stay 8-byte aligned)
    const size_t per_slot = (size_t)wn * (4 + 4 + 4 + 8) + (size_t)wn * 8 * 8;
    PinBuf w;
    if (w.reserve((size_t)wn * (stride + 12) + KU_RLE_MAX_IN_FLIGHT * per_slot + 4096) == 0) {
      memset(w.p, 0, w.cap);
      char *text = (char *)w.p;
      uint64_t *w_off = (uint64_t *)(text + ((wn * stride + 63) & ~63ull));
      uint32_t *w_len = (uint32_t *)(w_off + wn);
      char *slots = (char *)(w_len + wn);
      for (uint64_t r = 0; r < wn; ++r) {
        char *t = text + r * stride;
        for (int i = 0; i < 100; ++i) t[i] = "ACGTTGCAAGCTTCGA"[(i * 7 + i / 16 + r) & 15];
        t[100] = '\n';
        w_off[r] = r * stride;
        w_len[r] = 100;
      }
      const ku_opts wo = {KU_F_NO_COUNTS, 1, 100, 0};
      uint64_t w_runs = 0;
      int flying = 0;
      for (int rep = 0; rep < 12; ++rep) {
        if (flying == 3) { (void)ku_classify_batch_rle_finish(ctx, &w_runs); --flying; }
        char *sl = slots + (size_t)(rep % KU_RLE_MAX_IN_FLIGHT) * per_slot;
        uint64_t *roff = (uint64_t *)sl;
        uint32_t *calls = (uint32_t *)(roff + wn), *hits = calls + wn, *rcnt = hits + wn;
        ku_run *runs = (ku_run *)(rcnt + wn);
        if (ku_classify_batch_rle_enqueue(ctx, text, wn * stride, w_off, w_len, wn, &wo, calls, hits, roff, rcnt, runs, wn * 8) == KU_OK) ++flying;
      }
      while (flying-- > 0) (void)ku_classify_batch_rle_finish(ctx, &w_runs);
      if (ctx->fetch_stream) {  // (ku_fetch_runs' stream)
        (void)hipMemcpyAsync(w.p, ctx->rle[0].seqs.p, std::min<size_t>(ctx->rle[0].seqs.cap, 1u << 20), hipMemcpyDeviceToHost, ctx->fetch_stream);
        (void)hipStreamSynchronize(ctx->fetch_stream);
      }
      (void)hipDeviceSynchronize();
      w.release();
    }
    (void)hipGetLastError();
    memcpy(g_rle_t, keep_t, sizeof keep_t);
    memcpy(g_rle_x, keep_x, sizeof keep_x);
    g_rle_kernel_ms = keep_ms;
    g_rle_kernel_sum_ms = keep_sum;
    g_rle_reads = keep_reads;
  }
  return KU_OK;
}

extern "C" int ku_classify_batch_rle_in_flight(const ku_ctx *ctx) { return ctx ? ctx->rle_in_flight : 0; }
extern "C" uint64_t ku_classify_batch_rle_copied(const ku_ctx *ctx) { return ctx ? ctx->last_runs_copied : 0; }

extern "C" int ku_classify_batch_rle_enqueue(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                             const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                             uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, ku_run *runs, uint64_t runs_cap) {
  KU_TRY(check_ready(ctx));
  if (ctx->rle_in_flight >= KU_RLE_MAX_IN_FLIGHT)
    return fail(KU_ESTATE, "ku_classify_batch_rle_enqueue: " + std::to_string(KU_RLE_MAX_IN_FLIGHT) + " batches are in flight (ku_classify_batch_rle_finish first)");
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  bool monotonic = true;
  KU_TRY(rle_check_batch(seqs, n_bytes, seq_off, seq_len, n_reads, calls, run_off, run_cnt, o, monotonic));
  RleJob &j = rle_next_job(ctx);
  const uint32_t max_n = o.max_read_len >= ctx->m.db.k ? o.max_read_len - ctx->m.db.k + 1 : 0;
  // a unit in the staged form (carry buffers) is settled batch by batch, synchronously: such a batch goes in one step, too
  const bool in_steps = n_reads && rle_fused_eligible(ctx, o.flags, max_n, n_bytes, n_reads, monotonic) &&
                        !(ctx->sp.on && !(o.flags & KU_F_NO_COUNTS) && ctx->sp.open);
  if (in_steps) {
    int st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
    if (st == KU_ENOMEM && ctx->sp.on && !(o.flags & KU_F_NO_COUNTS)) {
      // no room for the emulation's tables: the classification itself does not depend on them (see classify_device_impl):
      // the run goes on with the dense registers alone; nothing of this batch had been started
      (void)hipStreamSynchronize(ctx->stream);
      for (hipStream_t ks : ctx->k_streams) if (ks) (void)hipStreamSynchronize(ks);
      if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
      (void)hipGetLastError();
      ctx_free_sparse(ctx);
      ctx->sp.gave_up = true;
      st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
    }
    KU_TRY(st);
    ++ctx->rle_in_flight;
    return KU_OK;
  }
  if (ctx->rle_in_flight) return fail(KU_ESTATE, "ku_classify_batch_rle_enqueue: this batch takes a path that cannot overlap with the batch in flight "
                                                 "(ku_classify_batch_rle_finish first, then enqueue it again)");
  // classified here and now; _finish hands the totals over
  uint64_t nr = 0;
  ctx->n_runs = 0;
  if (n_reads) {
    if (rle_fused_eligible(ctx, o.flags, max_n, n_bytes, n_reads, monotonic)) {  // (fused, but a unit in the staged form is open)
      int st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
      bool classified = false;
      if (st == KU_OK) st = rle_job_finish(ctx, j, &nr, &classified);
      j.busy = false;
      if (st == KU_ENOMEM && ctx->sp.on && !(o.flags & KU_F_NO_COUNTS)) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
        ctx_free_sparse(ctx);
        ctx->sp.gave_up = true;
        if (!classified) {
          st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
          if (st == KU_OK) st = rle_job_finish(ctx, j, &nr, &classified);
          j.busy = false;
        } else st = KU_OK;
      }
      KU_TRY(st);
    } else {
      KU_TRY(rle_staged_batch(ctx, seqs, n_bytes, seq_off, seq_len, n_reads, o, calls, hits, run_off, run_cnt, &nr));
      j.runs_in_ctx = true;
    }
  }
  j.settled = true;
  j.busy = true;
  j.n_runs = nr;
  j.runs_copied = 0;
  ++ctx->rle_in_flight;
  return KU_OK;
}

extern "C" int ku_classify_batch_rle_finish(ku_ctx *ctx, uint64_t *n_runs) {
  if (!ctx || !n_runs) return fail(KU_EINVAL, "ku_classify_batch_rle_finish: null argument");
  *n_runs = 0;
  if (!ctx->rle_in_flight) return fail(KU_ESTATE, "ku_classify_batch_rle_finish: no batch is in flight");
  KU_TRY(ctx_activate(ctx));
  RleJob &j = ctx->rle[ctx->rle_head];
  ctx->rle_head = (ctx->rle_head + 1) % KU_RLE_MAX_IN_FLIGHT;
  --ctx->rle_in_flight;
  bool classified = false;
  int st = rle_job_finish(ctx, j, n_runs, &classified);
  j.busy = false;
  if (st == KU_ENOMEM && classified && ctx->sp.on) {
    // the emulation ran out of room behind the classification: it is given up, the run goes on (ku_ctx_sparse_state says 2)
    (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t ks : ctx->k_streams) if (ks) (void)hipStreamSynchronize(ks);
    if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
    (void)hipGetLastError();
    ctx_free_sparse(ctx);
    ctx->sp.gave_up = true;
    st = KU_OK;
  }
  return st;
}

extern "C" int ku_classify_batch_rle(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                     const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                     uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs) {
  if (!n_runs) return fail(KU_EINVAL, "ku_classify_batch_rle: null buffer");
  *n_runs = 0;
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_classify_batch_rle"));
  KU_TRY(ku_classify_batch_rle_enqueue(ctx, seqs, n_bytes, seq_off, seq_len, n_reads, opts, calls, hits, run_off, run_cnt, nullptr, 0));
  return ku_classify_batch_rle_finish(ctx, n_runs);
}

extern "C" int ku_ctx_replace_calls(ku_ctx *ctx, const uint32_t *new_calls, uint64_t n_reads, uint64_t *n_dropped) {
  KU_TRY(check_ready(ctx));
  if (n_dropped) *n_dropped = 0;
  if (n_reads == 0) return KU_OK;
  if (!new_calls) return fail(KU_EINVAL, "ku_ctx_replace_calls: null argument");
  KU_TRY(rle_idle(ctx, "ku_ctx_replace_calls"));
  if (!ctx->last_calls_dev) return fail(KU_ESTATE, "ku_ctx_replace_calls: the context holds no batch");
  if (ctx->b_hits.reserve(n_reads * 4) != KU_OK) return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  unsigned long long *d_dropped = (unsigned long long *)(ctx->d_scalar + 16);
  HIP_TRY(hipMemsetAsync(d_dropped, 0, 8, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_hits.p, new_calls, n_reads * 4, hipMemcpyHostToDevice, s));
  KU_TRY(ku_launch_replace_calls((const uint32_t *)ctx->last_calls_dev, (const uint32_t *)ctx->b_hits.p, n_reads, ctx->d_node_taxid, ctx->tax.n_nodes,
                                 ctx->cnt.n_reads, d_dropped, s));
  HIP_TRY(hipMemcpyAsync((void *)ctx->last_calls_dev, ctx->b_hits.p, n_reads * 4, hipMemcpyDeviceToDevice, s));  // a second replacement starts from these
  unsigned long long dropped = 0;
  HIP_TRY(hipMemcpyAsync(&dropped, d_dropped, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n_dropped) *n_dropped = dropped;
  return KU_OK;
}

extern "C" int ku_fetch_runs(ku_ctx *ctx, ku_run *runs, uint64_t n_runs) {
  if (!ctx) return fail(KU_EINVAL, "ku_fetch_runs: null context");
  if (n_runs > ctx->n_runs) return fail(KU_EINVAL, "ku_fetch_runs: the last batch holds " + std::to_string(ctx->n_runs) + " runs");
  if (n_runs == 0) return KU_OK;
  if (!runs) return fail(KU_EINVAL, "ku_fetch_runs: null buffer");
  if (!ctx->fetch_runs_src) return fail(KU_ESTATE, "ku_fetch_runs: no batch was classified");
  // (a stream of its own: the copy queues neither behind the kernels of the batch in flight nor behind its copies back,
  // which wait for those kernels)
  KU_TRY(ctx_activate(ctx));
  if (!ctx->fetch_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
  HIP_TRY(hipMemcpyAsync(runs, ctx->fetch_runs_src, n_runs * 8, hipMemcpyDeviceToHost, ctx->fetch_stream));
  HIP_TRY(hipStreamSynchronize(ctx->fetch_stream));
  return KU_OK;
}
