// ku_ctx.h -- what the translation units of the C ABI share (ku_api*.cpp): the objects behind the opaque handles of
// include/krakenuniq_amd.h, the error helpers, and the few functions one area calls in another.
//   ku_api.cpp           errors, ku_db / ku_tax, the context: create / destroy, database load + probe table, taxonomy, reset
//   ku_api_sparse.cpp    HyperLogLog++ sparse-mode emulation: state, staged passes, export, the group's view
//   ku_api_classify.cpp  lookup / resolve / classify entry points on device and host buffers, owner routing
//   ku_api_rle.cpp       the host-batch call with run-length encoded output, in one step and in two (batches in flight)
//   ku_api_ooc.cpp       out-of-core runs: chunk swap / prefetch, device-resident batches, merging contexts
//   ku_api_report.cpp    counts export, the report from the resident state
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "ku_host.h"
#include "ku_internal.h"

void ku_set_error(const std::string &s);
static inline int fail(int code, const std::string &msg) {
  ku_set_error(msg);
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? KU_ENOMEM : KU_EHIP,                             \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                          \
  } while (0)
#define KU_TRY(expr)            \
  do {                          \
    int s_ = (expr);            \
    if (s_ != KU_OK) return s_; \
  } while (0)

struct ku_db {
  const uint8_t *pairs = nullptr;
  const uint64_t *offsets = nullptr;
  ku_db_info info{};
  void *map_kdb = nullptr, *map_idx = nullptr;
  size_t map_kdb_sz = 0, map_idx_sz = 0;
  // ku_db_values: the distinct values, scanned once (callers ask for the count first and the list second)
  mutable std::mutex values_mu;
  mutable std::vector<uint32_t> values;
  mutable bool values_ready = false;
};

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return KU_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {
      if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return KU_ENOMEM; }
      want = bytes;
    }
    cap = want;
    return KU_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// page-locked host scratch that grows on demand (sources and targets of asynchronous copies must outlive the call that
// enqueues them and must be page-locked for the copy to be asynchronous at all)
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return KU_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return KU_ENOMEM; }
    cap = want;
    return KU_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// One batch on its way through the fused kernel with run-length encoded output (ku_classify_batch_rle_enqueue / _finish,
// round 5): its own device buffers and page-locked scratch -- KU_RLE_MAX_IN_FLIGHT of them take turns, so that the uploads of
// the next batches and the copies back of the previous ones run under the kernel of batch b and the host waits for ONE event
// per batch -- and what
// _finish needs to know about the batch.
struct RleJob {
  DevBuf seqs, off, len, calls, runs, roff, rcnt, ws, unit, u_cnt, u_flag;
  PinBuf pin;       // [0] extent of the run array, [1] entries of the run-wide set, [2] the emulation's error word; byte 64 on: unit flags
  PinBuf pin_unit;  // work-unit number of every read (source of an asynchronous upload)
  unsigned long long *d_counter = nullptr;  // the kernel's bump counter of the run array (2 dwords of the context's scalars)
  hipEvent_t kernels_done = nullptr, done = nullptr;
  hipEvent_t t_k0 = nullptr, t_k1 = nullptr;  // KU_RLE_TIMES: around the batch's kernels on the stream they run on
  std::vector<hipEvent_t> seg_events;
  bool busy = false;
  bool settled = false;         // classified by a one-step path inside _enqueue: _finish only hands the totals over
  bool runs_in_ctx = false;     // the runs lie in the context's own run buffer (one-step paths, the overflow redo)
  uint64_t runs_copied = 0;     // entries of the run array already copied to the caller's buffer (0: ku_fetch_runs does it)
  uint64_t n_runs = 0;
  // the batch
  uint64_t n_bytes = 0, n_reads = 0, runs_cap = 0;
  uint32_t max_n = 0;
  ku_opts o{};
  const uint32_t *h_len = nullptr;
  uint32_t *h_calls = nullptr, *h_hits = nullptr, *h_rcnt = nullptr;
  uint64_t *h_roff = nullptr;
  // sparse-sketch emulation, fast path: the batch's work units
  bool sparse = false;
  bool cont_carry = false;      // unit 0 continues a unit whose state sits in the carry buffers (L / U entries; the staged form)
  bool cont_tail = false;       // unit 0 continues a unit kept as its reads + insert counts (the fast path's own form)
  bool open_after = false;      // the last unit is still open behind this batch
  uint32_t n_units = 0;
  uint64_t kmers = 0;           // upper bound of what the kernel may add to the run-wide set
  uint64_t acc_after = 0;
  std::vector<uint64_t> unit_first_read;
  std::vector<char> tail_text;      // cont_tail: the reads of the open unit BEFORE this batch (bases, each read followed by '\n')
  std::vector<uint32_t> tail_len;
  void release() {
    for (DevBuf *b : {&seqs, &off, &len, &calls, &runs, &roff, &rcnt, &ws, &unit, &u_cnt, &u_flag}) b->release();
    pin.release();
    pin_unit.release();
    if (kernels_done) (void)hipEventDestroy(kernels_done);
    if (done) (void)hipEventDestroy(done);
    if (t_k0) (void)hipEventDestroy(t_k0);
    if (t_k1) (void)hipEventDestroy(t_k1);
    for (hipEvent_t e : seg_events) (void)hipEventDestroy(e);
    kernels_done = done = t_k0 = t_k1 = nullptr;
    seg_events.clear();
  }
};

// one resident database (shard): the 12-byte pairs until the taxonomy is set, the probe table afterwards
struct DbStore {
  bool db_owned = false, offsets_owned = false;
  bool hash_layout = true;
  bool seen_dirty = false;  // SEEN marks of the probe table may be set (ku_device.h; the sparse-sketch emulation's fast path)
  void *d_table = nullptr;
  uint64_t n_dup = 0;
  uint64_t table_lines = 0;
  uint32_t *d_pairs = nullptr;
  uint64_t *d_offsets = nullptr;
  KuDbDev db{};
  std::vector<uint32_t> values;  // ascending distinct non-zero raw taxids of the shard
  // count_taxons (krakendb.cpp:90-113) per slot, taken while the 12-byte pairs are still there (store_finalize): one sequential
  // pass at load time instead of a scan of the whole probe table -- 27 ms of the first report over a database (round 6)
  std::vector<unsigned long long> slot_counts;
};

struct ku_ctx {
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  bool db_loaded = false, tax_set = false;
  bool hash_layout = true;   // KU_LAYOUT=sorted keeps the on-disk order + binary search (A/B and fallback for HBM-tight shards)
  double load_factor = 0.2;  // keys per bucket slot (8 slots per 128-byte line); KU_LOAD_FACTOR fixes it
  bool load_factor_set = false;
  DbStore m;                   // the (first) database: the only one that may be a strict minimizer-range shard
  std::vector<DbStore> extra;  // further whole databases of a hierarchical run, searched in order after `m`
  // taxonomy tables
  std::vector<uint32_t> h_node_taxid, h_slot_taxid;
  uint32_t *d_node_parent = nullptr, *d_node_slot = nullptr, *d_node_taxid = nullptr, *d_slot_node = nullptr,
           *d_slot_taxid = nullptr, *d_slot_anc_off = nullptr, *d_slot_anc = nullptr;
  KuTaxDev tax{};
  // run state
  KuCountsDev cnt{};
  // scratch for the host-buffer entry point
  DevBuf b_seqs, b_off, b_len, b_calls, b_taxa, b_hits, b_ws, b_runs, b_roff, b_rcnt;
  // ku_classify_batch_rle through the fused kernel: the batch goes up in segments on a stream of its own while the
  // segments before are classified (one event per segment)
  hipStream_t h2d_stream = nullptr, d2h_stream = nullptr, fetch_stream = nullptr;
  // Round 6: the kernels of consecutive batches in flight run on TWO streams in turn, so that the tail of one batch's launch --
  // its last waves, their counter flushes -- lies under the start of the next one's: launches of ~120 k reads then cost what the
  // bench's 10 M-read launch costs per read (scripts/launch_shape_probe.py: 30.3 -> 19.5 ms per 10 M reads; 19.8 in one launch).
  // What orders the batches: main_ev (work queued on the context's own stream before the batch), prev_kernels_done (the batch
  // enqueued before: the open work unit's insert counts travel from batch to batch behind the kernels; the exact pass of a batch
  // waits for the kernels of every batch in flight), and the host, which waits for a batch's event before it settles it.
  hipStream_t k_streams[2] = {nullptr, nullptr};
  hipEvent_t main_ev = nullptr;
  hipEvent_t prev_kernels_done = nullptr;  // (a job's event, not owned)
  std::vector<hipEvent_t> seg_events;
  uint32_t *d_scalar = nullptr;
  // ku_classify_batch_rle in two steps: up to two batches in flight (FIFO: rle_head is the oldest)
  RleJob rle[KU_RLE_MAX_IN_FLIGHT];
  int rle_head = 0, rle_in_flight = 0;
  const void *fetch_runs_src = nullptr;  // where the runs of the batch finished last lie (ku_fetch_runs)
  const void *last_calls_dev = nullptr;  // ... and its calls on the device (ku_ctx_replace_calls)
  uint64_t last_runs_copied = 0;         // ... and how many of its runs are in the caller's buffer already
  // ku_ctx_count_taxons of the store it was computed for (identified by its buffers)
  std::vector<unsigned long long> count_cache;
  const void *count_cache_store = nullptr, *count_cache_pairs = nullptr;
  uint64_t count_cache_lines = 0;
  uint64_t n_runs = 0;  // runs of the last ku_classify_batch_rle, still in b_runs
  // exact distinct counting (classifyExact): one global set of canonical k-mers + first-insertion counters per slot
  unsigned long long *d_exact_set = nullptr, *d_exact_unique = nullptr;
  uint64_t exact_mask = 0;
  // out-of-core runs: the NEXT chunk, uploaded and laid out on its own stream while the resident one is searched
  struct Prefetch {
    bool valid = false;
    const ku_db *db = nullptr;
    uint64_t bin_lo = 0, bin_hi = 0;
    DbStore store;
    hipStream_t stream = nullptr;
    uint32_t *d_scalar = nullptr;
  } pf;
  // HyperLogLog++ sparse-mode emulation (ku_sparse.hip)
  struct Sparse {
    bool on = false;
    uint64_t unit_nt = 500000;  // Work_unit_size (classify.cpp:38); 0 = the whole run is one unit (-x mode)
    uint64_t acc_nt = 0;        // nt of the unit that is still open
    bool open = false;          // ... whose encodings and statistics sit in the carry buffers
    KuSparseDev dev{};
    unsigned long long *d_counters = nullptr;  // [0] size of G, [1..2] carry sizes, [3] export size
    DevBuf unit, carry_l, carry_u, out;
    DevBuf u_cnt, u_flag, list;  // fast path: inserts per (unit, slot), per-unit flags, the reads of the flagged units
    uint64_t n_carry_l = 0, n_carry_u = 0, cap_carry_l = 0, cap_carry_u = 0;
    uint64_t g_count = 0;       // entries of the global set after the last pass (host copy of d_counters[0])
    bool gave_up = false;       // the emulation ran out of device memory during the run and was switched off
    // The open unit in TAIL form (round 5; the fast path's own): a host copy of its reads so far and its insert counts per
    // slot.  A unit can only turn a sketch dense when it gave it >= 1025 inserts (hyperloglogplus.cpp:496-498) -- known from
    // the counts once the unit closes, whichever batches it straddled; only then, and only for such a unit, does the exact
    // evaluation (L / U tables) run, over these reads + the closing batch's.  Rounds 3-4 ran it for the first and the last
    // unit of EVERY batch to carry their L / U entries along: two passes, ten launches, two host round trips per batch.
    // (acc_nt, open, tail_open and the tail describe the state behind the newest ENQUEUED batch.)
    bool tail_open = false;
    std::vector<char> tail_text;   // bases of the unit's reads, each read followed by '\n'
    std::vector<uint32_t> tail_len;
    DevBuf tail_row;               // inserts of the open unit so far, per slot
    DevBuf t_seqs, t_off, t_len, t_taxa, t_unit;  // the tail on the device, when it is evaluated
  } sp;
};


struct ku_batch {
  ku_ctx *ctx = nullptr;
  uint64_t n_bytes = 0, n_reads = 0;
  uint32_t max_len = 0;
  bool finished = false;  // ku_batch_finish translated the slots to taxids in place: no further passes
  void *d_seqs = nullptr;
  uint64_t *d_off = nullptr;
  uint32_t *d_len = nullptr, *d_taxa = nullptr;
  std::vector<uint64_t> h_off;  // host copies for the sparse-mode emulation's work-unit plan
  std::vector<uint32_t> h_len;
};

// ---- functions one area calls in another (defined in the file named behind them)
int ctx_activate(ku_ctx *ctx);  // ku_api.cpp
void store_free(DbStore &d);  // ku_api.cpp
void ctx_drop_count_cache(ku_ctx *ctx);  // ku_api.cpp
void ctx_free_sparse(ku_ctx *ctx);  // ku_api.cpp
int store_upload(ku_ctx *ctx, DbStore &d, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi, bool scan_values = true, hipStream_t stream = nullptr);  // ku_api.cpp
bool store_whole(const DbStore &d);  // ku_api.cpp
int store_finalize(ku_ctx *ctx, DbStore &d, hipStream_t stream = nullptr, uint32_t *d_scalar = nullptr);  // ku_api.cpp
int sparse_reserve_global(ku_ctx *ctx, uint64_t incoming, hipStream_t s);  // ku_api_sparse.cpp
int sparse_pass_tables(ku_ctx *ctx, uint64_t n_entries, KuSparseDev *view, hipStream_t s);  // ku_api_sparse.cpp
int sparse_pass(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off, const uint32_t *h_len, uint64_t n_reads, uint64_t n_bytes, const uint32_t *d_taxa, uint32_t quick_min_hits, hipStream_t s);  // ku_api_sparse.cpp
int sparse_close_open_unit(ku_ctx *ctx);  // ku_api_sparse.cpp
int ctx_seen_harvest(ku_ctx *ctx);  // ku_api_sparse.cpp
int check_ready(ku_ctx *ctx);  // ku_api_classify.cpp
int classify_device_impl(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off, const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls, uint32_t *d_taxa, uint32_t *d_hits, void *stream, const uint64_t *h_off, const uint32_t *h_len);  // ku_api_classify.cpp
int rle_and_fetch(ku_ctx *ctx, const uint32_t *d_taxa, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads, uint64_t runs_cap, bool quick, uint32_t *calls, uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs);  // ku_api_classify.cpp
int sparse_tail_to_carry(ku_ctx *ctx);  // ku_api_rle.cpp
int sparse_tail_close(ku_ctx *ctx);  // ku_api_rle.cpp
void rle_times_print();  // ku_api_rle.cpp
int rle_idle(const ku_ctx *ctx, const char *who);  // ku_api_rle.cpp
int rle_drain_kernels(ku_ctx *ctx);  // ku_api_rle.cpp
