// ku_internal.h -- structures shared by the C-ABI host code (ku_api.cpp) and the
// gfx950 kernels (ku_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/krakenuniq_amd.h"

// Device view of the resident DB shard.  Pairs keep the reference's on-disk
// 12-byte AoS record (8-byte LE key | 4-byte value, krakendb.cpp:176) so key and
// value of a probe arrive in the same cache line; after ku_ctx_set_taxonomy the
// value dword holds a *slot id* (rank of the taxid among the distinct DB values,
// 0 = taxid 0) instead of the raw taxid.
struct KuDbDev {
  const uint32_t *pairs;    // sorted layout: 3 dwords per pair: key_lo, key_hi, slot (nullptr once the table is built)
  const uint4 *table;       // hash layout: n_lines buckets of 128 B (16-byte tag header + 8 x 12-byte entries)
  uint64_t n_lines;         // 128-byte lines in the table
  const uint64_t *offsets;  // bin_hi - bin_lo + 1 global pair indices
  uint64_t pair_base;       // global index of pairs[0]
  uint64_t n_pairs;
  uint64_t bin_lo, bin_hi;  // owned minimizer range [bin_lo, bin_hi)
  uint32_t k, nt;
  uint32_t xor_mask;        // INDEX2_XOR_MASK & (4^nt - 1), 0 for KRAKIDX
  uint32_t pad;
};

// Dense taxonomy tables (all uint32, device memory).
//   node  = rank of a taxid in sorted(taxDB ids U DB values U {0}); node 0 = taxid 0
//   slot  = rank of a taxid in sorted(DB values U {0});             slot 0 = taxid 0
struct KuTaxDev {
  const uint32_t *node_parent;  // Parent_map in node space (0 = none)
  const uint32_t *node_slot;    // slot of the node's taxid, 0 if it is not a DB value
  const uint32_t *node_taxid;
  const uint32_t *slot_node;
  const uint32_t *slot_taxid;
  // root path of every slot, reduced to the nodes that are database values (only those can carry hit counts):
  // slot_anc[slot_anc_off[s] .. slot_anc_off[s + 1]) = s itself, then its ancestors' slots in walk order.  resolve_tree's
  // score (krakenutil.cpp:157-177) becomes a scan of one short list instead of a chain of dependent parent lookups.
  const uint32_t *slot_anc_off;
  const uint32_t *slot_anc;
  uint32_t n_nodes, n_slots;    // n_slots includes slot 0
  uint32_t node_one;            // node of taxid 1 (0xFFFFFFFF if absent)
  uint32_t pad;
};

// Per-taxon run state (device).
struct KuCountsDev {
  uint8_t *registers;                 // n_slots * 4096, dense HLL p = 12
  unsigned long long *n_kmers;        // n_slots
  unsigned long long *n_reads;        // n_nodes
};

// Owner routing of the sharded multi-GPU path (ku_mgpu.cpp, ku_route.hip).  A rank scans only its own slice of
// the reads (ku_lookup_kernel<3,...>) and sends every maximal run of consecutive unambiguous k-mers that share their
// ANCHOR occurrence (= minimizer occurrence, ku_device.h) to the rank that owns the run's minimizer bin -- as ONE 16-byte
// record ("super-k-mer": the run's k + n - 1 bases, 2 bits each, + n + the anchor's offset): ~2 B per k-mer on the wire where
// round 3 sent 12 B {k-mer, bucket prehash} + kept 4 B of position per k-mer.  The owner expands the records, probes its
// table, books the k-mers (HLL, n_kmers -- owner-computes) and returns one 4-byte slot per k-mer, in record order; the
// sender finds a k-mer's slot through the TICKET the scan left at the k-mer's position of the per-k-mer array:
//     record      d0..d2 = bases 0..47 (first base in bits 31..30 of d0), d3 = bases 48..55 << 16 | anchor offset << 8 | n
//                 (n = 0: padding; n <= KU_ROUTE_MAXN(k) so that k + n - 1 <= 56 bases; anchor offset = base index of
//                 the anchor m-mer within the record, read order)
//     ticket      record index in the sender's queue buffer << 5 | index of the k-mer within the record;
//                 KU_AMBIG: ambiguous k-mer; KU_ROUTE_MISS: nobody owns the k-mer's bin (slot 0, not booked)
//     slot of a ticket = returned[kb[record] + index], kb = exclusive prefix sum of n over the queue buffer
//                 (ku_launch_route_prefix: the owner numbers what it received the same way)
#define KU_ROUTE_CHUNK 64u           // unit of a queue: a block claims KuRouteDev::chunk records at a time, a multiple of this (the owner works on groups of 64)
#define KU_ROUTE_CURSOR_STRIDE 16u   // the queues' cursors lie 128 bytes apart
#define KU_ROUTE_NONE (~0ull)
#define KU_ROUTE_MISS 0xFFFFFFFEu
#define KU_ROUTE_MAX_RECORDS ((1u << 27) - 2u)  // per rank and step (27-bit record index in a ticket)
__host__ __device__ static inline uint32_t ku_route_maxn(uint32_t k, uint32_t nt) {  // k-mers per record
  const uint32_t w = k - nt + 1, room = 57u - k;
  const uint32_t a = w < room ? w : room;
  return a < 31u ? a : 31u;
}
struct KuRouteDev {
  const uint64_t *own_lo, *own_hi;  // [world] the ranks' minimizer ranges
  unsigned long long *cursor;       // [world * KU_ROUTE_CURSOR_STRIDE] records claimed in each owner's queue, whole chunks (ends as
                                    // the owner's total incl. padding, also beyond cap)
  uint4 *q_rec;                     // the queues: owner o's records at [o * cap, o * cap + cursor[o])
  uint64_t cap;                     // room per queue in records (a multiple of chunk)
  uint32_t world;
  uint32_t chunk;                   // records a block claims at a time (few owners: larger claims, fewer adds on the same cursor)
};
// HyperLogLog++ sparse-mode emulation (ku_sparse.hip): tables of one context
struct KuSparseDev {
  unsigned long long *l_key;  // (unit + 1) << 50 | slot << 32 | encoding; 0 = empty
  uint32_t *l_first;          // smallest position the encoding was seen at
  uint64_t l_mask;
  unsigned long long *u_key;  // (unit + 1) << 32 | slot; 0 = empty
  uint32_t *u_distinct, *u_last, *u_maxfirst;
  uint64_t u_mask;
  unsigned long long *g_key;  // (slot + 1) << 32 | encoding; 0 = empty
  uint64_t g_mask;
  uint32_t *dense;            // per slot
  uint32_t *err;              // bit 0: L full, bit 1: U full, bit 2: G full, bit 3: a run code that is no slot's taxid
  unsigned long long *g_count;
};
#define KU_SPARSE_MAX_UNITS 16000u  // per batch (14-bit unit field)
#define KU_SPARSE_MAX_SLOTS (1u << 18)
int ku_launch_sparse_insert(const KuSparseDev &s, uint32_t k, const uint8_t *d_seqs, const uint64_t *d_seq_off,
                            const uint32_t *d_seq_len, const uint32_t *d_unit, uint64_t n_reads, const uint32_t *d_taxa,
                            uint32_t quick_min_hits, int n_cu, hipStream_t stream);
int ku_launch_sparse_clear(const KuSparseDev &s, hipStream_t stream);
int ku_launch_zero3(void *a, uint64_t a_dwords, void *b, uint64_t b_dwords, void *c, uint64_t c_dwords, hipStream_t stream);
// dst[i] += src[i], i < n (the open work unit's insert counts joining a batch's first row)
int ku_launch_add_u32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t stream);
int ku_launch_sparse_close(const KuSparseDev &s, uint32_t n_closed, hipStream_t stream, bool skip_hits = false);
int ku_launch_sparse_carry_out(const KuSparseDev &s, uint32_t unit, unsigned long long *d_carry_l, uint32_t *d_carry_u,
                               unsigned long long *d_counters, uint64_t cap_l, uint64_t cap_u, hipStream_t stream);
int ku_launch_sparse_carry_in(const KuSparseDev &s, const unsigned long long *d_carry_l, uint64_t n_l, const uint32_t *d_carry_u,
                              uint64_t n_u, hipStream_t stream);
int ku_launch_sparse_flag_units(const uint32_t *d_u_cnt, uint64_t n_cells, uint32_t n_slots, const uint32_t *d_dense, uint8_t *d_unit_flag,
                                hipStream_t stream);
int ku_launch_sparse_insert_runs(const KuSparseDev &s, uint32_t k, const uint8_t *d_seqs, const uint64_t *d_seq_off, const uint32_t *d_seq_len,
                                 const uint32_t *d_list_read, const uint32_t *d_list_unit, const uint32_t *d_list_urow, uint64_t n_list,
                                 const void *d_runs, const uint64_t *d_run_off, const uint32_t *d_run_cnt, const uint32_t *d_slot_taxid,
                                 uint32_t n_slots, const uint32_t *d_u_cnt, int n_cu, hipStream_t stream, uint32_t pos_base = 0);
// s.g_key / s.g_mask: the new (zeroed) table; s.g_count zeroed by the caller
int ku_launch_sparse_rehash(const KuSparseDev &s, const unsigned long long *d_old_keys, uint64_t old_cells, hipStream_t stream);
int ku_launch_sparse_export(const KuSparseDev &s, unsigned long long *d_out, uint64_t cap, unsigned long long *d_counter,
                            hipStream_t stream);
// the SEEN marks of a probe table (ku_device.h): what = 0 count (entries of slots that are not dense), 1 insert them into the
// run-wide set, 2 clear all marks
int ku_launch_warm_scratch(int n_cu, hipStream_t stream);  // (a scratch-using kernel over every wave slot: see ku_sparse.hip)
int ku_launch_seen(int what, void *d_table, uint64_t n_lines, const KuSparseDev &s, unsigned long long *d_count, hipStream_t stream);

// clade roll-up of the report (ku_report.hip): histograms of KU_ROLLUP_BINS bins per clade
#define KU_ROLLUP_BINS 80
int ku_launch_rollup_dense(const uint8_t *d_registers, const uint32_t *d_member_off, const uint32_t *d_member_slot,
                           const uint8_t *d_clade_dense, uint32_t n_clades, uint32_t *d_hist, hipStream_t stream);
#define KU_ROLLUP_HOT 48  // clades whose histogram is pre-aggregated in LDS
// the union plan of one ku_ctx_report call (device pointers; ku_api.cpp builds it, ku_report.hip walks it)
struct KuRollupPlan {
  const uint32_t *dense;                // per slot
  const uint32_t *slot_fast;            // per slot: the bitmap its entries go to first, or KU_FAST_SKIP / KU_FAST_WALK (ku_report.hip)
  const uint32_t *slot_off, *slot_clade;  // CSR: the all-sparse clades on each slot's root path, leaf first
  const unsigned long long *set_off;    // per clade: its open-addressing table of 4-byte cells within `set` ...
  const uint32_t *set_cells;            // ... and its size (0: nothing is offered to the clade)
  const uint16_t *clade_hot;            // per clade: row in the block's LDS histogram (0xFFFF: none)
  const uint32_t *hot_clades;           // the reverse map
  uint32_t n_hot, pad;
  uint32_t *set, *hist, *err;
  const uint32_t *bm_of;                // per clade: its bitmap (KU_BM_NONE: a table clade)
  uint32_t *bm;
};
int ku_launch_rollup_sparse(const unsigned long long *d_g_key, uint64_t g_cells, const KuRollupPlan &plan, int n_cu, hipStream_t stream);
int ku_launch_rollup_table(const void *d_table, uint64_t n_lines, const KuRollupPlan &plan, int n_cu, hipStream_t stream);
// union bitmaps of the big all-sparse clades (ku_report.hip): one bit per 25-bit index, KU_BM_WORDS words per clade
#define KU_BM_WORDS (1u << 20)
#define KU_BM_NONE 0xFFFFFFFFu
#define KU_FAST_SKIP 0xFFFFFFFFu                 // KuRollupPlan::slot_fast: nothing of this slot is offered to anybody
#define KU_FAST_WALK 0xFFFFFFFEu                 //                          the slot's first clade keeps a table: the general walk
int ku_launch_bitmap_or_children(uint32_t *d_bm, const uint32_t *d_parents, uint32_t n_parents, const uint32_t *d_child_off,
                                 const uint32_t *d_child, hipStream_t stream);
int ku_launch_bitmap_hist(const uint32_t *d_bm, const uint32_t *d_bm_clade, uint32_t n_bm, uint32_t *d_hist, hipStream_t stream);
int ku_launch_replace_calls(const uint32_t *d_old, const uint32_t *d_new, uint64_t n, const uint32_t *d_node_taxid, uint32_t n_nodes,
                            unsigned long long *d_n_reads, unsigned long long *d_dropped, hipStream_t stream);

// host-side view of an opened database for the other translation units (ku_api.cpp owns the struct)
struct ku_db;
int ku_db_raw(const ku_db *db, const uint8_t **pairs, const uint64_t **offsets);

// the sparse-mode emulation over several GPUs (ku_api.cpp; used by ku_mgpu.cpp)
struct ku_ctx;
int ku_ctx_sparse_on(const ku_ctx *ctx);
uint64_t ku_ctx_sparse_unit_nt(const ku_ctx *ctx);
int ku_ctx_sparse_pass_slots(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off,
                             const uint32_t *h_len, uint64_t n_reads, uint64_t n_bytes, const uint32_t *d_taxa, uint32_t quick_min_hits,
                             hipStream_t s);
int ku_ctx_sparse_move_open_unit(ku_ctx *src, ku_ctx *dst);
int ku_ctx_sparse_finish(ku_ctx *ctx, uint32_t *h_dense);
int ku_ctx_sparse_set_dense(ku_ctx *ctx, const uint32_t *h_dense);
int ku_ctx_sparse_absorb(ku_ctx *dst, ku_ctx *src);
int ku_launch_sparse_absorb(const KuSparseDev &s, const unsigned long long *d_keys, uint64_t n, hipStream_t stream);

int ku_ctx_route_info(const ku_ctx *ctx, uint64_t *bin_lo, uint64_t *bin_hi, int *is_hash, int *single_db);
int ku_ctx_route_scan(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, uint32_t *d_taxa, const KuRouteDev &rt, hipStream_t s);
// gather + resolve in one kernel (reads of up to 65535 k-mers, no quick mode); prepare: KU_EUNSUP = take
// ku_launch_route_gather + ku_resolve_device
int ku_ctx_route_resolve_prepare(ku_ctx *ctx, const ku_opts *opts, hipStream_t s);
int ku_ctx_route_resolve(ku_ctx *ctx, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                         uint32_t *d_taxa, uint32_t *d_hits, const uint32_t *d_kb, const uint32_t *d_ret, hipStream_t s);
int ku_ctx_route_owner(ku_ctx *ctx, const void *d_rec, uint64_t n_rec, const uint32_t *d_kb, uint32_t *d_slots, bool do_counts, hipStream_t s);
// context internals the multi-GPU driver (ku_mgpu.cpp) needs
struct ku_ctx;
hipStream_t ku_ctx_stream_of(ku_ctx *ctx);
unsigned long long *ku_ctx_exact_unique_of(ku_ctx *ctx);  // first-insertion counters per slot (nullptr: exact counting is off)
int ku_ctx_device_of(const ku_ctx *ctx);
int ku_ctx_cus_of(const ku_ctx *ctx);
uint32_t ku_ctx_k_of(const ku_ctx *ctx);
void ku_set_error(const std::string &s);

// launch wrappers implemented in ku_kernels.hip (all asynchronous on `stream`)
int ku_launch_lookup(const KuDbDev &db, const KuCountsDev &cnt, const uint8_t *d_seqs, uint64_t n_bytes,
                     uint32_t *d_taxa, bool do_counts, bool prior, bool merge_chunk, int n_cu, hipStream_t stream);
// owner routing (ku_route.hip; the scan is ku_lookup_kernel<3, 1, true, false> in ku_kernels.hip)
int ku_launch_route_scan(const KuDbDev &db, const uint8_t *d_seqs, uint64_t n_bytes, uint32_t *d_taxa, const KuRouteDev &rt, int n_cu,
                         hipStream_t stream);
unsigned ku_route_scan_grid(uint64_t n_bytes, int n_cu);  // blocks of that launch (each may leave one padded chunk per queue)
// kb[i] = sum of n over the records before i, kb[n_rec] = the total.  cap != 0: the buffer is `n_rec / cap` queues of `cap`
// records of which only the first min(cursor[q * KU_ROUTE_CURSOR_STRIDE], cap) count (the sender's side); cap == 0: every
// record counts (what an owner received).  d_tot (optional, cap != 0): per queue {records = cursor, k-mers}, then cap.
// d_work: ku_route_prefix_work_bytes(n_rec) bytes of scratch.
uint64_t ku_route_prefix_work_bytes(uint64_t n_rec);
int ku_launch_route_prefix(const void *d_rec, uint64_t n_rec, uint64_t cap, const unsigned long long *d_cursor, uint32_t *d_kb,
                           unsigned long long *d_tot, void *d_work, hipStream_t stream);
// the owner's side: n_rec (a multiple of 64) received records -> d_slots[kb[i] + j] = slot of k-mer j of record i (+ HLL, n_kmers)
int ku_launch_route_owner(const KuDbDev &db, const KuCountsDev &cnt, const void *d_rec, uint64_t n_rec, const uint32_t *d_kb,
                          uint32_t *d_slots, bool do_counts, int n_cu, hipStream_t stream);
// the sender's side: tickets in d_taxa[0 .. n) become slots (KU_AMBIG stays)
int ku_launch_route_gather(uint32_t *d_taxa, uint64_t n, const uint32_t *d_kb, const uint32_t *d_ret, hipStream_t stream);
int ku_launch_lookup_stats(const KuDbDev &db, const uint8_t *d_seqs, uint64_t n_bytes, unsigned long long *d_stats,
                           int n_cu, hipStream_t stream);
int ku_launch_resolve(const KuDbDev &db, const KuTaxDev &tax, const KuCountsDev &cnt, const uint8_t *d_seqs,
                      const uint64_t *d_seq_off, const uint32_t *d_seq_len, uint64_t n_reads, uint32_t flags,
                      uint32_t min_hits, uint32_t max_read_len, uint32_t *d_calls, uint32_t *d_taxa,
                      uint32_t *d_hits, void *d_workspace, uint64_t workspace_bytes, int n_cu,
                      hipStream_t stream);
// Run-length encoded output of the fused kernel (instead of the per-k-mer array): {code, start} pairs as ku_run, every
// read's runs contiguous in `runs`; a wave claims `chunk` entries at a time from the bump counter (one atomic per
// ~chunk / runs-per-read reads) and fills them read by read, so the array has unused gaps -- (run_off, run_cnt) say
// where each read's runs are.  counter > cap afterwards = the array was too small (nothing was written out of bounds).
struct KuRunsOut {
  uint2 *runs;
  unsigned long long *counter;
  unsigned long long cap;
  uint64_t *run_off;
  uint32_t *run_cnt;
  uint32_t chunk;
  // 0: every chunk is claimed from the counter.  Else the launch's wave w OWNS chunk (pre_base1 - 1 + w) from the start and
  // only claims from the counter -- which the host then starts at (chunks owned by all launches) * chunk -- when that one is
  // full.  (Round 5: the batches of the `classify` executable are ~60 k reads = 5 reads per wave; every wave of a launch
  // claimed its first chunk at the same moment, 12 k adds to one address = ~150 of the launch's 290 microseconds.)
  uint32_t pre_base1;
};
// Sparse-mode emulation inside the fused kernel (fast path): every unambiguous k-mer of a slot whose
// sketch is not known to be dense goes into the run-wide set G straight away, and the number of inserts per (work unit,
// slot) is counted in a dense array -- a local sketch can only switch to the dense representation in a unit that gave it
// at least 1025 inserts, so the exact per-unit evaluation (ku_sparse.hip) runs on those units and slots only.
struct KuSparseFast {
  unsigned long long *g_key;
  uint64_t g_mask;
  unsigned long long *g_count;   // entries of G (one add per wave at the end of the kernel)
  const uint32_t *dense;         // per slot, sticky
  uint32_t *u_cnt;               // [unit - unit_base][n_slots] inserts of the (unit, slot) pair
  const uint32_t *unit_of;       // per read (indexed like seq_off)
  uint32_t *err;
  uint32_t n_slots;
  uint32_t unit_base;
};
// owner-routed path: where the fused kernel's ROUTE instances find the slot of a ticket (ku_route.hip)
struct KuRouteIn {
  const uint32_t *kb;   // k-mers before each record of this rank's queue buffer
  const uint32_t *ret;  // the slots the owners returned, in that numbering
};
#define KU_SPARSE_SWITCH_INSERTS 1025u  // fewest inserts with which a sparse sketch can turn dense (hyperloglogplus.cpp:496-498)
// fused wave-per-read path for short reads (ku_short.hip)
uint32_t ku_short_max_kmers(const KuDbDev &db);           // reads taken in one pass
uint32_t ku_short_max_kmers_windowed(const KuDbDev &db);  // ... in windows of 128 k-mers (ku_short.hip)
uint64_t ku_short_workspace_bytes(uint32_t max_kmers, uint32_t n_slots, uint64_t n_reads, int n_cu);
int ku_launch_classify_short(const KuDbDev &db, const KuTaxDev &tax, const KuCountsDev &cnt, const uint8_t *d_seqs,
                             uint64_t n_bytes, const uint64_t *d_seq_off, const uint32_t *d_seq_len, uint64_t n_reads,
                             uint32_t max_kmers, uint32_t flags, uint32_t *d_calls, uint32_t *d_taxa, uint32_t *d_hits,
                             void *d_workspace, uint64_t workspace_bytes, int n_cu, hipStream_t stream,
                             const KuRunsOut *runs_out = nullptr, const KuSparseFast *sparse = nullptr);
// resolve stage of the owner-routed path fused with the gather of the returned slots (ku_short.hip, ROUTE instances)
uint32_t ku_route_resolve_max_kmers();
int ku_launch_route_resolve(const KuDbDev &db, const KuTaxDev &tax, const KuCountsDev &cnt, const uint64_t *d_seq_off,
                            const uint32_t *d_seq_len, uint64_t n_reads, uint32_t max_kmers, uint32_t flags, uint32_t *d_calls,
                            uint32_t *d_taxa, uint32_t *d_hits, const uint32_t *d_kb, const uint32_t *d_ret, void *d_workspace,
                            uint64_t workspace_bytes, int n_cu, hipStream_t stream, const KuRunsOut *runs_out = nullptr);
// waves of the fused kernel's persistent grid for a batch of n_reads (sizing of the run array: every wave may leave one
// partly used chunk behind)
uint64_t ku_short_grid_waves(uint64_t n_reads, uint32_t max_kmers, int n_cu);
uint64_t ku_resolve_workspace_bytes(uint32_t max_read_len, uint32_t k, int n_cu);
int ku_launch_exact(uint32_t k, const uint8_t *d_seqs, const uint64_t *d_seq_off, const uint32_t *d_seq_len, uint64_t n_reads,
                    const uint32_t *d_taxa, unsigned long long *d_set, uint64_t mask, unsigned long long *d_unique,
                    uint32_t *d_overflow, int n_cu, hipStream_t stream, uint32_t quick_min_hits = 0);
int ku_launch_rle(const uint32_t *d_taxa, uint32_t k, const uint64_t *d_seq_off, const uint32_t *d_seq_len,
                  uint64_t n_reads, uint64_t n_bytes, void *d_runs, uint64_t runs_cap, unsigned long long *d_counter,
                  uint64_t *d_run_off, uint32_t *d_run_cnt, int n_cu, hipStream_t stream);
int ku_launch_max_len(const uint32_t *d_seq_len, uint64_t n_reads, uint32_t *d_out, hipStream_t stream);
int ku_launch_quick_chunked(const KuTaxDev &tax, const KuCountsDev &cnt, uint32_t k, const uint64_t *d_seq_off,
                            const uint32_t *d_seq_len, uint64_t n_reads, uint32_t flags, uint32_t min_hits, uint32_t *d_calls,
                            uint32_t *d_taxa, uint32_t *d_hits, int n_cu, hipStream_t stream);
// element-wise merges of the multi-GPU driver's same-process exchange: dst = max(dst, src) / dst += src
int ku_launch_merge_max_u32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t stream);
int ku_launch_merge_max_u8(uint8_t *dst, const uint8_t *src, uint64_t n, hipStream_t stream);
int ku_launch_replace_u32(uint32_t *p, uint64_t n, uint32_t from, uint32_t to, hipStream_t stream);
// per-k-mer array preset of a sharded exact-counting pass: positions another rank owns keep it (never a slot id: 0xFE bytes)
#define KU_FOREIGN_MARK 0xFEFEFEFEu
// classifyExact on a shard (ku_mgpu.cpp): lookup of the owned k-mers + their insertion into this rank's k-mer set
int ku_exact_owned_step(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads, uint64_t n_bytes,
                        const ku_opts *opts, uint32_t *d_taxa, hipStream_t s);
int ku_launch_merge_add_u64(unsigned long long *dst, const unsigned long long *src, uint64_t n, hipStream_t stream);
// DB preparation
int ku_launch_repack(const uint8_t *d_raw, uint64_t n_pairs, uint32_t key_len, uint32_t *d_pairs,
                     hipStream_t stream);
int ku_launch_mark_values(const uint32_t *d_pairs, uint64_t n_pairs, uint32_t *d_bitmap, hipStream_t stream);
int ku_launch_collect_values(const uint32_t *d_bitmap, uint32_t *d_out, uint32_t cap, uint32_t *d_count,
                             hipStream_t stream);
int ku_launch_remap_values(uint32_t *d_pairs, uint64_t n_pairs, const uint32_t *d_slot_taxid, uint32_t n_slots,
                           uint32_t *d_err, hipStream_t stream);
int ku_launch_build_table(const uint32_t *d_pairs, uint64_t n_pairs, void *d_table, uint64_t n_lines, uint32_t k,
                          uint32_t m, uint32_t xor_mask, unsigned long long *d_spilled, hipStream_t stream);
int ku_launch_count_table(const void *d_table, uint64_t n_lines, unsigned long long *d_counts, hipStream_t stream);
int ku_launch_count_slots(const uint32_t *d_pairs, uint64_t n_pairs, unsigned long long *d_counts,
                          uint32_t n_slots, hipStream_t stream);
