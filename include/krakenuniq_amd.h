/* krakenuniq_amd.h -- C ABI of the MI355X-native KrakenUniq classify hot path.
 *
 * The reference (fbreitwieser/krakenuniq v1.0.4) has no library API: the
 * boundary is the `classify` executable (src/classify.cpp) that
 * scripts/krakenuniq:248 spawns.  This header is the thin C layer between that
 * executable's host code (this repo's krakenuniq_amd/csrc/classify_{main,input,device,output}.cpp,
 * flag-compatible with src/classify.cpp:1074) and the HIP kernels.  Every entry
 * point names the reference code it replaces (file:line under the reference's
 * src/).  Conventions: extern "C", opaque handles, plain pointers + sizes, int
 * status (0 = KU_OK, negative = KU_E*), no exceptions / exit() inside the
 * library, calls on one ku_ctx are serialised by the caller.  There is NO CPU
 * fallback: every compute entry point fails with KU_EHIP when no gfx950 device
 * is usable.
 */
#ifndef KRAKENUNIQ_AMD_H
#define KRAKENUNIQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KU_ABI_VERSION 1

/* status codes (the CLI maps them to the reference's sysexits, classify.cpp:1085-1131) */
#define KU_OK 0
#define KU_EINVAL (-1)   /* bad argument / flag           -> EX_USAGE   64 */
#define KU_EDATA (-2)    /* malformed kdb/idx/taxDB       -> EX_DATAERR 65 */
#define KU_ENOINPUT (-3) /* cannot open a file            -> EX_NOINPUT 66 */
#define KU_ENOMEM (-4)   /* host or device allocation     -> EX_OSERR   71 */
#define KU_EHIP (-5)     /* HIP runtime / no usable GPU   -> EX_SOFTWARE 70 */
#define KU_ESTATE (-6)   /* call order violated           -> EX_SOFTWARE 70 */
#define KU_EUNSUP (-7)   /* valid in the reference, not built here */

/* Per-k-mer code for an ambiguous k-mer (classify.cpp:920-923 "ambig_list=1"). */
#define KU_AMBIG 0xFFFFFFFFu
/* HLL precision: the reference always runs p = 12 (hyperloglogplus.hpp:87; -p is a no-op). */
#define KU_HLL_P 12
#define KU_HLL_M 4096

const char *ku_strerror(int status);
/* thread-local detail for the last failing call on this thread ("" if none) */
const char *ku_last_error(void);
int ku_abi_version(void);
/* number of usable gfx950 devices (0 when none; never fails) */
int ku_device_count(void);

/* ------------------------------------------------------------------ database
 * Host view of database.kdb + database.idx.
 * Replaces KrakenDB::KrakenDB / KrakenDBIndex::KrakenDBIndex + QuickFile mmap
 * (krakendb.cpp:60-78,534-544; quickfile.cpp:44-78). */
typedef struct ku_db ku_db;
typedef struct ku_db_info {
  uint32_t k;          /* key_bits / 2                       (krakendb.cpp:75) */
  uint32_t nt;         /* minimizer length                   (krakendb.cpp:543) */
  uint32_t idx_type;   /* 1 = KRAKIDX, 2 = KRAKIX2 scrambled (krakendb.cpp:536-541) */
  uint32_t key_len;    /* bytes per key on disk              (krakendb.cpp:76) */
  uint64_t key_ct;     /* pairs                              (krakendb.cpp:72) */
  uint64_t n_bins;     /* 4^nt */
} ku_db_info;

int ku_db_open(const char *kdb_path, const char *idx_path, ku_db **out);
/* Wrap caller-owned host memory laid out as on disk (pairs: key_ct * (key_len+4)
 * bytes; offsets: 4^nt + 1 pair indices). */
int ku_db_wrap(const void *pairs, uint64_t key_ct, uint32_t k, const uint64_t *offsets, uint32_t nt,
               uint32_t idx_type, ku_db **out);
void ku_db_close(ku_db *db);
int ku_db_get_info(const ku_db *db, ku_db_info *out);
/* Split the bin space into n_shards contiguous minimizer ranges of balanced
 * bytes (8 per bin + pair_size per pair) -- the multi-GPU analogue of
 * KrakenDB::prepare_chunking / upper_bound (krakendb.cpp:430-522).
 * bin_bounds[n_shards + 1], bin_bounds[0] = 0, bin_bounds[n_shards] = 4^nt. */
int ku_db_shard_plan(const ku_db *db, uint32_t n_shards, uint64_t *bin_bounds);
/* Chunk plan for a byte budget, exactly as prepare_chunking computes it
 * (krakendb.cpp:463-522): returns the number of chunks and fills up to
 * cap + 1 bounds.  Chunks with zero pairs are skipped like the reference, so the
 * last bound may be < 4^nt: the bins behind it hold no pairs (a caller that
 * shards by these bounds extends the last chunk to 4^nt). */
int ku_db_chunk_plan(const ku_db *db, uint64_t max_bytes, uint64_t *bin_bounds, uint32_t cap,
                     uint32_t *n_chunks);

/* ------------------------------------------------------------------ database construction (offline)
 * db_sort on the GPU (src/db_sort.cpp:34-128 + KrakenDB::make_index, src/krakendb.cpp:118-148): reads a
 * Jellyfish-format k-mer list (JFLISTDN header, unsorted key/value records), orders the records by (minimizer bin
 * key, k-mer) and writes database.kdb (header copied verbatim, sorted records) and database.idx ("KRAKIX2", nt,
 * 4^nt + 1 offsets) -- byte for byte what the reference writes.  zero_vals = db_sort -z.  nt in [1, 15].  The whole
 * list has to fit the device (about 45 bytes of HBM per record at the peak). */
int ku_db_sort_files(int device, const char *in_path, const char *out_kdb_path, const char *out_idx_path, uint32_t nt,
                     int zero_vals);

/* ------------------------------------------------------------------ taxonomy
 * Host taxonomy: taxDB text -> entries + Parent_map
 * (taxdb.hpp:563-605 readTaxonomyIndex_, :411-433 createPointers, :383-398 getParentMap). */
typedef struct ku_tax ku_tax;
int ku_tax_open(const char *taxdb_path, ku_tax **out);
/* ids/parents as in the taxDB file (self-parent or unknown parent = no parent) */
int ku_tax_from_arrays(const uint32_t *ids, const uint32_t *parents, uint64_t n, ku_tax **out);
void ku_tax_close(ku_tax *tax);
uint64_t ku_tax_size(const ku_tax *tax);
/* the entries' taxids in the order of the object's rows (file order, duplicates dropped, "unclassified" = 0 added behind
 * them when the file has none): the layout ku_report_rows expects.  ids[ku_tax_size(tax)] */
int ku_tax_ids(const ku_tax *tax, uint32_t *ids);
/* Parent_map value, 0 = none/root; returns KU_AMBIG when the taxid has no entry */
uint32_t ku_tax_parent(const ku_tax *tax, uint32_t taxid);

/* ------------------------------------------------------------------ device context
 * One per GPU.  Holds the resident DB shard (12-byte pairs + offsets slice), the
 * dense taxonomy tables, and the run's per-taxon state (HLL registers, n_kmers,
 * n_reads) -- the device twin of the global `taxon_counts` map (classify.cpp:78). */
typedef struct ku_ctx ku_ctx;
int ku_ctx_create(int device, ku_ctx **out);
void ku_ctx_destroy(ku_ctx *ctx);

/* Upload the shard [bin_lo, bin_hi) of a host DB.  Device twin of
 * QuickFile::load_file (-M, quickfile.cpp:80-117) / KrakenDB::load_chunk (-x,
 * krakendb.cpp:411-425). */
int ku_ctx_load_db(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi);
/* Adopt a shard that is already in HBM (bench / on-device DB construction):
 * d_pairs = n_pairs 12-byte (8-byte LE key, 4-byte LE taxid) records of the
 * bins [bin_lo, bin_hi); d_offsets = bin_hi - bin_lo + 1 uint64 *global* pair
 * indices whose first entry is the global index of d_pairs[0].  The buffers
 * stay owned by the caller; d_pairs is remapped in place (taxid -> slot id) and
 * consumed by ku_ctx_set_taxonomy (it builds the probe table from it) and may be
 * released afterwards unless ku_ctx_db_layout reports the sorted layout;
 * d_offsets must outlive the context. */
int ku_ctx_adopt_db(ku_ctx *ctx, void *d_pairs, uint64_t n_pairs, const uint64_t *d_offsets, uint32_t k,
                    uint32_t nt, uint32_t idx_type, uint64_t bin_lo, uint64_t bin_hi);
/* Hierarchical multi-database classification ("classify -d A.kdb -i A.idx -d B.kdb -i B.idx", classify.cpp:163-177,
 * 928-936): every k-mer is searched in the databases in the order they were given and the first one that holds it
 * supplies the taxon; the k-mer is then accounted once under that taxon (or under 0).  ku_ctx_add_db appends a
 * whole database behind the one loaded with ku_ctx_load_db / ku_ctx_adopt_db; call it before ku_ctx_set_taxonomy.
 * All databases must share k (classify.cpp:199-208; KU_EINVAL), minimizer lengths and index types may differ.
 * Needs the first database resident as a whole (no minimizer-range shard: KU_EUNSUP); at most 8 databases. */
int ku_ctx_add_db(ku_ctx *ctx, const ku_db *db);
/* Which in-HBM layout the shard ended up in after ku_ctx_set_taxonomy: *is_hash = 1 for the bucketised probe
 * table (default), 0 for the sorted on-disk order + binary search (KU_LAYOUT=sorted, or the automatic fallback
 * when the table does not fit: an adopted d_pairs buffer then stays in use).  *resident_bytes = table or pairs
 * + offsets. */
int ku_ctx_db_layout(ku_ctx *ctx, uint32_t *is_hash, uint64_t *resident_bytes);
/* Distinct non-zero taxids stored in the resident shard, ascending (what
 * KrakenDB::count_taxons enumerates, krakendb.cpp:90-113).  Call with out = NULL
 * to get *n. */
int ku_ctx_db_values(ku_ctx *ctx, uint32_t *out, uint64_t *n);
/* Install the taxonomy and freeze the slot table.  all_values = ascending
 * distinct taxids over ALL shards of the database (NULL = this shard only; in
 * the sharded multi-GPU run the caller all-gathers ku_ctx_db_values first so
 * every rank numbers slots identically).  Remaps the resident values to slot
 * ids and allocates/zeroes the per-taxon state. */
int ku_ctx_set_taxonomy(ku_ctx *ctx, const ku_tax *tax, const uint32_t *all_values, uint64_t n_values);
/* Per-taxid pair counts of the resident shard = database.kdb.counts content
 * (KrakenDB::count_taxons, krakendb.cpp:90-113; classify.cpp:275-283), ascending
 * taxid, includes value 0 if present.  out arrays sized via n (NULL to query). */
int ku_ctx_count_taxons(ku_ctx *ctx, uint32_t *taxids, uint64_t *counts, uint64_t *n);
/* Same for the db_index-th database of a hierarchical run (0 = the first one); one .counts file per database. */
int ku_ctx_count_taxons_db(ku_ctx *ctx, uint32_t db_index, uint32_t *taxids, uint64_t *counts, uint64_t *n);
/* Exact distinct k-mer counting next to the sketches (classifyExact): one device-wide set of canonical k-mers with
 * 2^capacity_log2 cells (8 bytes each; keep the number of distinct k-mers of the run below ~70 % of that), filled by
 * ku_classify_batch / _rle / _device (plain mode only: KU_EUNSUP with quick mode, slot output, count-less runs, the
 * two-stage device API and resident batches).  Call after ku_ctx_set_taxonomy; ku_ctx_reset_counts empties the set.
 * ku_counts_export_exact returns the distinct count per slot (same order as ku_counts_export's slot_taxid), or
 * KU_ENOMEM when the set overflowed. */
int ku_ctx_enable_exact(ku_ctx *ctx, uint32_t capacity_log2);
int ku_counts_export_exact(ku_ctx *ctx, uint64_t *unique_kmers);
/* zero HLL registers / n_kmers / n_reads (start of a run) */
int ku_ctx_reset_counts(ku_ctx *ctx);
/* HyperLogLog++ sparse-mode emulation.  The reference's sketch starts as a set of 32-bit encoded
 * hashes at precision p' = 25 (hyperloglogplus.cpp:181-204) and turns dense when an insert finds 1024 entries
 * (:496-498); classify keeps one local sketch per taxon and work unit (`work_unit_nt` nt of reads, classify.cpp:487-564)
 * and merges it into the global one (sparse + sparse = set union of any size, :601-604).  So a taxon's global sketch
 * is dense iff one of its per-unit sketches switched, else it holds every distinct encoding of the run and the report
 * prints a near-exact count.  With the emulation enabled the batches of ku_classify_batch / _rle / ku_batch_finish (in
 * input order; the two-stage / fused device paths are bypassed) track exactly that next to the dense registers;
 * ku_sparse_export closes the last unit and returns slot_is_sparse[n_slots] and the (slot << 32 | encoded hash)
 * pairs of the sparse slots (pairs = NULL to query *n_pairs) for ku_report_sparse.  work_unit_nt = 0: the whole run
 * is one unit (what the reference's -x chunk mode amounts to: it inserts into the global sketches directly,
 * classify.cpp:719).  global_log2: initial cells of the run-wide (slot, encoding) set, 0 = 2^26; it moves to a larger
 * table whenever a pass could fill it beyond 1/2 (KU_ENOMEM when the device has no room for that).
 * Call after ku_ctx_set_taxonomy; single GPU; at most 2^18 database taxids. */
int ku_ctx_enable_sparse(ku_ctx *ctx, uint64_t work_unit_nt, uint32_t global_log2);
/* frees the emulation's tables: later batches only keep the dense registers (ku_ctx_sparse_state = 0).  Enabling it
 * again -- also on a context where it is still on -- starts afresh, with the per-taxon state of the run reset. */
int ku_ctx_disable_sparse(ku_ctx *ctx);
/* the work unit that is still open ends here: call between input files (the reference's units do not span files,
 * classify.cpp:487-564 runs once per file); no-op without the emulation or with work_unit_nt = 0 */
int ku_sparse_close_unit(ku_ctx *ctx);
/* 0 = emulation off, 1 = on, 2 = it was on and gave up: a batch found no device memory for its tables (they hold every
 * distinct k-mer of the taxa that stay sparse).  The batch and the run went on; ku_ctx_report and the exported state
 * are those of a run without the emulation (dense-register estimates), ku_sparse_export returns KU_ESTATE. */
int ku_ctx_sparse_state(const ku_ctx *ctx);
int ku_sparse_export(ku_ctx *ctx, uint8_t *slot_is_sparse, uint64_t *pairs, uint64_t *n_pairs);

/* ---- set_lcas on the GPU (src/set_lcas.cpp:429-476, the database build step after db_sort): every k-mer of a library
 * sequence that the database holds gets  value = lca(Parent_map, taxid of the sequence, value)  (krakenutil.cpp:90-118).
 * ku_setlcas_open uploads the database in its on-disk order and the taxonomy; ku_setlcas_add folds one sequence
 * (ASCII, any case; k-mers with other letters are skipped) and blocks until done; ku_setlcas_finish returns the
 * value of every pair in database order (and how many sequence k-mers the database did not hold: an error for
 * set_lcas without -x).  KU_SL_RESET = -R (the sequences' k-mers are set to 0 instead), KU_SL_FORCE_CONTAMINANT = -T
 * (a k-mer of a 'synthetic construct' / 'artificial sequences' sequence, taxids 32630 / 81077, keeps that taxid: the
 * first such sequence in call order wins, values that already are one of the two stay). */
#define KU_SL_RESET 0x1u
#define KU_SL_FORCE_CONTAMINANT 0x2u
/* -I, UID databases (src/set_lcas.cpp:451-455, src/uid_mapping.cpp:32-91): the value of a k-mer becomes the id of the SET
 * of taxids whose sequences hold it; a set gets the next UID when it first comes up, so the numbering follows the order of
 * the calls and of the k-mers within a sequence (the reference on one thread).  The GPU finds each k-mer's pair, the host
 * walks them in order.  ku_setlcas_finish then returns UIDs; ku_setlcas_uid_map the map file's content: per UID, in
 * creation order, {taxid added, UID of the set it was added to (0: none)} -- what classify -I reads (ku_uid_map_open). */
#define KU_SL_UIDS 0x4u
typedef struct ku_setlcas ku_setlcas;
int ku_setlcas_open(int device, const ku_db *db, const ku_tax *tax, uint32_t flags, ku_setlcas **out);
int ku_setlcas_add(ku_setlcas *s, const char *seq, uint64_t len, uint32_t taxid);
int ku_setlcas_finish(ku_setlcas *s, uint32_t *values_out, uint64_t *n_missing);
int ku_setlcas_uid_map(const ku_setlcas *s, const uint32_t **blocks /* 2 * n_uids words, owned by s */, uint64_t *n_uids);
void ku_setlcas_close(ku_setlcas *s);

/* ------------------------------------------------------------------ classification */
#define KU_F_QUICK 0x1u        /* -q : stop at min_hits hits (classify.cpp:943-944,962-963) */
#define KU_F_NO_COUNTS 0x2u    /* do not touch HLL / n_kmers / n_reads (pure lookup) */
#define KU_F_KEEP_SLOTS 0x4u   /* leave taxa[] as internal slot ids (multi-GPU reduce stage) */
#define KU_F_MERGE_CHUNK 0x8u  /* ku_lookup_device: positions whose bin this shard does not own keep their value
                                  (pass over one chunk of an out-of-core run; "non-zero wins", classify.cpp:445-452) */

typedef struct ku_opts {
  uint32_t flags;
  uint32_t min_hits;     /* -m, used with KU_F_QUICK (classify.cpp:101) */
  uint32_t max_read_len; /* upper bound of seq_len[] in the batch; 0 = let the library find it */
  uint32_t reserved;
} ku_opts;

/* Read batch layout (host or device): `seqs` is a byte buffer of n_bytes in which
 * read i occupies seqs[seq_off[i] .. seq_off[i]+seq_len[i]) and is followed by
 * at least one byte that is not one of ACGTacgt (raw FASTA/FASTQ text with its
 * line terminators qualifies).  Any byte outside ACGTacgt makes the k-mers that
 * cover it ambiguous (krakenutil.cpp:252-274); '\n'/'\r' *inside* a read are
 * not skipped: a '\r' closing a read (CRLF files) behaves as in the reference
 * (one more, ambiguous, k-mer), the CLI reader removes the ones inside
 * multi-line CRLF FASTA sequences before the batch is built.
 * Per-k-mer output `taxa` is parallel to `seqs`: taxa[seq_off[i] + j] is the
 * taxid (0 = miss, KU_AMBIG = ambiguous) of the k-mer starting at base j of read
 * i, j < seq_len[i]-k+1; other entries are unspecified.  `calls[i]` is the
 * read's taxid (0 = unclassified); `hits[i]` (optional, may be NULL) is the
 * quick-mode hit counter printed as "Q:n". */

/* Whole hot path for one batch on host buffers (H2D + kernels + D2H, blocking):
 * classify_sequence for every read (classify.cpp:897-968) + the merge into the
 * global per-taxon counts (classify.cpp:541-544). */
int ku_classify_batch(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                      const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                      uint32_t *taxa, uint32_t *hits);

/* As ku_classify_batch, but the per-k-mer codes come back run-length encoded -- exactly the information
 * hitlist_string prints (classify.cpp:826-861) at ~30 B instead of 4 B per base for a typical short read, which is
 * what the PCIe link and the output formatter want.  Two steps so that the caller can size the buffer exactly:
 * ku_classify_batch_rle classifies, encodes on the device and returns calls/hits, (run_off, run_cnt) per read and
 * the batch's total *n_runs; ku_fetch_runs then copies that many runs out of the context (valid until the next
 * batch call on it).  Read i owns runs[run_off[i] .. run_off[i] + run_cnt[i]); a run is {code, start}: code =
 * taxid / 0 / KU_AMBIG, start = index of its first k-mer; its length is the next run's start (or the read's k-mer
 * count) minus its own.  Reads shorter than k own no runs.  *n_runs is the extent of the context's run array, not the
 * number of runs: the fused kernel's waves claim the array in chunks, so there may be unused entries between the reads'
 * runs (sum of run_cnt <= *n_runs). */
typedef struct ku_run {
  uint32_t code;
  uint32_t start;
} ku_run;
int ku_classify_batch_rle(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                          const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                          uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs);
int ku_fetch_runs(ku_ctx *ctx, ku_run *runs, uint64_t n_runs);
#define KU_RLE_MAX_IN_FLIGHT 4
/* The same in two steps, up to KU_RLE_MAX_IN_FLIGHT batches in flight on a context (round 5; the reference's process_file keeps its team
 * busy by handing work units out under a critical section, classify.cpp:499-561 -- here the device is kept busy by having
 * the next batch's upload and the previous batch's copies back run under the current batch's kernels):
 *   ku_classify_batch_rle_enqueue  plans the batch and starts its upload (in segments, on a copy stream), its kernels and
 *       the copies back of calls / (run_off, run_cnt); it does not wait for the device.  Every array of the call, inputs
 *       and outputs, must stay where it is until _finish has returned for the batch; page-locked memory (ku_host_alloc)
 *       makes the copies asynchronous, pageable memory works but serialises.
 *   ku_classify_batch_rle_finish   waits for the OLDEST batch in flight (one event), settles the sparse-sketch emulation's
 *       bookkeeping for it and returns its *n_runs; ku_fetch_runs then copies that batch's runs (valid until the next
 *       _enqueue or _finish on the context).
 * Batches are finished in the order they were enqueued.  (A batch's way through the device is a chain of dependent steps --
 * upload, kernel, copies back, each with tens of microseconds of latency -- about 1 ms for a 60 k-read batch whose kernel takes 0.2 ms:
 * several batches in flight hide it.)  KU_ESTATE from _enqueue: KU_RLE_MAX_IN_FLIGHT batches are in flight already, or
 * the batch takes a path that cannot overlap (quick mode, several databases, sorted layout, a shard, exact counting, a
 * read beyond 65535 k-mers, an open work unit left by such a batch) while another is in flight: finish that one, then
 * enqueue again -- the batch is then classified inside _enqueue and _finish merely hands its totals over.  The entry
 * points that read or reset the run's state (reports, exports, ku_ctx_reset_counts, ku_sparse_close_unit,
 * ku_ctx_replace_calls, ku_classify_batch_rle itself) answer KU_ESTATE while batches are in flight. */
int ku_classify_batch_rle_enqueue(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                  const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                  uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, ku_run *runs, uint64_t runs_cap);
/* runs / runs_cap (optional, NULL / 0): where the batch's runs go.  The first runs_cap entries of the run array are copied there
 * along with the other results -- the extent in use is only known behind the kernel, a copy of exactly that size would have to
 * wait for the device.  After _finish, ku_classify_batch_rle_copied() says how many entries are in that buffer (0 when the batch
 * took a one-step path): when *n_runs is not larger, the runs are all there; else ku_fetch_runs brings them. */
int ku_classify_batch_rle_finish(ku_ctx *ctx, uint64_t *n_runs);
uint64_t ku_classify_batch_rle_copied(const ku_ctx *ctx); /* entries of the batch finished last that are in its `runs` buffer */
int ku_classify_batch_rle_in_flight(const ku_ctx *ctx); /* 0 .. KU_RLE_MAX_IN_FLIGHT */
/* optional: the buffers of n_jobs batches of up to n_bytes / n_reads (longest read max_read_len) ahead of the first batch --
 * device memory, page-locked scratch, streams and events that _enqueue would otherwise set up on first use -- and a warm-up:
 * a dozen synthetic batches (at most 65536 reads of 100 bases) go through the two-step path with KU_F_NO_COUNTS, three in
 * flight, so that what the runtime sets up lazily (the kernel's code object, copy queues, scratch memory) is there before
 * the caller's first batch: 40-60 ms of a `classify` run's first 60.  No state of the run changes.
 * The batches in flight take turns through all KU_RLE_MAX_IN_FLIGHT buffer sets: n_jobs = KU_RLE_MAX_IN_FLIGHT reserves them
 * all.  KU_NO_WARMUP=1 in the environment skips the warm-up. */
int ku_classify_batch_rle_reserve(ku_ctx *ctx, uint64_t n_bytes, uint64_t n_reads, uint32_t max_read_len, uint32_t n_jobs);

/* ---- out-of-core run: the database streamed through HBM chunk by chunk (classify -x SIZE; KrakenDB::prepare_chunking /
 * load_chunk / is_minimizer_in_chunk krakendb.cpp:411-526, process_file_with_db_chunk classify.cpp:566-791).  The
 * reference re-reads the input for every chunk and merges per-read taxa through temp files; here the read batches
 * stay resident on the device and each chunk is one lookup pass over them:
 *   ku_db_values(db) -> ku_ctx_load_db(ctx, db, chunk 0) -> ku_ctx_set_taxonomy(ctx, tax, values, n)
 *   b_i = ku_batch_create(...) for every batch of reads
 *   for every chunk c: ku_ctx_swap_shard(ctx, db, lo_c, hi_c) (c > 0); ku_batch_lookup(ctx, b_i, opts) for every i
 *   ku_batch_finish(ctx, b_i, ...) + ku_fetch_runs; ku_batch_destroy(b_i)
 * Every k-mer is searched and accounted (HLL, n_kmers; misses under taxon 0) by the one chunk that owns its
 * minimizer bin, so the chunks must tile [0, 4^nt).  Results equal a run with the whole database resident. */
/* ascending distinct non-zero taxids of the WHOLE database (host scan): the all_values of ku_ctx_set_taxonomy */
int ku_db_values(const ku_db *db, uint32_t *out, uint64_t *n);
/* replace the resident shard by bins [bin_lo, bin_hi) of `db`, keeping taxonomy, slot numbering and per-taxon state;
 * the slot table given to ku_ctx_set_taxonomy must cover the new shard's values (KU_EINVAL otherwise) */
int ku_ctx_swap_shard(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi);
/* Double buffering of the chunks: upload bins [bin_lo, bin_hi) and lay them out NEXT TO the resident shard, on a
 * stream of its own; a following ku_ctx_swap_shard with the same arguments then only exchanges the two.  The one
 * entry point that may run on a second host thread while the first one issues lookups on the same context (it touches
 * nothing the lookups use).  Blocking; at most one prefetched chunk at a time. */
int ku_ctx_prefetch_shard(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi);
/* free / total memory of the context's device (sizing of resident batches in out-of-core runs) */
int ku_ctx_mem_info(ku_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
typedef struct ku_batch ku_batch; /* reads + merged per-k-mer slots of one batch, resident on the context's device */
int ku_batch_create(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off, const uint32_t *seq_len,
                    uint64_t n_reads, ku_batch **out);
/* one pass against the resident shard.  Quick mode (KU_F_QUICK) follows the reference's CHUNKED run here
 * (classify.cpp:686-737), which differs from its normal quick mode: the passes book every unambiguous k-mer of every
 * read, ku_batch_finish counts the hits up to min_hits ("Q:n") and calls the taxon of the read's LAST unambiguous
 * k-mer when min_hits was reached */
int ku_batch_lookup(ku_ctx *ctx, ku_batch *b, const ku_opts *opts);
/* after the last chunk, once per batch (KU_ESTATE afterwards): resolve_tree / quick call per read + run-length
 * encoding; outputs as ku_classify_batch_rle */
int ku_batch_finish(ku_ctx *ctx, ku_batch *b, const ku_opts *opts, uint32_t *calls, uint32_t *hits, uint64_t *run_off,
                    uint32_t *run_cnt, uint64_t *n_runs);
void ku_batch_destroy(ku_batch *b);
/* Several GPUs on one out-of-core run (classify KU_DEVICES=0,1,... -x SIZE): every GPU keeps its own copy of the resident
 * batches and streams ITS share of the chunks over them (ku_batch_lookup: a pass only writes the positions whose bin the
 * resident chunk owns, so each position is written on exactly one GPU); ku_batch_absorb folds the slots another GPU's
 * copy collected into `dst` ("non-zero wins", src/classify.cpp:445-452), after which ku_batch_finish on dst's context
 * sees what a single GPU would have after all chunks.  ku_ctx_merge_state adds the per-taxon state the other GPU's passes
 * booked (registers MAX, n_kmers SUM, n_reads SUM) to dst's -- once, at the end of the run. */
int ku_batch_absorb(ku_ctx *ctx, ku_batch *dst, const ku_batch *src);
int ku_ctx_merge_state(ku_ctx *dst, ku_ctx *src);

/* Same on device-resident buffers, asynchronous on `stream` (a hipStream_t
 * passed as void*; NULL = the context's own, non-blocking stream).  The caller orders the work: whatever produced
 * the input buffers must have completed (or be ordered before `stream`), and the outputs are ready after
 * ku_ctx_synchronize / a synchronisation of `stream`. */
int ku_classify_batch_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                             const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts,
                             uint32_t *d_calls, uint32_t *d_taxa, uint32_t *d_hits, void *stream);

/* Device-resident batch in, the per-k-mer codes out as the Kraken line prints them (hitlist_string, classify.cpp:980-1010:
 * runs of equal codes): d_runs[runs_cap] {code, first k-mer}, per read d_run_off / d_run_cnt (as ku_classify_batch_rle
 * hands them to the host), *d_n_runs = the extent of d_runs in use.  The fused kernel writes the runs itself -- no
 * per-k-mer array exists at any time (6 GB per 10 M x 150 bp reads that ku_classify_batch_device writes).  max_read_len
 * in `opts` is REQUIRED here (the longest read of the batch; no host round trip to find it).  Asynchronous on `stream`.
 * ku_device_rle_runs_cap() is the capacity that holds any batch whose reads change taxon at most every sixth base; when
 * *d_n_runs comes back larger than runs_cap the run array was too small: calls and the per-taxon state are complete and
 * correct, the runs are not (redo the batch with KU_F_NO_COUNTS and a larger array, or through
 * ku_classify_batch_device).  KU_EUNSUP where the fused kernel does not apply (sorted layout, several databases, quick
 * mode, exact counting, the sparse-sketch emulation -- which needs the read lengths on the host --, reads beyond 65535
 * k-mers): use ku_classify_batch_device there. */
int ku_classify_batch_device_rle(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                                 const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                                 ku_run *d_runs, uint64_t runs_cap, uint64_t *d_run_off, uint32_t *d_run_cnt,
                                 uint64_t *d_n_runs, void *stream);
uint64_t ku_device_rle_runs_cap(const ku_ctx *ctx, uint64_t n_bytes, uint64_t n_reads, uint32_t max_read_len);

/* The two stages separately (sharded multi-GPU run: lookup on every rank, RCCL
 * max-reduce of d_taxa, resolve on the reads each rank owns):
 *  stage 1 = KmerScanner + canonical_representation + bin_key + kmer_query for
 *            every k-mer whose bin this context owns + ReadCounts::add_kmer
 *            (krakenutil.cpp:237-282, krakendb.cpp:200-321, classify.cpp:918-939;
 *            ownership test = is_minimizer_in_chunk, krakendb.cpp:524-526);
 *            writes slot ids (0 where not owned / miss, KU_AMBIG where ambiguous).
 *  stage 2 = hit_counts + resolve_tree + incrementReadCount (classify.cpp:941-968,
 *            krakenutil.cpp:149-200) and slot -> taxid translation of d_taxa. */
int ku_lookup_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const ku_opts *opts,
                     uint32_t *d_taxa, void *stream);
int ku_resolve_device(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_seq_off, const uint32_t *d_seq_len,
                      uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls, uint32_t *d_taxa,
                      uint32_t *d_hits, void *stream);
int ku_ctx_synchronize(ku_ctx *ctx);
/* Measurement aid (bench.py roofline): runs the scan + minimizer + idx stages only
 * and returns, for the k-mers this context owns, stats_out[4] = { lookups,
 * sum over lookups of ceil(log2(n_bin + 1)), lookups into non-empty bins,
 * sum of n_bin } -- the inputs of the algorithmic-bytes model
 * B(q) = 16 + 12 * ceil(log2(n_bin + 1)) + 4.  Blocking. */
int ku_lookup_stats_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, uint64_t *stats_out, void *stream);

/* ------------------------------------------------------------------ UID databases (classify -I)
 * A UID database (set_lcas -I, scripts/krakenuniq --uid-mapping) stores, instead of an LCA taxid, the id of the SET of
 * taxids whose library sequences hold the k-mer; uid_to_taxid.map lists the sets as {taxid, parent uid} blocks
 * (src/uid_mapping.cpp:32-91,279-302).  classify -I resolves a read from its per-UID hit counts with resolve_uids3
 * (src/uid_mapping.cpp:212-274; src/classify.cpp:953-960): the taxid with the most hits, ties by the larger sum of
 * count / |set| (double), then the LCA of the tied taxids -- summed and compared in the iteration order of the
 * std::unordered_maps the reference walks.  Lookup, per-k-mer codes (the UIDs), HLL / n_kmers accounting run on the GPU
 * as for any database (the values are just numbers); the resolve step is host code here: it uses the same container in
 * the same insertion order as the reference, which is what makes the result identical to a reference built with the
 * same C++ library.  ku_resolve_uids takes a batch's run-length encoded codes (ku_classify_batch_rle + ku_fetch_runs)
 * and returns the calls; ku_ctx_replace_calls then moves the device's read counts of the last batch from the calls
 * resolve_tree made to these (classify.cpp:968 counts the read under the UID call). */
typedef struct ku_uid_map ku_uid_map;
int ku_uid_map_open(const char *path, ku_uid_map **out);
/* blocks: n * 2 uint32 {taxid, parent uid} */
int ku_uid_map_from_blocks(const uint32_t *blocks, uint64_t n, ku_uid_map **out);
void ku_uid_map_close(ku_uid_map *m);
uint64_t ku_uid_map_size(const ku_uid_map *m);
/* seq_len: the reads' lengths (the last run of a read ends at its k-mer count); n_threads host threads (0 = 1).
 * KU_EDATA when a code is not a uid of the map. */
int ku_resolve_uids(const ku_tax *tax, const ku_uid_map *map, const ku_run *runs, const uint64_t *run_off, const uint32_t *run_cnt,
                    const uint32_t *seq_len, uint64_t n_reads, uint32_t k, uint32_t n_threads, uint32_t *calls);
/* the calls of the batch classified last on this context are replaced by new_calls[n_reads]: n_reads of the per-taxon
 * state moves accordingly (a call that is neither a taxDB id nor a database value cannot be counted: *n_dropped) */
int ku_ctx_replace_calls(ku_ctx *ctx, const uint32_t *new_calls, uint64_t n_reads, uint64_t *n_dropped);

/* ------------------------------------------------------------------ per-taxon state
 * Export of the run's `taxon_counts` (classify.cpp:78): one row per slot
 * (slot 0 = taxid 0 = "no hit") with its k-mer count and dense p=12 registers,
 * plus read counts per called taxid. */
typedef struct ku_counts_dims {
  uint64_t n_slots;  /* rows of slot_taxid / n_kmers / registers (includes slot 0) */
  uint64_t n_nodes;  /* rows of node_taxid / n_reads */
} ku_counts_dims;
int ku_counts_dims_get(ku_ctx *ctx, ku_counts_dims *out);
/* Any output pointer may be NULL.  registers: n_slots * 4096 bytes. */
int ku_counts_export(ku_ctx *ctx, uint32_t *slot_taxid, uint64_t *n_kmers, uint8_t *registers,
                     uint32_t *node_taxid, uint64_t *n_reads);
/* Device pointers of the live state for in-place RCCL reduction at the end of a
 * sharded run (max on registers, sum on the counters). */
int ku_counts_device_ptrs(ku_ctx *ctx, uint8_t **d_registers, uint64_t *n_register_bytes,
                          uint64_t **d_n_kmers, uint64_t *n_slots, uint64_t **d_n_reads, uint64_t *n_nodes);

/* ------------------------------------------------------------------ several GPUs
 * The database split into contiguous minimizer-bin ranges, one per GPU, all resident at the same time.  The reference
 * has this partitioning only in time (--preload-size chunk mode): ownership KrakenDB::prepare_chunking / upper_bound /
 * is_minimizer_in_chunk (krakendb.cpp:430-526), per-chunk lookup classify_sequence_with_db_chunk (classify.cpp:1014-1056),
 * "non-zero wins" merge (classify.cpp:445-452), final resolve pass (classify.cpp:676-785).  Here, per batch, either
 *   OWNER ROUTING (default): rank r takes slice r of the reads, scans it once (minimizers), sends every RUN of unambiguous
 *   k-mers that share their minimizer occurrence -- one 16-byte record: the run's bases at 2 bits, its length, the offset
 *   of the minimizer -- to the rank that owns the bin (all-to-all), which expands the records, probes its shard and
 *   accounts the k-mers -- HLL, n_kmers, misses under taxon 0: owner-computes, nothing is counted twice -- and sends one
 *   slot per k-mer back; or
 *   POSITION-WISE (KU_MGPU_EXCHANGE=slots; shards in the sorted layout, several passes): broadcast of the read batch from
 *   rank 0 -> ku_lookup_device(KU_F_KEEP_SLOTS) on every rank (each searches and accounts only the k-mers whose bin it
 *   owns) -> max-reduce of the per-k-mer slots, scattered over the read dimension;
 *   then ku_resolve_device + run-length encoding of the slice on rank r;
 * at the end of the run ku_mgpu_reduce_state merges the per-taxon state (registers MAX, n_kmers / n_reads SUM).
 * A ku_mgpu drives `n_local` ranks of a `world` of ranks from this process, one host thread per rank:
 *   - one process, all ranks (first_rank = 0, n_local = world, id = NULL): what the classify executable does for
 *     KU_DEVICES=0,1,...; the collectives are RCCL (ncclBroadcast, grouped ncclSend / ncclRecv + merge = reduce-scatter
 *     with read-aligned slices, ncclAllReduce) over xGMI when the devices are distinct, and device-to-device copies + merge kernels when
 *     a device is listed more than once (several ranks on one GPU: tests, 1-GPU boxes; RCCL allows one rank per device);
 *   - one process per GPU (n_local = 1, the launcher hands every process the id rank 0 made with ku_mgpu_unique_id):
 *     RCCL through ncclCommInitRank; bench.py under torch.distributed.run.
 * KU_MGPU_REPLICAS: every rank holds the whole database instead and classifies its own reads; only the per-taxon
 * state is merged.  Collective calls (create, load / set_taxonomy, every batch, reduce_state) must be made by every
 * process of the world in the same order. */
typedef struct ku_mgpu ku_mgpu;
#define KU_MGPU_ID_BYTES 128
#define KU_MGPU_REPLICAS 0x1u
#define KU_MGPU_NO_RCCL 0x2u /* single-process groups: use the copy + merge-kernel exchange even between distinct devices */
int ku_mgpu_unique_id(uint8_t *id /* [KU_MGPU_ID_BYTES] */);
int ku_mgpu_create(const int *devices, uint32_t n_local, uint32_t first_rank, uint32_t world, const uint8_t *id,
                   uint32_t flags, ku_mgpu **out);
void ku_mgpu_destroy(ku_mgpu *m);
/* the context of local rank i (0 <= i < n_local): adopt a device-resident shard, export counts, ... */
ku_ctx *ku_mgpu_ctx(ku_mgpu *m, uint32_t local_index);
/* 1 when the ranks exchange through RCCL, 0 for the same-process copy + merge exchange */
int ku_mgpu_uses_rccl(const ku_mgpu *m);
/* 1 when sharded batches are owner-routed (valid after the load / ku_mgpu_set_taxonomy): a rank scans only its own slice of
 * the reads and sends every run of k-mers to the rank that owns its minimizer bin (~2 B per k-mer out, a 4-B slot back)
 * instead of every rank scanning every read and exchanging 4 B per base position.  The default for sharded groups whose ranks all hold a
 * probe table of one database; KU_MGPU_EXCHANGE=slots keeps the position-wise exchange. */
int ku_mgpu_uses_routing(const ku_mgpu *m);
/* Measurement aid (bench.py): with timing on, every owner-routed step brackets its stages with HIP events on the streams
 * they run on; ku_mgpu_step_times waits for the last step of local rank `local_index` and returns, in milliseconds,
 * out[0] the scan (+ the numbering of its records), out[1] the owner side (numbering + probe / accounting kernel),
 * out[2] the resolve stage (tickets -> slots -> calls), out[3] the number of rounds, out[4] records received, out[5]
 * k-mers received (both summed over the rounds).  The exchanges themselves are not in these figures; a timed step runs its
 * rounds on one stream (no overlap between the owner kernel of one round and the scan of the next), so the figures add up. */
int ku_mgpu_set_timing(ku_mgpu *m, int on);
int ku_mgpu_step_times(ku_mgpu *m, uint32_t local_index, double *out /* [6] */);
/* shard plan (ku_db_shard_plan over the world) + upload of every local rank's range + taxonomy with the slot table of
 * the whole database (the ranks' distinct values are all-gathered); KU_MGPU_REPLICAS: the whole database everywhere */
int ku_mgpu_load(ku_mgpu *m, const ku_db *db, const ku_tax *tax);
/* several databases searched in order per k-mer (classify -d A -d B, classify.cpp:928-936): replicas only -- with the
 * first database in shards a later one could not tell that another rank had found the k-mer (the reference's own chunk
 * mode searches the first database only, classify.cpp:639); KU_EUNSUP otherwise */
int ku_mgpu_load_dbs(ku_mgpu *m, const ku_db *const *dbs, uint32_t n_dbs, const ku_tax *tax);
/* The report modes of one context, over the group (single-process groups; call after the load):
 *  - HyperLogLog++ sparse-mode emulation (ku_ctx_enable_sparse): host batches are cut at WORK UNIT boundaries, every rank
 *    runs the emulation on whole units (replicas: the fused kernel's fast path; shards: on the merged per-k-mer slots of
 *    its slice), the unit that is still open when a batch ends continues on rank 0 with the next batch;
 *    ku_mgpu_reduce_state folds the ranks' states into rank 0's context (dense if any rank says so, else the union of the
 *    ranks' sets), whose ku_ctx_report then equals the single-GPU report.  ku_mgpu_sparse_close_unit between input files.
 *  - exact distinct counts (classifyExact; sharded mode): a k-mer is put into the set of the rank that owns its bin, the
 *    per-slot counts add up in ku_mgpu_reduce_state. */
int ku_mgpu_enable_sparse(ku_mgpu *m, uint64_t work_unit_nt, uint32_t global_log2);
int ku_mgpu_sparse_close_unit(ku_mgpu *m);
int ku_mgpu_sparse_state(const ku_mgpu *m); /* as ku_ctx_sparse_state; 2 as soon as one rank gave up */
int ku_mgpu_enable_exact(ku_mgpu *m, uint32_t capacity_log2);
/* the same last step alone, after the local shards were adopted through ku_mgpu_ctx() + ku_ctx_adopt_db */
int ku_mgpu_set_taxonomy(ku_mgpu *m, const ku_tax *tax);
/* One batch on host buffers, arguments and results as ku_classify_batch_rle (+ ku_mgpu_fetch_runs for the runs).
 * Single-process groups only.  Sharded: broadcast / lookup / reduce-scatter / resolve as above; replicas: the reads are
 * cut into `world` slices, rank r classifies slice r. */
int ku_mgpu_classify_batch_rle(ku_mgpu *m, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                               const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                               uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs);
int ku_mgpu_fetch_runs(ku_mgpu *m, ku_run *runs, uint64_t n_runs);
/* One sharded batch on device buffers, asynchronous on each local rank's stream (measurement path; multi-process
 * capable).  Per local rank: d_seqs / d_seq_off / d_seq_len hold the batch on rank 0 and receive it elsewhere
 * (n_bytes + 16, n_reads * 8, n_reads * 4 bytes), d_taxa (4 * (n_bytes + 16) bytes) and d_calls (4 * n_reads) are
 * outputs: rank r leaves the calls and per-k-mer taxids of the reads [read_bounds[r], read_bounds[r+1]) in place
 * (same indices as an unsharded run).  pos_bounds[r] = seq_off[read_bounds[r]] (pos_bounds[world] = n_bytes): the
 * slice of the per-k-mer array rank r receives.  stream = NULL: the context's own. */
typedef struct ku_mgpu_dev_batch {
  void *d_seqs;
  uint64_t *d_seq_off;
  uint32_t *d_seq_len;
  uint32_t *d_calls;
  uint32_t *d_taxa;
  void *stream;
} ku_mgpu_dev_batch;
int ku_mgpu_step_device(ku_mgpu *m, const ku_mgpu_dev_batch *local /* [n_local] */, uint64_t n_bytes, uint64_t n_reads,
                        const uint64_t *read_bounds /* [world + 1] */, const uint64_t *pos_bounds /* [world + 1] */,
                        const ku_opts *opts);
/* merge the per-taxon state over all ranks in place (every rank ends up with the whole run's state): HLL registers
 * MAX, n_kmers and n_reads SUM -- taxon_counts[t] += local[t] (classify.cpp:541-544) across GPUs.  Call once, at the
 * end of the run: a second call without a batch in between is refused (KU_ESTATE; it would add the sums again --
 * a caller that resets the contexts' counts itself classifies a batch before it reduces again);
 * streams[i] = NULL: the context's own stream. */
int ku_mgpu_reduce_state(ku_mgpu *m, void *const *streams /* [n_local] or NULL */);
/* database.kdb.counts over all local shards (ku_ctx_count_taxons summed; single-process groups) */
int ku_mgpu_count_taxons(ku_mgpu *m, uint32_t *taxids, uint64_t *counts, uint64_t *n);

/* ------------------------------------------------------------------ host-side helpers
 * (double arithmetic / text; the reference does these on the host too) */
/* Ertl improved estimator on dense registers, clipped to n_observed
 * (HyperLogLogPlusMinus::ertlCardinality, hyperloglogplus.cpp:722-753). */
uint64_t ku_hll_cardinality(const uint8_t *registers, uint32_t p, uint64_t n_observed);
/* hitlist_string (classify.cpp:826-861): RLE of one read's codes; returns bytes
 * written (buf must hold 24 * n + 8 bytes). */
size_t ku_hitlist_string(const uint32_t *taxa, size_t n, char *buf);
/* Kraken output lines for a batch (classify.cpp:980-1010).  ids: NUL-separated
 * read ids (header up to first whitespace, seqreader.cpp:57-58).  flags:
 * KU_P_ONLY_CLASSIFIED (-c), KU_P_SEQUENCE (-s), KU_P_QUICK.  Returns a
 * malloc'ed buffer in *out (free with ku_free) and its length. */
#define KU_P_ONLY_CLASSIFIED 0x1u
#define KU_P_SEQUENCE 0x2u
#define KU_P_QUICK 0x4u
int ku_format_kraken(const char *seqs, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                     const char *ids, uint32_t k, const uint32_t *calls, const uint32_t *taxa,
                     const uint32_t *hits, uint32_t flags, char **out, size_t *out_len);
/* ku_format_kraken fed with the run-length encoded codes of ku_classify_batch_rle (taxa -> runs/run_off/run_cnt). */
int ku_format_kraken_rle(const char *seqs, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                         const char *ids, uint32_t k, const uint32_t *calls, const ku_run *runs,
                         const uint64_t *run_off, const uint32_t *run_cnt, const uint32_t *hits, uint32_t flags,
                         char **out, size_t *out_len);
/* Report with the reference's default columns
 * "%  reads  taxReads  kmers  dup  cov  taxID  rank  taxName"
 * (TaxReport, taxdb.hpp:928-1123; classify.cpp:288-325).  counts_path =
 * database.kdb.counts (NULL -> cov = NA).  Rows for sibling taxa with equal
 * (reads, kmers) are ordered by ascending taxid (unspecified in the reference). */
int ku_report(const ku_tax *tax, const char *counts_path, const uint32_t *slot_taxid, const uint64_t *n_kmers,
              const uint8_t *registers, uint64_t n_slots, const uint32_t *node_taxid, const uint64_t *n_reads,
              uint64_t n_nodes, char **out, size_t *out_len);
/* Same with one counts file per database of a hierarchical run: genome sizes add up in the order given
 * (readGenomeSizes once per database, classify.cpp:263-285; taxdb.hpp:850-885). */
int ku_report_multi(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint32_t *slot_taxid,
                    const uint64_t *n_kmers, const uint8_t *registers, uint64_t n_slots, const uint32_t *node_taxid,
                    const uint64_t *n_reads, uint64_t n_nodes, char **out, size_t *out_len);
/* The report with the reference's sparse sketches: slot_is_sparse[n_slots] and sparse_pairs[n_pairs] (slot << 32 |
 * encoded hash) as ku_sparse_export returns them.  A clade's sketch is the merge of its members' by the reference's
 * rules (HyperLogLogPlusMinus::merge, hyperloglogplus.cpp:586-665): dense as soon as one member is, else the union of
 * the members' sets; `kmers` is the Ertl estimate of that state (sparse: m = 2^25, q = 39; :356-366,726-753). */
int ku_report_sparse(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint32_t *slot_taxid,
                     const uint64_t *n_kmers, const uint8_t *registers, const uint8_t *slot_is_sparse,
                     const uint64_t *sparse_pairs, uint64_t n_pairs, uint64_t n_slots, const uint32_t *node_taxid,
                     const uint64_t *n_reads, uint64_t n_nodes, char **out, size_t *out_len);
/* Ertl estimate of a sparse sketch: its n distinct 32-bit encoded hashes (p = 12, p' = 25), clipped to n_observed */
uint64_t ku_hll_cardinality_sparse(const uint32_t *encoded, uint64_t n, uint64_t n_observed);
/* classifyExact's report (classify built with EXACT_COUNTING, classify.cpp:46-53): `kmers` is the exact number of
 * distinct k-mers (ku_counts_export_exact) instead of the HyperLogLog estimate; a clade's count is the sum over its
 * members (a k-mer has one database value, the members' sets are disjoint). */
int ku_report_exact(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint32_t *slot_taxid,
                    const uint64_t *n_kmers, const uint64_t *unique_kmers, uint64_t n_slots, const uint32_t *node_taxid,
                    const uint64_t *n_reads, uint64_t n_nodes, char **out, size_t *out_len);
/* The report text from per-entry clade summaries, arrays parallel to the taxDB entries (n_rows = their number, file
 * order): present[r] = the clade of entry r was counted, clade_reads / clade_kmers = sums over the clade, tax_reads =
 * the entry's own reads, clade_uniq = distinct k-mers of the clade's merged sketch.  Rows, order and number formats
 * of TaxReport::printReport (taxdb.hpp:1004-1123); what ku_report* and ku_ctx_report end in. */
int ku_report_rows(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint8_t *present,
                   const uint64_t *clade_reads, const uint64_t *tax_reads, const uint64_t *clade_kmers,
                   const uint64_t *clade_uniq, uint64_t n_rows, char **out, size_t *out_len);
/* ... with a choice of columns.  flags = KU_R_NO_KMER_COLS: "%  reads  taxReads  taxID  rank  taxName", the report of
 * `classify -p 0` (HLL_PRECISION <= 0, classify.cpp:289,316-323); same rows in the same order, clade_uniq may be NULL. */
#define KU_R_NO_KMER_COLS 1u
int ku_report_rows_cols(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint8_t *present,
                        const uint64_t *clade_reads, const uint64_t *tax_reads, const uint64_t *clade_kmers,
                        const uint64_t *clade_uniq, uint64_t n_rows, uint32_t flags, char **out, size_t *out_len);
/* The report straight from the context's device-resident state: the clade roll-up of the TaxReport
 * constructor (taxdb.hpp:928-982: every counted taxon's sketch merged into each of its ancestors') runs on the GPU --
 * one workgroup per counted clade takes the byte-wise maximum over its members' HLL registers and reduces it to the
 * register histogram the estimator needs; clades whose members all stayed sparse (ku_ctx_enable_sparse) get the
 * histogram of the union of their members' encoded hashes from a device hash set.  Only the histograms (320 B per
 * clade) and the per-taxon counters come back; estimator and text are host work as in the reference.  Same text as
 * ku_report_multi / ku_report_sparse / ku_report_exact on the exported state (whichever mode the context is in).  With the
 * sparse-mode emulation on it ends the work unit that is still open, like ku_sparse_export: a call for the end of a run. */
int ku_ctx_report(ku_ctx *ctx, const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, char **out,
                  size_t *out_len);
/* ... with the flags of ku_report_rows_cols: KU_R_NO_KMER_COLS needs no sketch at all -- no roll-up runs */
int ku_ctx_report_cols(ku_ctx *ctx, const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, uint32_t flags,
                       char **out, size_t *out_len);
void ku_free(void *p);
/* Page-locked host memory for batch buffers (fast, truly asynchronous H2D / D2H in ku_classify_batch). */
int ku_host_alloc(size_t bytes, void **out);
void ku_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* KRAKENUNIQ_AMD_H */
