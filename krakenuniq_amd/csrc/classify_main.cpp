// classify_main.cpp -- drop-in for the reference's `classify` executable (src/classify.cpp), host side only:
// flag parsing, FASTA/FASTQ(+gz) ingest, batching, output files, stderr summary, report.  Every per-read
// computation goes through the C ABI (include/krakenuniq_amd.h) to the HIP kernels; there is no CPU
// classification path in this program.
//
// Honoured getopt string (src/classify.cpp:1074): d:i:t:u:n:m:o:qcC:U:Ma:r:sI:p:x:
//   -d kdb  -i idx  -a taxDB            as the reference; several -d/-i pairs = hierarchical run, searched in order
//                                       (src/classify.cpp:163-177,928-936)
//   -o file|off|-                       Kraken output ("-" and "off" both disable it, src/classify.cpp:234-235)
//   -r file|off                         report, opened in APPEND mode like the reference (:286)
//   -C/-U file                          classified / unclassified reads;  -c only classified lines;  -s print sequence
//   -q -m N                             quick mode
//   -t N                                host threads parsing the input and formatting the output (0 < N <= processors,
//                                       src/classify.cpp:1085-1088; default 4); the GPU replaces the OpenMP classification team
//   -u N                                work unit size in nt as in the reference (default 500000, src/classify.cpp:38): it decides
//                                       which per-taxon sketches stay sparse, i.e. which `kmers` of the report are near
//                                       exact (HLL sparse-mode emulation; KU_NO_SPARSE=1 switches it off: dense estimates).
//                                       The GPU batch size is separate: KU_BATCH_NT (default 64 Mi nt)
//   -M                                  accepted: the database is always preloaded (into HBM)
//   -x SIZE                             the database is streamed through HBM in minimizer-range chunks of at most SIZE
//                                       bytes (src/krakendb.cpp:463-522) when that yields more than one chunk
//   -p N                                accepted and ignored exactly like the reference
//   KU_DEVICES=0,1,...                  several GPUs (see the end of this comment): every flag keeps its meaning -- -r reports
//                                       with the reference's sparse sketches (each GPU takes whole work units), several
//                                       -d run as replicas, classifyExact on the sharded database; not with -x / -I
//   -I file                             UID database (set_lcas -I / --uid-mapping): the values of the (single) database are
//                                       UIDs, reads are resolved with resolve_uids3 on the host from the device's
//                                       run-length encoded codes (src/classify.cpp:953-960, src/uid_mapping.cpp:212-274); no
//                                       quick mode (the reference exits there too), one GPU, database resident
// Extensions: -P (mate pairs merged on the fly); env KU_DEVICE selects the GPU (default 0); env KU_DEVICES=0,1,...
// runs on several GPUs through the multi-GPU driver (ku_mgpu: database sharded by minimizer range, read batches
// broadcast, per-k-mer slots reduce-scattered, per-taxon state reduced at the end; KU_MGPU_MODE=replicas keeps the
// whole database on every GPU and splits the reads instead).
#include "classify_run.h"

void ku_seqio::fatal(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "classify: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  leave(code);
}


static void usage(int code) {  // text of src/classify.cpp:1164-1189
  fprintf(stderr,
          "Usage: classify [options] <fasta/fastq file(s)>\n\n"
          "Options: (*mandatory)\n"
          "* -d filename      Kraken DB filename\n"
          "* -i filename      Kraken DB index filename\n"
          "  -o filename      Output file for Kraken output\n"
          "  -r filename      Output file for Kraken report output\n"
          "  -a filename      TaxDB\n"
          "  -I filename      UID to TaxId map\n"
          "  -p #             Precision for unique k-mer counting, between 10 and 18\n"
          "  -t #             Number of threads\n"
          "  -u #             Thread work unit size (in bp)\n"
          "  -q               Quick operation\n"
          "  -m #             Minimum hit count (ignored w/o -q)\n"
          "  -C filename      Print classified sequences\n"
          "  -U filename      Print unclassified sequences\n"
          "  -c               Only include classified reads in output\n"
          "  -M               Preload database files\n"
          "  -x size          Preload database files using x amount of RAM (e.g. 10G)\n"
          "  -s               Print read sequence in Kraken output\n"
          "  -P               (extension) input files are mate pairs: merged on the fly as read_merger.pl does\n"
          "  -h               Print this message\n\n"
          "Kraken output is to standard output by default.\n");
  exit(code);
}

// parse_human_readable_size (src/krakenutil.cpp:30-55)
static uint64_t parse_size(const char *s) {
  char *end = nullptr;
  errno = 0;
  unsigned long long x = strtoull(s, &end, 10);
  if (errno || end == s) return 0;
  int sh;
  switch (*end) {
    case 'k': case 'K': sh = 10; break;
    case 'm': case 'M': sh = 20; break;
    case 'g': case 'G': sh = 30; break;
    case 0: sh = 0; break;
    default: return 0;
  }
  if (x > (UINT64_MAX >> sh)) return 0;
  return (uint64_t)x << sh;
}

int main(int argc, char **argv) {
  static Run run;  // (static: its queues and counters outlive every thread that may still look at them when a fatal error leaves)
  return run.run(argc, argv);
}

int Run::run(int argc_, char **argv_) {
  argc = argc_;
  argv = argv_;
  std::vector<std::string> dbs, idxs;
  std::string kraken_out, report_out, taxdb, cls_out, ucls_out, uid_map_file;
  bool only_classified = false, print_seq = false, populate = false;
  uint32_t min_hits = 1;
  unit_nt = 64ull << 20;             // GPU batch size in nt (KU_BATCH_NT; round 5: 64 Mi = regions of 16 Mi nt, one launch of ~120 k reads each -- with the
                                     // batch call in two steps the window is flat from 40 to 96 Mi and the kernel's cost per read falls with the launch size); plain and .gz files travel in regions of a quarter of it.
                                     // 10 M x 150 bp end to end (scripts/e2e_sweep.py, profiles/r04_e2e_sweep.log): 128 Mi 0.53 s, 64 Mi 0.44,
                                     // 32 Mi 0.29, 24 Mi 0.31, 16 Mi 0.36, 8 Mi 0.46 -- larger batches fill and drain the three stages slowly,
                                     // smaller ones pay the device stage's ~1 ms per call too often
  uint64_t work_unit_nt = 500000;   // -u: the reference's Work_unit_size (src/classify.cpp:38)
  uint64_t chunk_bytes = 0;  // -x SIZE: stream the database through HBM in chunks of at most SIZE bytes
  int hll_precision = 1;  // -p: only its sign matters (six or nine report columns)
  fmt_threads = 4;  // -t: host threads that format the Kraken lines (the GPU replaces the OpenMP team)
  if (argc > 1 && strcmp(argv[1], "-h") == 0) usage(0);
  // The formatted lines' buffers (~1 MB each, sixteen per batch, allocated by the helpers and freed by the writer) come from the
  // heap and stay there: by default malloc gives blocks of that size an mmap / munmap pair each -- 2 600 exclusive acquisitions
  // of the address space's lock per 10 M reads, each waiting for (and holding up) the page faults of the parser team.
  // KU_MALLOPT=0: malloc's defaults.
  if (!(getenv("KU_MALLOPT") && atoi(getenv("KU_MALLOPT")) == 0)) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, getenv("KU_TOP_PAD_MB") ? atoi(getenv("KU_TOP_PAD_MB")) << 20 : 16 << 20);  // (a thread's heap grows by mprotect -- exclusive, too: in few large steps)
  }
  int opt;
  while ((opt = getopt(argc, argv, "d:i:t:u:n:m:o:qcC:U:Ma:r:sI:p:x:P")) != -1) {
    long long sig;
    switch (opt) {
      case 'd': dbs.push_back(optarg); break;
      case 'i': idxs.push_back(optarg); break;
      case 't':
        sig = atoll(optarg);
        if (sig <= 0) die(EX_USAGE, "can't use nonpositive thread count");
        {  // src/classify.cpp:1085-1088 (omp_get_num_procs there)
          const long procs = sysconf(_SC_NPROCESSORS_ONLN);
          if (procs > 0 && sig > procs) die(EX_USAGE, "thread count exceeds number of processors");
        }
        fmt_threads = (int)(sig > 64 ? 64 : sig);
        break;
      case 'p': {  // HLL_PRECISION only selects the report's columns in the reference (the sketch is p = 12 whatever it says):
                   // <= 0 drops kmers / dup / cov (classify.cpp:289,316-323,1093-1095).  std::stoi's reading: leading blanks, a
                   // sign, digits, the rest ignored; nothing to read is its std::invalid_argument (the reference aborts there)
        char *end = nullptr;
        const long v = strtol(optarg, &end, 10);
        if (end == optarg) die(EX_USAGE, "-p: not a number: %s", optarg);
        hll_precision = v > 0 ? 1 : 0;
        break;
      }
      case 'q': quick = true; break;
      case 'm':
        sig = atoll(optarg);
        if (sig <= 0) die(EX_USAGE, "can't use nonpositive minimum hit count");
        min_hits = (uint32_t)sig;
        break;
      case 'c': only_classified = true; break;
      case 'C': print_cls = true; cls_out = optarg; break;
      case 'U': print_ucls = true; ucls_out = optarg; break;
      case 'o': kraken_out = optarg; break;
      case 'r': report_out = optarg; break;
      case 's': print_seq = true; break;
      case 'a': taxdb = optarg; break;
      case 'u':
        sig = atoll(optarg);
        if (sig <= 0) die(EX_USAGE, "can't use nonpositive work unit size");
        work_unit_nt = (uint64_t)sig;
        break;
      case 'M': populate = true; break;
      case 'x':
        populate = true;
        chunk_bytes = parse_size(optarg);
        if (chunk_bytes == 0) die(EX_USAGE, "can't parse preload size %s", optarg);
        break;
      case 'I': uid_map_file = optarg; break;
      case 'n': break;
      case 'P': paired = true; break;  // extension: the input files are mate pairs, merged on the fly (scripts/read_merger.pl)
      default: usage(EX_USAGE);
    }
  }
  if (const char *e = getenv("KU_BATCH_NT")) {
    const long long v = atoll(e);
    if (v > 0) unit_nt = (uint64_t)v < (1ull << 16) ? (1ull << 16) : (uint64_t)v;
  }
  if (dbs.empty()) { fprintf(stderr, "Missing mandatory option -d\n"); usage(EX_USAGE); }
  if (idxs.empty()) { fprintf(stderr, "Missing mandatory option -i\n"); usage(EX_USAGE); }
  if (dbs.size() != idxs.size()) die(EX_USAGE, "every -d needs its -i (%zu databases, %zu indexes)", dbs.size(), idxs.size());
  if (dbs.size() > 8) die(EX_SOFTWARE, "at most 8 databases");
  // installed as `classifyExact` (the reference's EXACT_COUNTING build, src/classify.cpp:46-53) the report counts
  // distinct k-mers exactly instead of estimating them
  const char *base = strrchr(argv[0], '/');
  const bool exact = strcmp(base ? base + 1 : argv[0], "classifyExact") == 0 || getenv("KU_EXACT") != nullptr;
  map_uids = !uid_map_file.empty();
  if (map_uids && dbs.size() > 1) { fprintf(stderr, "Cannot use more than one database with UID mapping!\n"); return 1; }  // src/classify.cpp:158-160
  if (map_uids && quick) { fprintf(stderr, "Quick mode not available when mapping UIDs\n"); return 1; }                  // :954-956
  if (optind == argc && !populate) fprintf(stderr, "No sequence data files specified\n");
  if (paired && (argc - optind) % 2) die(EX_USAGE, "-P needs the input files in pairs (mate 1, mate 2)");
  if (taxdb.empty()) { fprintf(stderr, "TaxDB argument is required!\n"); return 1; }  // src/classify.cpp:221-222

  // hierarchical run: the databases are searched in command-line order (src/classify.cpp:163-177,928-936)
  std::vector<ku_db *> db_handles(dbs.size(), nullptr);
  for (size_t i = 0; i < dbs.size(); ++i) {
    fprintf(stderr, " Database %s\n", dbs[i].c_str());
    KU_CHECK(ku_db_open(dbs[i].c_str(), idxs[i].c_str(), &db_handles[i]));
    ku_db_info inf;
    KU_CHECK(ku_db_get_info(db_handles[i], &inf));
    fprintf(stderr, "Loaded database with %" PRIu64 " keys with k of %u [val_len 4, key_len %u].\n", inf.key_ct, inf.k, inf.key_len);
    if (i == 0) info = inf;
    else if (inf.k != info.k) {  // src/classify.cpp:199-208
      fprintf(stderr, "Different k-mer sizes in databases 1 and %zu: %i vs %i!\n", i + 1, (int)info.k, (int)inf.k);
      return 1;
    }
  }
  db = db_handles[0];
  KU_CHECK(ku_tax_open(taxdb.c_str(), &tax));
  if (map_uids) {
    fprintf(stderr, "Reading UID mapping file %s\n", uid_map_file.c_str());  // src/classify.cpp:163
    KU_CHECK(ku_uid_map_open(uid_map_file.c_str(), &uid_map));
  }
  // -x SIZE (src/krakendb.cpp:463-522): the chunk plan of the reference.  The reference runs its chunk mode whenever -x is given
  // (src/classify.cpp:196-198,251-252), also when the plan has ONE chunk -- and that mode differs from the plain one in what it
  // counts: every k-mer goes into the run's global sketches (one work unit: :719), and quick mode calls the taxon of the read's
  // last unambiguous k-mer (:686-737).  One chunk and no -q: everything resident, the plain pipeline with that accounting;
  // one chunk with -q: the chunked pipeline over its single chunk (its own quick kernel).
  bool one_chunk = false;
  if (chunk_bytes) {
    chunk_bounds.resize(info.n_bins + 2 < (1u << 20) ? info.n_bins + 2 : (1u << 20));
    uint32_t n_chunks = 0;
    KU_CHECK(ku_db_chunk_plan(db, chunk_bytes, chunk_bounds.data(), (uint32_t)chunk_bounds.size() - 1, &n_chunks));
    chunk_bounds.resize(n_chunks + 1);
    chunk_bounds.back() = info.n_bins;  // the bins behind the last chunk hold no pairs
    one_chunk = n_chunks <= 1;
    if (n_chunks == 0 || (one_chunk && !quick)) chunk_bounds.clear();
  }
  chunked = !chunk_bounds.empty();
  // KU_DEVICES with -x (more chunks than one): the first GPU runs the out-of-core pipeline, the others are HELPERS -- each
  // streams its share of the chunks (chunk c on GPU c mod N) over its own copies of the resident batches; the slots they
  // collect are folded into the first GPU's batches before the finish, their per-taxon state at the end of the run
  std::vector<int> devices;
  if (const char *dl = getenv("KU_DEVICES")) {
    for (const char *p = dl; *p;) {
      char *end = nullptr;
      const long d = strtol(p, &end, 10);
      if (end == p || d < 0) die(EX_USAGE, "can't parse KU_DEVICES=%s", dl);
      devices.push_back((int)d);
      p = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') die(EX_USAGE, "can't parse KU_DEVICES=%s", dl);
    }
  }
  if (devices.size() > 1 && !chunked) {
    const char *mode = getenv("KU_MGPU_MODE");
    uint32_t mflags = (mode && strcmp(mode, "replicas") == 0) ? KU_MGPU_REPLICAS : 0u;
    if (db_handles.size() > 1 && !mflags) {
      // "the first database that holds the k-mer wins" (src/classify.cpp:928-936) needs every database whole on a rank
      fprintf(stderr, "classify: several databases on several GPUs: every GPU holds all of them (replicas), the reads are split\n");
      mflags = KU_MGPU_REPLICAS;
    }
    if (exact && quick) die(EX_SOFTWARE, "exact counting in quick mode runs on one GPU (the shards' foreign-mark pass needs whole reads)");
    if (exact && mflags) die(EX_SOFTWARE, "exact counting on several GPUs needs the database sharded by minimizer range (not KU_MGPU_MODE=replicas / several databases)");
    KU_CHECK(ku_mgpu_create(devices.data(), (uint32_t)devices.size(), 0, (uint32_t)devices.size(), nullptr, mflags, &mg));
    fprintf(stderr, "Running on %zu GPU ranks (%s, %s exchange)\n", devices.size(), mflags ? "replicas" : "database sharded by minimizer range",
            ku_mgpu_uses_rccl(mg) ? "RCCL" : "same-process");
    ctx = ku_mgpu_ctx(mg, 0);
  } else {
    const char *dev_env = getenv("KU_DEVICE");
    KU_CHECK(ku_ctx_create(!devices.empty() ? devices[0] : (dev_env ? atoi(dev_env) : 0), &ctx));
    for (size_t r = 1; r < devices.size(); ++r) {  // (only with -x chunks, see above)
      ku_ctx *h = nullptr;
      KU_CHECK(ku_ctx_create(devices[r], &h));
      helpers.push_back(h);
    }
    if (!helpers.empty()) fprintf(stderr, "Running on %zu GPUs: the database chunks of the out-of-core run are dealt out among them\n", devices.size());
  }
  // The batch buffers of the host pipeline (below) are page-locked memory, which is slow to allocate (a few hundred MB
  // take longer than classifying the first millions of reads).  A helper sizes the pool's buffers for plain-text
  // regions while this thread loads the database; it is joined before the first read is looked at.
  // (round 5: up to 12 parsers -- with the device stage out of the way the reader is the longest stage: 8 -> 12 members took the
  // 10 M-read window from 0.17 to 0.15 s, 16 bought nothing more; profiles/r05_e2e_sweep.log)
  const int team_cap = getenv("KU_PARSE_TEAM") ? std::max(1, atoi(getenv("KU_PARSE_TEAM"))) : 12;
  parse_team = paired ? 1 : (fmt_threads < team_cap ? fmt_threads : team_cap);
  const int n_batches = 7 + (parse_team > 1 ? parse_team : 0);  // one per team member + up to four on the device, formatter, writer and one queued
  ku_seqio::PinSwitch::enabled = !chunk_bytes;  // per-read arrays of the batches page-locked too (before any batch exists)
  pool.resize(n_batches);
  std::thread pool_setup([&] {
    if (chunk_bytes) return;  // -x runs allocate a batch per region (plain memory), the pool stays empty
    const size_t seq_bytes = (size_t)((double)(unit_nt / 4) * 1.2) + 8192;  // a region's sequences (see region_bytes below)
    const size_t reads = seq_bytes / 100 + 1024;                           // per-read arrays: grow on demand for shorter reads
    for (auto &bt : pool) {
      bt.reserve_seq(seq_bytes);
      bt.reserve_runs(seq_bytes / 32);  // ~ 3 runs per 100 bases; grows on demand
      bt.off.reserve(reads); bt.len.reserve(reads); bt.calls.reserve(reads); bt.hits.reserve(reads);
      bt.run_off.reserve(reads); bt.run_cnt.reserve(reads);
    }
  });
  if (map_uids && (mg || chunked)) die(EX_SOFTWARE, "UID mapping (-I) runs on one GPU with the database resident (no KU_DEVICES, no -x chunks)");
  // database.kdb.counts of a chunked run is summed up chunk by chunk while each one is resident
  auto counts_file_good = [](const std::string &name, bool say) {
    bool good = false;
    if (FILE *cf = fopen(name.c_str(), "r")) {
      good = fgetc(cf) != EOF;
      fclose(cf);
      if (!good && say) fprintf(stderr, "Kmer counts file is empty - trying to regenerate ...\n");
    }
    return good;
  };
  const bool want_report = !report_out.empty() && report_out != "off";
  const bool sum_chunk_counts = chunked && want_report && !counts_file_good(dbs[0] + ".counts", false);
  std::map<uint32_t, uint64_t> chunk_counts;
  std::mutex chunk_counts_mu;
  add_chunk_counts = [&](ku_ctx *c) {  // the chunk that is resident on c
    if (!sum_chunk_counts) return;
    uint64_t nc = 0;
    KU_CHECK(ku_ctx_count_taxons(c, nullptr, nullptr, &nc));
    std::vector<uint32_t> ct(nc + 1); std::vector<uint64_t> cc(nc + 1);
    uint64_t cap = nc;
    KU_CHECK(ku_ctx_count_taxons(c, ct.data(), cc.data(), &cap));
    std::lock_guard<std::mutex> l(chunk_counts_mu);
    for (uint64_t i = 0; i < cap; ++i) chunk_counts[ct[i]] += cc[i];
  };
  // chunk c belongs to GPU c mod N (N = 1 + helpers): rank_chunks[r] in ascending order; rank 0 starts with chunk 0
  n_ranks_x = 1 + helpers.size();
  rank_chunks.assign(n_ranks_x, std::vector<size_t>());
  if (chunked)
    for (size_t c = 0; c + 1 < chunk_bounds.size(); ++c) rank_chunks[c % n_ranks_x].push_back(c);
  if (chunked) {
    if (db_handles.size() > 1) die(EX_SOFTWARE, "-x with several databases is not supported (the reference only searches the first one there)");
    fprintf(stderr, "Streaming the database through the GPU in %zu chunks of at most %" PRIu64 " bytes\n", chunk_bounds.size() - 1, chunk_bytes);
    uint64_t nv = 0;
    KU_CHECK(ku_db_values(db, nullptr, &nv));
    std::vector<uint32_t> values(nv + 1);
    uint64_t cap = nv;
    KU_CHECK(ku_db_values(db, values.data(), &cap));
    KU_CHECK(ku_ctx_load_db(ctx, db, chunk_bounds[0], chunk_bounds[1]));
    KU_CHECK(ku_ctx_set_taxonomy(ctx, tax, values.data(), cap));
    add_chunk_counts(ctx);
    for (size_t r = 1; r < n_ranks_x; ++r) {  // the helpers: their first chunk, the slot table of the whole database as everywhere
      if (rank_chunks[r].empty()) continue;
      const size_t c0 = rank_chunks[r][0];
      KU_CHECK(ku_ctx_load_db(helpers[r - 1], db, chunk_bounds[c0], chunk_bounds[c0 + 1]));
      KU_CHECK(ku_ctx_set_taxonomy(helpers[r - 1], tax, values.data(), cap));
      add_chunk_counts(helpers[r - 1]);
    }
  } else if (mg) {
    KU_CHECK(ku_mgpu_load_dbs(mg, db_handles.data(), (uint32_t)db_handles.size(), tax));
  } else {
    KU_CHECK(ku_ctx_load_db(ctx, db, 0, info.n_bins));
    for (size_t i = 1; i < db_handles.size(); ++i) KU_CHECK(ku_ctx_add_db(ctx, db_handles[i]));
    KU_CHECK(ku_ctx_set_taxonomy(ctx, tax, nullptr, 0));
  }
  // HLL sparse-mode emulation (single GPU): the report's `kmers` as the reference prints them -- so only when a report
  // was asked for; the emulation keeps every distinct k-mer of the taxa whose sketches stay sparse and is by far the most
  // expensive part of a run with many low-abundance taxa.  -x runs insert into the global sketches directly
  // (src/classify.cpp:719): one unit for the whole run.
  sparse = want_report && hll_precision > 0 && !exact && !getenv("KU_NO_SPARSE");  // (-p 0: no k-mer columns, no sketches needed)
  if (sparse) {
    const char *e = getenv("KU_SPARSE_LOG2");
    uint32_t g_log2 = e ? (uint32_t)atoi(e) : 0u;
    // (Rounds 3-4 sized the run-wide (slot, encoding) set from the input files here, up to 16 GB: every k-mer went into it.
    // Since round 5 the k-mers the database holds are marked in the probe table itself and the set only takes the misses of
    // the first work units: the default of 2^26 cells, which grows on demand, does.)
    int st = mg ? ku_mgpu_enable_sparse(mg, one_chunk ? 0 : work_unit_nt, g_log2)
                : ku_ctx_enable_sparse(ctx, chunked || one_chunk ? 0 : work_unit_nt, g_log2);
    if (st == KU_EUNSUP) { fprintf(stderr, "classify: %s -- the report will carry dense estimates\n", ku_last_error()); sparse = false; }
    else KU_CHECK(st);
  }
  if (exact) {  // 2^30 cells = 8 GiB hold ~750 M distinct k-mers (per GPU); KU_EXACT_LOG2 sizes it for larger runs
    const char *e = getenv("KU_EXACT_LOG2");
    if (mg) KU_CHECK(ku_mgpu_enable_exact(mg, e ? (uint32_t)atoi(e) : 30u));
    else KU_CHECK(ku_ctx_enable_exact(ctx, e ? (uint32_t)atoi(e) : 30u));
  }

  if (!mg && !chunked && !map_uids && !getenv("KU_RLE_ONE_STEP")) {
    // the device-side buffers of the batches in flight, ahead of the timing window (sized like the pool's batches; a batch that
    // is larger makes its own room).  Not fatal: the batch calls allocate on demand.
    const uint64_t b_bytes = (uint64_t)((double)(unit_nt / 4) * 1.2) + 8192;
    // (three in flight take turns through all four sets)
    if (ku_classify_batch_rle_reserve(ctx, b_bytes, b_bytes / 100 + 1024, 400, KU_RLE_MAX_IN_FLIGHT) != KU_OK) fprintf(stderr, "classify: note: %s\n", ku_last_error());
  }
  print_kraken = true;
  if (!kraken_out.empty()) {
    if (kraken_out == "off" || kraken_out == "-") print_kraken = false;
    else {
      fprintf(stderr, "Writing Kraken output to %s\n", kraken_out.c_str());
      if (!s_kraken.open(kraken_out, false, /*team=*/true)) die(EX_OSERR, "can't open %s", kraken_out.c_str());
    }
  } else s_kraken.open("-");
  if (print_cls && !s_cls.open(cls_out)) die(EX_OSERR, "can't open %s", cls_out.c_str());
  if (print_ucls && !s_ucls.open(ucls_out)) die(EX_OSERR, "can't open %s", ucls_out.c_str());
  if (print_kraken) fmt_team.start(fmt_threads);  // (ahead of the timing window, like the batch pool)
  pool_setup.join();
  timeval tv1, tv2;
  gettimeofday(&tv1, nullptr);
  double cpu_sys0 = 0;
  const double cpu_user0 = process_cpu_s(&cpu_sys0), cpu_device0 = thread_cpu_s();
  base_opts = ku_opts{quick ? KU_F_QUICK : 0u, min_hits, 0, 0};
  pflags = (only_classified ? KU_P_ONLY_CLASSIFIED : 0u) | (print_seq ? KU_P_SEQUENCE : 0u) | (quick ? KU_P_QUICK : 0u);

  // Three-stage host pipeline: reader thread (FASTA/FASTQ(+gz) -> pinned batch) | this thread
  // (ku_classify_batch_rle: H2D, kernels, run-length encoding, D2H) | writer thread (Kraken lines formatted by `fmt_threads` helpers,
  // files written in input order).  Batches circulate through two bounded queues.
  // a team of parser threads for plain-text inputs (-t, at most 8): every member owns one batch while it parses
  // (team and pool are set up above, next to the database load)
  for (auto &bt : pool) free_q.push(&bt);
  gate.unit_nt = work_unit_nt;
  keep_records = print_cls || print_ucls;
  double cpu_device = 0;  // CPU seconds of this (the device stage's) thread in the window
  // KU_CLI_TRACE=1: a line per batch on stderr behind the run -- when it reached each step (ms from the window's start): region
  // claimed, parsed, handed on in file order, enqueue begins / ends, finished on the device, formatting begins / ends, write begins / ends
  cli_trace = getenv("KU_CLI_TRACE") != nullptr;
  // KU_CLI_STACKS=a-b (ms): every 2 ms of that stretch of the window, where each thread of the process is -- its state and the
  // top of its kernel stack (/proc/self/task/*/stack, root only) -- grouped, printed behind the run.  For stalls that hit every stage at once.
  std::vector<std::string> stack_samples;
  std::atomic<bool> stacks_stop{false};
  std::thread stack_sampler;
  if (const char *e = getenv("KU_CLI_STACKS")) {
    double a = 0, b = 40;
    sscanf(e, "%lf-%lf", &a, &b);
    stack_sampler = std::thread([&, a, b] {
      prctl(PR_SET_NAME, "ku-sampler");
      const double t0 = (double)tv1.tv_sec + (double)tv1.tv_usec / 1e6;
      while (!stacks_stop && (now_s() - t0) * 1e3 < a) usleep(200);
      while (!stacks_stop && (now_s() - t0) * 1e3 < b) {
        const double ts = (now_s() - t0) * 1e3;
        std::map<std::string, int> groups;
        if (DIR *d = opendir("/proc/self/task")) {
          while (dirent *de = readdir(d)) {
            if (de->d_name[0] == '.') continue;
            char path[320], buf[1024];
            std::string key;
            for (const char *what : {"comm", "stat", "syscall", "stack"}) {
              snprintf(path, sizeof path, "/proc/self/task/%s/%s", de->d_name, what);
              FILE *f = fopen(path, "r");
              if (!f) continue;
              const size_t n = fread(buf, 1, sizeof buf - 1, f);
              fclose(f);
              buf[n] = 0;
              if (what[0] == 'c') { key = buf; if (!key.empty() && key.back() == '\n') key.pop_back(); }
              else if (what[1] == 't' && what[2] == 'a' && what[3] == 't') { const char *r = strrchr(buf, ')'); key += r && r[1] ? std::string(" ") + r[2] : " ?"; }
              else if (what[1] == 'y') {  // the system call the thread is in: number, first three arguments (an ioctl: descriptor, request)
                char *sp = buf;
                int fields = 0;
                for (; *sp && fields < 4; ++sp) if (*sp == ' ' || *sp == '\n') { ++fields; if (fields == 4) *sp = 0; }
                if (!key.empty() && key.back() != 'R') key += std::string(" sys ") + buf;
                if (!key.empty() && key.back() == '\n') key.pop_back();
              } else {  // the first four frames, function names only
                int frames = 0;
                for (char *line = strtok(buf, "\n"); line && frames < 4; line = strtok(nullptr, "\n"), ++frames) {
                  const char *fn = strchr(line, ']');
                  std::string name = fn ? fn + 2 : line;
                  const size_t plus = name.find('+');
                  if (plus != std::string::npos) name.resize(plus);
                  key += " < " + name;
                }
              }
            }
            ++groups[key];
          }
          closedir(d);
        }
        char head[64];
        snprintf(head, sizeof head, "stacks: t = %.1f ms\n", ts);
        std::string out = head;
        for (auto &g : groups) out += "stacks:   " + std::to_string(g.second) + " x " + g.first + "\n";
        stack_samples.push_back(out);
        usleep(1500);
      }
    });
  }
  const double trace_t0 = (double)tv1.tv_sec + (double)tv1.tv_usec / 1e6;  // (now_s()'s clock)

  // Three-stage host pipeline: reader (classify_input.cpp) | this thread: the device stage (classify_device.cpp) |
  // formatter + writer (classify_output.cpp).  Batches circulate through the queues of `Run`.
  std::thread reader([this] { reader_stage(); });
  std::thread formatter([this] { formatter_stage(); });
  std::thread writer([this] { writer_stage(); });
  if (chunked) device_stage_chunked();
  else device_stage_resident();
  done_q.push(nullptr);
  cpu_device = thread_cpu_s() - cpu_device0;
  reader.join();
  formatter.join();
  writer.join();
  fmt_team.stop();
  gettimeofday(&tv2, nullptr);
  {  // report_stats (src/classify.cpp:361-375)
    double seconds = seconds_between(tv1, tv2);
    fprintf(stderr, "\r");
    fprintf(stderr, "%llu sequences (%.2f Mbp) processed in %.3fs (%.1f Kseq/m, %.2f Mbp/m).\n", total_sequences,
            total_bases / 1.0e6, seconds, total_sequences / 1.0e3 / (seconds / 60), total_bases / 1.0e6 / (seconds / 60));
    fprintf(stderr, "  %llu sequences classified (%.2f%%)\n", total_classified, total_classified * 100.0 / total_sequences);
    fprintf(stderr, "  %llu sequences unclassified (%.2f%%)\n", total_sequences - total_classified,
            (total_sequences - total_classified) * 100.0 / total_sequences);
  }
  s_kraken.close(); s_cls.close(); s_ucls.close();
  stacks_stop = true;
  if (stack_sampler.joinable()) stack_sampler.join();
  for (auto &o : stack_samples) fputs(o.c_str(), stderr);
  if (cli_trace) {
    fprintf(stderr, "trace: window ends at %.2f ms\n", ((double)tv2.tv_sec + (double)tv2.tv_usec / 1e6 - trace_t0) * 1e3);
    fprintf(stderr, "trace: batch claimed parsed handed enq0 enq1 finished fmt0 fmt1 wr0 wr1 (ms)\n");
    for (size_t i = 0; i < trace_rows.size(); ++i) {
      fprintf(stderr, "trace: %zu", i);
      for (double v : trace_rows[i]) fprintf(stderr, " %.2f", v > 0 ? (v - trace_t0) * 1e3 : -1.0);
      fprintf(stderr, "\n");
    }
  }
  if (getenv("KU_CLI_TIMES")) {
    fprintf(stderr, "stage busy seconds: reader %.3f, device %.3f, writer %.3f (formatting %.3f + writing %.3f; device: batch call %.3f + runs back %.3f)\n",
            busy_reader, busy_gpu, busy_format + busy_writer, busy_format, busy_writer, busy_gpu_classify, busy_gpu_fetch);
    double sys1 = 0;
    const double user1 = process_cpu_s(&sys1);
    fprintf(stderr, "cpu seconds in the window: user %.2f + sys %.2f in all; parser team %.2f, formatting helpers %.2f, writer %.2f, device thread %.2f\n",
            user1 - cpu_user0, sys1 - cpu_sys0, cpu_parse, cpu_format, cpu_write, cpu_device);
    // (others: the reader that hands the batches on in file order, the device thread, the formatter's dispatcher, the writer; a .gz /
    // .bz2 input adds its inflating team, sized from the usable CPUs: ku_seqio.h)
    fprintf(stderr, "threads in the window: parser team %d, formatting helpers %d, others %d; usable CPUs %d\n", parse_team, print_kraken ? fmt_threads : 0, 4,
            ku_seqio::usable_cpus());
  }

  if (!report_out.empty() && report_out != "off") {
    gettimeofday(&tv1, nullptr);
    fprintf(stderr, "Writing report file to %s  ..\n", report_out.c_str());
    // database.kdb.counts, one per database: regenerate when missing or empty (src/classify.cpp:263-285)
    std::vector<std::string> cnames;
    for (size_t di = 0; di < dbs.size(); ++di) {
      const std::string cname = dbs[di] + ".counts";
      cnames.push_back(cname);
      const bool good = counts_file_good(cname, true);
      if (!good && chunked) {
        fprintf(stderr, "Writing kmer counts to %s... [only once for this database, may take a while] \n", cname.c_str());
        FILE *cf = fopen(cname.c_str(), "w");
        if (!cf) die(EX_OSERR, "can't write %s", cname.c_str());
        for (const auto &kv : chunk_counts) fprintf(cf, "%u\t%" PRIu64 "\n", kv.first, kv.second);
        fclose(cf);
      } else if (!good) {
        fprintf(stderr, "Writing kmer counts to %s... [only once for this database, may take a while] \n", cname.c_str());
        uint64_t nc = 0;
        const bool group_counts = mg && dbs.size() == 1;  // shards: summed over the ranks; several databases: replicas, rank 0 holds them all
        if (group_counts) KU_CHECK(ku_mgpu_count_taxons(mg, nullptr, nullptr, &nc));
        else KU_CHECK(ku_ctx_count_taxons_db(ctx, (uint32_t)di, nullptr, nullptr, &nc));
        std::vector<uint32_t> ct(nc + 1); std::vector<uint64_t> cc(nc + 1);
        uint64_t cap = nc;
        if (group_counts) KU_CHECK(ku_mgpu_count_taxons(mg, ct.data(), cc.data(), &cap));
        else KU_CHECK(ku_ctx_count_taxons_db(ctx, (uint32_t)di, ct.data(), cc.data(), &cap));
        FILE *cf = fopen(cname.c_str(), "w");
        if (!cf) die(EX_OSERR, "can't write %s", cname.c_str());
        for (uint64_t i = 0; i < cap; ++i) fprintf(cf, "%u\t%" PRIu64 "\n", ct[i], cc[i]);
        fclose(cf);
      }
    }
    std::vector<const char *> cpaths;
    for (const std::string &c : cnames) cpaths.push_back(c.c_str());
    const bool sparse_gave_up = sparse && (mg ? ku_mgpu_sparse_state(mg) : ku_ctx_sparse_state(ctx)) == 2;
    if (mg) KU_CHECK(ku_mgpu_reduce_state(mg, nullptr));  // every rank's registers / counters / sparse sets into rank 0's context
    if (sparse_gave_up)
      fprintf(stderr, "classify: warning: the sparse-sketch emulation ran out of device memory during the run -- the report's kmers / dup / "
                      "cov columns are dense-register estimates (within 3 sigma = 4.9 %% of the reference's)\n");
    // clade roll-up on the device, from the registers / counters / sparse sets where they lie (ku_ctx_report); in a
    // group rank 0's context holds the reduced state
    char *text = nullptr; size_t tn = 0;
    KU_CHECK(ku_ctx_report_cols(ctx, tax, cpaths.data(), (uint32_t)cpaths.size(), hll_precision > 0 ? 0u : KU_R_NO_KMER_COLS, &text, &tn));
    if (tn == 0) fprintf(stderr, "total number of reads is zero - not creating a report!\n");
    Sink rs;
    if (!rs.open(report_out, /*append=*/true)) die(EX_OSERR, "can't open %s", report_out.c_str());
    if (sparse_gave_up && tn) {  // the file says so too (a comment line, where the wrapper's "# CL:" header lines are)
      static const char note[] = "# NOTE: kmers / dup / cov are dense HyperLogLog estimates (the sparse-sketch emulation ran out of device memory)\n";
      rs.write(note, sizeof note - 1);
    }
    rs.write(text, tn);
    rs.close();
    ku_free(text);
    gettimeofday(&tv2, nullptr);
    fprintf(stderr, "Report finished in %.3f seconds.\n", seconds_between(tv1, tv2));
  }
  fprintf(stderr, "Finishing up ...\n");
  // The pool's page-locked buffers and the input mappings go back HERE, behind the window and the report: a hundred hipHostFree
  // calls took 45 ms of the window, the unmapping 30 ms -- giving memory back is no part of classifying (the reference's window
  // ends behind its last work unit as well, classify.cpp:248-258) -- and with 2 GB of page-locked memory being released right
  // in front of it, the report's one large device allocation took 1-3 s in two of thirty runs (2 ms otherwise).
  for (auto &m : input_maps) munmap(m.first, m.second);
  for (auto &bt : pool) bt.release();
  if (mg) ku_mgpu_destroy(mg);
  else ku_ctx_destroy(ctx);
  for (ku_ctx *h : helpers) ku_ctx_destroy(h);
  ku_tax_close(tax);
  ku_uid_map_close(uid_map);
  for (ku_db *h : db_handles) ku_db_close(h);
  return 0;
}

