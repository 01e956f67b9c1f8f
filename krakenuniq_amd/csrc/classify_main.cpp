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
#include <fcntl.h>
#include <malloc.h>
#include <getopt.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <dirent.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sysexits.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/krakenuniq_amd.h"
#include "ku_seqio.h"
#include "ku_pgzout.h"

// Fatal errors are raised by whichever thread meets them (the reader finds a damaged input while the main thread still
// loads the database): exit() would run the static destructors -- the HIP runtime's among them -- under the feet of the
// other threads (a truncated .bz2 file ended in SIGSEGV instead of EX_DATAERR).  Flush what is buffered and leave.
[[noreturn]] static void leave(int code) {
  fflush(nullptr);
  _exit(code);
}
static void die(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3), noreturn));
static void die(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "classify: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  leave(code);
}
void ku_seqio::fatal(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "classify: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  leave(code);
}

static int exit_code_of(int st) {
  switch (st) {
    case KU_EINVAL: return EX_USAGE;
    case KU_EDATA: return EX_DATAERR;
    case KU_ENOINPUT: return EX_NOINPUT;
    case KU_ENOMEM: return EX_OSERR;
    default: return EX_SOFTWARE;
  }
}
#define KU_CHECK(call)                                                              \
  do {                                                                              \
    int st_ = (call);                                                               \
    if (st_ != KU_OK) die(exit_code_of(st_), "%s: %s", ku_strerror(st_), ku_last_error()); \
  } while (0)

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

// ---- output sink: plain file, stdout, or gzip when the name ends in .gz (src/classify.cpp:133-148)
struct Sink {
  FILE *f = nullptr;
  gzFile g = nullptr;
  ku_pgzout::Member pg;  // `team`: a .gz file whose parts arrive deflated (ku_pgzout.h: the formatting helpers compress what they
  bool pgz = false;      // formatted; ogzstream's one deflate on the writing thread would be twenty times slower than the pipeline)
  bool open(const std::string &name, bool append = false, bool team = false) {
    if (name == "-") { f = stdout; return true; }
    if (name.size() > 3 && name.compare(name.size() - 3, 3, ".gz") == 0) {
      if (team && !getenv("KU_NO_PGZOUT")) { pgz = pg.open(name.c_str()); return pgz; }
      g = gzopen(name.c_str(), "wb");
      return g != nullptr;
    }
    f = fopen(name.c_str(), append ? "a" : "w");
    return f != nullptr;
  }
  void write(const char *p, size_t n) {
    if (!n) return;
    if (pgz) {  // (text for a team-written file: deflated here)
      size_t cl = 0;
      uLong crc = 0;
      unsigned char *c = ku_pgzout::deflate_part(p, n, &cl, &crc);
      if (!c) die(EX_OSERR, "gz write error");
      write_deflated(c, cl, crc, n);
      free(c);
    } else if (g) { if (gzwrite(g, p, (unsigned)n) <= 0) die(EX_OSERR, "gz write error"); }
    else if (f && fwrite(p, 1, n, f) != n) die(EX_OSERR, "write error: %s", strerror(errno));
  }
  void write_deflated(const unsigned char *c, size_t clen, uLong crc, size_t raw_len) {
    if (!pg.put(c, clen, crc, raw_len)) die(EX_OSERR, "write error: %s", strerror(errno));
  }
  void close() {
    if (pgz && !pg.close()) die(EX_OSERR, "write error: %s", strerror(errno));
    pgz = false;
    if (g) gzclose(g);
    if (f && f != stdout) fclose(f);
    if (f == stdout) fflush(stdout);
    f = nullptr; g = nullptr;
  }
};

using ku_seqio::Batch;
using ku_seqio::Reader;

struct Queue {  // unbounded MPSC-ish queue; the number of Batch objects bounds what is in flight
  std::mutex m;
  std::condition_variable cv;
  std::deque<Batch *> q;
  void push(Batch *b) { { std::lock_guard<std::mutex> l(m); q.push_back(b); } cv.notify_one(); }
  Batch *pop() {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return !q.empty(); });
    Batch *b = q.front();
    q.pop_front();
    return b;
  }
  bool try_pop(Batch **b) {  // false: nothing queued right now
    std::lock_guard<std::mutex> l(m);
    if (q.empty()) return false;
    *b = q.front();
    q.pop_front();
    return true;
  }
};

static double now_s() {
  timeval t;
  gettimeofday(&t, nullptr);
  return (double)t.tv_sec + (double)t.tv_usec / 1e6;
}
// CPU seconds the calling thread has used (KU_CLI_TIMES: where the cores of a quota-limited host go)
static double thread_cpu_s() {
  timespec t;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static double process_cpu_s(double *sys_s) {
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  if (sys_s) *sys_s = (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec;
  return (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec;
}
static double seconds_between(const timeval &a, const timeval &b) {
  return (double)(b.tv_sec - a.tv_sec) + (double)(b.tv_usec - a.tv_usec) / 1e6;
}

// KU_CRLF_REFERENCE=1: the Kraken lines of the reads that had carriage returns inside (multi-line FASTA with CRLF line ends)
// as the reference prints them (ku_seqio.h, crlf_note()).  `text` holds the lines of the reads [lo, hi) of `bt` as
// ku_format_kraken_rle wrote them, one per read; the lines of the listed reads are replaced.  Returns a malloc'ed buffer.
static char *rewrite_crlf_lines(const Batch &bt, uint64_t lo, uint64_t hi, uint32_t k, char *text, size_t *len) {
  size_t a = std::lower_bound(bt.crlf_read.begin(), bt.crlf_read.end(), (uint32_t)lo) - bt.crlf_read.begin();
  const size_t b = std::lower_bound(bt.crlf_read.begin(), bt.crlf_read.end(), (uint32_t)hi) - bt.crlf_read.begin();
  if (a == b) return text;
  std::string out;
  out.reserve(*len + 64);
  const char *p = text, *end = text + *len;
  for (uint64_t r = lo; r < hi && p < end; ++r) {
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    const char *line_end = nl ? nl + 1 : end;
    if (a < b && bt.crlf_read[a] == r) {
      // the line's five columns: C/U, id, call, length, hit list
      std::vector<std::string> col;
      const char *q = p;
      for (int c = 0; c < 4 && q < line_end; ++c) {
        const char *t = (const char *)memchr(q, '\t', (size_t)(line_end - q));
        if (!t) break;
        col.emplace_back(q, t);
        q = t + 1;
      }
      std::string hits(q, line_end - (nl ? 1 : 0));
      const uint32_t s0 = bt.crlf_off[a], s1 = a + 1 < bt.crlf_off.size() ? bt.crlf_off[a + 1] : (uint32_t)bt.crlf_start.size();
      const uint32_t L1 = bt.len[r];           // bases + the carriage return that closes the record
      const uint32_t L = L1 ? L1 - 1 : 0;      // bases
      if (col.size() == 4 && L1 >= k && bt.seqs[bt.off[r] + L1 - 1] == '\r') {
        // per-k-mer codes of the joined sequence (L1 - k + 1 of them, the last one holds the '\r')
        std::vector<std::string> codes;
        codes.reserve(L1 - k + 1);
        for (size_t i = 0; i < hits.size();) {
          size_t sp = hits.find(' ', i);
          if (sp == std::string::npos) sp = hits.size();
          const size_t colon = hits.find(':', i);
          if (colon != std::string::npos && colon < sp) {
            const std::string code = hits.substr(i, colon - i);
            const unsigned long cnt = strtoul(hits.c_str() + colon + 1, nullptr, 10);
            for (unsigned long j = 0; j < cnt; ++j) codes.push_back(code);
          }
          i = sp + 1;
        }
        if (codes.size() == (size_t)L1 - k + 1) {
          std::vector<const std::string *> kept;
          uint32_t nb = 0, si = s0;
          for (uint32_t t = 0; t < L; ++t) {
            if (si < s1 && bt.crlf_start[si] == t) { ++nb; ++si; continue; }  // the first base behind a line break is not counted
            if (t + 1 - nb >= k) kept.push_back(&codes[t - k + 1]);
          }
          kept.push_back(&codes.back());  // the scanner's last, ambiguous k-mer behind the closing '\r'
          std::string h;
          for (size_t i = 0; i < kept.size();) {
            size_t j = i;
            while (j < kept.size() && *kept[j] == *kept[i]) ++j;
            if (!h.empty()) h += ' ';
            h += *kept[i];
            h += ':';
            h += std::to_string(j - i);
            i = j;
          }
          out += col[0]; out += '\t'; out += col[1]; out += '\t'; out += col[2]; out += '\t';
          out += std::to_string(L + (s1 - s0) + 1);  // every line's '\r' counts (taxdb / classify.cpp print dna.seq.size())
          out += '\t'; out += h; out += '\n';
          ++a;
          p = line_end;
          continue;
        }
      }
      ++a;  // (not the shape this emulation knows: the line stays)
    }
    out.append(p, line_end);
    p = line_end;
  }
  out.append(p, end);
  char *nb = (char *)malloc(out.size() + 1);
  if (!nb) return text;
  memcpy(nb, out.data(), out.size());
  ku_free(text);
  *len = out.size();
  return nb;
}

int main(int argc, char **argv) {
  std::vector<std::string> dbs, idxs;
  std::string kraken_out, report_out, taxdb, cls_out, ucls_out, uid_map_file;
  bool paired = false, warned_pairs = false, warned_uid_calls = false;
  bool quick = false, only_classified = false, print_seq = false, print_cls = false, print_ucls = false, populate = false;
  uint32_t min_hits = 1;
  uint64_t unit_nt = 64ull << 20;    // GPU batch size in nt (KU_BATCH_NT; round 5: 64 Mi = regions of 16 Mi nt, one launch of ~120 k reads each -- with the
                                     // batch call in two steps the window is flat from 40 to 96 Mi and the kernel's cost per read falls with the launch size); plain and .gz files travel in regions of a quarter of it.
                                     // 10 M x 150 bp end to end (scripts/e2e_sweep.py, profiles/r04_e2e_sweep.log): 128 Mi 0.53 s, 64 Mi 0.44,
                                     // 32 Mi 0.29, 24 Mi 0.31, 16 Mi 0.36, 8 Mi 0.46 -- larger batches fill and drain the three stages slowly,
                                     // smaller ones pay the device stage's ~1 ms per call too often
  uint64_t work_unit_nt = 500000;   // -u: the reference's Work_unit_size (src/classify.cpp:38)
  uint64_t chunk_bytes = 0;  // -x SIZE: stream the database through HBM in chunks of at most SIZE bytes
  int hll_precision = 1;  // -p: only its sign matters (six or nine report columns)
  int fmt_threads = 4;  // -t: host threads that format the Kraken lines (the GPU replaces the OpenMP team)
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
  const bool map_uids = !uid_map_file.empty();
  if (map_uids && dbs.size() > 1) { fprintf(stderr, "Cannot use more than one database with UID mapping!\n"); return 1; }  // src/classify.cpp:158-160
  if (map_uids && quick) { fprintf(stderr, "Quick mode not available when mapping UIDs\n"); return 1; }                  // :954-956
  if (optind == argc && !populate) fprintf(stderr, "No sequence data files specified\n");
  if (paired && (argc - optind) % 2) die(EX_USAGE, "-P needs the input files in pairs (mate 1, mate 2)");
  if (taxdb.empty()) { fprintf(stderr, "TaxDB argument is required!\n"); return 1; }  // src/classify.cpp:221-222

  // hierarchical run: the databases are searched in command-line order (src/classify.cpp:163-177,928-936)
  std::vector<ku_db *> db_handles(dbs.size(), nullptr);
  ku_db_info info;
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
  ku_db *db = db_handles[0];
  ku_tax *tax = nullptr;
  KU_CHECK(ku_tax_open(taxdb.c_str(), &tax));
  ku_uid_map *uid_map = nullptr;
  if (map_uids) {
    fprintf(stderr, "Reading UID mapping file %s\n", uid_map_file.c_str());  // src/classify.cpp:163
    KU_CHECK(ku_uid_map_open(uid_map_file.c_str(), &uid_map));
  }
  // -x SIZE (src/krakendb.cpp:463-522): the chunk plan of the reference; one chunk = everything resident as usual
  std::vector<uint64_t> chunk_bounds;
  if (chunk_bytes) {
    chunk_bounds.resize(info.n_bins + 2 < (1u << 20) ? info.n_bins + 2 : (1u << 20));
    uint32_t n_chunks = 0;
    KU_CHECK(ku_db_chunk_plan(db, chunk_bytes, chunk_bounds.data(), (uint32_t)chunk_bounds.size() - 1, &n_chunks));
    chunk_bounds.resize(n_chunks + 1);
    chunk_bounds.back() = info.n_bins;  // the bins behind the last chunk hold no pairs
    if (n_chunks <= 1) chunk_bounds.clear();
  }
  const bool chunked = !chunk_bounds.empty();
  ku_ctx *ctx = nullptr;
  ku_mgpu *mg = nullptr;  // KU_DEVICES=0,1,...: several GPUs through the multi-GPU driver
  // KU_DEVICES with -x (more chunks than one): the first GPU runs the out-of-core pipeline, the others are HELPERS -- each
  // streams its share of the chunks (chunk c on GPU c mod N) over its own copies of the resident batches; the slots they
  // collect are folded into the first GPU's batches before the finish, their per-taxon state at the end of the run
  std::vector<ku_ctx *> helpers;
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
  const int parse_team = paired ? 1 : (fmt_threads < team_cap ? fmt_threads : team_cap);
  const int n_batches = 7 + (parse_team > 1 ? parse_team : 0);  // one per team member + up to four on the device, formatter, writer and one queued
  ku_seqio::PinSwitch::enabled = !chunk_bytes;  // per-read arrays of the batches page-locked too (before any batch exists)
  std::vector<Batch> pool(n_batches);
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
  auto add_chunk_counts = [&](ku_ctx *c) {  // the chunk that is resident on c
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
  const size_t n_ranks_x = 1 + helpers.size();
  std::vector<std::vector<size_t>> rank_chunks(n_ranks_x);
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
  bool sparse = want_report && hll_precision > 0 && !exact && !getenv("KU_NO_SPARSE");  // (-p 0: no k-mer columns, no sketches needed)
  if (sparse) {
    const char *e = getenv("KU_SPARSE_LOG2");
    uint32_t g_log2 = e ? (uint32_t)atoi(e) : 0u;
    // (Rounds 3-4 sized the run-wide (slot, encoding) set from the input files here, up to 16 GB: every k-mer went into it.
    // Since round 5 the k-mers the database holds are marked in the probe table itself and the set only takes the misses of
    // the first work units: the default of 2^26 cells, which grows on demand, does.)
    int st = mg ? ku_mgpu_enable_sparse(mg, work_unit_nt, g_log2)
                : ku_ctx_enable_sparse(ctx, chunked ? 0 : work_unit_nt, g_log2);
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
  Sink s_kraken, s_cls, s_ucls;
  bool print_kraken = true;
  if (!kraken_out.empty()) {
    if (kraken_out == "off" || kraken_out == "-") print_kraken = false;
    else {
      fprintf(stderr, "Writing Kraken output to %s\n", kraken_out.c_str());
      if (!s_kraken.open(kraken_out, false, /*team=*/true)) die(EX_OSERR, "can't open %s", kraken_out.c_str());
    }
  } else s_kraken.open("-");
  if (print_cls && !s_cls.open(cls_out)) die(EX_OSERR, "can't open %s", cls_out.c_str());
  if (print_ucls && !s_ucls.open(ucls_out)) die(EX_OSERR, "can't open %s", ucls_out.c_str());

  struct FmtTeam {  // the helpers: they take tasks -- a slice of a batch each -- from one queue, across batches (a team that ran
                    // one batch at a time behind a barrier was busy 0.63 ms of every 0.93: the batch waited for its slowest slice)
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv;
    std::deque<std::function<void()>> tasks; bool quit = false;
    void start(int n) {
      for (int t = 0; t < n; ++t) th.emplace_back([this] {
        prctl(PR_SET_NAME, "ku-fmt");
        // (the member's part of the heap, touched once: its first buffers for formatted lines -- ~1 MB each -- then come without
        // page faults; sixteen members faulting 12 MB in while the parser team maps the input made the first batch's
        // formatting take 6-19 ms instead of 1.4)
        if (void *w = malloc((size_t)3 << 20)) { memset(w, 1, (size_t)3 << 20); free(w); }
        for (;;) {
          std::function<void()> task;
          {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return quit || !tasks.empty(); });
            if (tasks.empty()) return;  // (quit, and nothing left)
            task = std::move(tasks.front());
            tasks.pop_front();
          }
          task();
        }
      });
    }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> l(m); tasks.push_back(std::move(f)); } cv.notify_one(); }
    void stop() { { std::lock_guard<std::mutex> l(m); quit = true; } cv.notify_all(); for (auto &x : th) x.join(); th.clear(); }
  } fmt_team;
  if (print_kraken) fmt_team.start(fmt_threads);  // (ahead of the timing window, like the batch pool)
  unsigned long long total_sequences = 0, total_classified = 0, total_bases = 0;
  pool_setup.join();
  timeval tv1, tv2;
  gettimeofday(&tv1, nullptr);
  double cpu_sys0 = 0;
  const double cpu_user0 = process_cpu_s(&cpu_sys0), cpu_device0 = thread_cpu_s();
  const ku_opts base_opts = {quick ? KU_F_QUICK : 0u, min_hits, 0, 0};
  const uint32_t pflags = (only_classified ? KU_P_ONLY_CLASSIFIED : 0u) | (print_seq ? KU_P_SEQUENCE : 0u) | (quick ? KU_P_QUICK : 0u);

  // Three-stage host pipeline: reader thread (FASTA/FASTQ(+gz) -> pinned batch) | this thread
  // (ku_classify_batch_rle: H2D, kernels, run-length encoding, D2H) | writer thread (Kraken lines formatted by `fmt_threads` helpers,
  // files written in input order).  Batches circulate through two bounded queues.
  // a team of parser threads for plain-text inputs (-t, at most 8): every member owns one batch while it parses
  // (team and pool are set up above, next to the database load)
  Queue free_q, parsed_q, done_q;
  for (auto &bt : pool) free_q.push(&bt);
  ku_seqio::UnitGate gate;  // (reader thread only)
  gate.unit_nt = work_unit_nt;
  const bool keep_records = print_cls || print_ucls;
  // -x runs allocate a batch per region instead of recycling a pool: the nucleotides between reader and writer are
  // bounded instead (set once the device budget is known)
  uint64_t chunk_budget_nt = ~0ull, inflight_nt = 0;
  std::mutex inflight_mu;
  std::condition_variable inflight_cv;
  auto inflight_add = [&](uint64_t nt) {
    std::unique_lock<std::mutex> l(inflight_mu);
    inflight_cv.wait(l, [&] { return inflight_nt == 0 || inflight_nt + nt <= 2 * chunk_budget_nt; });
    inflight_nt += nt;
  };
  auto inflight_sub = [&](uint64_t nt) {
    { std::lock_guard<std::mutex> l(inflight_mu); inflight_nt -= nt; }
    inflight_cv.notify_all();
  };
  double busy_reader = 0, busy_gpu = 0, busy_writer = 0, busy_format = 0;  // seconds each pipeline stage spent working (KU_CLI_TIMES)
  double busy_gpu_classify = 0, busy_gpu_fetch = 0;                       // ... of the device stage: the batch call, the runs' copy back
  std::mutex cpu_mu;
  double cpu_parse = 0, cpu_format = 0, cpu_write = 0, cpu_device = 0;    // CPU seconds of the stages' threads (KU_CLI_TIMES)
  // KU_CLI_TRACE=1: a line per batch on stderr behind the run -- when it reached each step (ms from the window's start): region
  // claimed, parsed, handed on in file order, enqueue begins / ends, finished on the device, formatting begins / ends, write begins / ends
  std::vector<std::pair<void *, size_t>> input_maps;  // mappings of the input files the parser team read from (reader thread; unmapped behind the window)
  const bool cli_trace = getenv("KU_CLI_TRACE") != nullptr;
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
  std::vector<std::vector<double>> trace_rows;
  const double trace_t0 = (double)tv1.tv_sec + (double)tv1.tv_usec / 1e6;  // (now_s()'s clock)
  auto cpu_add = [&](double &acc, double t0) { const double d = thread_cpu_s() - t0; std::lock_guard<std::mutex> l(cpu_mu); acc += d; };

  // Regular files, plain or .gz: the text is cut into record-aligned regions of about a quarter work unit and parsed by
  // `parse_team` threads, each into its own batch; the batches go on in file order.  A plain file is mapped; a .gz file
  // (BGZF or one gzip stream, ku_pgzip.h) is inflated by its own team into text that grows while it is parsed
  // (ku_seqio::GrowingText) -- the single reader below managed 5 M reads/s of it, with zlib's one inflate 1.7.  A member
  // takes a batch BEFORE it takes a region number, so the lowest outstanding region always owns one and the team cannot
  // starve itself.  false: neither (a pipe, an empty file, no room) -> the sequential reader below handles it.
  auto parse_file_in_regions = [&](const char *path) -> bool {
    struct stat st;
    if (::stat(path, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) return false;  // pipes: one sequential reader
    bool direct;
    {
      gzFile g = gzopen(path, "rb");
      if (!g) die(EX_NOINPUT, "can't open %s", path);
      direct = gzdirect(g) != 0 && !ku_seqio::Reader::file_is_bzip2(path);
      gzclose(g);
    }
    ku_seqio::GrowingText gtext;
    ku_seqio::GzTextStream gz;
    ku_seqio::RegionCutter cut;
    void *map = MAP_FAILED;
    const size_t n = (size_t)st.st_size;
    if (direct) {
      int fd = ::open(path, O_RDONLY);
      if (fd < 0) return false;
      map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      ::close(fd);
      if (map == MAP_FAILED) return false;
      cut.data = (const char *)map;
      cut.n = n;
    } else {
      if (getenv("KU_NO_GZ_REGIONS") || !gz.open(path, gtext)) return false;
      bool complete;
      if (gtext.wait_for(1, &complete) == 0) {  // no text at all
        gz.close();
        if (!gtext.error.empty()) die(EX_DATAERR, "%s: %s", path, gtext.error.c_str());
        return true;
      }
      cut.data = gtext.base;
      cut.gt = &gtext;
    }
    const char *data = cut.data;
    const bool fastq = data[0] == '@';  // determine_input_file_type (src/classify.cpp:377-388)
    cut.fastq = fastq;
    const double t_parse = now_s();
    // a quarter of a work unit per region: the team's batches are pinned memory, smaller ones are quicker to set up
    // and keep the three stages busier.  FASTQ text is ~2.2 bytes per base (header, '+', qualities), FASTA ~1.05
    cut.region_bytes = std::max<size_t>((size_t)1 << 16, (size_t)((double)(unit_nt / 4) * (fastq ? 2.3 : 1.05)));
    cut.ramp = getenv("KU_REGION_RAMP") ? (size_t)atoi(getenv("KU_REGION_RAMP")) : 12;  // (smaller first regions: the first batch reaches the device after 2.5 ms instead of 8-10; round 5, once the start-up stalls were gone: windows of 0.117-0.133 s against 0.099-0.148)
    std::mutex mu;
    std::condition_variable cv;
    size_t next_out = 0;
    struct Parsed { Batch *bt; size_t lo, hi; ku_seqio::RegionParse res; };
    std::map<size_t, Parsed> ready;
    ku_seqio::GrowingText *const gtp = direct ? nullptr : &gtext;
    auto member = [&] {
      prctl(PR_SET_NAME, "ku-parse");
      const double cpu0 = thread_cpu_s();
      for (;;) {
        Batch *bt = chunked ? new Batch() : free_q.pop();
        size_t lo, hi, idx;
        if (!cut.claim(lo, hi, idx)) {
          if (chunked) { bt->release(); delete bt; } else free_q.push(bt);
          { std::lock_guard<std::mutex> l(mu); }
          cv.notify_all();
          cpu_add(cpu_parse, cpu0);
          return;
        }
        bt->clear();
        bt->trace[0] = now_s();
        bt->fastq = fastq;
        bt->first_of_file = false;  // (set where the batches go on in file order)
        bt->reserve_seq(fastq ? (hi - lo) / 2 + 4096 : hi - lo);  // one allocation: the sequences are at most that long
#ifdef MADV_POPULATE_READ
        if (direct) {  // the region's pages into this process's page table with one call instead of one fault per 4 KiB (eight
                       // threads faulting in one address space queue on its locks: a third of the team's time); failure is harmless
          static const bool populate = !(getenv("KU_NO_POPULATE") && atoi(getenv("KU_NO_POPULATE")));
          // (2 MiB per call: the call holds the address space's lock shared for as long as it runs, and a thread that wants it
          // exclusively -- any mmap / munmap, e.g. under malloc or in the GPU runtime -- waits for every holder while it keeps all
          // new ones out, page faults included.  With a region per call, twelve members held it ~10 ms each at the start of a
          // file and the whole process stood still for ~20 ms: the pipeline trace in profiles/r05_e2e_sweep.log)
          static const size_t step = (size_t)std::max(1, getenv("KU_POPULATE_MB") ? atoi(getenv("KU_POPULATE_MB")) : 2) << 20;
          const size_t pg = 4096, a0 = lo & ~(pg - 1);
          if (populate)
            for (size_t a = a0; a < hi; a += step) (void)madvise((void *)(data + a), std::min(step, hi - a), MADV_POPULATE_READ);
        }
#endif
        // (the records that START in the region, each read to its end wherever that lies: ku_seqio::parse_region)
        const ku_seqio::RegionParse res = ku_seqio::parse_region(data, n, gtp, lo, hi, fastq, *bt, keep_records);
        bt->trace[1] = now_s();
        { std::lock_guard<std::mutex> l(mu); ready[idx] = Parsed{bt, lo, hi, res}; }
        cv.notify_all();
      }
    };
    std::vector<std::thread> team;
    // (a .gz / .bz2 file: the inflating team is the slowest stage and wants the cores -- six parsers keep up with it; measured
    // on the 16-CPU quota of the GPU box, 10 M reads from one gzip stream: 0.368-0.372 s with 6, 0.381-0.412 with 12)
    const int members = direct ? parse_team : std::min(parse_team, getenv("KU_PARSE_TEAM_GZ") ? std::max(1, atoi(getenv("KU_PARSE_TEAM_GZ"))) : 6);
    for (int t = 0; t < members; ++t) team.emplace_back(member);
    // The batches go on in file order.  A region counts iff the parse of the region before it stopped exactly at its start
    // (ku_seqio::RegionChain: a region cut inside a record -- damaged FASTQ -- is parsed again from there, by this thread); the
    // reference's "a work unit without nucleotides ends the file" is applied to work units, not to batches (ku_seqio::UnitGate).
    ku_seqio::RegionChain chain;
    gate.begin_file();
    bool file_start_pending = true;  // the next batch that goes on opens the file (work units do not span files)
    auto recycle = [&](Batch *bt) { if (chunked) { bt->release(); delete bt; } else free_q.push(bt); };
    auto forward = [&](Batch *bt) {
      if (bt->off.empty()) { recycle(bt); return; }
      bt->first_of_file = file_start_pending;
      file_start_pending = false;
      if (chunked) inflight_add(bt->nt);
      bt->trace[2] = now_s();
      parsed_q.push(bt);
    };
    for (;;) {
      std::unique_lock<std::mutex> l(mu);
      size_t handed = 0;
      cv.wait(l, [&] { return ready.count(next_out) || (cut.finished(&handed) && next_out == handed); });
      auto it = ready.find(next_out);
      if (it == ready.end()) break;  // every region handed out and forwarded
      const Parsed p = it->second;
      ready.erase(it);
      ++next_out;
      l.unlock();
      switch (chain.judge(p.lo, p.hi, p.res)) {
        case ku_seqio::RegionChain::REPARSE:
          p.bt->clear();
          chain.accept(ku_seqio::parse_region(data, n, gtp, chain.expect, p.hi, fastq, *p.bt, keep_records));
          gate.push(p.bt, forward, recycle);
          break;
        case ku_seqio::RegionChain::ACCEPT: gate.push(p.bt, forward, recycle); break;
        case ku_seqio::RegionChain::SKIP: recycle(p.bt); break;
      }
      if (chain.ended) { cut.halt(); if (!direct) gtext.cancel(); break; }  // malformed record / end of the file: nothing behind it counts
      if (!direct) gtext.release_before(std::min(p.hi, chain.expect));  // (its sequences are in the batch: the text's pages go back)
    }
    gate.finish(forward, recycle);
    for (auto &t : team) t.join();
    {  // batches parsed behind the end of the stream are dropped
      std::lock_guard<std::mutex> l(mu);
      for (auto &kv : ready) { Batch *bt = kv.second.bt; if (chunked) { bt->release(); delete bt; } else free_q.push(bt); }
    }
    // (the mapping is taken down behind the timing window: unmapping 3 GB of populated pages took the reader 30 ms AFTER the last
    // line was written -- giving memory back is no part of classifying, as for the pool below)
    if (direct) input_maps.emplace_back(map, n);
    else {
      gz.close();
      // damage of the compressed file (a parser that stopped early cancels the producer: that leaves no error behind)
      if (!gtext.error.empty()) die(EX_DATAERR, "%s: %s", path, gtext.error.c_str());
    }
    busy_reader += now_s() - t_parse;
    return true;
  };

  std::thread reader([&] {
    prctl(PR_SET_NAME, "ku-read");
    std::string header, quals, header2;
    auto add_record_meta = [&](Batch *bt, const std::string &hdr, size_t id_lo, size_t id_hi, const std::string &q) {
      bt->add_meta(hdr, id_lo, id_hi, q, keep_records);
    };
    for (int fi = optind; fi < argc; fi += paired ? 2 : 1) {
      if (parse_team > 1 && parse_file_in_regions(argv[fi])) continue;  // a regular file, plain or .gz: the parser team took it
      Reader rd, rd2;
      rd.open(argv[fi], /*prefetch=*/true);
      if (paired) rd2.open(argv[fi + 1], /*prefetch=*/true);
      bool more = true, file_start_pending = true;
      gate.begin_file();
      auto recycle = [&](Batch *b) { if (chunked) { b->release(); delete b; } else free_q.push(b); };
      auto forward = [&](Batch *b) {
        if (b->off.empty()) { recycle(b); return; }
        b->first_of_file = file_start_pending;
        file_start_pending = false;
        if (chunked) inflight_add(b->nt);
        parsed_q.push(b);
      };
      while (more) {
        Batch *bt = chunked ? new Batch() : free_q.pop();  // -x: every batch stays alive until the last chunk
        const double t_parse = now_s();
        bt->clear();
        bt->first_of_file = false;
        bt->fastq = paired ? false : rd.fastq;  // mate pairs travel as merged FASTA records (read_merger.pl:187-197)
        while (bt->nt < unit_nt) {
          size_t n1 = 0, n2 = 0, lo, hi;
          bt->begin_read();
          if (!paired) {
            if (!ku_seqio::next_record(rd, *bt, &header, keep_records ? &quals : nullptr, &n1)) { bt->off.pop_back(); more = false; break; }
            bt->end_read();
            ku_seqio::split_id(header.data(), header.size(), lo, hi);
            add_record_meta(bt, header, lo, hi, quals);
            continue;
          }
          // mate pairs: id of mate 1 without its /1 suffix, seq1 + "N" + seq2 (read_merger.pl:102-117,182,187-191);
          // when one file runs out the other's remaining reads go through unpaired, with the script's warning
          const bool got1 = ku_seqio::next_record(rd, *bt, &header, nullptr, &n1);
          if (got1) {
            const size_t mark = bt->seqs_len;
            bt->append("N", 1);
            if (!ku_seqio::next_record(rd2, *bt, &header2, nullptr, &n2)) {
              if (!warned_pairs) fprintf(stderr, "classify: mismatched sequence counts - file 1 has more reads\n\n  Outputting the further reads unpaired\n");
              warned_pairs = true;
              bt->seqs_len = mark;  // drop the joining N
            }
          } else if (ku_seqio::next_record(rd2, *bt, &header, nullptr, &n2)) {
            if (!warned_pairs) fprintf(stderr, "classify: mismatched sequence counts - file 2 has more reads\n\n  Outputting the further reads unpaired\n");
            warned_pairs = true;
          } else { bt->off.pop_back(); more = false; break; }
          bt->end_read();
          ku_seqio::split_id(header.data(), header.size(), lo, hi);
          hi = lo + ku_seqio::strip_mate_suffix(header.data() + lo, hi - lo);
          header.erase(hi);  // -C/-U records carry the merged id only
          header.erase(0, lo);
          quals.clear();
          add_record_meta(bt, header, 0, header.size(), quals);
        }
        busy_reader += now_s() - t_parse;
        gate.push(bt, forward, recycle);
      }
      gate.finish(forward, recycle);  // a work unit without nucleotides ends the file, its reads are dropped (src/classify.cpp:522-523)
      rd.close();
      rd2.close();
    }
    parsed_q.push(nullptr);
  });

  // Output stage in two steps that overlap: the formatting helpers (a standing team of `fmt_threads`) take slices of the
  // finished batches from one queue -- `fmt_threads` slices per batch, disjoint read ranges, across batch borders -- while the
  // writer writes the batches in input order, each as soon as its slices are through.  (One thread doing both, with a team
  // spawned per batch, was the slowest stage of the pipeline in round 1: 16 thread starts and a serial 12 MB write per batch;
  // a team behind a barrier per batch left its members idle a third of the time in round 5.)
  struct Formatted {
    Batch *bt; std::vector<char *> parts; std::vector<size_t> len; std::vector<uLong> crc; std::vector<size_t> raw;
    std::vector<double> t_end;     // when each slice was done (the writer takes the latest for the trace)
    std::atomic<int> pending{0};   // slices still being formatted: the writer waits for 0 (fmt_done_cv)
    double t0 = 0;
  };
  std::mutex fmt_done_mu;
  std::condition_variable fmt_done_cv;
  struct FQueue {
    std::mutex m; std::condition_variable cv; std::deque<Formatted *> q;
    void push(Formatted *f) { { std::lock_guard<std::mutex> l(m); q.push_back(f); } cv.notify_one(); }
    Formatted *pop() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty(); }); Formatted *f = q.front(); q.pop_front(); return f; }
  } write_q;
  std::thread formatter([&] {
    prctl(PR_SET_NAME, "ku-format");
    for (;;) {
      Batch *bt = done_q.pop();
      if (!bt) break;
      const uint64_t n = bt->off.size();
      Formatted *f = new Formatted;
      f->bt = bt;
      f->parts.assign(fmt_threads, nullptr); f->len.assign(fmt_threads, 0); f->crc.assign(fmt_threads, 0); f->raw.assign(fmt_threads, 0);
      f->t_end.assign(fmt_threads, 0.0);
      f->t0 = now_s();
      f->pending.store(print_kraken ? fmt_threads : 0);
      write_q.push(f);  // (in batch order; the writer waits until the batch's slices are through)
      if (!print_kraken) continue;
      for (int t = 0; t < fmt_threads; ++t)
        fmt_team.submit([&, f, bt, n, t] {
          {
            const uint64_t lo = n * t / fmt_threads, hi = n * (t + 1) / fmt_threads;
            const double cpu0 = thread_cpu_s();
            int status = KU_OK;
            if (hi > lo) {
              status = ku_format_kraken_rle(bt->seqs, bt->off.data() + lo, bt->len.data() + lo, hi - lo, bt->ids.data() + bt->idoff[lo], info.k,
                                            bt->calls.data() + lo, bt->runs, bt->run_off.data() + lo, bt->run_cnt.data() + lo, bt->hits.data() + lo,
                                            pflags, &f->parts[t], &f->len[t]);
              if (status == KU_OK && !bt->crlf_read.empty() && pflags == 0)  // KU_CRLF_REFERENCE=1: the reference's lines for such reads
                f->parts[t] = rewrite_crlf_lines(*bt, lo, hi, info.k, f->parts[t], &f->len[t]);
              if (status == KU_OK && s_kraken.pgz && f->len[t]) {  // -o x.gz: the helper deflates its own lines
                size_t cl = 0;
                unsigned char *c = ku_pgzout::deflate_part(f->parts[t], f->len[t], &cl, &f->crc[t]);
                if (!c) status = KU_ENOMEM;
                else {
                  ku_free(f->parts[t]);
                  f->parts[t] = (char *)c;  // (malloc'ed like the text: the writer frees either the same way)
                  f->raw[t] = f->len[t];
                  f->len[t] = cl;
                }
              }
            }
            if (status != KU_OK) die(exit_code_of(status), "%s", ku_strerror(status));
            cpu_add(cpu_format, cpu0);
            f->t_end[t] = now_s();
          }
          // (nothing of f or bt is touched behind this line: the writer may take them the moment the count reaches 0)
          if (f->pending.fetch_sub(1) == 1) { { std::lock_guard<std::mutex> l(fmt_done_mu); } fmt_done_cv.notify_all(); }
        });
    }
    write_q.push(nullptr);
  });
  std::thread writer([&] {
    prctl(PR_SET_NAME, "ku-write");
    const double cpu0 = thread_cpu_s();
    for (;;) {
      Formatted *f = write_q.pop();
      if (!f) { cpu_add(cpu_write, cpu0); break; }
      if (f->pending.load() != 0) {
        std::unique_lock<std::mutex> l(fmt_done_mu);
        fmt_done_cv.wait(l, [&] { return f->pending.load() == 0; });
      }
      Batch *bt = f->bt;
      bt->trace[6] = f->t0;
      bt->trace[7] = f->t0;
      for (double e : f->t_end) if (e > bt->trace[7]) bt->trace[7] = e;
      busy_format += bt->trace[7] - f->t0;
      const uint64_t n = bt->off.size();
      const double t_write = now_s();
      // (one thread, one write after the other: ~1.2 ms per 12 MB batch into a tmpfs file, the pipeline's slowest step since round 5;
      // a team of four pwrite()-ing a batch's parts side by side took 1.9 ms -- the file's pages are allocated under one lock;
      // a second thread allocating them ahead of the writer, fallocate(KEEP_SIZE) 32-512 MB ahead, made the writer slower
      // as well: 0.102-0.115 s of writing per run instead of 0.086)
      for (int t = 0; t < fmt_threads; ++t)
        if (f->parts[t]) {
          if (s_kraken.pgz) s_kraken.write_deflated((const unsigned char *)f->parts[t], f->len[t], f->crc[t], f->raw[t]);
          else s_kraken.write(f->parts[t], f->len[t]);
          ku_free(f->parts[t]);
        }
      delete f;
      if (keep_records) {  // print_sequence (src/classify.cpp:794-805)
        std::string rec;
        for (uint64_t i = 0; i < n; ++i) {
          Sink &sk = bt->calls[i] ? s_cls : s_ucls;
          if (bt->calls[i] ? !print_cls : !print_ucls) continue;
          rec.clear();
          rec += bt->fastq ? '@' : '>';
          rec += bt->headers.c_str() + bt->hoff[i];
          rec += '\n';
          rec.append(bt->seqs + bt->off[i], bt->len[i]);
          rec += '\n';
          if (bt->fastq) { rec += "+\n"; rec += bt->quals.c_str() + bt->qoff[i]; rec += '\n'; }
          sk.write(rec.data(), rec.size());
        }
      }
      bt->trace[8] = t_write;
      bt->trace[9] = now_s();
      busy_writer += bt->trace[9] - t_write;
      if (cli_trace) trace_rows.push_back(std::vector<double>(bt->trace, bt->trace + 10));
      for (uint64_t i = 0; i < n; ++i) total_classified += bt->calls[i] != 0;
      total_sequences += n;
      total_bases += bt->nt;
      fprintf(stderr, "\r Processed %llu sequences (%.2f%% classified)", total_sequences, total_classified * 100.0 / total_sequences);
      if (chunked) { inflight_sub(bt->nt); bt->release(); delete bt; } else free_q.push(bt);
    }
  });

  if (chunked) {
    // Out-of-core run (src/classify.cpp:566-791).  The reference re-reads the input once per database chunk; here the
    // read batches stay on the device and the CHUNKS cycle: the input is taken in super-batches that fit a fixed share
    // of the HBM; for each super-batch chunk 0 is searched while the reads still arrive, then one pass per further
    // chunk over the resident batches, then calls + hit lists.  Device and host memory are bounded by the super-batch
    // whatever the input size; the next chunk is uploaded and laid out by a helper thread (ku_ctx_prefetch_shard)
    // while the current one is searched.
    uint64_t free_b = 0, total_b = 0;
    KU_CHECK(ku_ctx_mem_info(ctx, &free_b, &total_b));
    uint64_t budget = free_b / 4;  // device bytes of resident batches (5 B per base: text + one slot per position)
    if (const char *e = getenv("KU_SUPERBATCH_BYTES")) { const long long v = atoll(e); if (v > 0) budget = (uint64_t)v; }
    {  // the reader stops this far ahead of the writer
      std::lock_guard<std::mutex> l(inflight_mu);
      chunk_budget_nt = budget / 5 > unit_nt ? budget / 5 : unit_nt;
    }
    inflight_cv.notify_all();
    const size_t n_chunks = chunk_bounds.size() - 1;
    ku_opts opts = base_opts;
    // One stream of chunks per GPU: its context, its share of the chunks, and the helper thread that uploads and lays out the
    // NEXT chunk while the resident one is searched.
    // -x SIZE is the reference's bound on ONE resident chunk (src/krakendb.cpp:463-522).  Double buffering needs room for a
    // second one next to it: when the device has none (KU_ENOMEM from the helper) the run goes on with one chunk at a time --
    // ku_ctx_swap_shard then uploads synchronously, as before there was a prefetch
    struct ChunkStream {
      ku_ctx *c = nullptr;
      const std::vector<size_t> *list = nullptr;
      std::thread prefetcher;
      int status = KU_OK;
      std::string error;
      bool off = false;
    };
    std::vector<ChunkStream> cs(n_ranks_x);
    for (size_t r = 0; r < n_ranks_x; ++r) {
      cs[r].c = r == 0 ? ctx : helpers[r - 1];
      cs[r].list = &rank_chunks[r];
      cs[r].off = getenv("KU_NO_PREFETCH") != nullptr;
    }
    auto start_prefetch = [&](ChunkStream &st, size_t c) {
      if (st.off) return;
      st.prefetcher = std::thread([&st, c, db, &chunk_bounds] {
        st.status = ku_ctx_prefetch_shard(st.c, db, chunk_bounds[c], chunk_bounds[c + 1]);
        if (st.status != KU_OK) st.error = ku_last_error();
      });
    };
    auto join_prefetch = [&](ChunkStream &st) {
      if (st.prefetcher.joinable()) st.prefetcher.join();
      if (st.status == KU_ENOMEM) {
        fprintf(stderr, "\rclassify: no device memory for a second database chunk next to the resident one: chunks are uploaded one at a time from here on\n");
        st.status = KU_OK;
        st.off = true;
      }
      if (st.status != KU_OK) die(exit_code_of(st.status), "%s: %s", ku_strerror(st.status), st.error.c_str());
    };
    bool input_done = false, first_super = true;
    size_t n_super = 0;
    // the further chunks of one GPU's list over its copies of the super-batch, then (input still coming) its first chunk back
    auto further_passes = [&](ChunkStream &st, const std::vector<ku_batch *> &mine, bool more_input, bool say) {
      const std::vector<size_t> &L = *st.list;
      for (size_t i = 1; i < L.size(); ++i) {
        if (say) fprintf(stderr, "\r Database chunk %zu of %zu", L[i] + 1, n_chunks);
        join_prefetch(st);
        KU_CHECK(ku_ctx_swap_shard(st.c, db, chunk_bounds[L[i]], chunk_bounds[L[i] + 1]));
        if (first_super) add_chunk_counts(st.c);
        // the chunk after this one -- or the list's first again for the next super-batch -- comes in underneath the passes
        if (i + 1 < L.size()) start_prefetch(st, L[i + 1]);
        else if (more_input) start_prefetch(st, L[0]);
        for (ku_batch *b : mine) KU_CHECK(ku_batch_lookup(st.c, b, &opts));
      }
      if (more_input && L.size() > 1) {
        join_prefetch(st);
        KU_CHECK(ku_ctx_swap_shard(st.c, db, chunk_bounds[L[0]], chunk_bounds[L[0] + 1]));
      }
    };
    while (!input_done) {
      // every GPU's first chunk is resident here (loaded at start-up, or swapped back in at the end of the previous super-batch)
      for (auto &st : cs)
        if (st.list->size() > 1) start_prefetch(st, (*st.list)[1]);
      std::vector<Batch *> all;
      uint64_t resident = 0;
      while (resident < budget) {
        Batch *bt = parsed_q.pop();
        if (!bt) { input_done = true; break; }
        KU_CHECK(ku_batch_create(ctx, bt->seqs, bt->seqs_len, bt->off.data(), bt->len.data(), bt->off.size(), &bt->dev));
        KU_CHECK(ku_batch_lookup(ctx, bt->dev, &opts));
        resident += 5 * (uint64_t)bt->seqs_len + 12 * (uint64_t)bt->off.size();
        all.push_back(bt);
      }
      if (all.empty()) { for (auto &st : cs) join_prefetch(st); break; }
      ++n_super;
      // the helpers: their copies of the super-batch, every chunk of their lists over them
      std::vector<std::vector<ku_batch *>> copies(n_ranks_x);
      std::vector<std::thread> team;
      for (size_t r = 1; r < n_ranks_x; ++r) {
        if (cs[r].list->empty()) continue;
        team.emplace_back([&, r] {
          ChunkStream &st = cs[r];
          for (Batch *bt : all) {
            ku_batch *b = nullptr;
            KU_CHECK(ku_batch_create(st.c, bt->seqs, bt->seqs_len, bt->off.data(), bt->len.data(), bt->off.size(), &b));
            KU_CHECK(ku_batch_lookup(st.c, b, &opts));
            copies[r].push_back(b);
          }
          further_passes(st, copies[r], !input_done, false);
        });
      }
      std::vector<ku_batch *> mine;
      for (Batch *bt : all) mine.push_back(bt->dev);
      further_passes(cs[0], mine, !input_done, true);
      for (auto &t : team) t.join();
      for (size_t i = 0; i < all.size(); ++i) {
        Batch *bt = all[i];
        for (size_t r = 1; r < n_ranks_x; ++r)
          if (i < copies[r].size()) {  // "non-zero wins" (src/classify.cpp:445-452): what the other GPUs' chunks found
            KU_CHECK(ku_batch_absorb(ctx, bt->dev, copies[r][i]));
            ku_batch_destroy(copies[r][i]);
          }
        const uint64_t n = bt->off.size();
        bt->calls.assign(n, 0); bt->hits.assign(n, 0); bt->run_off.assign(n, 0); bt->run_cnt.assign(n, 0);
        uint64_t n_runs = 0;
        KU_CHECK(ku_batch_finish(ctx, bt->dev, &opts, bt->calls.data(), bt->hits.data(), bt->run_off.data(), bt->run_cnt.data(), &n_runs));
        if (print_kraken && !quick) {
          bt->reserve_runs(n_runs);
          KU_CHECK(ku_fetch_runs(ctx, bt->runs, n_runs));
        }
        ku_batch_destroy(bt->dev);
        bt->dev = nullptr;
        done_q.push(bt);
      }
      first_super = false;
    }
    for (auto &st : cs) join_prefetch(st);
    // what the helpers' passes booked (HLL registers, k-mer counts) joins the first GPU's state: the report is written from there
    for (size_t r = 1; r < n_ranks_x; ++r)
      if (!rank_chunks[r].empty()) KU_CHECK(ku_ctx_merge_state(ctx, helpers[r - 1]));
    if (n_super > 1) fprintf(stderr, "\r %zu passes over the %zu database chunks (the input did not fit the device at once)\n", n_super, n_chunks);
  } else {
  // GPU stage.  One GPU, the database resident: the batches go through ku_classify_batch_rle in its two-step form with THREE
  // in flight -- the uploads of the next batches and the copies back of the previous one run under the kernels of batch b, and
  // this thread waits for one event per batch (one step per batch cost ~1 ms of fixed time each, four times the kernels'; VERDICT
  // r04 weak #3).  Groups (KU_DEVICES) and UID mapping (whose calls are replaced batch by batch) go one batch at a time.
  const bool two_step = !mg && !map_uids && !getenv("KU_RLE_ONE_STEP");
  // (a batch's way through the device is ~1 ms of dependent steps around a 0.2 ms kernel: three in flight hide it)
  const size_t depth = getenv("KU_RLE_DEPTH") ? (size_t)std::min(std::max(atoi(getenv("KU_RLE_DEPTH")), 1), KU_RLE_MAX_IN_FLIGHT) : 3;
  std::deque<Batch *> flying;
  uint64_t runs_seen_max = 0;  // extent of the largest run array so far: the next batches' buffers take it in one go
  auto finish_oldest = [&] {
    Batch *ft = flying.front();
    flying.pop_front();
    const double t0 = now_s();
    uint64_t n_runs = 0;
    KU_CHECK(ku_classify_batch_rle_finish(ctx, &n_runs));
    const double t1 = now_s();
    busy_gpu_classify += t1 - t0;
    if (print_kraken && !quick && n_runs > ku_classify_batch_rle_copied(ctx)) {  // the runs feed the Kraken lines: usually they came
      ft->reserve_runs(n_runs);                                                   // with the calls; a batch with more runs than expected
      KU_CHECK(ku_fetch_runs(ctx, ft->runs, n_runs));                            // fetches them (the next ones make more room)
    }
    if (n_runs > runs_seen_max) runs_seen_max = n_runs;
    const double t2 = now_s();
    busy_gpu_fetch += t2 - t1;
    busy_gpu += t2 - t0;
    ft->trace[5] = t2;
    done_q.push(ft);
  };
  for (;;) {
    Batch *bt = nullptr;
    if (two_step && !flying.empty() && !parsed_q.try_pop(&bt)) {  // nothing parsed yet: the time goes to the batch in flight
      finish_oldest();
      continue;
    }
    if (!two_step || flying.empty()) { if (!bt) bt = parsed_q.pop(); }
    if (!bt) { while (!flying.empty()) finish_oldest(); break; }
    const uint64_t n = bt->off.size();
    const double t_gpu = now_s();
    bt->calls.resize(n);  // every element is written by the copies back from the device
    bt->hits.resize(n);
    bt->run_off.resize(n);
    bt->run_cnt.resize(n);
    ku_opts opts = base_opts;
    uint64_t n_runs = 0;
    if (sparse && bt->first_of_file) {  // work units do not span input files
      while (!flying.empty()) finish_oldest();
      if (mg) KU_CHECK(ku_mgpu_sparse_close_unit(mg));
      else if (ku_ctx_sparse_state(ctx) == 1) KU_CHECK(ku_sparse_close_unit(ctx));
    }
    if (two_step) {
      if (flying.size() >= depth) finish_oldest();
      const double t_enq0 = now_s();
      // the runs come back with the calls when their buffer holds the batch's run array: a quarter more than the largest so far
      // (the first batches: 3 runs per 100 bases, what the pool's buffers were sized for)
      const bool want_runs = print_kraken && !quick;
      const uint64_t r_est = std::max<uint64_t>(runs_seen_max + runs_seen_max / 4, bt->seqs_len / 32);
      if (want_runs) bt->reserve_runs(r_est);
      ku_run *rbuf = want_runs ? bt->runs : nullptr;
      const uint64_t rcap = want_runs ? std::min<uint64_t>(bt->runs_cap, r_est) : 0;  // (what is copied, not what the buffer could hold)
      int st = ku_classify_batch_rle_enqueue(ctx, bt->seqs, bt->seqs_len, bt->off.data(), bt->len.data(), n, &opts, bt->calls.data(),
                                             bt->hits.data(), bt->run_off.data(), bt->run_cnt.data(), rbuf, rcap);
      if (st == KU_ESTATE && !flying.empty()) {  // a batch that cannot overlap with the one in flight (quick mode, a very long read, ...)
        while (!flying.empty()) finish_oldest();
        st = ku_classify_batch_rle_enqueue(ctx, bt->seqs, bt->seqs_len, bt->off.data(), bt->len.data(), n, &opts, bt->calls.data(),
                                           bt->hits.data(), bt->run_off.data(), bt->run_cnt.data(), rbuf, rcap);
      }
      KU_CHECK(st);
      flying.push_back(bt);
      const double t_enq = now_s();
      bt->trace[3] = t_enq0;
      bt->trace[4] = t_enq;
      busy_gpu_classify += t_enq - t_enq0;
      busy_gpu += t_enq - t_enq0;
      continue;
    }
    if (mg)
      KU_CHECK(ku_mgpu_classify_batch_rle(mg, bt->seqs, bt->seqs_len, bt->off.data(), bt->len.data(), n, &opts,
                                          bt->calls.data(), bt->hits.data(), bt->run_off.data(), bt->run_cnt.data(), &n_runs));
    else
      KU_CHECK(ku_classify_batch_rle(ctx, bt->seqs, bt->seqs_len, bt->off.data(), bt->len.data(), n, &opts,
                                     bt->calls.data(), bt->hits.data(), bt->run_off.data(), bt->run_cnt.data(), &n_runs));
    const double t_fetch = now_s();
    busy_gpu_classify += t_fetch - t_gpu;
    if ((print_kraken && !quick) || map_uids) {  // the runs feed the Kraken lines -- and the UID resolution
      bt->reserve_runs(n_runs);
      if (mg) KU_CHECK(ku_mgpu_fetch_runs(mg, bt->runs, n_runs));
      else KU_CHECK(ku_fetch_runs(ctx, bt->runs, n_runs));
    }
    busy_gpu_fetch += now_s() - t_fetch;
    if (map_uids) {  // the calls of resolve_tree give way to resolve_uids3's; the read counts on the device follow
      KU_CHECK(ku_resolve_uids(tax, uid_map, bt->runs, bt->run_off.data(), bt->run_cnt.data(), bt->len.data(), n, info.k,
                               (uint32_t)fmt_threads, bt->calls.data()));
      uint64_t dropped = 0;
      KU_CHECK(ku_ctx_replace_calls(ctx, bt->calls.data(), n, &dropped));
      if (dropped && !warned_uid_calls) {
        fprintf(stderr, "\rclassify: reads were called with taxids that are neither in taxDB nor values of the database: they are missing from the report\n");
        warned_uid_calls = true;
      }
    }
    busy_gpu += now_s() - t_gpu;
    done_q.push(bt);
  }
  }
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
