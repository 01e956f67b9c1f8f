// classify_run.h -- the state one run of the `classify` executable shares between its stages, and the small tools they use.
// The program is host C++ over the C ABI (include/krakenuniq_amd.h), in four translation units:
//   classify_main.cpp    flags (getopt string of src/classify.cpp:1074), database / taxonomy / device set-up, the timing window,
//                        stderr summary, report, teardown
//   classify_input.cpp   reader stage: FASTA / FASTQ (+ .gz / .bz2) -> page-locked read batches, by a parser team over record-aligned
//                        regions or by one sequential reader (pipes, mate pairs) -- src/seqreader.cpp, src/classify.cpp:499-525
//   classify_device.cpp  device stage: batches through ku_classify_batch_rle in its two-step form (database resident), through the
//                        multi-GPU driver, or over database chunks streamed through HBM (-x, src/classify.cpp:566-791)
//   classify_output.cpp  output stage: Kraken lines formatted by a team of helpers, files written in input order
//                        (src/classify.cpp:826-861,980-1010), -C / -U read files
#pragma once
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
[[noreturn]] inline void leave(int code) {
  fflush(nullptr);
  _exit(code);
}
inline void die(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3), noreturn));
inline void die(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "classify: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  leave(code);
}
inline int exit_code_of(int st) {
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

inline double now_s() {
  timeval t;
  gettimeofday(&t, nullptr);
  return (double)t.tv_sec + (double)t.tv_usec / 1e6;
}
// CPU seconds the calling thread has used (KU_CLI_TIMES: where the cores of a quota-limited host go)
inline double thread_cpu_s() {
  timespec t;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
inline double process_cpu_s(double *sys_s) {
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  if (sys_s) *sys_s = (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec;
  return (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec;
}
inline double seconds_between(const timeval &a, const timeval &b) {
  return (double)(b.tv_sec - a.tv_sec) + (double)(b.tv_usec - a.tv_usec) / 1e6;
}

// the formatting helpers: they take tasks -- a slice of a batch each -- from one queue, across batches
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
};

// a batch on its way through the output stage: the text of its slices
struct Formatted {
  Batch *bt; std::vector<char *> parts; std::vector<size_t> len; std::vector<uLong> crc; std::vector<size_t> raw;
  std::vector<double> t_end;     // when each slice was done (the writer takes the latest for the trace)
  std::atomic<int> pending{0};   // slices still being formatted: the writer waits for 0 (fmt_done_cv)
  double t0 = 0;
};
struct FQueue {
  std::mutex m; std::condition_variable cv; std::deque<Formatted *> q;
  void push(Formatted *f) { { std::lock_guard<std::mutex> l(m); q.push_back(f); } cv.notify_one(); }
  Formatted *pop() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty(); }); Formatted *f = q.front(); q.pop_front(); return f; }
};

struct Run {
  // ---- flags the stages look at
  bool paired = false, quick = false, print_cls = false, print_ucls = false;
  bool warned_pairs = false, warned_uid_calls = false;
  uint64_t unit_nt = 64ull << 20;  // GPU batch size in nt (KU_BATCH_NT)
  int fmt_threads = 4;             // -t: host threads that format the Kraken lines (the GPU replaces the OpenMP team)
  int parse_team = 1;              // region parsers of a plain / compressed input file
  int argc = 0;
  char **argv = nullptr;
  // ---- handles
  ku_db *db = nullptr;
  ku_db_info info{};
  ku_tax *tax = nullptr;
  ku_uid_map *uid_map = nullptr;
  ku_ctx *ctx = nullptr;
  ku_mgpu *mg = nullptr;              // KU_DEVICES=0,1,...: several GPUs through the multi-GPU driver
  std::vector<ku_ctx *> helpers;      // KU_DEVICES with -x: the other GPUs, each streams its share of the chunks
  std::vector<uint64_t> chunk_bounds; // -x SIZE: the chunk plan (empty: everything resident)
  std::vector<std::vector<size_t>> rank_chunks;
  size_t n_ranks_x = 1;
  bool chunked = false, map_uids = false, sparse = false;
  std::function<void(ku_ctx *)> add_chunk_counts;  // database.kdb.counts of a chunked run, summed chunk by chunk
  // ---- the pipeline: batches circulate reader -> device -> formatter / writer -> reader
  std::vector<Batch> pool;
  Queue free_q, parsed_q, done_q;
  ku_seqio::UnitGate gate;  // (reader thread only)
  bool keep_records = false;
  // -x runs allocate a batch per region instead of recycling a pool: the nucleotides between reader and writer are bounded instead
  uint64_t chunk_budget_nt = ~0ull, inflight_nt = 0;
  std::mutex inflight_mu;
  std::condition_variable inflight_cv;
  void inflight_add(uint64_t nt) {
    std::unique_lock<std::mutex> l(inflight_mu);
    inflight_cv.wait(l, [&] { return inflight_nt == 0 || inflight_nt + nt <= 2 * chunk_budget_nt; });
    inflight_nt += nt;
  }
  void inflight_sub(uint64_t nt) {
    { std::lock_guard<std::mutex> l(inflight_mu); inflight_nt -= nt; }
    inflight_cv.notify_all();
  }
  // ---- output
  Sink s_kraken, s_cls, s_ucls;
  bool print_kraken = true;
  uint32_t pflags = 0;
  ku_opts base_opts{};
  FmtTeam fmt_team;
  FQueue write_q;
  std::mutex fmt_done_mu;
  std::condition_variable fmt_done_cv;
  unsigned long long total_sequences = 0, total_classified = 0, total_bases = 0;
  // ---- what the stages did (KU_CLI_TIMES, KU_CLI_TRACE)
  double busy_reader = 0, busy_gpu = 0, busy_writer = 0, busy_format = 0;  // seconds each pipeline stage spent working
  double busy_gpu_classify = 0, busy_gpu_fetch = 0;                       // ... of the device stage: the batch call, the runs' copy back
  std::mutex cpu_mu;
  double cpu_parse = 0, cpu_format = 0, cpu_write = 0;                    // CPU seconds of the stages' threads
  void cpu_add(double &acc, double t0) { const double d = thread_cpu_s() - t0; std::lock_guard<std::mutex> l(cpu_mu); acc += d; }
  bool cli_trace = false;
  std::vector<std::vector<double>> trace_rows;
  std::vector<std::pair<void *, size_t>> input_maps;  // mappings of the input files the parser team read from (unmapped behind the window)

  int run(int argc_, char **argv_);           // classify_main.cpp
  void reader_stage();                        // classify_input.cpp
  bool parse_file_in_regions(const char *path);
  void device_stage_resident();               // classify_device.cpp
  void device_stage_chunked();
  void formatter_stage();                     // classify_output.cpp
  void writer_stage();
};
