// Host-side sequence input of the classify executable: FASTA/FASTQ records (plain, .gz, .bz2) parsed straight into the pinned
// read batch the C ABI takes (no per-record std::string round trips), with the record semantics of the reference's
// readers (src/seqreader.cpp:26-133) and, for mate pairs, of scripts/read_merger.pl:100-197.
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <atomic>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/krakenuniq_amd.h"
#include "ku_pgzip.h"
#include "ku_pbzip2.h"

namespace ku_seqio {

[[noreturn]] void fatal(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));  // provided by the program

// ---- a batch of reads travelling through the pipeline; the big arrays live in pinned host memory
// Allocator of the per-read arrays that cross PCIe (offsets, lengths, calls, hit counts, run index): page-locked memory
// when `enabled` (set once at program start, before any batch exists), so that their copies are DMA transfers instead
// of staged ones; plain memory otherwise (tools without a device, -x runs with one batch per region).
struct PinSwitch { static inline bool enabled = false; };

// CPUs this process may really use: the affinity mask, capped by the container's CFS quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us) -- a box that shows 256 CPUs under a quota of 16 runs 16 threads' worth, and a team
// sized for 256 is throttled (measured: the gzip team at 16 on such a box lost to the team at 12)
inline int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n > 0 ? n : 1 << 30, CPU_COUNT(&set));
  long long quota = -1, period = 0;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else {
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
  }
  if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
  return std::max(1, n);
}
template <class T> struct PinAlloc {
  using value_type = T;
  PinAlloc() = default;
  template <class U> PinAlloc(const PinAlloc<U> &) {}
  T *allocate(size_t n) {
    void *p = nullptr;
    if (PinSwitch::enabled ? ku_host_alloc(n * sizeof(T), &p) != KU_OK : (p = malloc(n * sizeof(T))) == nullptr) fatal(71, "out of host memory");
    return (T *)p;
  }
  void deallocate(T *p, size_t) { if (PinSwitch::enabled) ku_host_free(p); else free(p); }
  template <class U> bool operator==(const PinAlloc<U> &) const { return true; }
  template <class U> bool operator!=(const PinAlloc<U> &) const { return false; }
};
template <class T> using PinVec = std::vector<T, PinAlloc<T>>;

struct Batch {
  char *seqs = nullptr;       // reads, each followed by '\n' (the separator the C ABI asks for)
  size_t seqs_len = 0, seqs_cap = 0;
  ku_run *runs = nullptr;     // run-length encoded per-k-mer codes (ku_classify_batch_rle)
  size_t runs_cap = 0;
  std::string ids, headers, quals;
  PinVec<uint64_t> off, run_off;
  PinVec<uint32_t> len, calls, hits, run_cnt;
  std::vector<uint64_t> idoff, hoff, qoff;
  bool fastq = false;
  bool first_of_file = false;  // the batch opens an input file (the reference's work units do not span files)
  uint64_t nt = 0;
  double trace[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // KU_CLI_TRACE=1: when the batch reached each step of classify's pipeline
  ku_batch *dev = nullptr;    // -x runs: the batch stays on the device across the chunk passes
  bool pinned = true;         // page-locked buffers for the copies to the device; false: plain host memory
  void *alloc_bytes(size_t n) {
    void *p = nullptr;
    if (pinned ? ku_host_alloc(n, &p) != KU_OK : (p = malloc(n)) == nullptr) fatal(71, "out of host memory");
    return p;
  }
  void free_bytes(void *p) { if (pinned) ku_host_free(p); else free(p); }
  // KU_CRLF_REFERENCE=1 (multi-line FASTA with CRLF line ends, see crlf_note()): the reads that had carriage returns inside,
  // and where their lines began (base index in the joined sequence) -- crlf_off is a CSR over crlf_start
  std::vector<uint32_t> crlf_read, crlf_off, crlf_start;
  void clear() {
    seqs_len = 0; nt = 0;
    ids.clear(); headers.clear(); quals.clear();
    off.clear(); idoff.clear(); hoff.clear(); qoff.clear(); len.clear();
    crlf_read.clear(); crlf_off.clear(); crlf_start.clear();
  }
  void reserve_seq(size_t extra) {
    const size_t need = seqs_len + extra + 1;
    if (need <= seqs_cap) return;
    size_t ncap = seqs_cap ? seqs_cap : (size_t)1 << 24;
    while (ncap < need) ncap *= 2;
    void *np = alloc_bytes(ncap);
    if (seqs_len) memcpy(np, seqs, seqs_len);
    if (seqs) free_bytes(seqs);
    seqs = (char *)np;
    seqs_cap = ncap;
  }
  // a read is built from one or more pieces (FASTA lines, mate 1 + 'N' + mate 2) and closed with end_read()
  void begin_read() { off.push_back(seqs_len); }
  void append(const char *p, size_t n) {
    reserve_seq(n);
    memcpy(seqs + seqs_len, p, n);
    seqs_len += n;
  }
  void end_read() {
    const uint64_t l = seqs_len - off.back();
    len.push_back((uint32_t)l);
    nt += l;
    reserve_seq(0);
    seqs[seqs_len++] = '\n';
  }
  // id (a slice of the header line) and, for -C/-U, the whole header and the quality line of the read just closed
  void add_meta(const std::string &hdr, size_t id_lo, size_t id_hi, const std::string &q, bool keep_records) {
    idoff.push_back(ids.size());
    ids.append(hdr, id_lo, id_hi - id_lo);
    ids.push_back('\0');
    if (keep_records) {
      hoff.push_back(headers.size()); headers += hdr; headers.push_back('\0');
      qoff.push_back(quals.size()); quals += q; quals.push_back('\0');
    }
  }
  void reserve_runs(size_t n) {
    if (n <= runs_cap) return;
    if (runs) free_bytes(runs);
    size_t ncap = runs_cap ? runs_cap : (size_t)1 << 20;
    while (ncap < n) ncap *= 2;
    runs = (ku_run *)alloc_bytes(ncap * sizeof(ku_run));
    runs_cap = ncap;
  }
  void release() {
    if (seqs) free_bytes(seqs);
    if (runs) free_bytes(runs);
    seqs = nullptr; runs = nullptr;
    seqs_cap = runs_cap = 0;
  }
};

// ---- compressed input parsed in regions: the text of a .gz file arrives in one contiguous (virtual) range, written front
// to back by the inflating team while the parser team already cuts and parses record-aligned regions of it, exactly as it
// does with a mapped plain file.  Pages behind the parsed regions go back to the system (MADV_DONTNEED), the producer
// stays at most `max_ahead` bytes in front of them.
struct GrowingText {
  char *base = nullptr;
  size_t reserved = 0;
  std::mutex m;
  std::condition_variable cv;
  size_t avail = 0, freed = 0, max_ahead = (size_t)1 << 30;
  int starving = 0;  // consumers waiting for more text: the producer may then run further ahead than max_ahead
  bool done = false, cancelled = false;
  std::string error;
  GrowingText() = default;
  GrowingText(const GrowingText &) = delete;
  GrowingText &operator=(const GrowingText &) = delete;
  ~GrowingText() { if (base) munmap(base, reserved); }
  bool reserve(size_t bytes) {
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) return false;
#ifdef MADV_HUGEPAGE
    (void)madvise(p, bytes, MADV_HUGEPAGE);  // 2 MiB faults where the system allows them
#endif
    base = (char *)p;
    reserved = bytes;
    if (const char *e = getenv("KU_TEXT_AHEAD_MB")) max_ahead = (size_t)std::max(1L, atol(e)) << 20;
    return true;
  }
  // producer: where the next n bytes go (nullptr: cancelled, or the reservation is used up)
  char *place(size_t n) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return cancelled || starving > 0 || avail == freed || avail + n - freed <= max_ahead; });
    if (cancelled || avail + n > reserved) return nullptr;
    return base + avail;
  }
  void publish(size_t n) {
    { std::lock_guard<std::mutex> l(m); avail += n; }
    cv.notify_all();
  }
  void finish(const std::string &err = std::string()) {
    { std::lock_guard<std::mutex> l(m); done = true; if (!err.empty() && !cancelled) error = err; }  // (a cancelled producer's complaint is no error)
    cv.notify_all();
  }
  // consumers: wait until `need` bytes are there or the text is complete; the bytes there now
  size_t wait_for(size_t need, bool *complete) {
    std::unique_lock<std::mutex> l(m);
    if (avail < need && !done && !cancelled) {
      ++starving;
      cv.notify_all();
      cv.wait(l, [&] { return avail >= need || done || cancelled; });
      --starving;
    }
    *complete = done || cancelled;
    return avail;
  }
  void release_before(size_t pos) {
    pos &= ~(size_t)((2u << 20) - 1);
    std::unique_lock<std::mutex> l(m);
    if (pos <= freed) return;
    const size_t lo = freed;
    freed = pos;
    l.unlock();
    (void)madvise(base + lo, pos - lo, MADV_DONTNEED);
    cv.notify_all();
  }
  void cancel() {
    { std::lock_guard<std::mutex> l(m); cancelled = true; }
    cv.notify_all();
  }
};

// ---- FASTA/FASTQ reader (gz transparently via zlib).  Lines are handed out as ranges of one large buffer.  With
// prefetch on, a producer thread per file does the read(2) / inflate into a small ring of blocks, so decompression
// overlaps with parsing and the two files of a mate pair are inflated concurrently.
struct Reader {
  gzFile g = nullptr;
  int fd = -1;  // plain (uncompressed) files are read with read(2): one copy less than through zlib
  bool fastq = false, valid = true, eof = false;
  // no record of this file has been handed out yet.  The reference's FASTA reader takes the header of every record but the
  // first from the loop that read the record before it (its `linebuffer`, src/seqreader.cpp:62-71): a header line that ends the
  // file WITHOUT a line end leaves that loop with the stream at its end, and the next call returns "no more sequences"
  // (:37-40) -- the record is dropped.  The first header of a file is read by the call itself and goes through.
  bool first_record = true;
  std::vector<char> buf;
  const char *mem = nullptr;  // memory mode: parse [mem, mem + len) in place (the text of a file from a record start on)
  GrowingText *grow = nullptr;  // memory mode over text that is still being written: [mem, mem + len) is what has arrived,
  size_t grow_base = 0;         // mem = grow->base + grow_base; more() waits for the producer
  size_t pos = 0, len = 0;  // unconsumed bytes: buf[pos, len)
  // producer side (prefetch)
  static constexpr size_t BLOCK = (size_t)4 << 20;
  struct Block { std::vector<char> data; size_t n = 0; };
  std::thread producer;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Block *> ready, spare;
  std::vector<Block> blocks;
  bool produced_all = false, stop = false;
  // BGZF input (blocked gzip as bgzip / htslib write it: every block a gzip member of at most 64 KiB with its size in an
  // extra field, RFC 1952 2.3.1.1 + the SAM specification 4.1): the blocks are independent deflate streams, so a team of
  // threads inflates them side by side -- a plain .gz file is one stream and one zlib inflate (1.7 M reads/s), a BGZF
  // file scales with the team.  The compressed file is mapped, cut into tasks of BGZF_TASK blocks at block boundaries
  // (a walk over the block headers), the tasks' outputs are handed on in file order through a ring of slots.
  static constexpr size_t BGZF_TASK = 64;   // blocks per task (~4 MiB of text)
  struct BgzfTask { size_t in_lo = 0, in_hi = 0; };
  struct BgzfSlot { std::vector<char> out; size_t n = 0; size_t task = (size_t)-1; bool done = false, bad = false; };
  const unsigned char *bz_map = nullptr;
  size_t bz_len = 0, bz_next_claim = 0, bz_next_out = 0;
  std::vector<BgzfTask> bz_tasks;
  std::vector<BgzfSlot> bz_slots;
  std::vector<std::thread> bz_team;

  // plain .gz input (ONE deflate stream): inflated by a team all the same, ku_pgzip.h -- spans of the compressed file are
  // decoded side by side from block starts found by search, their unknown 32 KiB of history resolved afterwards.  A
  // coordinator thread runs the rounds and hands their text on through two slots; zlib (gzread) remains for pipes,
  // small files and KU_NO_PGZIP=1.
  ku_pgzip::ParallelGunzip *pgz = nullptr;
  struct PgzSlot { ku_pgzip::RawBuf<char> text; size_t n = 0; bool full = false; };
  PgzSlot pgz_slot[2];
  size_t pgz_put = 0, pgz_get = 0;
  bool pgz_done = false;
  std::thread pgz_thread;
  const unsigned char *pgz_map = nullptr;
  size_t pgz_len = 0;

  // .bz2 input: the blocks of the file decoded by a team (ku_pbzip2.h); a pipe is read to its end first (the blocks can
  // only be found in memory), a regular file is mapped
  ku_pbzip2::ParallelBunzip2 *pbz = nullptr;
  const unsigned char *pbz_map = nullptr;
  size_t pbz_len = 0;
  std::vector<unsigned char> pbz_mem;
  static int bzip2_team() {
    int team = std::max(1, std::min((int)std::thread::hardware_concurrency(), 16));
    if (const char *e = getenv("KU_PBZIP2_TEAM")) team = std::max(1, std::min(atoi(e), 64));
    return team;
  }
  static bool file_is_bzip2(const char *path) {
    unsigned char h[14];
    const int f = ::open(path, O_RDONLY);
    if (f < 0) return false;
    const ssize_t got = ::pread(f, h, sizeof h, 0);
    ::close(f);
    return got == (ssize_t)sizeof h && ku_pbzip2::ParallelBunzip2::is_bzip2(h, sizeof h);
  }

  Reader() = default;
  Reader(const Reader &) = delete;
  Reader &operator=(const Reader &) = delete;
  ~Reader() { close(); }

  // zlib's reader ends a damaged stream (wrong CRC-32, broken deflate data, a file that stops in mid-member) with -1: that is
  // no end of file.  (It ended the input silently in rounds 1-4; the team readers have always refused such files.)
  std::atomic<bool> gz_failed{false};
  long raw_read(char *dst, size_t want) {
    if (fd >= 0) return (long)::read(fd, dst, want);
    const long n = (long)gzread(g, dst, (unsigned)want);
    if (n < 0) gz_failed = true;
    else if (n == 0) {  // zlib reports a file that stops in mid-member as a plain 0 with Z_BUF_ERROR pending: no end of file either
      int e = Z_OK;
      (void)gzerror(g, &e);
      if (e != Z_OK && e != Z_STREAM_END) gz_failed = true;
    }
    return n;
  }
  void check_gz() {
    if (gz_failed) fatal(65, "corrupt gzip data in the input (zlib: wrong CRC-32 / length, broken deflate stream or truncated file)");
  }
  // size of the BGZF block at p (0: not one)
  static size_t bgzf_block_size(const unsigned char *p, size_t avail) {
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = p[10] | ((size_t)p[11] << 8);
    if (12 + xlen > avail) return 0;
    for (size_t o = 12; o + 4 <= 12 + xlen;) {  // the subfields of the extra field: "BC", length 2, block size - 1
      const size_t slen = p[o + 2] | ((size_t)p[o + 3] << 8);
      if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) return (size_t)(p[o + 4] | ((size_t)p[o + 5] << 8)) + 1;
      o += 4 + slen;
    }
    return 0;
  }
  // one task: its blocks inflated into the slot's buffer (raw deflate payload between the member header and the 8-byte trailer)
  void bgzf_inflate(const BgzfTask &t, BgzfSlot &sl) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) { sl.bad = true; return; }
    if (sl.out.size() < BGZF_TASK * 65536) sl.out.resize(BGZF_TASK * 65536);
    sl.n = 0;
    for (size_t p = t.in_lo; p < t.in_hi;) {
      const size_t bs = bgzf_block_size(bz_map + p, bz_len - p);
      if (bs < 26 || p + bs > t.in_hi) { sl.bad = true; break; }
      const size_t xlen = bz_map[p + 10] | ((size_t)bz_map[p + 11] << 8);
      if (bs < 12 + xlen + 8) { sl.bad = true; break; }  // (an extra field that claims more than the block holds)
      const size_t isize = (size_t)bz_map[p + bs - 4] | ((size_t)bz_map[p + bs - 3] << 8) | ((size_t)bz_map[p + bs - 2] << 16) | ((size_t)bz_map[p + bs - 1] << 24);
      const uint32_t crc = (uint32_t)bz_map[p + bs - 8] | ((uint32_t)bz_map[p + bs - 7] << 8) | ((uint32_t)bz_map[p + bs - 6] << 16) | ((uint32_t)bz_map[p + bs - 5] << 24);
      if (isize > 65536 || sl.n + isize > sl.out.size()) { sl.bad = true; break; }
      z.next_in = const_cast<unsigned char *>(bz_map + p + 12 + xlen);
      z.avail_in = (unsigned)(bs - 12 - xlen - 8);
      z.next_out = (unsigned char *)sl.out.data() + sl.n;
      z.avail_out = (unsigned)isize;
      const int rc = isize ? inflate(&z, Z_FINISH) : Z_STREAM_END;
      // the block's CRC-32, as zlib's gzread checks it (and GzTextStream::inflate_task on the region path): a damaged block
      // must not be classified silently (a flipped byte in a stored block went through with -P / -t 1)
      if (rc != Z_STREAM_END || z.avail_out != 0 || (uint32_t)ku_pgzip::crc_of((const uint8_t *)sl.out.data() + sl.n, isize) != crc) { sl.bad = true; break; }
      sl.n += isize;
      inflateReset(&z);
      p += bs;
    }
    inflateEnd(&z);
  }
  // BGZF from its first to its last byte?  Then: map, task list, team.  false: leave it to zlib's gzread
  bool open_bgzf(const char *path, size_t n) {
    if (n < 28) return false;
    const int f = ::open(path, O_RDONLY);
    if (f < 0) return false;
    void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, f, 0);
    ::close(f);
    if (m == MAP_FAILED) return false;
    const unsigned char *p = (const unsigned char *)m;
    std::vector<BgzfTask> tasks;
    size_t at = 0, in_task = 0;
    BgzfTask cur{0, 0};
    while (at < n) {
      const size_t bs = bgzf_block_size(p + at, n - at);
      // (a plain gzip member in between, or an extra field longer than its block: not for this path)
      if (bs < 26 || at + bs > n || bs < 12 + ((size_t)p[at + 10] | ((size_t)p[at + 11] << 8)) + 8) { munmap(m, n); return false; }
      at += bs;
      if (++in_task == BGZF_TASK || at == n) { cur.in_hi = at; tasks.push_back(cur); cur.in_lo = at; in_task = 0; }
    }
    bz_map = p;
    bz_len = n;
    bz_tasks.swap(tasks);
    bz_next_claim = bz_next_out = 0;
    int team = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("KU_BGZF_TEAM")) team = atoi(e);
    team = std::max(1, std::min(team, 8));
    bz_slots.assign((size_t)2 * team, BgzfSlot());
    for (int t = 0; t < team; ++t)
      bz_team.emplace_back([this] {
        for (;;) {
          size_t ti;
          BgzfSlot *sl;
          {
            std::unique_lock<std::mutex> l(mu);
            // a task is claimed when its slot is free: task ti uses slot ti % slots, free once task ti - slots went out
            cv.wait(l, [&] { return stop || bz_next_claim >= bz_tasks.size() || bz_next_claim < bz_next_out + bz_slots.size(); });
            if (stop || bz_next_claim >= bz_tasks.size()) return;
            ti = bz_next_claim++;
            sl = &bz_slots[ti % bz_slots.size()];
            sl->task = ti;
            sl->done = false;
          }
          bgzf_inflate(bz_tasks[ti], *sl);
          { std::lock_guard<std::mutex> l(mu); sl->done = true; }
          cv.notify_all();
        }
      });
    gzclose(g);
    g = nullptr;
    return true;
  }
  bool open_pgzip(const char *path, size_t n) {
    if (n < ((size_t)64 << 10)) return false;
    const int f = ::open(path, O_RDONLY);
    if (f < 0) return false;
    void *mp = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, f, 0);
    ::close(f);
    if (mp == MAP_FAILED) return false;
    const unsigned char *p = (const unsigned char *)mp;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) { munmap(mp, n); return false; }
    int team = (int)std::thread::hardware_concurrency();
    team = std::max(1, std::min(team, 8));
    if (const char *e = getenv("KU_PGZIP_TEAM")) team = std::max(1, std::min(atoi(e), 64));
    pgz_map = p;
    pgz_len = n;
    pgz = new ku_pgzip::ParallelGunzip;
    pgz->open(p, n, team);
    pgz_put = pgz_get = 0;
    pgz_done = false;
    pgz_slot[0].full = pgz_slot[1].full = false;
    pgz_thread = std::thread([this] {
      for (;;) {
        PgzSlot *sl;
        {
          std::unique_lock<std::mutex> l(mu);
          cv.wait(l, [&] { return stop || !pgz_slot[pgz_put & 1].full; });
          if (stop) return;
          sl = &pgz_slot[pgz_put & 1];
        }
        size_t got = 0;
        const bool more = pgz->round(sl->text, got);
        {
          std::lock_guard<std::mutex> l(mu);
          if (!more) pgz_done = true;
          else { sl->n = got; sl->full = true; ++pgz_put; }
        }
        cv.notify_all();
        if (!more) return;
      }
    });
    gzclose(g);
    g = nullptr;
    return true;
  }
  void open(const char *path, bool prefetch = false) {
    g = gzopen(path, "rb");
    if (!g) fatal(66, "can't open %s", path);
    gzbuffer(g, 1 << 20);
    fd = -1;
    struct stat st;
    memset(&st, 0, sizeof st);
    if (::stat(path, &st) == 0 && S_ISREG(st.st_mode) && gzdirect(g)) {  // a regular file without gzip data: bypass zlib
      fd = ::open(path, O_RDONLY);  // (a pipe such as <(cat library/*.fna) must stay with the one reader that opened it)
      if (fd >= 0) { gzclose(g); g = nullptr; }
    }
    buf.resize((size_t)1 << 24);
    pos = len = 0;
    valid = true; eof = false;
    first_record = true;
    produced_all = stop = false;
    if (fd >= 0 && file_is_bzip2(path)) {  // a regular .bz2 file: mapped
      void *mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (mp == MAP_FAILED) fatal(71, "can't map %s", path);
      ::close(fd);
      fd = -1;
      pbz_map = (const unsigned char *)mp;
      pbz_len = (size_t)st.st_size;
      pbz = new ku_pbzip2::ParallelBunzip2;
      pbz->open(pbz_map, pbz_len, bzip2_team());
      more();
      fastq = len > 0 && buf[0] == '@';
      return;
    }
    if (g && gzdirect(g)) {  // uncompressed bytes through a pipe -- or bzip2 ones: the first bytes tell
      const long got = gzread(g, buf.data(), 14);
      len = got > 0 ? (size_t)got : 0;
      if (ku_pbzip2::ParallelBunzip2::is_bzip2((const unsigned char *)buf.data(), len)) {
        pbz_mem.assign((const unsigned char *)buf.data(), (const unsigned char *)buf.data() + len);
        len = 0;
        for (;;) {
          const size_t at = pbz_mem.size();
          pbz_mem.resize(at + ((size_t)16 << 20));
          const long r = gzread(g, pbz_mem.data() + at, (unsigned)((size_t)16 << 20));
          pbz_mem.resize(at + (r > 0 ? (size_t)r : 0));
          if (r <= 0) break;
        }
        gzclose(g);
        g = nullptr;
        pbz = new ku_pbzip2::ParallelBunzip2;
        pbz->open(pbz_mem.data(), pbz_mem.size(), bzip2_team());
        more();
        fastq = len > 0 && buf[0] == '@';
        return;
      }
    }
    if (prefetch && g && S_ISREG(st.st_mode) && !getenv("KU_NO_BGZF") && open_bgzf(path, (size_t)st.st_size)) {
      more();
      fastq = len > 0 && buf[0] == '@';
      return;
    }
    if (prefetch && g && S_ISREG(st.st_mode) && !getenv("KU_NO_PGZIP") && open_pgzip(path, (size_t)st.st_size)) {
      more();
      fastq = len > 0 && buf[0] == '@';
      return;
    }
    if (prefetch) {
      blocks.resize(4);
      for (Block &b : blocks) { b.data.resize(BLOCK); spare.push_back(&b); }
      producer = std::thread([this] {
        for (;;) {
          Block *b;
          {
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return stop || !spare.empty(); });
            if (stop) return;
            b = spare.front();
            spare.pop_front();
          }
          long n = raw_read(b->data.data(), BLOCK);
          std::lock_guard<std::mutex> l(mu);
          if (n <= 0) { spare.push_back(b); produced_all = true; cv.notify_all(); return; }
          b->n = (size_t)n;
          ready.push_back(b);
          cv.notify_all();
        }
      });
    }
    more();
    fastq = len > 0 && buf[0] == '@';  // determine_input_file_type (src/classify.cpp:377-388)
  }
  // parse text that already sits in memory, from a record start to the end of the file: [p, p + n)
  void open_memory(const char *p, size_t n, bool is_fastq, bool file_start = true) {
    mem = p;
    pos = 0; len = n;
    valid = true; eof = true;
    fastq = is_fastq;
    first_record = file_start;
  }
  // the same over a text that is still growing: from offset `at` of gt, of which `have` bytes (absolute) are known to be there
  void open_growing(GrowingText *gt, size_t at, size_t have, bool is_fastq, bool file_start) {
    open_memory(gt->base + at, have > at ? have - at : 0, is_fastq, file_start);
    grow = gt;
    grow_base = at;
    eof = false;
  }
  void close() {
    mem = nullptr;
    grow = nullptr;
    if (!bz_team.empty()) {
      { std::lock_guard<std::mutex> l(mu); stop = true; }
      cv.notify_all();
      for (auto &t : bz_team) t.join();
      bz_team.clear();
    }
    if (pbz) { pbz->close(); delete pbz; pbz = nullptr; }
    if (pbz_map) munmap((void *)pbz_map, pbz_len);
    pbz_map = nullptr;
    std::vector<unsigned char>().swap(pbz_mem);
    if (pgz_thread.joinable()) {
      { std::lock_guard<std::mutex> l(mu); stop = true; }
      cv.notify_all();
      pgz_thread.join();
    }
    delete pgz;
    pgz = nullptr;
    if (pgz_map) munmap((void *)pgz_map, pgz_len);
    pgz_map = nullptr;
    for (PgzSlot &sl : pgz_slot) { sl.full = false; sl.n = 0; }
    if (bz_map) munmap((void *)bz_map, bz_len);
    bz_map = nullptr;
    bz_tasks.clear(); bz_slots.clear();
    if (producer.joinable()) {
      { std::lock_guard<std::mutex> l(mu); stop = true; }
      cv.notify_all();
      producer.join();
    }
    ready.clear(); spare.clear(); blocks.clear();
    if (g) gzclose(g);
    if (fd >= 0) ::close(fd);
    g = nullptr; fd = -1;
  }
  // append file data behind the unconsumed bytes (which move to the front); false at end of file
  bool more() {
    if (eof) return false;
    if (grow) {  // memory mode over growing text: nothing moves, the window's end does
      bool complete = false;
      const size_t avail = grow->wait_for(grow_base + len + 1, &complete);
      if (avail > grow_base + len) { len = avail - grow_base; return true; }
      eof = true;
      return false;
    }
    if (pos > 0) {
      if (len > pos) memmove(buf.data(), buf.data() + pos, len - pos);
      len -= pos;
      pos = 0;
    }
    if (pbz) {  // the next block's text
      const uint8_t *blk;
      size_t nb = 0;
      if (!pbz->next(blk, nb)) {
        if (!pbz->error.empty()) fatal(65, "%s", pbz->error.c_str());
        eof = true;
        return false;
      }
      if (len + nb > buf.size()) buf.resize(std::max(buf.size() * 2, len + nb));
      memcpy(buf.data() + len, blk, nb);
      len += nb;
      return nb > 0 || more();
    }
    if (pgz) {  // the next round's text
      PgzSlot &sl = pgz_slot[pgz_get & 1];
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return sl.full || pgz_done; });
        if (!sl.full) {
          if (!pgz->error.empty()) fatal(65, "%s", pgz->error.c_str());
          eof = true;
          return false;
        }
      }
      if (len + sl.n > buf.size()) buf.resize(std::max(buf.size() * 2, len + sl.n));
      memcpy(buf.data() + len, sl.text.d, sl.n);
      len += sl.n;
      const bool any = sl.n > 0;
      { std::lock_guard<std::mutex> l(mu); sl.full = false; ++pgz_get; }
      cv.notify_all();
      return any || more();
    }
    if (bz_map) {  // the next task's text, in file order
      if (bz_next_out >= bz_tasks.size()) { eof = true; return false; }
      BgzfSlot &sl = bz_slots[bz_next_out % bz_slots.size()];
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return sl.task == bz_next_out && sl.done; });
      }
      if (sl.bad) fatal(65, "corrupt BGZF block in the input (deflate stream or block header)");
      if (len + sl.n > buf.size()) buf.resize(std::max(buf.size() * 2, len + sl.n));
      memcpy(buf.data() + len, sl.out.data(), sl.n);
      len += sl.n;
      const bool any = sl.n > 0;
      { std::lock_guard<std::mutex> l(mu); ++bz_next_out; }
      cv.notify_all();
      return any || more();  // (an empty task -- the 28-byte end-of-file block alone -- is no data)
    }
    if (producer.joinable()) {
      Block *b = nullptr;
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return !ready.empty() || produced_all; });
        if (!ready.empty()) { b = ready.front(); ready.pop_front(); }
      }
      if (!b) { check_gz(); eof = true; return false; }
      if (len + b->n > buf.size()) buf.resize(std::max(buf.size() * 2, len + b->n));
      memcpy(buf.data() + len, b->data.data(), b->n);
      len += b->n;
      { std::lock_guard<std::mutex> l(mu); spare.push_back(b); }
      cv.notify_all();
      return true;
    }
    if (len == buf.size()) buf.resize(buf.size() * 2);
    const size_t room = buf.size() - len;
    const size_t want = room < ((size_t)1 << 30) ? room : ((size_t)1 << 30);
    long n = raw_read(buf.data() + len, want);
    if (n <= 0) { check_gz(); eof = true; return false; }
    len += (size_t)n;
    return true;
  }
  // line starting `from` bytes behind pos: [lo, hi) without the '\n', `next` = start of the following line (all
  // relative to pos, so they survive more()).  false when the file ends before `from` (std::getline's failure).
  bool line_at(size_t from, size_t &hi, size_t &next) {
    for (;;) {
      const size_t avail = len - pos > from ? len - pos - from : 0;
      const char *b = (mem ? mem : buf.data()) + pos + from;
      const char *nl = avail ? (const char *)memchr(b, '\n', avail) : nullptr;
      if (nl) { hi = from + (size_t)(nl - b); next = hi + 1; return true; }
      if (!more()) {
        const size_t rest = len - pos > from ? len - pos - from : 0;
        if (rest == 0) return false;
        hi = from + rest; next = hi;
        return true;
      }
    }
  }
  const char *at(size_t rel) const { return (mem ? mem : buf.data()) + pos + rel; }
};

// id = first whitespace-delimited token of the header line ("istringstream >> id", src/seqreader.cpp:56-58,114-116)
inline void split_id(const char *h, size_t n, size_t &lo, size_t &hi) {
  auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
  lo = 0;
  while (lo < n && ws(h[lo])) ++lo;
  hi = lo;
  while (hi < n && !ws(h[hi])) ++hi;
}

// Multi-line FASTA with CRLF line ends.  The reference keeps every '\r' in the sequence string (getline,
// src/seqreader.cpp:60-75) and its scanner steps over them (src/krakenutil.cpp:266-270) -- but that step also loses
// the k-mer ending at the first base of the next line, and the reported length counts the '\r's.  Here the line ends
// inside a sequence are dropped and every k-mer is classified; said once, because the per-read output then differs
// from the reference's for such files (FASTQ and one-line-per-sequence FASTA are byte-identical, tests/golden/f10).
// KU_CRLF_REFERENCE=1 reproduces the reference's Kraken line for such reads (round 5): the parser notes where the lines began,
// and the output stage rewrites the line -- the k-mer ending on the first base behind every line break is taken out of the hit
// list (and, near the start of a read, the k-mers the scanner had not counted enough bases for yet: it does not count that
// base, src/krakenutil.cpp:240-249,265-269), the length column counts the carriage returns.  What stays as computed from
// every k-mer: the call and the per-taxon accounting (the reference would not count the dropped k-mers either).
inline bool crlf_reference() {
  static const bool on = getenv("KU_CRLF_REFERENCE") && atoi(getenv("KU_CRLF_REFERENCE"));
  return on;
}
inline void crlf_note() {
  static std::atomic<bool> said{false};
  if (!said.exchange(true) && !crlf_reference())
    fprintf(stderr, "classify: warning: multi-line FASTA with CRLF line ends: the carriage returns inside sequences are "
                    "removed (the reference skips one k-mer per line break on such input; KU_CRLF_REFERENCE=1 prints its lines)\n");
}

// One record of `rd`: sequence pieces appended to the read currently open in `bt`; header (without '>'/'@') and
// quality line to `header` / `quals` when wanted.  false when the stream is exhausted or malformed
// (reader->is_valid() == false); a malformed record ends the stream with the reference's warning.
inline bool next_record(Reader &rd, Batch &bt, std::string *header, std::string *quals, size_t *seq_bytes) {
  size_t h_hi, h_next;
  *seq_bytes = 0;
  if (rd.fastq) {  // FastqReader::next_sequence (src/seqreader.cpp:96-133)
    if (!rd.valid || !rd.line_at(0, h_hi, h_next) || h_hi == 0) { rd.valid = false; return false; }
    if (*rd.at(0) != '@') {
      if (*rd.at(0) != '\r') fprintf(stderr, "classify: malformed fastq file - sequence header (%.*s)\n", (int)h_hi, rd.at(0));
      rd.valid = false;
      return false;
    }
    size_t s_hi = h_next, s_next = h_next, p_hi = h_next, p_next = h_next, q_hi, q_next;
    const bool has_seq = rd.line_at(h_next, s_hi, s_next);
    const bool has_plus = has_seq && rd.line_at(s_next, p_hi, p_next);
    if (!has_plus || p_hi == s_next || *rd.at(s_next) != '+') {
      if (has_plus && p_hi > s_next && *rd.at(s_next) != '\r')
        fprintf(stderr, "classify: malformed fastq file - quality header (%.*s)\n", (int)(p_hi - s_next), rd.at(s_next));
      else if (!has_plus || p_hi == s_next)
        fprintf(stderr, "classify: malformed fastq file - quality header ()\n");
      rd.valid = false;
      return false;
    }
    const bool has_q = rd.line_at(p_next, q_hi, q_next);
    if (!has_q) { q_hi = p_next; q_next = p_next; }
    if (header) header->assign(rd.at(1), h_hi - 1);
    bt.append(rd.at(h_next), s_hi - h_next);
    *seq_bytes = s_hi - h_next;
    if (quals) quals->assign(rd.at(p_next), q_hi - p_next);
    rd.pos += q_next;
    return true;
  }
  // FastaReader::next_sequence (src/seqreader.cpp:34-80)
  if (!rd.valid || !rd.line_at(0, h_hi, h_next)) { rd.valid = false; return false; }
  if (h_hi == 0 || *rd.at(0) != '>') {
    fprintf(stderr, "classify: malformed fasta file - expected header char > not found\n");
    rd.valid = false;
    return false;
  }
  // a header line the file ends in, without a line end: only the first record of a file survives that (see first_record)
  if (h_next == h_hi && !rd.first_record) { rd.valid = false; return false; }
  rd.first_record = false;
  if (header) header->assign(rd.at(1), h_hi - 1);
  rd.pos += h_next;
  size_t l_hi, l_next;
  const size_t read_lo = bt.seqs_len;
  while (rd.line_at(0, l_hi, l_next)) {
    if (l_hi > 0 && *rd.at(0) == '>') break;  // next record: not consumed
    // CRLF files: the '\r' closing a line that another sequence line follows is dropped (see crlf_note()); the one
    // closing the record stays -- it is what the reference scans too (one more, ambiguous, k-mer; length + 1)
    if (l_hi > 0 && bt.seqs_len > read_lo && bt.seqs[bt.seqs_len - 1] == '\r') {
      --bt.seqs_len;
      crlf_note();
      if (crlf_reference() && !bt.off.empty()) {  // (the read in progress is the one begin_read() opened last)
        const uint32_t r = (uint32_t)bt.off.size() - 1;
        if (bt.crlf_read.empty() || bt.crlf_read.back() != r) { bt.crlf_read.push_back(r); bt.crlf_off.push_back((uint32_t)bt.crlf_start.size()); }
        bt.crlf_start.push_back((uint32_t)(bt.seqs_len - read_lo));
      }
    }
    bt.append(rd.at(0), l_hi);
    *seq_bytes += l_hi;
    rd.pos += l_next;
  }
  return true;
}

// First record start at or behind offset x of a file image [data, data + n) whose offset 0 starts a record; n when
// there is none.  FASTA: a line starting with '>'.  FASTQ (4-line records as the reference reads them): a line
// starting with '@' whose second successor starts with '+' -- a quality line may start with '@' too, but then the
// second line behind it is a sequence line, which never starts with '+'.
inline size_t find_record_start(const char *data, size_t n, size_t x, bool fastq) {
  if (x == 0) return 0;
  if (x >= n) return n;
  const char *nl = (const char *)memchr(data + x - 1, '\n', n - (x - 1));  // x itself starts a line if data[x-1] == '\n'
  size_t p = nl ? (size_t)(nl - data) + 1 : n;
  while (p < n) {
    if (!fastq) {
      if (data[p] == '>') return p;
    } else if (data[p] == '@') {
      const char *l1 = (const char *)memchr(data + p, '\n', n - p);
      const char *l2 = l1 ? (const char *)memchr(l1 + 1, '\n', n - (size_t)(l1 + 1 - data)) : nullptr;
      if (!l2 || (size_t)(l2 + 1 - data) >= n) return p;  // the tail of the file: let the parser judge it
      if (l2[1] == '+') return p;
    }
    const char *e = (const char *)memchr(data + p, '\n', n - p);
    p = e ? (size_t)(e - data) + 1 : n;
  }
  return n;
}

// Every record of the record-aligned region [data, data + n) of a plain-text file into `bt`.  false when the stream
// ends inside the region (malformed record / empty FASTQ line: the reference stops reading there).
// FASTQ records laid out the way sequencers write them -- "@header\n sequence\n +...\n quality\n" -- straight from the
// region into the batch: two line searches per record (the '+' line is mostly two bytes, the quality line mostly as long as
// the sequence: looked at where they should end before they are searched), the id cut out of the header in place, no
// std::string round trip.  Stops in front of the first record that is anything else (a missing '+', an empty line, the
// ragged end of the file) and returns how far it got: the general parser takes it from there, with the reference's
// diagnostics.  8.4 -> 16 M reads/s per thread on the build container (scripts/seqio_rate.sh, -j 1 -w).
inline size_t parse_fastq_fast(const char *data, size_t n, Batch &bt, bool keep_records) {
  const char *p = data, *const end = data + n;
  bt.reserve_seq(n / 2 + 64);
  while (p < end) {
    if (*p != '@') break;
    const char *h_end = (const char *)memchr(p, '\n', (size_t)(end - p));
    if (!h_end || h_end == p + 0) break;
    const char *sq = h_end + 1;
    const char *s_end = sq < end ? (const char *)memchr(sq, '\n', (size_t)(end - sq)) : nullptr;
    if (!s_end) break;
    const char *pl = s_end + 1;
    if (pl >= end || *pl != '+') break;
    const char *pl_end = (pl + 1 < end && pl[1] == '\n') ? pl + 1 : (const char *)memchr(pl, '\n', (size_t)(end - pl));
    if (!pl_end) break;
    const char *q = pl_end + 1;
    const size_t L = (size_t)(s_end - sq);
    const char *q_end = (q + L < end && q[L] == '\n' && (L == 0 || !memchr(q, '\n', L))) ? q + L : (q < end ? (const char *)memchr(q, '\n', (size_t)(end - q)) : nullptr);
    if (!q_end) break;  // the last line of the file without its line end: the general parser's case
    // ---- the record
    bt.off.push_back(bt.seqs_len);
    bt.reserve_seq(L + 1);
    memcpy(bt.seqs + bt.seqs_len, sq, L);
    bt.seqs_len += L;
    bt.len.push_back((uint32_t)L);
    bt.nt += L;
    bt.seqs[bt.seqs_len++] = '\n';
    const char *h = p + 1;
    size_t lo, hi;
    split_id(h, (size_t)(h_end - h), lo, hi);
    bt.idoff.push_back(bt.ids.size());
    bt.ids.append(h + lo, hi - lo);
    bt.ids.push_back('\0');
    if (keep_records) {
      bt.hoff.push_back(bt.headers.size()); bt.headers.append(h, (size_t)(h_end - h)); bt.headers.push_back('\0');
      bt.qoff.push_back(bt.quals.size()); bt.quals.append(q, (size_t)(q_end - q)); bt.quals.push_back('\0');
    }
    p = q_end + 1;
  }
  return (size_t)(p - data);
}

// The records that START inside [lo, hi) of a text, parsed as the sequential reader would parse them from `lo` on -- `lo` must be
// a position that reader reaches as a record start; `hi` is where the next region was cut, a record start by the cutter's
// judgement (find_record_start) -- exact for FASTA, a heuristic for FASTQ that damaged files defeat.  So the parser is not
// confined to the region: the text is visible to its end (`text_n` bytes of `text`, or all that `gt` will ever deliver), a record
// that starts before `hi` is read to ITS end wherever that lies, and the caller learns where the parse stopped.  Regions chain:
// region i + 1 counts iff region i stopped exactly at its start (RegionChain below); otherwise the stretch is parsed again from
// where region i stopped.  (Round 5 handed the parser [lo, hi) alone: a region cut inside a record then produced records the
// reference never sees -- `seqio_dump -j 3` on a FASTQ file with a deleted sequence line printed a quality string as an id.)
struct RegionParse {
  bool ended = false;  // the stream ended inside the region (malformed record, an empty FASTQ line, the end of the file):
                       // nothing behind `end` is ever read by the reference (src/seqreader.cpp:51-55,103-112)
  size_t end = 0;      // where the parse stopped (absolute offset in the text; >= hi unless `ended`)
};
inline RegionParse parse_region(const char *text, size_t text_n, GrowingText *gt, size_t lo, size_t hi, bool fastq, Batch &bt,
                                bool keep_records) {
  RegionParse res;
  size_t at = lo;
  if (fastq && !getenv("KU_SEQIO_GENERAL")) {  // whole four-line records inside [lo, hi): the fast path
    at += parse_fastq_fast(text + lo, hi - lo, bt, keep_records);
    if (at == hi) { res.end = hi; return res; }
  }
  Reader rd;
  if (gt) rd.open_growing(gt, at, hi, fastq, at == 0);
  else rd.open_memory(text + at, text_n - at, fastq, at == 0);
  std::string header, quals;
  size_t nb, id_lo, id_hi;
  while (at + rd.pos < hi) {
    bt.begin_read();
    if (!next_record(rd, bt, &header, keep_records ? &quals : nullptr, &nb)) { bt.off.pop_back(); res.ended = true; break; }
    bt.end_read();
    split_id(header.data(), header.size(), id_lo, id_hi);
    bt.add_meta(header, id_lo, id_hi, quals, keep_records);
  }
  res.end = at + rd.pos;
  return res;
}

// Work units at the reference's granularity (src/classify.cpp:510-523): a unit takes reads until it holds Work_unit_size
// nucleotides; a unit WITHOUT nucleotides ends the file's processing and its reads are never classified or printed -- it can
// only be the file's last unit, made of empty records.  Batches of one file pass through in order; a batch whose tail is such a
// run of empty reads in a unit that holds no nucleotide yet is held back until the next batch (or the end of the file) decides.
struct UnitGate {
  uint64_t unit_nt = 500000;  // Work_unit_size (-u)
  uint64_t acc = 0;           // nucleotides of the open unit
  struct Held { Batch *bt; size_t keep; };  // reads [keep, n) of bt are empty and belong to the open unit, which holds no nucleotide
  std::vector<Held> held;     // (non-empty only while acc == 0)
  void begin_file() { acc = 0; held.clear(); }
  // `out(bt)`: the batch goes on as it is; `recycle(bt)`: its reads went into a batch that is held already (so that a long
  // stretch of empty records cannot take every batch of a caller's pool out of circulation)
  template <class Out, class Recycle> void push(Batch *bt, Out &&out, Recycle &&recycle) {
    const size_t n = bt->off.size();
    if (bt->nt == 0) {  // empty reads only: they join the open unit
      if (acc > 0 || n == 0) out(bt);
      else if (held.empty()) held.push_back(Held{bt, 0});
      else {
        Batch *dst = held.back().bt;
        const bool records = !dst->hoff.empty() || !bt->hoff.empty();
        const std::string none;
        for (size_t i = 0; i < n; ++i) {
          dst->begin_read();
          dst->end_read();
          const std::string id(bt->ids.c_str() + bt->idoff[i]);
          if (records) {
            const std::string hdr(bt->headers.c_str() + bt->hoff[i]), q(bt->quals.c_str() + bt->qoff[i]);
            dst->idoff.push_back(dst->ids.size());
            dst->ids += id; dst->ids.push_back('\0');
            dst->hoff.push_back(dst->headers.size()); dst->headers += hdr; dst->headers.push_back('\0');
            dst->qoff.push_back(dst->quals.size()); dst->quals += q; dst->quals.push_back('\0');
          } else dst->add_meta(id, 0, id.size(), none, false);
        }
        recycle(bt);
      }
      return;
    }
    // these nucleotides share a unit with the reads held so far
    for (Held &h : held) out(h.bt);
    held.clear();
    size_t last_close = 0;
    uint64_t a = acc;
    const uint32_t *len = bt->len.data();
    for (size_t i = 0; i < n; ++i) {
      a += len[i];
      if (a >= unit_nt) { a = 0; last_close = i + 1; }
    }
    acc = a;
    if (a == 0 && last_close < n) held.push_back(Held{bt, last_close});  // a unit closed, empty reads follow it
    else out(bt);
  }
  // end of the file's stream: the held reads are a unit without nucleotides -- never classified, never printed.  `out(bt)` for a
  // batch that keeps its front part, `drop(bt)` for one that is left with no reads
  template <class Out, class Drop> void finish(Out &&out, Drop &&drop) {
    for (Held &h : held) {
      if (h.keep == 0) { drop(h.bt); continue; }
      Batch *bt = h.bt;
      bt->seqs_len = bt->off[h.keep];  // (the dropped reads are empty: only their separators go)
      bt->off.resize(h.keep); bt->len.resize(h.keep); bt->idoff.resize(h.keep);
      if (!bt->hoff.empty()) { bt->hoff.resize(h.keep); bt->qoff.resize(h.keep); }
      out(bt);
    }
    held.clear();
    acc = 0;
  }
};

// The chain of regions of one file, in order: region i + 1 counts iff the parse of region i stopped exactly at its start.
struct RegionChain {
  size_t expect = 0;   // where the sequential parse stands
  bool ended = false;  // the stream has ended: nothing behind counts
  // what to do with the region [lo, hi) whose parse (started at lo) gave `r`:
  enum Verdict { ACCEPT, REPARSE, SKIP };
  //   ACCEPT  -- its records count;  REPARSE -- it was cut inside a record: parse [expect, hi) again, then call accept() with that
  //   result;  SKIP -- the record before it ran past its end (or the stream has ended): its records do not exist
  Verdict judge(size_t lo, size_t hi, const RegionParse &r) {
    static const bool say = getenv("KU_SEQIO_DEBUG") != nullptr;
    if (ended || expect >= hi) { if (say) fprintf(stderr, "regions: [%zu, %zu) skipped (the parse stands at %zu%s)\n", lo, hi, expect, ended ? ", ended" : ""); return SKIP; }
    if (lo != expect) { if (say) fprintf(stderr, "regions: [%zu, %zu) was cut inside a record: parsed again from %zu\n", lo, hi, expect); return REPARSE; }
    accept(r);
    return ACCEPT;
  }
  void accept(const RegionParse &r) { expect = r.end; ended |= r.ended; }
};

// the producer side: a regular .gz file (BGZF, or any gzip stream through ku_pgzip.h) or .bz2 file (ku_pbzip2.h) inflated
// into a GrowingText
struct GzTextStream {
  GrowingText *gt = nullptr;
  const unsigned char *map = nullptr;
  size_t len = 0;
  ku_pgzip::ParallelGunzip *pgz = nullptr;
  ku_pbzip2::ParallelBunzip2 *pbz = nullptr;
  std::thread coord;
  struct Task { size_t in_lo, in_hi, out_off, out_len; };
  std::vector<Task> tasks;
  std::vector<char> task_done;
  std::vector<std::thread> team;
  std::mutex tm;
  size_t next_claim = 0, next_pub = 0;
  bool bad = false;
  GzTextStream() = default;
  GzTextStream(const GzTextStream &) = delete;
  GzTextStream &operator=(const GzTextStream &) = delete;
  ~GzTextStream() { close(); }

  bool plan_bgzf() {
    size_t at = 0, in_task = 0, out = 0;
    Task cur{0, 0, 0, 0};
    while (at < len) {
      const size_t bs = Reader::bgzf_block_size(map + at, len - at);
      if (bs < 26 || at + bs > len) return false;
      const size_t isize = (size_t)map[at + bs - 4] | ((size_t)map[at + bs - 3] << 8) | ((size_t)map[at + bs - 2] << 16) | ((size_t)map[at + bs - 1] << 24);
      if (isize > 65536) return false;
      at += bs;
      out += isize;
      if (++in_task == Reader::BGZF_TASK || at == len) {
        cur.in_hi = at;
        cur.out_len = out - cur.out_off;
        tasks.push_back(cur);
        cur = Task{at, 0, out, 0};
        in_task = 0;
      }
    }
    return !tasks.empty();
  }
  bool inflate_task(const Task &t, char *dst) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) return false;
    bool ok = true;
    size_t o = 0;
    for (size_t p = t.in_lo; p < t.in_hi && ok;) {
      const size_t bs = Reader::bgzf_block_size(map + p, len - p);
      const size_t xlen = map[p + 10] | ((size_t)map[p + 11] << 8);
      const size_t isize = (size_t)map[p + bs - 4] | ((size_t)map[p + bs - 3] << 8) | ((size_t)map[p + bs - 2] << 16) | ((size_t)map[p + bs - 1] << 24);
      const uint32_t crc = (uint32_t)map[p + bs - 8] | ((uint32_t)map[p + bs - 7] << 8) | ((uint32_t)map[p + bs - 6] << 16) | ((uint32_t)map[p + bs - 5] << 24);
      if (bs < 12 + xlen + 8 || o + isize > t.out_len) { ok = false; break; }
      z.next_in = const_cast<unsigned char *>(map + p + 12 + xlen);
      z.avail_in = (unsigned)(bs - 12 - xlen - 8);
      z.next_out = (unsigned char *)dst + o;
      z.avail_out = (unsigned)isize;
      const int rc = isize ? inflate(&z, Z_FINISH) : Z_STREAM_END;
      if (rc != Z_STREAM_END || z.avail_out != 0 || (uint32_t)ku_pgzip::crc_of((const uint8_t *)dst + o, isize) != crc) ok = false;
      o += isize;
      inflateReset(&z);
      p += bs;
    }
    inflateEnd(&z);
    return ok && o == t.out_len;
  }

  // false: not a regular gzip file (or no room to reserve): the sequential reader takes it
  bool open(const char *path, GrowingText &text) {
    struct stat st;
    if (::stat(path, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 20) return false;
    const int f = ::open(path, O_RDONLY);
    if (f < 0) return false;
    len = (size_t)st.st_size;
    void *mp = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, f, 0);
    ::close(f);
    if (mp == MAP_FAILED) return false;
    map = (const unsigned char *)mp;
    const bool bz2 = ku_pbzip2::ParallelBunzip2::is_bzip2(map, len);
    if (!bz2 && (map[0] != 0x1f || map[1] != 0x8b || map[2] != 8)) { munmap(mp, len); map = nullptr; return false; }
    // virtual room for the text: deflate expands at most 1032-fold; nothing of it is touched before it is written
    const size_t room = std::min((size_t)8 << 40, std::max((size_t)1 << 30, len * 1100));
    if (!text.reserve(room)) { munmap(mp, len); map = nullptr; return false; }
    gt = &text;
    int hw = (int)std::thread::hardware_concurrency();
    if (bz2) {  // the blocks of a .bz2 file, decoded by a team, copied behind one another in file order
      pbz = new ku_pbzip2::ParallelBunzip2;
      pbz->open(map, len, Reader::bzip2_team());
      coord = std::thread([this] {
        const uint8_t *blk;
        size_t nb = 0;
        std::string err;
        while (pbz->next(blk, nb)) {
          char *dst = gt->place(nb);
          if (!dst) { err = "no room for the text"; break; }
          memcpy(dst, blk, nb);
          gt->publish(nb);
        }
        gt->finish(err.empty() ? pbz->error : err);
      });
      return true;
    }
    if (!getenv("KU_NO_BGZF") && plan_bgzf()) {
      int n = std::max(1, std::min(hw, 8));
      if (const char *e = getenv("KU_BGZF_TEAM")) n = std::max(1, std::min(atoi(e), 64));
      task_done.assign(tasks.size(), 0);
      for (int t = 0; t < n; ++t)
        team.emplace_back([this] {
          for (;;) {
            size_t ti;
            {
              std::lock_guard<std::mutex> l(tm);
              if (bad || next_claim >= tasks.size()) return;
              ti = next_claim++;
            }
            const Task &tk = tasks[ti];
            // (tasks are claimed in order, so waiting for room here cannot block an earlier task)
            bool room_ok;
            {
              std::unique_lock<std::mutex> l(gt->m);
              gt->cv.wait(l, [&] { return gt->cancelled || gt->starving > 0 || tk.out_off == gt->freed || tk.out_off + tk.out_len - gt->freed <= gt->max_ahead; });
              room_ok = !gt->cancelled && tk.out_off + tk.out_len <= gt->reserved;
            }
            const bool ok = room_ok && inflate_task(tk, gt->base + tk.out_off);
            {  // published in task order, and "complete" only behind the last publication: both under the one lock
              std::lock_guard<std::mutex> l(tm);
              if (!ok) bad = true;
              task_done[ti] = 1;
              size_t add = 0;
              while (!bad && next_pub < tasks.size() && task_done[next_pub]) add += tasks[next_pub++].out_len;
              if (add) gt->publish(add);
              if (bad || next_pub == tasks.size()) gt->finish(bad ? "corrupt BGZF block in the input (deflate stream, length or crc)" : "");
              if (bad) return;
            }
          }
        });
      return true;
    }
    tasks.clear();
    if (getenv("KU_NO_PGZIP")) {  // zlib's one inflate is asked for: the sequential reader
      gt = nullptr;
      munmap(mp, len);
      map = nullptr;
      return false;
    }
    // (the inflating team next to the parser team and the formatting helpers: three quarters of the usable CPUs, at most 12)
    int n = std::max(1, std::min(usable_cpus() * 3 / 4, 12));
    (void)hw;
    if (const char *e = getenv("KU_PGZIP_TEAM")) n = std::max(1, std::min(atoi(e), 64));
    pgz = new ku_pgzip::ParallelGunzip;
    pgz->open(map, len, n);
    coord = std::thread([this] {
      size_t got = 0;
      while (pgz->round_to([this](size_t total) { return gt->place(total); }, got)) gt->publish(got);
      gt->finish(pgz->error);
    });
    return true;
  }
  void close() {
    if (gt) gt->cancel();
    if (coord.joinable()) coord.join();
    for (auto &t : team) t.join();
    team.clear();
    delete pgz;
    pgz = nullptr;
    if (pbz) { pbz->close(); delete pbz; pbz = nullptr; }
    if (map) munmap((void *)map, len);
    map = nullptr;
    gt = nullptr;
  }
};

// Record-aligned regions of a text, handed out in order to the members of a parser team: a fixed text (mapped file) or
// one that is still growing.  A region ends at the first record start at or behind its nominal end; with growing text
// the decision is only taken on lines that are completely there.
struct RegionCutter {
  const char *data = nullptr;
  size_t n = 0;            // fixed text: its size
  GrowingText *gt = nullptr;
  bool fastq = false;
  size_t region_bytes = (size_t)1 << 20;
  size_t ramp = 0;         // regions of a quarter / half the size at the start: this many each (0: every region full size)
  std::mutex mu;
  size_t next_cut = 0, next_region = 0;
  bool stop = false, exhausted = false;

  void halt() { std::lock_guard<std::mutex> l(mu); stop = true; }
  bool finished(size_t *handed_out) {
    std::lock_guard<std::mutex> l(mu);
    *handed_out = next_region;
    if (!gt) exhausted |= next_cut >= n;
    else {
      std::lock_guard<std::mutex> g(gt->m);
      exhausted |= (gt->done || gt->cancelled) && next_cut >= gt->avail;
    }
    return exhausted || stop;
  }
  // is `p` (a line start in [0, avail)) decided as a record start by lines that are completely visible?
  bool decided(size_t p, size_t avail) const {
    if (p >= avail) return false;
    if (!fastq) return true;
    const char *l1 = (const char *)memchr(data + p, '\n', avail - p);
    const char *l2 = l1 ? (const char *)memchr(l1 + 1, '\n', avail - (size_t)(l1 + 1 - data)) : nullptr;
    return l2 && (size_t)(l2 + 1 - data) < avail;
  }
  bool claim(size_t &lo, size_t &hi, size_t &idx) {
    for (;;) {
      size_t cand;
      {
        std::lock_guard<std::mutex> l(mu);
        if (stop || exhausted) return false;
        cand = next_cut;
      }
      // the first regions are smaller (a quarter, then half of region_bytes): the stage behind the parsers gets its first batch
      // after a quarter of the time, the team's members do not all finish their first region at the same moment
      size_t want = region_bytes;
      if (ramp) {
        size_t idx_now;
        { std::lock_guard<std::mutex> l(mu); idx_now = next_region; }
        if (idx_now < ramp) want = std::max<size_t>(region_bytes / 4, 1);
        else if (idx_now < 2 * ramp) want = std::max<size_t>(region_bytes / 2, 1);
      }
      size_t avail = n, end;
      bool complete = true;
      if (gt) avail = gt->wait_for(cand + want + ((size_t)64 << 10), &complete);
      if (complete && cand >= avail) {
        std::lock_guard<std::mutex> l(mu);
        if (next_cut == cand) exhausted = true;
        continue;
      }
      if (complete && cand + want >= avail) end = avail;
      else {
        end = find_record_start(data, avail, cand + want, fastq);
        if (!complete && !decided(end, avail)) {  // the boundary lies in lines that are not all there yet
          bool c2;
          gt->wait_for(avail + 1, &c2);
          continue;
        }
      }
      std::lock_guard<std::mutex> l(mu);
      if (stop) return false;
      if (next_cut != cand) continue;  // another member took it meanwhile
      lo = cand;
      hi = end;
      next_cut = end;
      idx = next_region++;
      return true;
    }
  }
};

// read_merger.pl:182 "$id =~ s/[\/_.][12]$//"
inline size_t strip_mate_suffix(const char *id, size_t n) {
  if (n >= 2 && (id[n - 1] == '1' || id[n - 1] == '2') && (id[n - 2] == '/' || id[n - 2] == '_' || id[n - 2] == '.')) return n - 2;
  return n;
}

}  // namespace ku_seqio
