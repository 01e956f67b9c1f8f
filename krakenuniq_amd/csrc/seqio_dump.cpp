// Test / measurement aid for the classify executable's input stage (ku_seqio.h): parses FASTA/FASTQ(+gz) files
// exactly as the reader thread does and prints "id<TAB>sequence" per read, or with -n only the parsing rate.
// -j N parses N record-aligned regions of each (plain) file independently, as the executable's parser team does.
// -z J inflates each file with the J-thread gzip team alone (ku_pgzip.h) and writes the bytes (with -n: the rate).
// -G out.gz: the sequential reader's records go to a gzip file written as the executable writes `-o x.gz` (ku_pgzout.h: parts
// deflated side by side, one stream).
// Host-only: does not link the GPU library (pinned allocation is replaced by malloc here), classifies nothing.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <map>
#include <thread>

#include <cstdarg>
#include <cstdlib>

#include "ku_seqio.h"
#include "ku_pgzip.h"
#include "ku_pbzip2.h"
#include "ku_pgzout.h"

// the batches of this tool are plain host memory (Batch::pinned = false); these are never called
extern "C" int ku_host_alloc(size_t, void **out) { *out = nullptr; return KU_ENOMEM; }
extern "C" void ku_host_free(void *) {}

void ku_seqio::fatal(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  exit(code);
}

int main(int argc, char **argv) {
  bool paired = false, quiet = false, prefetch = false, warm = false;
  int regions = 0, gunzip_team = 0, bunzip_team = 0;
  const char *gz_out = nullptr;
  std::string gz_text;
  ku_seqio::UnitGate gate;  // -u N: the reference's work unit size (a unit without nucleotides ends a file, src/classify.cpp:522-523)
  int a = 1;
  for (; a < argc && argv[a][0] == '-' && argv[a][1]; ++a) {
    if (argv[a][1] == 'P') paired = true;
    else if (argv[a][1] == 'n') quiet = true;
    else if (argv[a][1] == 'T') prefetch = true;
    else if (argv[a][1] == 'w') warm = true;  // -j: parse every region twice into the same batch, time the second pass (buffers and pages warm)
    else if (argv[a][1] == 'z' && a + 1 < argc) gunzip_team = atoi(argv[++a]);
    else if (argv[a][1] == 'G' && a + 1 < argc) gz_out = argv[++a];
    else if (argv[a][1] == 'Z' && a + 1 < argc) bunzip_team = atoi(argv[++a]);  // as -z, for .bz2 (ku_pbzip2.h)
    else if (argv[a][1] == 'u' && a + 1 < argc) gate.unit_nt = (uint64_t)std::max(1LL, atoll(argv[++a]));
    else if (argv[a][1] == 'j' && a + 1 < argc) regions = atoi(argv[++a]);  // producer thread per file, as the classify executable runs
  }
  timeval t0, t1;
  gettimeofday(&t0, nullptr);
  uint64_t n_reads = 0, n_bytes = 0;
  std::string header, header2;
  for (; bunzip_team > 0 && a < argc; ++a) {
    int fd = ::open(argv[a], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) ku_seqio::fatal(66, "can't open %s", argv[a]);
    const size_t n = (size_t)st.st_size;
    const uint8_t *data = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : (const uint8_t *)"";
    ::close(fd);
    if (!ku_pbzip2::ParallelBunzip2::is_bzip2(data, n)) ku_seqio::fatal(65, "%s: not a bzip2 file", argv[a]);
    ku_pbzip2::ParallelBunzip2 pb;
    pb.open(data, n, bunzip_team);
    const uint8_t *blk;
    size_t got = 0;
    uint64_t total = 0;
    while (pb.next(blk, got)) {
      total += got;
      if (!quiet) fwrite(blk, 1, got, stdout);
    }
    pb.close();
    if (!pb.error.empty()) ku_seqio::fatal(65, "%s: %s", argv[a], pb.error.c_str());
    gettimeofday(&t1, nullptr);
    const double s = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_usec - t0.tv_usec) / 1e6;
    fprintf(stderr, "%llu bytes in %.3f s: %.1f MB/s; %llu blocks, %llu magic numbers inside data\n", (unsigned long long)total, s, total / s / 1e6,
            (unsigned long long)pb.n_blocks, (unsigned long long)pb.n_skipped);
    if (n) munmap((void *)data, n);
    if (a + 1 == argc) return 0;
  }
  for (; gunzip_team > 0 && a < argc; ++a) {
    int fd = ::open(argv[a], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) ku_seqio::fatal(66, "can't open %s", argv[a]);
    const size_t n = (size_t)st.st_size;
    const uint8_t *data = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : (const uint8_t *)"";
    ::close(fd);
    ku_pgzip::ParallelGunzip pg;
    pg.open(data, n, gunzip_team);
    ku_pgzip::RawBuf<char> out;
    size_t got = 0;
    uint64_t total = 0;
    while (pg.round(out, got)) {
      total += got;
      if (!quiet) fwrite(out.d, 1, got, stdout);
    }
    if (!pg.error.empty()) ku_seqio::fatal(65, "%s: %s", argv[a], pg.error.c_str());
    gettimeofday(&t1, nullptr);
    const double s = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_usec - t0.tv_usec) / 1e6;
    fprintf(stderr, "%llu bytes in %.3f s: %.1f MB/s; %llu rounds, %llu spans, %llu dropped\n", (unsigned long long)total, s, total / s / 1e6,
            (unsigned long long)pg.n_rounds, (unsigned long long)pg.n_spans, (unsigned long long)pg.n_dropped);
    fprintf(stderr, "  decode %.3f s (span 0 alone %.3f; block search %.3f summed over spans), windows %.3f, translate + crc %.3f\n", pg.t_decode, pg.t_first,
            pg.search_us.load() / 1e6, pg.t_resolve, pg.t_translate);
    if (n) munmap((void *)data, n);
    if (a + 1 == argc) return 0;
  }
  for (; regions > 0 && a < argc; ++a) {  // region-parallel parse: plain files mapped, .gz files as text that grows while it is parsed
    ku_seqio::GrowingText gt;
    ku_seqio::GzTextStream gz;
    const bool growing = gz.open(argv[a], gt);
    const char *data = "";
    size_t n = 0;
    if (!growing) {
      int fd = ::open(argv[a], O_RDONLY);
      struct stat st;
      if (fd < 0 || fstat(fd, &st) != 0) ku_seqio::fatal(66, "can't open %s", argv[a]);
      n = (size_t)st.st_size;
      if (n) data = (const char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      ::close(fd);
    }
    ku_seqio::RegionCutter cut;
    bool complete = true;
    if (growing) {
      cut.gt = &gt;
      cut.data = gt.base;
      if (gt.wait_for(1, &complete) == 0) { gz.close(); if (!gt.error.empty()) ku_seqio::fatal(65, "%s", gt.error.c_str()); continue; }
      cut.region_bytes = (size_t)(getenv("KU_REGION_KB") ? atol(getenv("KU_REGION_KB")) : 8192) << 10;
    } else {
      cut.data = data;
      cut.n = n;
      cut.region_bytes = std::max<size_t>(1, (n + (size_t)regions - 1) / (size_t)regions);
    }
    cut.fastq = growing ? gt.base[0] == '@' : (n && data[0] == '@');
    if (const char *e = getenv("KU_REGION_RAMP")) cut.ramp = (size_t)atoi(e);  // smaller first regions, as the classify executable cuts them
    std::mutex rm;
    struct Parsed { ku_seqio::Batch *bt; size_t lo, hi; ku_seqio::RegionParse res; };
    std::map<size_t, Parsed> parsed;
    std::vector<ku_seqio::Batch *> ordered;  // the batches whose records count, in file order
    ku_seqio::RegionChain chain;
    size_t next_out = 0;
    ku_seqio::GrowingText *gtp = growing ? &gt : nullptr;
    auto member = [&] {
      size_t lo, hi, idx;
      while (cut.claim(lo, hi, idx)) {
        ku_seqio::Batch *bt = new ku_seqio::Batch;
        bt->pinned = false;
        ku_seqio::RegionParse res = ku_seqio::parse_region(cut.data, n, gtp, lo, hi, cut.fastq, *bt, false);
        if (warm && !growing) { bt->clear(); res = ku_seqio::parse_region(cut.data, n, gtp, lo, hi, cut.fastq, *bt, false); }
        std::lock_guard<std::mutex> l(rm);
        parsed[idx] = Parsed{bt, lo, hi, res};
        for (auto it = parsed.find(next_out); it != parsed.end(); it = parsed.find(next_out)) {  // in file order: does the region count?
          Parsed &p = it->second;
          switch (chain.judge(p.lo, p.hi, p.res)) {
            case ku_seqio::RegionChain::REPARSE:  // cut inside a record: again from where the region before it stopped
              p.bt->clear();
              chain.accept(ku_seqio::parse_region(cut.data, n, gtp, chain.expect, p.hi, cut.fastq, *p.bt, false));
              ordered.push_back(p.bt);
              break;
            case ku_seqio::RegionChain::ACCEPT: ordered.push_back(p.bt); break;
            case ku_seqio::RegionChain::SKIP: p.bt->release(); delete p.bt; break;
          }
          if (chain.ended) cut.halt();
          else if (growing) gt.release_before(std::min(p.hi, chain.expect));  // the text behind the regions parsed so far is not needed any more
          parsed.erase(it);
          ++next_out;
        }
      }
    };
    std::vector<std::thread> team;
    for (int r = 0; r < regions; ++r) team.emplace_back(member);
    for (auto &t : team) t.join();
    if (growing) {
      gz.close();
      if (!gt.error.empty()) ku_seqio::fatal(65, "%s", gt.error.c_str());  // (damage of the compressed file, not the parser stopping early)
    }
    for (auto &kv : parsed) { kv.second.bt->release(); delete kv.second.bt; }  // (parsed behind the end of the stream)
    auto print = [&](ku_seqio::Batch *btp) {
      ku_seqio::Batch &bt = *btp;
      for (size_t i = 0; i < bt.off.size(); ++i) {
        ++n_reads;
        n_bytes += bt.len[i];
        if (!quiet) {
          fputs(bt.ids.c_str() + bt.idoff[i], stdout);
          fputc('\t', stdout);
          fwrite(bt.seqs + bt.off[i], 1, bt.len[i], stdout);
          fputc('\n', stdout);
        }
      }
      bt.release();
      delete btp;
    };
    auto drop = [&](ku_seqio::Batch *btp) { btp->release(); delete btp; };
    gate.begin_file();
    for (ku_seqio::Batch *bt : ordered) gate.push(bt, print, drop);
    gate.finish(print, drop);
    if (!growing && n) munmap((void *)data, n);
  }
  for (; a < argc; a += paired ? 2 : 1) {
    ku_seqio::Reader rd, rd2;
    rd.open(argv[a], prefetch);
    if (paired) {
      if (a + 1 >= argc) ku_seqio::fatal(64, "-P needs the files in pairs");
      rd2.open(argv[a + 1], prefetch);
    }
    auto emit = [&](ku_seqio::Batch *btp) {
      ku_seqio::Batch &bt = *btp;
      for (size_t i = 0; i < bt.off.size(); ++i) {
        ++n_reads;
        const char *id = bt.ids.c_str() + bt.idoff[i];
        if (gz_out) {
          gz_text.append(id);
          gz_text += '\t';
          gz_text.append(bt.seqs + bt.off[i], bt.len[i]);
          gz_text += '\n';
        } else if (!quiet) {
          fputs(id, stdout);
          fputc('\t', stdout);
          fwrite(bt.seqs + bt.off[i], 1, bt.len[i], stdout);
          fputc('\n', stdout);
        }
      }
      n_bytes += bt.nt;
      bt.release();
      delete btp;
    };
    auto drop = [&](ku_seqio::Batch *btp) { btp->release(); delete btp; };
    gate.begin_file();
    const std::string no_quals;
    for (bool more = true; more;) {
      ku_seqio::Batch *btp = new ku_seqio::Batch;
      ku_seqio::Batch &bt = *btp;
      bt.pinned = false;
      while (bt.nt < (64u << 20)) {
        size_t n1, n2, lo, hi;
        bt.begin_read();
        bool got = ku_seqio::next_record(rd, bt, &header, nullptr, &n1);
        if (paired) {
          if (got) {
            const size_t mark = bt.seqs_len;
            bt.append("N", 1);
            if (!ku_seqio::next_record(rd2, bt, &header2, nullptr, &n2)) bt.seqs_len = mark;
          } else got = ku_seqio::next_record(rd2, bt, &header, nullptr, &n2);
        }
        if (!got) { bt.off.pop_back(); more = false; break; }
        bt.end_read();
        ku_seqio::split_id(header.data(), header.size(), lo, hi);
        if (paired) hi = lo + ku_seqio::strip_mate_suffix(header.data() + lo, hi - lo);
        bt.add_meta(header, lo, hi, no_quals, false);
      }
      gate.push(btp, emit, drop);  // (a unit without nucleotides ends the file: decided at work-unit granularity)
    }
    gate.finish(emit, drop);
    rd.close();
    rd2.close();
  }
  if (gz_out) {  // parts of ~200 kB (what a formatting helper of the executable holds), deflated by four threads, written in order
    const size_t part = getenv("KU_GZ_PART") ? (size_t)atol(getenv("KU_GZ_PART")) : 200000;
    const size_t n_parts = (gz_text.size() + part - 1) / part;
    std::vector<unsigned char *> comp(n_parts, nullptr);
    std::vector<size_t> clen(n_parts, 0);
    std::vector<uLong> crcs(n_parts, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
      th.emplace_back([&, t] {
        for (size_t i = (size_t)t; i < n_parts; i += 4)
          comp[i] = ku_pgzout::deflate_part(gz_text.data() + i * part, std::min(part, gz_text.size() - i * part), &clen[i], &crcs[i]);
      });
    for (auto &x : th) x.join();
    ku_pgzout::Member mem;
    if (!mem.open(gz_out)) ku_seqio::fatal(73, "can't create %s", gz_out);
    for (size_t i = 0; i < n_parts; ++i) {
      if (!comp[i] || !mem.put(comp[i], clen[i], crcs[i], std::min(part, gz_text.size() - i * part))) ku_seqio::fatal(74, "deflate / write error");
      free(comp[i]);
    }
    if (!mem.close()) ku_seqio::fatal(74, "write error");
  }
  gettimeofday(&t1, nullptr);
  const double s = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_usec - t0.tv_usec) / 1e6;
  fprintf(stderr, "%llu reads, %.1f Mbp in %.3f s: %.2f M reads/s\n", (unsigned long long)n_reads, n_bytes / 1e6, s, n_reads / s / 1e6);
  return 0;
}
