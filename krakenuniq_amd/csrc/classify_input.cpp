// classify_input.cpp -- reader stage of the `classify` executable (classify_run.h): FASTA / FASTQ files, plain or compressed, into
// page-locked read batches.  Record semantics of src/seqreader.cpp:26-133 and of src/classify.cpp:499-525 (ku_seqio.h).
#include "classify_run.h"

// Regular files, plain or .gz: the text is cut into record-aligned regions of about a quarter work unit and parsed by
// `parse_team` threads, each into its own batch; the batches go on in file order.  A plain file is mapped; a .gz file
// (BGZF or one gzip stream, ku_pgzip.h) is inflated by its own team into text that grows while it is parsed
// (ku_seqio::GrowingText).  A member takes a batch BEFORE it takes a region number, so the lowest outstanding region always
// owns one and the team cannot starve itself.  false: neither (a pipe, an empty file, no room) -> the sequential reader
// handles it.
bool Run::parse_file_in_regions(const char *path) {
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
}

void Run::reader_stage() {
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
}
