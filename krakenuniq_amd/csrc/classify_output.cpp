// classify_output.cpp -- output stage of the `classify` executable (classify_run.h) in two steps that overlap: the formatting
// helpers (a standing team of `fmt_threads`) take slices of the finished batches from one queue -- `fmt_threads` slices per batch,
// disjoint read ranges, across batch borders -- while the writer writes the batches in input order, each as soon as its slices are
// through.  (One thread doing both, with a team spawned per batch, was the slowest stage of the pipeline in round 1; a team behind
// a barrier per batch left its members idle a third of the time in round 5.)  Text: src/classify.cpp:826-861,980-1010.
#include "classify_run.h"

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

void Run::formatter_stage() {
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
}

void Run::writer_stage() {
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
}
