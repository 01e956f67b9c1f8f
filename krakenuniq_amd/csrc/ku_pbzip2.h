// bzip2 input inflated by a team of threads, written from the format (no libbz2: the image has the library but not its
// header, and one bzlib stream decodes at 40 MB/s).  (host side of the classify executable's input stage, ku_seqio.h; the
// reference reads .bz2 through bxz::ifstream over bzlib, src/seqreader.hpp:48)
//
// A .bz2 file is a sequence of blocks of at most 900 kB, each compressed on its own (Burrows-Wheeler transform, move to
// front, run lengths, up to six Huffman tables switched every 50 symbols) and introduced by a 48-bit magic number at ANY
// bit offset.  The members of a team find the magic numbers (the file in pieces, by whichever member has no block to decode)
// and decode the blocks behind them side by side; the consumer takes them in file order.  A block counts only where the block before it ended (a magic number
// inside compressed data is skipped that way), with its CRC right, and every stream's combined CRC is checked at its end
// marker: what bzip2 -t refuses is refused.  Several streams in one file (pbzip2's output, `cat a.bz2 b.bz2`) follow one
// another; bytes behind the last stream that do not start a stream are ignored as bzip2 ignores them.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ku_pbzip2 {

static constexpr uint64_t MAGIC_BLOCK = 0x314159265359ull, MAGIC_END = 0x177245385090ull;
static constexpr uint32_t MAX_BLOCK = 900000, MAX_SELECTORS = 18002;

// bits most significant first
struct Bits {
  const uint8_t *base = nullptr, *p = nullptr, *end = nullptr;
  uint64_t buf = 0;
  int cnt = 0;
  size_t pad = 0;  // zero bytes taken from behind the end
  void init(const uint8_t *b, size_t n, size_t bitpos) {
    base = b; end = b + n; p = b + (bitpos >> 3);
    buf = 0; cnt = 0; pad = 0;
    if (bitpos & 7) (void)take((unsigned)(bitpos & 7));
  }
  inline void refill() {  // at least 32 valid bits afterwards
    if (cnt > 32) return;
    if (end - p >= 4) {
      buf = (buf << 32) | ((uint64_t)p[0] << 24 | (uint64_t)p[1] << 16 | (uint64_t)p[2] << 8 | (uint64_t)p[3]);
      p += 4;
      cnt += 32;
    } else {
      while (cnt <= 56) {
        uint64_t b = 0;
        if (p < end) b = *p; else ++pad;
        ++p;
        buf = (buf << 8) | b;
        cnt += 8;
      }
    }
  }
  inline uint32_t peek(unsigned n) const { return (uint32_t)((buf >> (cnt - (int)n)) & (((uint64_t)1 << n) - 1)); }
  inline void drop(unsigned n) { cnt -= (int)n; }
  inline uint32_t take(unsigned n) {  // n <= 32
    refill();
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  size_t bitpos() const { return (size_t)(p - base) * 8 - (size_t)cnt; }
  bool overrun() const { return pad * 8 > (size_t)cnt; }
};

static inline const uint32_t *crc_table() {
  static const uint32_t *t = [] {
    uint32_t *tab = new uint32_t[256];
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i << 24;
      for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : c << 1;
      tab[i] = c;
    }
    return tab;
  }();
  return t;
}

struct Scratch {
  std::vector<uint32_t> tt;
  std::vector<uint8_t> selector;
  Scratch() : tt(MAX_BLOCK), selector(MAX_SELECTORS) {}
};

struct Huff {
  uint16_t fast[1 << 10];  // sym << 5 | len for codes of at most 10 bits; 0: longer
  uint32_t first[22], count[22], offset[22];
  uint16_t perm[258];
  int min_len = 1, max_len = 20;
  bool build(const uint8_t *len, int n) {
    memset(count, 0, sizeof count);
    min_len = 32; max_len = 0;
    for (int i = 0; i < n; ++i) {
      ++count[len[i]];
      if (len[i] < min_len) min_len = len[i];
      if (len[i] > max_len) max_len = len[i];
    }
    uint32_t code = 0, off = 0;
    for (int l = min_len; l <= max_len; ++l) {
      first[l] = code;
      offset[l] = off;
      if ((uint64_t)code + count[l] > ((uint64_t)1 << l)) return false;  // over-subscribed
      code = (code + count[l]) << 1;
      off += count[l];
    }
    uint32_t fill[22];
    memcpy(fill, offset, sizeof fill);
    for (int i = 0; i < n; ++i) perm[fill[len[i]]++] = (uint16_t)i;
    memset(fast, 0, sizeof fast);
    for (int l = min_len; l <= max_len && l <= 10; ++l)
      for (uint32_t k = 0; k < count[l]; ++k) {
        const uint32_t c = first[l] + k, sym = perm[offset[l] + k];
        for (uint32_t f = c << (10 - l); f < ((c + 1) << (10 - l)); ++f) fast[f] = (uint16_t)(sym << 5 | (uint32_t)l);
      }
    return true;
  }
  // -1: not a code
  inline int decode(Bits &b) const {
    b.refill();
    const uint16_t e = fast[b.peek(10)];
    if (e) { b.drop(e & 31); return e >> 5; }
    for (int l = min_len > 11 ? min_len : 11; l <= max_len; ++l) {
      const uint32_t v = b.peek((unsigned)l) - first[l];
      if (v < count[l]) { b.drop((unsigned)l); return perm[offset[l] + v]; }
    }
    return -1;
  }
};

// One block whose 48-bit magic number has just been read.  Its bytes go to `out`; false: damaged (err says how)
static inline bool decode_block(Bits &b, Scratch &S, std::vector<uint8_t> &out, uint32_t *stored_crc, uint32_t *calc_crc, const char **err) {
  *stored_crc = b.take(32);
  if (b.take(1)) { *err = "randomised bzip2 block (bzip2 0.9.0 and older)"; return false; }
  const uint32_t orig_ptr = b.take(24);
  uint8_t seq_to_unseq[256];
  int n_in_use = 0;
  const uint32_t used16 = b.take(16);
  for (int i = 0; i < 16; ++i)
    if (used16 & (0x8000u >> i)) {
      const uint32_t mm = b.take(16);
      for (int j = 0; j < 16; ++j)
        if (mm & (0x8000u >> j)) seq_to_unseq[n_in_use++] = (uint8_t)(i * 16 + j);
    }
  if (n_in_use == 0) { *err = "bzip2 block without symbols"; return false; }
  const int alpha = n_in_use + 2;
  const int n_groups = (int)b.take(3);
  const uint32_t n_sel = b.take(15);
  if (n_groups < 2 || n_groups > 6 || n_sel < 1) { *err = "bzip2 block header"; return false; }
  {
    uint8_t pos[6] = {0, 1, 2, 3, 4, 5};
    for (uint32_t i = 0; i < n_sel; ++i) {
      int j = 0;
      while (b.take(1)) {
        if (++j >= n_groups) { *err = "bzip2 selector"; return false; }
      }
      const uint8_t tmp = pos[j];
      for (; j > 0; --j) pos[j] = pos[j - 1];
      pos[0] = tmp;
      if (i < MAX_SELECTORS) S.selector[i] = tmp;  // (bzip2 1.0.8: what lies beyond is read and dropped)
    }
  }
  const uint32_t n_selectors = n_sel < MAX_SELECTORS ? n_sel : MAX_SELECTORS;
  Huff H[6];
  for (int t = 0; t < n_groups; ++t) {
    uint8_t len[258];
    int curr = (int)b.take(5);
    for (int i = 0; i < alpha; ++i) {
      for (;;) {
        if (curr < 1 || curr > 20) { *err = "bzip2 code length"; return false; }
        if (!b.take(1)) break;
        curr += b.take(1) ? -1 : 1;
      }
      len[i] = (uint8_t)curr;
    }
    if (!H[t].build(len, alpha)) { *err = "bzip2 code lengths"; return false; }
  }
  if (b.overrun()) { *err = "truncated bzip2 block"; return false; }
  // ---- the symbols: run lengths of the front of the move-to-front list, list positions, end of block
  uint32_t *tt = S.tt.data();
  uint32_t unzftab[256] = {0};
  uint8_t yy[256];
  for (int i = 0; i < 256; ++i) yy[i] = (uint8_t)i;
  uint32_t nblock = 0, group_no = 0, group_pos = 0;
  const Huff *h = nullptr;
  const int eob = n_in_use + 1;
  uint64_t es = 0;
  unsigned run_bit = 0;
  for (;;) {
    if (group_pos == 0) {
      if (group_no >= n_selectors) { *err = "bzip2 block runs past its selectors"; return false; }
      h = &H[S.selector[group_no++]];
      group_pos = 50;
    }
    --group_pos;
    const int sym = h->decode(b);
    if (sym < 0) { *err = "bzip2 data"; return false; }
    if (sym <= 1) {
      if (run_bit > 21) { *err = "bzip2 run length"; return false; }
      es += (uint64_t)(sym + 1) << run_bit;
      ++run_bit;
      continue;
    }
    if (run_bit) {
      const uint8_t uc = seq_to_unseq[yy[0]];
      if (es > MAX_BLOCK - nblock) { *err = "bzip2 block too long"; return false; }
      unzftab[uc] += (uint32_t)es;
      for (uint32_t e = 0; e < (uint32_t)es; ++e) tt[nblock + e] = uc;
      nblock += (uint32_t)es;
      es = 0;
      run_bit = 0;
    }
    if (sym == eob) break;
    if (nblock >= MAX_BLOCK) { *err = "bzip2 block too long"; return false; }
    const int nn = sym - 1;
    if (nn >= n_in_use) { *err = "bzip2 data"; return false; }
    const uint8_t uc = yy[nn];
    memmove(yy + 1, yy, (size_t)nn);
    yy[0] = uc;
    const uint8_t byte = seq_to_unseq[uc];
    ++unzftab[byte];
    tt[nblock++] = byte;
    if (b.overrun()) { *err = "truncated bzip2 block"; return false; }
  }
  if (b.overrun()) { *err = "truncated bzip2 block"; return false; }
  if (orig_ptr >= nblock && !(nblock == 0 && orig_ptr == 0)) { *err = "bzip2 block origin"; return false; }
  // ---- the Burrows-Wheeler transform undone: tt[i] >> 8 = where the byte behind byte i of the sorted column sits
  uint32_t cftab[257];
  cftab[0] = 0;
  for (int i = 0; i < 256; ++i) cftab[i + 1] = cftab[i] + unzftab[i];
  for (uint32_t i = 0; i < nblock; ++i) {
    const uint32_t uc = tt[i] & 0xff;
    tt[cftab[uc]++] |= i << 8;
  }
  out.clear();
  out.reserve(nblock + nblock / 4 + 64);
  const uint32_t *ct = crc_table();
  uint32_t crc = 0xffffffffu;
  if (nblock) {
    uint32_t t_pos = tt[orig_ptr] >> 8;
    int run = 0;
    uint8_t prev = 0;
    for (uint32_t i = 0; i < nblock; ++i) {
      t_pos = tt[t_pos];
      const uint8_t ch = (uint8_t)t_pos;
      t_pos >>= 8;
      if (run == 4) {  // four equal bytes are followed by the number of further copies
        for (uint8_t k = 0; k < ch; ++k) crc = (crc << 8) ^ ct[(crc >> 24) ^ prev];
        out.insert(out.end(), (size_t)ch, prev);
        run = 0;
        continue;
      }
      if (run > 0 && ch == prev) ++run; else { run = 1; prev = ch; }
      crc = (crc << 8) ^ ct[(crc >> 24) ^ ch];
      out.push_back(ch);
    }
  }
  *calc_crc = ~crc;
  return true;
}

struct ParallelBunzip2 {
  const uint8_t *m = nullptr;
  size_t n = 0;
  int team = 1;
  struct Cand { size_t bit; bool end_marker; };
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Cand> cands;   // in file order, appended by the scanner
  bool scanned = false, stop = false;
  struct Slot {
    std::vector<uint8_t> out;
    size_t task = (size_t)-1, start_bit = 0, end_bit = 0;
    uint32_t crc_stored = 0, crc_calc = 0;
    bool end_marker = false, ok = false, done = false;
    const char *err = "";
  };
  std::vector<Slot> slots;
  size_t next_claim = 0, next_out = 0;
  std::vector<std::thread> workers;
  // consumer
  size_t expect_bit = 32;
  uint32_t combined = 0;
  bool finished = false, holding = false;
  std::string error;
  uint64_t n_blocks = 0, n_skipped = 0;

  ParallelBunzip2() = default;
  ParallelBunzip2(const ParallelBunzip2 &) = delete;
  ParallelBunzip2 &operator=(const ParallelBunzip2 &) = delete;
  ~ParallelBunzip2() { close(); }

  static bool is_bzip2(const uint8_t *p, size_t len) { return len >= 14 && p[0] == 'B' && p[1] == 'Z' && p[2] == 'h' && p[3] >= '1' && p[3] <= '9'; }

  // ---- finding the magic numbers: the file in pieces, scanned by whichever member has no block to decode, published in
  // file order
  static constexpr size_t PIECE = (size_t)4 << 20;
  size_t n_pieces = 0, next_scan = 0, next_pub = 0;
  std::vector<std::vector<Cand>> pending;
  std::vector<char> piece_done;

  void scan_piece(size_t k, std::vector<Cand> &found) const {
    const size_t lo = k * PIECE, hi = std::min(n, lo + PIECE);
    const uint64_t mask = ((uint64_t)1 << 48) - 1;
    uint64_t w = 0;
    for (size_t i = lo >= 7 ? lo - 7 : 0; i < lo; ++i) w = (w << 8) | m[i];
    for (size_t i = lo; i < hi; ++i) {
      w = (w << 8) | m[i];  // the last eight bytes: a magic number may end at any of the newest eight bits
      for (unsigned s = 0; s < 8; ++s) {
        const uint64_t v = (w >> s) & mask;
        if (v == MAGIC_BLOCK || v == MAGIC_END) {
          const size_t end_bit = (i + 1) * 8 - s;
          if (end_bit >= 48) found.push_back(Cand{end_bit - 48, v == MAGIC_END});
        }
      }
    }
  }

  void open(const uint8_t *map, size_t len, int threads) {
    m = map; n = len;
    team = threads < 1 ? 1 : threads;
    slots.assign((size_t)2 * team, Slot());
    n_pieces = (n + PIECE - 1) / PIECE;
    pending.assign(n_pieces, std::vector<Cand>());
    piece_done.assign(n_pieces, 0);
    scanned = n_pieces == 0;
    for (int t = 0; t < team; ++t)
      workers.emplace_back([this] {
        Scratch S;
        for (;;) {
          size_t ti = 0, piece = (size_t)-1;
          Slot *sl = nullptr;
          Cand c{0, false};
          {
            std::unique_lock<std::mutex> l(mu);
            for (;;) {
              if (stop) return;
              if (next_claim < cands.size() && next_claim < next_out + slots.size()) {  // a block to decode
                ti = next_claim++;
                c = cands[ti];
                sl = &slots[ti % slots.size()];
                sl->task = ti;
                sl->done = false;
                break;
              }
              if (next_scan < n_pieces && cands.size() - next_claim < (size_t)4 * team) { piece = next_scan++; break; }  // look further ahead
              if (scanned && next_claim >= cands.size()) return;
              cv.wait(l);
            }
          }
          if (!sl) {
            std::vector<Cand> found;
            scan_piece(piece, found);
            std::lock_guard<std::mutex> l(mu);
            pending[piece].swap(found);
            piece_done[piece] = 1;
            while (next_pub < n_pieces && piece_done[next_pub]) {
              cands.insert(cands.end(), pending[next_pub].begin(), pending[next_pub].end());
              std::vector<Cand>().swap(pending[next_pub]);
              ++next_pub;
            }
            if (next_pub == n_pieces) scanned = true;
            cv.notify_all();
            continue;
          }
          sl->start_bit = c.bit;
          sl->end_marker = c.end_marker;
          sl->err = "";
          Bits b;
          b.init(m, n, c.bit + 48);
          if (c.end_marker) {
            sl->crc_stored = b.take(32);
            sl->ok = !b.overrun();
            sl->out.clear();
          } else {
            sl->ok = decode_block(b, S, sl->out, &sl->crc_stored, &sl->crc_calc, &sl->err);
          }
          sl->end_bit = b.bitpos();
          { std::lock_guard<std::mutex> l(mu); sl->done = true; }
          cv.notify_all();
        }
      });
  }

  void close() {
    { std::lock_guard<std::mutex> l(mu); stop = true; }
    cv.notify_all();
    for (auto &t : workers) t.join();
    workers.clear();
  }

  void release_held() {
    if (!holding) return;
    holding = false;
    { std::lock_guard<std::mutex> l(mu); ++next_out; }
    cv.notify_all();
  }

  // the next block's bytes (valid until the next call); false: the end, or `error`
  bool next(const uint8_t *&data, size_t &len) {
    release_held();
    for (;;) {
      if (finished || !error.empty()) return false;
      Slot *sl = nullptr;
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] {
          Slot &s = slots[next_out % slots.size()];
          return (s.task == next_out && s.done) || (scanned && next_out >= cands.size());
        });
        Slot &s = slots[next_out % slots.size()];
        if (s.task == next_out && s.done) sl = &s;
      }
      if (!sl) { error = "truncated bzip2 data (no block where the last one ended)"; return false; }
      if (sl->start_bit < expect_bit) {  // a magic number inside the data of the block before
        ++n_skipped;
        holding = true;
        release_held();
        continue;
      }
      if (sl->start_bit > expect_bit) { error = "corrupt bzip2 data (no block header where the previous block ended)"; return false; }
      if (sl->end_marker) {
        if (!sl->ok || sl->crc_stored != combined) { error = "corrupt bzip2 data (combined crc of a stream)"; return false; }
        size_t at = (sl->end_bit + 7) >> 3;
        holding = true;
        release_held();
        if (at + 4 <= n && is_bzip2(m + at, 14)) {  // another stream follows
          expect_bit = (at + 4) * 8;
          combined = 0;
          continue;
        }
        finished = true;  // (anything else behind the last stream is ignored)
        return false;
      }
      if (!sl->ok) { error = std::string("corrupt bzip2 data (") + sl->err + ")"; return false; }
      if (sl->crc_stored != sl->crc_calc) { error = "corrupt bzip2 data (crc of a block)"; return false; }
      combined = ((combined << 1) | (combined >> 31)) ^ sl->crc_calc;
      expect_bit = sl->end_bit;
      ++n_blocks;
      data = sl->out.data();
      len = sl->out.size();
      holding = true;
      return true;
    }
  }
};

}  // namespace ku_pbzip2
