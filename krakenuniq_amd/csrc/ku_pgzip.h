// One gzip stream inflated by a team of threads.  (host side of the classify executable's input stage, ku_seqio.h)
//
// A .gz file as gzip(1) writes it is ONE deflate stream: no index, no restart points, every match may reach 32 KiB back.
// zlib therefore inflates it on one thread (1.7 M reads/s of FASTQ on the GPU box's host) while the
// device takes 500 M reads/s.  Here the compressed file is cut into spans and every span is decoded at the same time:
//
//   * the thread of a span looks for the first deflate block that starts in it -- a bit position where a dynamic-Huffman
//     block header parses (complete code-length, literal/length and distance codes, RFC 1951 3.2.7), the block decodes to
//     its end-of-block symbol, every literal in it is text, and the header of the block behind it parses as well;
//   * from there it decodes WITHOUT the 32 KiB of history it cannot know: the output is 16-bit symbols, a byte (< 256)
//     or "byte j of the unknown window" (256 + j); matches copy symbols, so unknowns propagate;
//   * it stops at the block start the next span's thread found.  The first span of a round starts at a known block with
//     a known window and decodes straight to bytes;
//   * then the windows are resolved in order (only the last 32 KiB of every span, sequential), and every span translates
//     its symbols to bytes in parallel; CRC-32 and ISIZE of every gzip member are checked from the per-span CRCs
//     (crc32_combine), as zlib's gzread checks them.
//
// Nothing depends on the block search being right: a span is accepted only if its predecessor's decoder arrives at
// exactly its start bit, at a block boundary; otherwise the span is dropped and the predecessor's end starts the next
// round.  Data that is not text (no span ever validates) falls back to one decoder; rounds then stop searching.
// Stored and fixed-Huffman blocks, several members and trailing garbage are handled as zlib handles them
// (RFC 1952; zlib's gzread ignores what follows the last member when it does not start with the gzip magic).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace ku_pgzip {

// ---------------------------------------------------------------------------------------------------- CRC-32
// zlib's table-driven crc32 runs at about 1 GB/s, a third of what a span's thread spends per byte; with carry-less
// multiplication (folding four 128-bit lanes, then Barrett reduction: Gopal et al., "Fast CRC computation for generic
// polynomials using PCLMULQDQ", Intel 2009, constants of the reflected polynomial 0xEDB88320) it is memory speed.
// Checked once against zlib on first use; any other CPU, or a mismatch, stays with zlib.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_fold(const uint8_t *buf, size_t len, uint32_t crc) {
  // len >= 64 and a multiple of 16; crc = register state (not inverted)
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
  const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
  const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124ll);
  const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
  __m128i x1 = _mm_xor_si128(_mm_loadu_si128((const __m128i *)buf), _mm_cvtsi32_si128((int)crc));
  __m128i x2 = _mm_loadu_si128((const __m128i *)(buf + 16)), x3 = _mm_loadu_si128((const __m128i *)(buf + 32)),
          x4 = _mm_loadu_si128((const __m128i *)(buf + 48));
  buf += 64; len -= 64;
  while (len >= 64) {
    const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00),
                  a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
    x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)buf));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(buf + 16)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(buf + 32)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(buf + 48)));
    buf += 64; len -= 64;
  }
#define KU_PGZIP_FOLD(nxt)                                                                         \
  {                                                                                                \
    const __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);                                        \
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), nxt);               \
  }
  KU_PGZIP_FOLD(x2) KU_PGZIP_FOLD(x3) KU_PGZIP_FOLD(x4)
  while (len >= 16) {
    KU_PGZIP_FOLD(_mm_loadu_si128((const __m128i *)buf))
    buf += 16; len -= 16;
  }
#undef KU_PGZIP_FOLD
  const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
  __m128i y = _mm_clmulepi64_si128(x1, k3k4, 0x10);
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), y);
  y = _mm_srli_si128(x1, 4);
  x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, m32), k5k0, 0x00), y);
  y = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, m32), poly, 0x10), m32);
  x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(y, poly, 0x00));
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

static inline uLong crc_of(const uint8_t *p, size_t n) {
  uLong c = crc32(0L, Z_NULL, 0);
#if defined(__x86_64__)
  static const bool fast = [] {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1") || getenv("KU_PGZIP_ZLIB_CRC")) return false;
    uint8_t t[64 * 5 + 16];
    for (size_t i = 0; i < sizeof t; ++i) t[i] = (uint8_t)(i * 131 + 7);
    return (uLong)(~crc32_fold(t, sizeof t, ~0u) & 0xffffffffu) == crc32_z(crc32(0L, Z_NULL, 0), t, sizeof t);
  }();
  if (fast && n >= 64) {
    const size_t body = n & ~(size_t)15;
    c = (uLong)(~crc32_fold(p, body, ~0u) & 0xffffffffu);
    p += body; n -= body;
  }
#endif
  return n ? crc32_z(c, p, n) : c;
}

// ---------------------------------------------------------------------------------------------------- bit reader
struct Bits {
  const uint8_t *base = nullptr, *p = nullptr, *end = nullptr;
  uint64_t buf = 0;
  unsigned cnt = 0;  // valid bits in buf
  size_t pad = 0;    // zero bytes taken from behind the end of the input (a consumer of those has overrun)
  void init(const uint8_t *b, size_t n, size_t bitpos) {
    base = b; end = b + n; p = b + (bitpos >> 3);
    buf = 0; cnt = 0; pad = 0;
    refill();
    drop((unsigned)(bitpos & 7));
  }
  // at least 56 valid bits afterwards
  inline void refill() {
    if (__builtin_expect(end - p >= 8, 1)) {
      uint64_t w;
      memcpy(&w, p, 8);
      buf |= w << cnt;
      p += (63 - cnt) >> 3;
      cnt |= 56;
    } else {
      while (cnt <= 56) {
        uint64_t b = 0;
        if (p < end) b = *p; else ++pad;
        ++p;
        buf |= b << cnt;
        cnt += 8;
      }
    }
  }
  inline void drop(unsigned n) { buf >>= n; cnt -= n; }
  inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & (((uint64_t)1 << n) - 1)); }
  inline uint32_t take(unsigned n) { const uint32_t v = peek(n); drop(n); return v; }
  size_t bitpos() const { return (size_t)(p - base) * 8 - cnt; }
  bool overrun() const { return pad * 8 > cnt; }
  void align_byte() { drop(cnt & 7); }
};

// ---------------------------------------------------------------------------------------------------- Huffman tables
// two-level lookup: the low LB (DB) bits of the stream index the primary table; codes longer than that go through a
// sub-table.  Entry: value << 16 | extra bits << 8 | kind << 5 | code bits to drop.
static constexpr int LB = 11, DB = 9, PB = 7;
enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
static inline uint32_t mk(uint32_t value, uint32_t extra, uint32_t kind) { return value << 16 | extra << 8 | kind << 5; }
static inline uint32_t e_kind(uint32_t e) { return (e >> 5) & 7; }
static inline uint32_t e_bits(uint32_t e) { return e & 31; }
static inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31; }
static inline uint32_t e_value(uint32_t e) { return e >> 16; }

struct Tables {
  uint32_t lit[(1 << LB) + (1 << 15)];
  uint32_t dist[(1 << DB) + (1 << 15)];
  uint32_t pre[1 << PB];
};

static inline uint32_t lit_entry(int s) {
  static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  if (s < 256) return mk((uint32_t)s, 0, K_LIT);
  if (s == 256) return mk(0, 0, K_EOB);
  if (s < 286) return mk(base[s - 257], extra[s - 257], K_LEN);
  return mk(0, 0, K_BAD);
}
static inline uint32_t dist_entry(int s) {
  static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  if (s < 30) return mk(base[s], extra[s], K_LEN);
  return mk(0, 0, K_BAD);
}
static inline uint32_t pre_entry(int s) { return mk((uint32_t)s, 0, K_LIT); }

static inline uint32_t rev_bits(uint32_t c, int l) {
  c = ((c & 0x5555) << 1) | ((c >> 1) & 0x5555);
  c = ((c & 0x3333) << 2) | ((c >> 2) & 0x3333);
  c = ((c & 0x0f0f) << 4) | ((c >> 4) & 0x0f0f);
  c = ((c & 0x00ff) << 8) | ((c >> 8) & 0x00ff);
  return c >> (16 - l);
}

// canonical code of `lens` into `tab` (pb primary bits).  0: complete, 1: incomplete (max_len says how long its codes
// are; 0 = no code at all), -1: over-subscribed
template <class F> static int build(const uint8_t *lens, int n, int pb, uint32_t *tab, F entry_of, int *max_len, bool need_complete = false) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  count[0] = 0;
  int maxl = 15;
  while (maxl && !count[maxl]) --maxl;
  *max_len = maxl;
  const uint32_t bad = mk(0, 0, K_BAD) | 1;
  long left = 1;
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= count[l];
    if (left < 0) return -1;
  }
  if (maxl && left > 0 && need_complete) return 1;  // (the block search asks before anything is filled in)
  for (int i = 0; i < (1 << pb); ++i) tab[i] = bad;
  if (!maxl) return 1;
  uint32_t next[16], code = 0;
  for (int l = 1; l <= 15; ++l) { next[l] = code; code = (code + (uint32_t)count[l]) << 1; }
  uint16_t rc[320];
  const uint32_t pmask = ((uint32_t)1 << pb) - 1;
  uint32_t off = (uint32_t)1 << pb;
  if (maxl > pb) {  // sub-tables: as deep as the longest code behind each primary index
    uint8_t submax[1 << LB];
    memset(submax, 0, (size_t)1 << pb);
    uint32_t nx[16];
    memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
      const int l = lens[s];
      if (!l) continue;
      const uint32_t r = rev_bits(nx[l]++, l);
      rc[s] = (uint16_t)r;
      if (l > pb && submax[r & pmask] < l) submax[r & pmask] = (uint8_t)l;
    }
    for (uint32_t g = 0; g <= pmask; ++g)
      if (submax[g]) {
        const uint32_t sb = (uint32_t)submax[g] - (uint32_t)pb;
        tab[g] = mk(off, sb, K_SUB) | (uint32_t)pb;
        for (uint32_t i = 0; i < ((uint32_t)1 << sb); ++i) tab[off + i] = bad;
        off += (uint32_t)1 << sb;
      }
  } else {
    for (int s = 0; s < n; ++s) {
      const int l = lens[s];
      if (l) rc[s] = (uint16_t)rev_bits(next[l]++, l);
    }
  }
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t e = entry_of(s), r = rc[s];
    if (l <= pb) {
      for (uint32_t i = r; i <= pmask; i += (uint32_t)1 << l) tab[i] = e | (uint32_t)l;
    } else {
      const uint32_t head = tab[r & pmask], sb = e_extra(head), so = e_value(head);
      for (uint32_t i = r >> pb; i < ((uint32_t)1 << sb); i += (uint32_t)1 << (l - pb)) tab[so + i] = e | (uint32_t)(l - pb);
    }
  }
  return left > 0 ? 1 : 0;
}

// header of a dynamic block behind its three type bits.  strict (block search): every code complete (the distance code
// may consist of at most one code); otherwise zlib's rule: incomplete only when all codes are one bit long.
static inline bool read_dynamic(Bits &b, Tables &t, bool strict) {
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  b.refill();
  const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  uint8_t pl[19] = {0};
  for (int i = 0; i < hclen; ++i) {
    if (b.cnt < 3) b.refill();
    pl[order[i]] = (uint8_t)b.take(3);
  }
  int maxl;
  if (build(pl, 19, PB, t.pre, pre_entry, &maxl, true) != 0) return false;
  uint8_t lens[320];
  const int total = hlit + hdist;
  for (int i = 0; i < total;) {
    b.refill();
    const uint32_t e = t.pre[b.peek(PB)];
    if (e_kind(e) != K_LIT) return false;
    b.drop(e_bits(e));
    const int sym = (int)e_value(e);
    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
    int rep;
    uint8_t val = 0;
    if (sym == 16) {
      if (!i) return false;
      val = lens[i - 1];
      rep = 3 + (int)b.take(2);
    } else if (sym == 17) rep = 3 + (int)b.take(3);
    else rep = 11 + (int)b.take(7);
    if (i + rep > total) return false;
    memset(lens + i, val, (size_t)rep);
    i += rep;
  }
  if (b.overrun() || lens[256] == 0) return false;
  const int rl = build(lens, hlit, LB, t.lit, lit_entry, &maxl, strict);
  if (rl < 0 || (rl == 1 && (strict || maxl != 1))) return false;
  const int rd = build(lens + hlit, hdist, DB, t.dist, dist_entry, &maxl);
  if (rd < 0) return false;
  if (rd == 1) {
    if (strict) {
      int used = 0;
      for (int i = 0; i < hdist; ++i) used += lens[hlit + i] != 0;
      if (used > 1) return false;
    } else if (maxl > 1) return false;
  }
  return true;
}

static inline const Tables &fixed_tables() {
  static const Tables *ft = [] {
    Tables *t = new Tables;
    uint8_t l[288], d[32];
    for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    for (int i = 0; i < 32; ++i) d[i] = 5;
    int m;
    build(l, 288, LB, t->lit, lit_entry, &m);
    build(d, 32, DB, t->dist, dist_entry, &m);
    return t;
  }();
  return *ft;
}

// ---------------------------------------------------------------------------------------------------- output
template <class T> struct RawBuf {
  T *d = nullptr;
  size_t cap = 0;
  RawBuf() = default;
  RawBuf(const RawBuf &) = delete;
  RawBuf &operator=(const RawBuf &) = delete;
  ~RawBuf() { free(d); }
  bool ensure(size_t n) {
    if (n <= cap) return true;
    size_t nc = cap ? cap : (size_t)1 << 20;
    while (nc < n) nc += nc / 2;
    T *nd = (T *)realloc(d, nc * sizeof(T));
    if (!nd) return false;
    d = nd; cap = nc;
    return true;
  }
};

static constexpr size_t WIN = 32768;
enum { RC_DONE = 0, RC_ROOM = 1, RC_ERR = 2 };

// the symbols of one Huffman-coded block.  out[0, pos) is what exists (the first WIN elements are the window before the
// span); `floor` = first position a match may reach (start of the member, or of the known part of the window).
template <class T> static int inflate_codes(Bits &b, const Tables &t, T *out, size_t &pos_io, size_t limit, size_t floor) {
  size_t pos = pos_io;
  const uint32_t *lt = t.lit, *dt = t.dist;
  int rc = RC_DONE;
  for (;;) {
    if (pos > limit) { rc = RC_ROOM; break; }
    b.refill();
    uint32_t e = lt[b.buf & ((1u << LB) - 1)];
    if (e_kind(e) == K_SUB) {
      b.drop(LB);
      e = lt[e_value(e) + b.peek(e_extra(e))];
    }
    b.drop(e_bits(e));
    if ((e & 0xE0) == 0) {  // literal; up to two more from the same 56 bits
      out[pos++] = (T)e_value(e);
      e = lt[b.buf & ((1u << LB) - 1)];
      if ((e & 0xE0) == 0) {
        b.drop(e_bits(e));
        out[pos++] = (T)e_value(e);
        e = lt[b.buf & ((1u << LB) - 1)];
        if ((e & 0xE0) == 0) {
          b.drop(e_bits(e));
          out[pos++] = (T)e_value(e);
        }
      }
      continue;
    }
    const uint32_t kind = e_kind(e);
    if (kind == K_EOB) break;
    if (kind != K_LEN) { rc = RC_ERR; break; }
    const size_t len = e_value(e) + b.take(e_extra(e));
    uint32_t d = dt[b.buf & ((1u << DB) - 1)];
    if (e_kind(d) == K_SUB) {
      b.drop(DB);
      d = dt[e_value(d) + b.peek(e_extra(d))];
    }
    if (e_kind(d) != K_LEN) { rc = RC_ERR; break; }
    b.drop(e_bits(d));
    const size_t dist = e_value(d) + b.take(e_extra(d));
    if (dist > pos - floor) { rc = RC_ERR; break; }
    T *dst = out + pos;
    const T *src = dst - dist;
    pos += len;
    constexpr size_t V = 16 / sizeof(T);
    if (dist >= V) {  // (may write up to V - 1 elements behind the match: the caller keeps that slack)
      T *const stop = dst + len;
      do {
        memcpy(dst, src, 16);
        dst += V; src += V;
      } while (dst < stop);
    } else if (dist >= V / 2) {
      T *const stop = dst + len;
      do {
        memcpy(dst, src, 8);
        dst += V / 2; src += V / 2;
      } while (dst < stop);
    } else if (dist == 1) {
      const T v = *src;
      for (size_t i = 0; i < len; ++i) dst[i] = v;
    } else {
      for (size_t i = 0; i < len; ++i) dst[i] = src[i];
    }
  }
  pos_io = pos;
  return rc;
}

struct Event { size_t pos; uint32_t crc, isize; };  // a member ended after `pos` bytes of this span
enum { ST_STOP = 0, ST_END = 1, ST_ERROR = 2 };

// gzip member header at byte `at` (RFC 1952 2.3).  0: parsed, `at` behind it; 1: no further member (end of input, or
// bytes that do not start with the magic: ignored as zlib does); -1: broken
static inline int skip_member_header(const uint8_t *m, size_t n, size_t &at) {
  if (at >= n) return 1;
  if (n - at < 2 || m[at] != 0x1f || m[at + 1] != 0x8b) return 1;
  if (n - at < 10 || m[at + 2] != 8 || (m[at + 3] & 0xE0)) return -1;
  const unsigned flg = m[at + 3];
  size_t q = at + 10;
  if (flg & 4) {
    if (q + 2 > n) return -1;
    q += 2 + (m[q] | ((size_t)m[q + 1] << 8));
  }
  for (unsigned f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
    if (flg & f) {
      while (q < n && m[q]) ++q;
      ++q;
    }
  if (flg & 2) {  // FHCRC: the low half of the CRC-32 of the header before it (zlib checks it: "header crc mismatch")
    if (q + 2 > n) return -1;
    const uint32_t want = m[q] | ((uint32_t)m[q + 1] << 8);
    if ((crc32_z(crc32(0L, Z_NULL, 0), m + at, q - at) & 0xffff) != want) return -1;
    q += 2;
  }
  if (q > n) return -1;
  at = q;
  return 0;
}

template <class T> struct SpanDecoder {
  RawBuf<T> out;       // [0, WIN): the window before the span; [WIN, pos): the span's output
  size_t pos = WIN, floor = 0;
  std::vector<Event> events;
  size_t end_bit = 0;
  int status = ST_ERROR;
  const char *err = "";
  Tables *tb = nullptr;
  uint8_t lut[256 + WIN];  // symbolic spans: symbol -> byte (identity, then the resolved window; set by the coordinator)

  void reset() { pos = WIN; floor = 0; events.clear(); status = ST_ERROR; err = ""; }
  bool room(size_t extra) { return out.ensure(pos + extra + 64); }

  // one block whose three type bits are still ahead.  *final_block = BFINAL
  bool block(Bits &b, const uint8_t *m, size_t n, bool *final_block, size_t max_pos) {
    b.refill();
    *final_block = b.take(1);
    const uint32_t type = b.take(2);
    if (type == 0) {
      b.align_byte();
      b.refill();
      const uint32_t len = b.take(16), nlen = b.take(16);
      if ((len ^ 0xffff) != nlen || b.overrun()) { err = "stored block length"; return false; }
      const size_t at = b.bitpos() >> 3;
      if (at + len > n) { err = "truncated stored block"; return false; }
      if (pos + len > max_pos || !room(len)) { err = "out of memory"; return false; }
      for (size_t i = 0; i < len; ++i) out.d[pos + i] = (T)m[at + i];
      pos += len;
      b.init(m, n, (at + len) * 8);
      return true;
    }
    const Tables *t;
    if (type == 1) t = &fixed_tables();
    else if (type == 2) {
      if (!read_dynamic(b, *tb, false)) { err = "dynamic block header"; return false; }
      t = tb;
    } else { err = "block type 3"; return false; }
    for (;;) {
      if (pos > max_pos) { err = "span output too large"; return false; }
      if (!room((size_t)1 << 20)) { err = "out of memory"; return false; }
      const int rc = inflate_codes<T>(b, *t, out.d, pos, out.cap - 64 - 258 - 16, floor);
      if (rc == RC_ERR) { err = "deflate data"; return false; }
      if (b.overrun()) { err = "truncated deflate stream"; return false; }  // (zeros behind the end decode for ever)
      if (rc == RC_DONE) break;
    }
    return true;
  }

  // decode from the reader's position (a member header when at_header, else a block start) until stop(bitpos) says so
  // at a block start, or the stream ends
  // (and once the span holds more than `soft_cap` elements: a span of highly compressible data -- the N runs of a genome --
  //  hands over at the next block start instead of growing without bound; the spans behind it are then decoded again)
  size_t soft_cap = (size_t)48 << 20;
  void run(Bits &b, const uint8_t *m, size_t n, bool at_header, bool check_first, const std::function<bool(size_t)> &stop) {
    bool first = !check_first;
    for (;;) {
      if (at_header) {
        size_t at = b.bitpos() >> 3;
        const int r = skip_member_header(m, n, at);
        if (r == 1) { end_bit = at * 8; status = ST_END; return; }
        if (r < 0) { err = "gzip member header"; status = ST_ERROR; return; }
        b.init(m, n, at * 8);
        floor = pos;
        at_header = false;
      }
      if (!first && (pos - WIN >= soft_cap || stop(b.bitpos()))) { end_bit = b.bitpos(); status = ST_STOP; return; }
      first = false;
      bool fin = false;
      if (!block(b, m, n, &fin, (size_t)-1 / 4)) { status = ST_ERROR; return; }
      if (fin) {
        b.align_byte();
        b.refill();
        const uint32_t crc = b.take(32);
        b.refill();
        const uint32_t isize = b.take(32);
        if (b.overrun()) { err = "truncated gzip trailer"; status = ST_ERROR; return; }
        events.push_back({pos - WIN, crc, isize});
        at_header = true;
      }
    }
  }
};

static inline const uint8_t *text_table() {
  static const uint8_t *tt = [] {
    uint8_t *t = new uint8_t[256]();
    for (int c = 32; c < 127; ++c) t[c] = 1;
    t[9] = t[10] = t[13] = 1;
    return t;
  }();
  return tt;
}

// ---------------------------------------------------------------------------------------------------- the team
struct ParallelGunzip {
  const uint8_t *m = nullptr;
  size_t n = 0;
  int team = 1;
  size_t span = (size_t)2 << 20;          // compressed bytes per span
  size_t search_max = (size_t)512 << 10;  // a span gives up looking for its block behind this many bytes
  // bytes of text after which a span hands over at the next block start.  (48 Mi: a span's symbol buffer is 2 bytes per byte
  // of text and only grows -- 256 Mi let one highly compressible stretch pin 512 MiB per span, times the team, times two sets:
  // ADVICE r04.  FASTQ spans of 2 MiB compressed hold 8-10 MiB of text: the cap does not touch them.)
  size_t span_cap = (size_t)48 << 20;
  // stream position between rounds (decode side)
  size_t cur_bit = 0;
  bool at_header = true, finished = false;
  uint8_t window[WIN];
  size_t win_len = 0;
  int dry_rounds = 0;  // rounds in which no span found a block (not text?): the search pauses
  int pause = 0;
  // member check (emit side)
  uLong run_crc = 0;
  uint64_t run_len = 0;
  std::string error;
  // statistics
  uint64_t n_rounds = 0, n_spans = 0, n_dropped = 0;
  double t_decode = 0, t_first = 0, t_resolve = 0, t_translate = 0;  // wall seconds (t_first: span 0 alone)
  std::atomic<long long> search_us{0};                               // block search, summed over the spans

  // the spans of one round.  Two sets: while one round is translated and checked, the next one is being decoded
  struct Set {
    SpanDecoder<uint8_t> d0;
    std::vector<SpanDecoder<uint16_t>> ds;
    std::vector<int> acc;        // the spans that follow one another exactly, in order (0 = d0)
    std::vector<size_t> o_at;    // their offsets in the round's text
    bool last_of_stream = false;
    std::string error;
    size_t len_of(size_t a) const { return o_at[a + 1] - o_at[a]; }
  };
  Set sets[2];
  int cur = 0;
  bool primed = false;
  std::vector<Tables> tabs;

  void open(const uint8_t *map, size_t len, int threads) {
    m = map; n = len;
    team = threads < 1 ? 1 : threads;
    // smaller spans for smaller files: several rounds, so that decoding and emitting overlap and the last round is short
    span = std::min((size_t)2 << 20, std::max((size_t)256 << 10, n / ((size_t)team * 6)));
    if (const char *e = getenv("KU_PGZIP_SPAN_KB")) span = (size_t)atol(e) << 10;
    if (const char *e = getenv("KU_PGZIP_SPAN_CAP_KB")) span_cap = (size_t)std::max(1L, atol(e)) << 10;  // test hook
    if (span < 4096) span = 4096;
    if (search_max > span) search_max = span;
    cur_bit = 0; at_header = true; finished = false;
    win_len = 0; run_crc = crc32(0L, Z_NULL, 0); run_len = 0;
    tabs = std::vector<Tables>((size_t)team);
    for (Set &st : sets) {
      st.ds = std::vector<SpanDecoder<uint16_t>>((size_t)team);
      st.d0.tb = &tabs[0];
      st.d0.soft_cap = span_cap;
      for (int i = 1; i < team; ++i) { st.ds[(size_t)i].tb = &tabs[(size_t)i]; st.ds[(size_t)i].soft_cap = span_cap; }
      st.acc.clear();
    }
    primed = false; cur = 0;
  }

  // first block start at or behind byte `lo` (searched up to byte `hi`) that validates; decodes that block into sd.
  // Returns the bit position or (size_t)-1; on success `b` stands behind the block.
  size_t find_block(size_t lo, size_t hi, SpanDecoder<uint16_t> &sd, Bits &b) {
    const uint8_t *tt = text_table();
    if (hi + 16 > n) hi = n > 16 ? n - 16 : 0;
    for (size_t bit = lo * 8; bit < hi * 8; ++bit) {
      uint64_t w;
      memcpy(&w, m + (bit >> 3), 8);
      w >>= bit & 7;
      if ((w & 7) != 4) continue;                               // BFINAL 0, BTYPE 10
      if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;  // HLIT, HDIST
      b.init(m, n, bit + 3);
      if (!read_dynamic(b, *sd.tb, true)) continue;
      sd.reset();
      bool ok = true;
      for (;;) {
        if (sd.pos > WIN + ((size_t)8 << 20) || !sd.room((size_t)1 << 20)) { ok = false; break; }
        const int rc = inflate_codes<uint16_t>(b, *sd.tb, sd.out.d, sd.pos, sd.out.cap - 64 - 258 - 16, 0);
        if (rc == RC_ERR || b.overrun()) { ok = false; break; }
        if (rc == RC_DONE) break;
      }
      if (!ok || b.overrun() || sd.pos - WIN < 512) continue;
      const uint16_t *o = sd.out.d;
      for (size_t i = WIN; i < sd.pos; ++i)
        if (o[i] < 256 && !tt[o[i]]) { ok = false; break; }
      if (!ok) continue;
      // the block behind it must start sensibly too
      Bits c = b;
      c.refill();
      c.take(1);
      const uint32_t type = c.take(2);
      if (type == 3) continue;
      if (type == 0) {
        c.align_byte();
        c.refill();
        const uint32_t len = c.take(16), nlen = c.take(16);
        if ((len ^ 0xffff) != nlen) continue;
      } else if (type == 2) {
        Tables *scratch = sd.tb;  // (the block's own tables are not needed any more)
        if (!read_dynamic(c, *scratch, true)) continue;
      }
      return bit;
    }
    return (size_t)-1;
  }

  static constexpr long long PENDING = -1, NONE = -2;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  // decode side of a round: up to `team` spans from cur_bit on into `st`; moves cur_bit / window behind them
  void decode(Set &st) {
    st.acc.clear();
    st.o_at.assign(1, 0);
    st.last_of_stream = false;
    st.error.clear();
    if (finished) return;
    ++n_rounds;
    const size_t lo_byte = cur_bit >> 3;
    int T = team;
    if (pause > 0) { T = 1; --pause; }
    std::vector<size_t> start((size_t)T, 0);
    int used = 1;
    for (int i = 1; i < T; ++i) {
      start[(size_t)i] = lo_byte + (size_t)i * span;
      if (start[(size_t)i] + 64 < n) used = i + 1;
    }
    T = used;
    const size_t round_end = lo_byte + (size_t)T * span;
    const bool to_the_end = round_end >= n;
    std::vector<std::atomic<long long>> sync((size_t)T);
    for (auto &s : sync) s.store(PENDING);
    // span k stops at the first block start at or behind the one the next span (that found any) starts with
    auto stop_for = [&](int k) {
      return [&, nxt = k + 1](size_t bitpos) mutable -> bool {
        while (nxt < T) {
          if (bitpos < start[(size_t)nxt] * 8) return false;
          long long s;
          while ((s = sync[(size_t)nxt].load(std::memory_order_acquire)) == PENDING) std::this_thread::yield();
          if (s == NONE) { ++nxt; continue; }
          return bitpos >= (size_t)s;
        }
        return !to_the_end && bitpos >= round_end * 8;
      };
    };
    const double tA = now();
    std::vector<std::thread> th;
    for (int i = 1; i < T; ++i)
      th.emplace_back([&, i] {
        SpanDecoder<uint16_t> &sd = st.ds[(size_t)i];
        sd.reset();
        // the window the span cannot see: symbols 256 + j (nothing ever writes there)
        if (!sd.out.ensure(WIN + ((size_t)1 << 20))) { sync[(size_t)i].store(NONE, std::memory_order_release); return; }
        for (size_t j = 0; j < WIN; ++j) sd.out.d[j] = (uint16_t)(256 + j);
        Bits b;
        const double ts = now();
        const size_t bit = find_block(start[(size_t)i], start[(size_t)i] + search_max, sd, b);
        search_us += (long long)((now() - ts) * 1e6);
        if (bit == (size_t)-1) { sync[(size_t)i].store(NONE, std::memory_order_release); return; }
        sync[(size_t)i].store((long long)bit, std::memory_order_release);
        sd.run(b, m, n, false, true, stop_for(i));
      });
    SpanDecoder<uint8_t> &d0 = st.d0;
    d0.reset();
    if (!d0.out.ensure(WIN + ((size_t)1 << 20))) {
      d0.err = "out of memory";
    } else {
      memset(d0.out.d, 0, WIN - win_len);
      memcpy(d0.out.d + WIN - win_len, window + WIN - win_len, win_len);
      d0.floor = WIN - win_len;
      Bits b;
      b.init(m, n, cur_bit);
      d0.run(b, m, n, at_header, false, stop_for(0));
    }
    t_first += now() - tA;
    for (auto &t : th) t.join();
    const double tB = now();
    t_decode += tB - tA;
    // ---- which spans follow one another exactly
    st.acc.push_back(0);
    bool found_any = false;
    for (int k = 0;;) {
      if ((k ? st.ds[(size_t)k].status : d0.status) != ST_STOP) break;
      const size_t eb = k ? st.ds[(size_t)k].end_bit : d0.end_bit;
      int j = k + 1;
      while (j < T && sync[(size_t)j].load() == NONE) ++j;
      if (j >= T || (size_t)sync[(size_t)j].load() != eb) break;
      st.acc.push_back(j);
      k = j;
    }
    for (int i = 1; i < T; ++i)
      if (sync[(size_t)i].load() >= 0) {
        found_any = true;
        bool in = false;
        for (int a : st.acc) in |= a == i;
        n_dropped += !in;
      }
    n_spans += st.acc.size();
    if (T > 1 && !found_any) {
      if (++dry_rounds >= 2) pause = 16 << std::min(dry_rounds - 2, 6);
    } else if (T > 1) dry_rounds = 0;
    const int last = st.acc.back();
    const int last_status = last ? st.ds[(size_t)last].status : d0.status;
    if (last_status == ST_ERROR) {
      st.error = std::string("corrupt gzip data (") + (last ? st.ds[(size_t)last].err : d0.err) + ")";
      st.acc.clear();  // (what was decoded in front of the damage is not handed out)
      finished = true;
      return;
    }
    // ---- the windows in order: only the last 32 KiB of every span are resolved here
    for (size_t a = 0; a < st.acc.size(); ++a) st.o_at.push_back(st.o_at[a] + (st.acc[a] ? st.ds[(size_t)st.acc[a]].pos : d0.pos) - WIN);
    for (size_t a = 0; a < st.acc.size(); ++a) {
      const size_t len = st.len_of(a);
      uint8_t nw[WIN];
      if (st.acc[a] == 0) {
        if (len >= WIN) memcpy(nw, d0.out.d + d0.pos - WIN, WIN);
        else { memcpy(nw, window + len, WIN - len); memcpy(nw + WIN - len, d0.out.d + WIN, len); }
      } else {
        SpanDecoder<uint16_t> &sd = st.ds[(size_t)st.acc[a]];
        for (int c = 0; c < 256; ++c) sd.lut[c] = (uint8_t)c;
        memcpy(sd.lut + 256, window, WIN);
        const size_t take = std::min(len, WIN);
        if (len < WIN) memcpy(nw, window + len, WIN - len);
        const uint16_t *s = sd.out.d + sd.pos - take;
        for (size_t i = 0; i < take; ++i) nw[WIN - take + i] = sd.lut[s[i]];
      }
      memcpy(window, nw, WIN);
      win_len = std::min(WIN, win_len + len);
    }
    cur_bit = last ? st.ds[(size_t)last].end_bit : d0.end_bit;
    at_header = false;
    if (last_status == ST_END) { finished = true; st.last_of_stream = true; }
    t_resolve += now() - tB;
  }

  // emit side: every accepted span to bytes (and its CRCs) side by side into out[0, n_out); members checked
  bool emit(Set &st, const std::function<char *(size_t)> &place, size_t &n_out) {
    n_out = 0;
    const double tC = now();
    const size_t total = st.o_at.back();
    char *const out_d = place(total);
    if (!out_d) { error = "no room for the text"; return false; }
    struct Seg { uLong crc; size_t len; };
    std::vector<std::vector<Seg>> segs(st.acc.size());
    auto translate = [&](size_t a) {
      char *dst = out_d + st.o_at[a];
      const size_t len = st.len_of(a);
      const std::vector<Event> *ev;
      if (st.acc[a] == 0) {
        memcpy(dst, st.d0.out.d + WIN, len);
        ev = &st.d0.events;
      } else {
        SpanDecoder<uint16_t> &sd = st.ds[(size_t)st.acc[a]];
        const uint16_t *s = sd.out.d + WIN;
        const uint8_t *lut = sd.lut;  // (most symbols of a span are unknowns in FASTQ: one table, no branch)
        size_t i = 0;
        for (; i + 8 <= len; i += 8) {
          uint64_t q0, q1;
          memcpy(&q0, s + i, 8);
          memcpy(&q1, s + i + 4, 8);
          const uint64_t r = (uint64_t)lut[q0 & 0xffff] | (uint64_t)lut[(q0 >> 16) & 0xffff] << 8 | (uint64_t)lut[(q0 >> 32) & 0xffff] << 16 |
                             (uint64_t)lut[q0 >> 48] << 24 | (uint64_t)lut[q1 & 0xffff] << 32 | (uint64_t)lut[(q1 >> 16) & 0xffff] << 40 |
                             (uint64_t)lut[(q1 >> 32) & 0xffff] << 48 | (uint64_t)lut[q1 >> 48] << 56;
          memcpy(dst + i, &r, 8);
        }
        for (; i < len; ++i) dst[i] = (char)lut[s[i]];
        ev = &sd.events;
      }
      size_t at = 0;
      for (size_t e = 0; e <= ev->size(); ++e) {
        const size_t hi = e < ev->size() ? (*ev)[e].pos : len;
        segs[a].push_back({crc_of((const uint8_t *)dst + at, hi - at), hi - at});
        at = hi;
      }
    };
    std::vector<std::thread> th;
    for (size_t a = 1; a < st.acc.size(); ++a) th.emplace_back(translate, a);
    translate(0);
    for (auto &t : th) t.join();
    t_translate += now() - tC;
    for (size_t a = 0; a < st.acc.size(); ++a) {
      const std::vector<Event> &ev = st.acc[a] ? st.ds[(size_t)st.acc[a]].events : st.d0.events;
      for (size_t e = 0; e < segs[a].size(); ++e) {
        run_crc = crc32_combine(run_crc, segs[a][e].crc, (z_off_t)segs[a][e].len);
        run_len += segs[a][e].len;
        if (e < ev.size()) {
          if ((uint32_t)run_crc != ev[e].crc || (uint32_t)run_len != ev[e].isize) { error = "corrupt gzip data (crc or length of a member)"; return false; }
          run_crc = crc32(0L, Z_NULL, 0);
          run_len = 0;
        }
      }
    }
    if (st.last_of_stream && run_len != 0) { error = "truncated gzip data"; return false; }
    n_out = total;
    return true;
  }

  // the next piece of text: out[0, n_out).  false: nothing more (end of the stream, or `error` says what broke).
  // The round behind it is decoded meanwhile.
  bool round(RawBuf<char> &out, size_t &n_out) {
    return round_to([&](size_t total) { return out.ensure(total + 8) ? out.d : nullptr; }, n_out);
  }
  // ... to where `place(bytes)` says (nullptr: give up)
  bool round_to(const std::function<char *(size_t)> &place, size_t &n_out) {
    n_out = 0;
    if (!error.empty()) return false;
    if (!primed) { decode(sets[0]); primed = true; cur = 0; }
    Set &st = sets[cur];
    if (st.acc.empty()) {  // decode had nothing left, or ran into damage
      if (!st.error.empty()) error = st.error;
      return false;
    }
    std::thread ahead;
    Set &nx = sets[1 - cur];
    ahead = std::thread([&] { decode(nx); });
    const bool ok = emit(st, place, n_out);
    ahead.join();
    if (!ok) return false;
    cur = 1 - cur;
    return true;
  }
};

}  // namespace ku_pgzip
