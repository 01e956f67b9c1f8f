// gzip output written by a team: the Kraken lines of a batch are formatted in parts by the -t helpers; each helper also
// deflates its part (raw deflate closed with a sync flush, so that it ends on a byte boundary), the writer puts the parts
// behind one another -- ONE deflate stream in one gzip member, as pigz builds it -- and keeps the member's CRC-32 with
// crc32_combine.  (The reference writes `-o x.gz` through ogzstream: one deflate on the writing thread,
// src/classify.cpp:133-148; at 100 MB/s that is 6 s for the 600 MB of lines of 10 M reads, twenty times the pipeline.)
// A run that dies in the middle (a fatal input error: the executable leaves through _exit) leaves a member without its last
// block and trailer -- `gzip -t` then reports an unexpected end of file, as it does for the reference's ogzstream file.
#pragma once
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ku_pgzout {

// [p, p + n) deflated (raw, not the stream's last block) into a malloc'ed buffer; the caller frees it.  nullptr: zlib failed
inline unsigned char *deflate_part(const char *p, size_t n, size_t *out_len, uLong *crc) {
  // one deflate state per thread, given back when the thread ends (~260 KB of zlib state per formatting helper otherwise)
  struct State {
    z_stream z;
    bool ready = false;
    ~State() { if (ready) deflateEnd(&z); }
  };
  thread_local State st;
  z_stream &z = st.z;
  if (!st.ready) {
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return nullptr;
    st.ready = true;
  } else if (deflateReset(&z) != Z_OK) return nullptr;
  size_t cap = deflateBound(&z, (uLong)n) + 64;
  unsigned char *out = (unsigned char *)malloc(cap);
  if (!out) return nullptr;
  z.next_in = (Bytef *)const_cast<char *>(p);
  z.next_out = out;
  size_t in_left = n, used = 0;
  for (;;) {
    z.avail_in = (uInt)(in_left > ((size_t)1 << 30) ? ((size_t)1 << 30) : in_left);
    in_left -= z.avail_in;
    const bool last = in_left == 0;
    for (;;) {
      z.avail_out = (uInt)((cap - used) > ((size_t)1 << 30) ? ((size_t)1 << 30) : (cap - used));
      const uInt before = z.avail_out;
      const int rc = deflate(&z, last ? Z_SYNC_FLUSH : Z_NO_FLUSH);
      if (rc != Z_OK && rc != Z_BUF_ERROR) { free(out); return nullptr; }
      used += before - z.avail_out;
      if (z.avail_out != 0 && z.avail_in == 0) break;  // everything handed over has been consumed and flushed
      if (z.avail_out == 0) {
        cap += cap / 2 + 4096;
        unsigned char *nb = (unsigned char *)realloc(out, cap);
        if (!nb) { free(out); return nullptr; }
        out = nb;
        z.next_out = out + used;
      }
    }
    if (last) break;
  }
  *out_len = used;
  *crc = crc32_z(crc32(0L, Z_NULL, 0), (const Bytef *)p, n);
  return out;
}

// the member around the parts
struct Member {
  FILE *f = nullptr;
  uLong crc = 0;
  unsigned long long total = 0;
  bool open(const char *path) {
    f = fopen(path, "wb");
    if (!f) return false;
    static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    crc = crc32(0L, Z_NULL, 0);
    total = 0;
    return fwrite(header, 1, sizeof header, f) == sizeof header;
  }
  bool put(const unsigned char *comp, size_t comp_len, uLong part_crc, size_t raw_len) {
    if (comp_len && fwrite(comp, 1, comp_len, f) != comp_len) return false;
    crc = crc32_combine(crc, part_crc, (z_off_t)raw_len);
    total += raw_len;
    return true;
  }
  bool close() {
    if (!f) return true;
    const unsigned char tail[10] = {0x03, 0x00,  // an empty final block (fixed codes: BFINAL 1, BTYPE 01, end of block)
                                    (unsigned char)crc, (unsigned char)(crc >> 8), (unsigned char)(crc >> 16), (unsigned char)(crc >> 24),
                                    (unsigned char)total, (unsigned char)(total >> 8), (unsigned char)(total >> 16), (unsigned char)(total >> 24)};
    const bool ok = fwrite(tail, 1, sizeof tail, f) == sizeof tail;
    const bool closed = fclose(f) == 0;
    f = nullptr;
    return ok && closed;
  }
};

}  // namespace ku_pgzout
