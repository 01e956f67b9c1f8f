#!/usr/bin/env python3
"""path -> path.gz as ONE deflate stream inside one gzip member, written the way pigz writes: chunks deflated side by
side, each closed with a sync flush, concatenated (no member boundaries, no index -- what gzip(1) gives, only faster to make:
`gzip -6` of 3 GB takes a minute).    python scripts/write_one_stream_gz.py <file> [<out.gz>] [level]"""
import multiprocessing
import os
import struct
import sys
import zlib

CH = 32 << 20


def _deflate(args):
    path, off, last, level = args
    with open(path, "rb") as f:
        f.seek(off)
        d = f.read(CH)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(d) + c.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(d), len(d)


def _combine(crc1, crc2, len2):
    """crc32 of A + B from crc32(A), crc32(B), len(B) (zlib's crc32_combine: the operator "append len2 zero bytes" by squaring)"""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    if len2 == 0:
        return crc1
    odd = [0xedb88320] + [1 << n for n in range(31)]  # one zero bit
    even = square(odd)   # two
    odd = square(even)   # four
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def write(path, out=None, level=6, procs=None):
    out = out or path + ".gz"
    size = os.path.getsize(path)
    offs = list(range(0, size, CH)) or [0]
    crc = 0
    with multiprocessing.Pool(procs or min(32, os.cpu_count() or 1)) as pool, open(out, "wb") as g:
        g.write(b"\x1f\x8b\x08\0\0\0\0\0\0\x03")
        for comp, c1, n1 in pool.imap(_deflate, [(path, o, o == offs[-1], level) for o in offs]):
            g.write(comp)
            crc = _combine(crc, c1, n1)
        g.write(struct.pack("<II", crc, size & 0xffffffff))
    return out


if __name__ == "__main__":
    o = write(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 6)
    print(f"{o}: {os.path.getsize(o)} bytes")
