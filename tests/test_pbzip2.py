""".bz2 input decoded by a team of threads (krakenuniq_amd/csrc/ku_pbzip2.h; the reference reads .bz2 through bxz::ifstream
over bzlib, src/seqreader.hpp:48): the bytes handed to the parser are the bytes bzlib gives, whatever the block size, the
number of streams in the file and the team; damaged files are refused.  Checked through bin/seqio_dump -Z J (the decoder
alone, J threads) against Python's bz2, and through the reader / the region parsers against the plain text."""
import bz2
import os
import subprocess

import numpy as np
import pytest

from test_pgzip import fasta, fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "krakenuniq_amd", "bin", "seqio_dump")

FQ = fastq(24000, seed=31)
FA = fasta(1200, seed=32)
RUNS = (b"\0" * 3_000_000 + b"A" * 260 + bytes(range(256)) * 1000 + b"B" * 4 + b"C" * 5 + b"D" * 259 + b"E" * 3 + b"F" * 255 + b"xyz" * 70000 +
        b"G" * 1_000_003)
RND = np.random.default_rng(7).integers(0, 256, 2_500_000, dtype=np.uint8).tobytes()
STREAMS = {
    "fastq_900k": lambda: (bz2.compress(FQ, 9), FQ),
    "fastq_500k": lambda: (bz2.compress(FQ, 5), FQ),
    "fastq_100k": lambda: (bz2.compress(FQ, 1), FQ),
    "fasta": lambda: (bz2.compress(FA), FA),
    "two_streams": lambda: (bz2.compress(FQ[:3000000]) + bz2.compress(FQ[3000000:], 3), FQ),
    "many_streams": lambda: (b"".join(bz2.compress(FQ[i:i + 150000], 1) for i in range(0, len(FQ), 150000)), FQ),  # pbzip2 writes such files
    "runs": lambda: (bz2.compress(RUNS), RUNS),              # the run-length stage in front of the block sort, at its edges (4, 5, 255, 259 ...)
    "random_bytes": lambda: (bz2.compress(RND), RND),        # all 256 byte values, long codes
    "empty": lambda: (bz2.compress(b""), b""),
    "empty_then_data": lambda: (bz2.compress(b"") + bz2.compress(FA), FA),
    "tiny": lambda: (bz2.compress(b"@a\nACGT\n+\nIIII\n"), b"@a\nACGT\n+\nIIII\n"),
    "trailing_garbage": lambda: (bz2.compress(FQ) + b"\0" * 1000, FQ),  # bzip2: "trailing garbage after EOF ignored"
}


@pytest.mark.parametrize("shape", sorted(STREAMS))
def test_team_decode_equals_bzlib(shape, tmp_path):
    assert os.path.exists(DUMP), "build with make -C krakenuniq_amd/csrc"
    blob, want = STREAMS[shape]()
    p = tmp_path / "t.bz2"
    p.write_bytes(blob)
    for team in (1, 3, 8):
        r = subprocess.run([DUMP, "-Z", str(team), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, (shape, team, r.stderr.decode()[-300:])
        assert r.stdout == want, (shape, team, len(r.stdout), len(want))


@pytest.mark.parametrize("damage", ["truncated", "flipped_bit", "block_crc", "combined_crc", "second_stream"])
def test_damaged_files_are_refused(damage, tmp_path):
    blob = bz2.compress(FQ, 3)
    at = len(blob) // 2
    bad = {"truncated": blob[:at], "flipped_bit": blob[:at] + bytes([blob[at] ^ 0x10]) + blob[at + 1:],
           "block_crc": blob[:10] + bytes([blob[10] ^ 1]) + blob[11:],          # (the first block's crc follows "BZh3" and the magic number)
           "combined_crc": blob[:-5] + bytes([blob[-5] ^ 0x40]) + blob[-4:],   # (the last 32 bits in front of the padding)
           "second_stream": blob + bz2.compress(FA)[:-20]}[damage]
    with pytest.raises((OSError, ValueError, EOFError)):
        d = bz2.BZ2Decompressor()
        d.decompress(bad)
        if not d.eof:
            raise EOFError
        if d.unused_data:
            d2 = bz2.BZ2Decompressor()
            d2.decompress(d.unused_data)
            if not d2.eof:
                raise EOFError
    p = tmp_path / "t.bz2"
    p.write_bytes(bad)
    for team in (1, 4):
        r = subprocess.run([DUMP, "-Z", str(team), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode != 0, (damage, team)


def test_reader_and_region_parsers_take_bz2(tmp_path):
    """records of a .bz2 file == records of the text: through the sequential reader (with and without its producer side),
    through a pipe (read to its end first: the blocks are found in memory), as mate pairs, and through the region parsers
    over the growing text; a truncated file is a data error"""
    text = fastq(9000, seed=41)
    plain = tmp_path / "r.fq"
    plain.write_bytes(text)
    z1 = tmp_path / "r1.fq.bz2"
    z1.write_bytes(bz2.compress(text, 1))
    t2 = fastq(9000, seed=42)
    p2 = tmp_path / "r2.fq"
    p2.write_bytes(t2)
    z2 = tmp_path / "r2.fq.bz2"
    z2.write_bytes(bz2.compress(t2[:1000000], 2) + bz2.compress(t2[1000000:], 9))

    def run(args, **env):
        r = subprocess.run([DUMP] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        return r.stdout
    want = run([str(plain)])
    assert run([str(z1)]) == want
    assert run(["-T", str(z1)], KU_PBZIP2_TEAM="3") == want
    assert run(["-j", "4", str(z1)]) == want
    assert run(["-j", "5", str(z1)], KU_REGION_KB="64", KU_TEXT_AHEAD_MB="1", KU_PBZIP2_TEAM="2") == want
    r = subprocess.run(["bash", "-c", f"{DUMP} <(cat {z1})"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == want
    assert run(["-P", "-T", str(z1), str(z2)]) == run(["-P", str(plain), str(p2)])
    bad = tmp_path / "bad.fq.bz2"
    bad.write_bytes(z1.read_bytes()[:len(z1.read_bytes()) // 2])
    for args in (["-T", str(bad)], ["-j", "4", str(bad)]):
        r = subprocess.run([DUMP] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 65 and b"bzip2" in r.stderr, args


def test_random_damage_never_yields_other_bytes(tmp_path):
    """as for .gz (tests/test_pgzip.py): the run fails or hands out exactly the original text"""
    from test_pgzip import corrupt
    rng = np.random.default_rng(78)
    src = FQ[:1_200_000]
    blob = bz2.compress(src, 1)
    p = tmp_path / "d.bz2"
    refused = 0
    for it in range(24):
        p.write_bytes(corrupt(blob, rng))
        r = subprocess.run([DUMP, "-Z", "3", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode != 0 or r.stdout == src, it
        refused += r.returncode != 0
    assert refused >= 18
