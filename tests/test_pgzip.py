"""One gzip stream inflated by a team of threads (krakenuniq_amd/csrc/ku_pgzip.h), the .gz side of the classify executable's
input stage.  The reference reads .gz through zlib's gzread (src/seqreader.cpp:26-133 via kseq/gzFile); the bytes handed
to the parser must be the bytes zlib would hand over, for every shape a deflate stream can take, and damaged files must be
refused.  Checked through bin/seqio_dump -z J (the inflater alone, J threads) against Python's zlib."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "krakenuniq_amd", "bin", "seqio_dump")


def fastq(n, seed=1):
    r = np.random.default_rng(seed)
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, (n, 150))]
    quals = (r.integers(0, 40, (n, 150)) // 8 * 8 + 33).astype(np.uint8)
    return b"".join(b"@read%d some/description\n" % i + seqs[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n" for i in range(n))


def fasta(n, seed=3):
    r = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(r.integers(100, 5000))
        s = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[r.integers(0, 9, L)].tobytes()
        out.append(b">seq%d desc\n" % i + b"\n".join(s[j:j + 60] for j in range(0, L, 60)) + b"\n")
    return b"".join(out)


def deflate_raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = []
    for i in range(0, len(data), flush_every):  # (what pigz and "gzip --rsyncable"-like writers leave: empty stored blocks)
        out.append(c.compress(data[i:i + flush_every]))
        out.append(c.flush(zlib.Z_SYNC_FLUSH if (i // flush_every) % 2 else zlib.Z_FULL_FLUSH))
    out.append(c.flush())
    return b"".join(out)


def member(data, raw=None, flg=0, extra=b"", hcrc_xor=0):
    raw = deflate_raw(data) if raw is None else raw
    head = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\x03" + extra
    if flg & 2:  # FHCRC
        head += struct.pack("<H", (zlib.crc32(head) & 0xffff) ^ hcrc_xor)
    return head + raw + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


def zlib_accepts(blob):
    """every member decodes to its end with a matching trailer; bytes behind the last member that do not start with the gzip
    magic are ignored (gzread's rule)"""
    while blob[:2] == b"\x1f\x8b":
        d = zlib.decompressobj(31)
        try:
            d.decompress(blob)
        except zlib.error:
            return False
        if not d.eof:
            return False
        blob = d.unused_data
    return True


def inflate(path, team, span_kb=None):
    env = dict(os.environ)
    if span_kb:
        env["KU_PGZIP_SPAN_KB"] = str(span_kb)
    return subprocess.run([DUMP, "-z", str(team), str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)


FQ = fastq(24000)
FA = fasta(1500)
RND = np.random.default_rng(5).integers(0, 256, 1_500_000, dtype=np.uint8).tobytes()
BINC = (np.random.default_rng(6).integers(0, 4, 3_000_000, dtype=np.uint8) * 50).tobytes()
STREAMS = {
    "fastq_level1": lambda: (member(FQ, deflate_raw(FQ, 1)), FQ),
    "fastq_level6": lambda: (member(FQ), FQ),
    "fastq_level9": lambda: (member(FQ, deflate_raw(FQ, 9)), FQ),
    "fasta": lambda: (member(FA), FA),
    "two_members": lambda: (member(FQ[:3000000]) + member(FQ[3000000:]), FQ),
    "many_members": lambda: (b"".join(member(FQ[i:i + 200000]) for i in range(0, len(FQ), 200000)), FQ),
    "stored_blocks": lambda: (member(FQ, deflate_raw(FQ, 0)), FQ),
    "fixed_huffman": lambda: (member(FQ, deflate_raw(FQ, 6, zlib.Z_FIXED)), FQ),
    "huffman_only": lambda: (member(FQ, deflate_raw(FQ, 6, zlib.Z_HUFFMAN_ONLY)), FQ),
    "rle": lambda: (member(FQ, deflate_raw(FQ, 6, zlib.Z_RLE)), FQ),
    "flush_points": lambda: (member(FQ, deflate_raw(FQ, 6, flush_every=131072)), FQ),
    "random_bytes": lambda: (member(RND), RND),                 # not text, incompressible: stored blocks, no span validates
    "binary_compressible": lambda: (member(BINC), BINC),        # not text: the search never validates, one decoder
    "empty": lambda: (member(b""), b""),
    "tiny": lambda: (member(b"@a\nACGT\n+\nIIII\n"), b"@a\nACGT\n+\nIIII\n"),
    "trailing_zeros": lambda: (member(FQ) + b"\0" * 1000, FQ),  # zlib's gzread ignores what does not start with the magic
    "header_fields": lambda: (member(FQ, flg=4 | 8 | 16 | 2, extra=b"\x05\0hello" + b"name.fq\0" + b"a comment\0"), FQ),
}


@pytest.mark.parametrize("shape", sorted(STREAMS))
def test_team_inflate_equals_zlib(shape, tmp_path):
    """every stream shape, with one, two, four and seven threads and with spans of 2 MiB (default), 64 KiB and 4 KiB of
    compressed data (many rounds, spans smaller than a deflate block, spans that find no block start)"""
    assert os.path.exists(DUMP), "build with make -C krakenuniq_amd/csrc"
    blob, want = STREAMS[shape]()
    assert zlib_accepts(blob)
    p = tmp_path / "t.gz"
    p.write_bytes(blob)
    for span_kb in (None, 64, 4):
        for team in (1, 2, 4, 7):
            r = inflate(p, team, span_kb)
            assert r.returncode == 0, (shape, team, span_kb, r.stderr.decode()[-300:])
            assert r.stdout == want, (shape, team, span_kb, len(r.stdout), len(want))


def test_highly_compressible_spans_hand_over_early(tmp_path):
    """the N runs of a genome FASTA expand a thousandfold: a span that holds more text than its cap stops at the next block
    start and the next round goes on from there (memory stays bounded); same bytes"""
    text = b">chr\n" + (b"N" * 60 + b"\n") * 400000 + FA + (b"N" * 60 + b"\n") * 300000 + FA[:100000]
    p = tmp_path / "n.fa.gz"
    p.write_bytes(member(text))
    for cap_kb, span_kb in ((256, 16), (1024, 4), (64, 64)):
        r = subprocess.run([DUMP, "-z", "4", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, KU_PGZIP_SPAN_CAP_KB=str(cap_kb), KU_PGZIP_SPAN_KB=str(span_kb)))
        assert r.returncode == 0 and r.stdout == text, (cap_kb, span_kb, r.stderr.decode()[-200:])


def test_spans_are_really_decoded_side_by_side(tmp_path):
    """the block search finds the deflate blocks of FASTQ text and the chain of spans holds: with four threads about four
    spans per round, none dropped"""
    p = tmp_path / "t.gz"
    p.write_bytes(member(FQ))
    r = subprocess.run([DUMP, "-n", "-z", "4", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KU_PGZIP_SPAN_KB="256"))
    assert r.returncode == 0
    import re
    m = re.search(r"(\d+) rounds, (\d+) spans, (\d+) dropped", r.stderr.decode())
    rounds, spans, dropped = map(int, m.groups())
    assert spans >= 3 * rounds and dropped == 0, r.stderr.decode()


@pytest.mark.parametrize("damage", ["truncated", "no_trailer", "flipped_bit", "bad_crc", "bad_isize", "garbage_member", "header_crc"])
def test_damaged_files_are_refused(damage, tmp_path):
    """as gzread fails on them (Z_DATA_ERROR / Z_BUF_ERROR): nonzero exit, whatever the team"""
    blob = member(FQ)
    at = len(blob) // 2
    bad = {"truncated": blob[:at], "no_trailer": blob[:-8], "flipped_bit": blob[:at] + bytes([blob[at] ^ 0x10]) + blob[at + 1:],
           "bad_crc": blob[:-8] + b"\0\0\0\0" + blob[-4:], "bad_isize": blob[:-4] + b"\1\0\0\0",
           "garbage_member": blob + b"\x1f\x8b\x08\0\0\0\0\0\0\x03" + b"\xff" * 64,
           "header_crc": member(FQ, flg=2 | 8, extra=b"name\0", hcrc_xor=1)}[damage]
    p = tmp_path / "t.gz"
    p.write_bytes(bad)
    assert not zlib_accepts(bad)
    for team in (1, 4):
        for span_kb in (None, 64):
            r = inflate(p, team, span_kb)
            assert r.returncode != 0, (damage, team, span_kb)


def test_reader_takes_plain_gz_through_the_team(tmp_path):
    """the executable's reader (-T: with its producer side) parses a plain .gz through the team: same records as the text and
    as zlib's reader (KU_NO_PGZIP=1), single file and mate pairs"""
    text = fastq(9000, seed=11)
    plain = tmp_path / "r.fq"
    plain.write_bytes(text)
    z1 = tmp_path / "r1.fq.gz"
    z1.write_bytes(member(text))
    z2 = tmp_path / "r2.fq.gz"
    z2.write_bytes(member(fastq(9000, seed=12), deflate_raw(fastq(9000, seed=12), 1)))

    def run(args, **env):
        r = subprocess.run([DUMP] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        return r.stdout
    want = run([str(plain)])
    assert run(["-T", str(z1)]) == want
    assert run(["-T", str(z1)], KU_PGZIP_SPAN_KB="64", KU_PGZIP_TEAM="3") == want
    assert run(["-T", str(z1)], KU_NO_PGZIP="1") == want
    pairs = run(["-P", "-T", str(z1), str(z2)], KU_NO_PGZIP="1")
    assert run(["-P", "-T", str(z1), str(z2)]) == pairs
    assert run(["-P", "-T", str(z1), str(z2)], KU_PGZIP_SPAN_KB="128") == pairs
    # damage is an error of the run (gzread's -1 ended the input silently before)
    bad = tmp_path / "bad.fq.gz"
    blob = member(text)
    bad.write_bytes(blob[:len(blob) // 2])
    r = subprocess.run([DUMP, "-T", str(bad)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 65 and b"gzip" in r.stderr
    # ... through zlib's reader too (a file that stops in mid-member makes gzread return 0 with Z_BUF_ERROR pending,
    # not -1 -- the sequential reader took that for the end of the input): the plain reader, the producer side, mate pairs
    for args in ([str(bad)], ["-T", str(bad)], ["-P", "-T", str(bad), str(z2)], ["-P", str(z1), str(bad)]):
        r = subprocess.run([DUMP] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KU_NO_PGZIP="1"))
        assert r.returncode == 65 and b"gzip" in r.stderr, args


def test_gz_text_is_parsed_in_regions_while_it_arrives(tmp_path):
    """-j N on a .gz file: the team inflates into one growing text, the parser team cuts record-aligned regions off its front
    as it arrives (decisions only on lines that are completely there) and releases the pages behind them; same records, in
    order, as the sequential reader -- FASTQ and multi-line FASTA (records longer than a region), plain gzip and BGZF, tiny
    regions / spans / look-ahead, a malformed record in the middle, a truncated file"""
    from test_seqio import bgzf
    text = fastq(20000, seed=21)
    fa = fasta(400, seed=22)
    bad_rec = text[:1500000] + b"@broken\nACGT\nIIII\n" + text[1500000:]  # ('+' line missing: the reference stops reading there)
    # long reads whose quality lines start with '@' or '+' (Q31, Q10): a record start is only decided by the lines behind it,
    # and those may not have arrived yet (lines far longer than a region and than the look-ahead)
    r = np.random.default_rng(23)
    long_fq = b""
    for i in range(24):
        L = int(r.integers(60_000, 400_000))
        q = (r.integers(0, 40, L) + 33).astype(np.uint8)
        q[0] = ord("@") if i % 2 else ord("+")
        long_fq += b"@long%d\n" % i + np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, L)].tobytes() + b"\n+\n" + q.tobytes() + b"\n"
    cases = {"long_fq": long_fq, "fq": text, "fa": fa, "broken": bad_rec[:bad_rec.index(b"\n", 1500000 - 400) + 1] + bad_rec[bad_rec.index(b"\n@", 1500000 - 400) + 1:]}

    def run(args, ok=True, **env):
        r = subprocess.run([DUMP] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert (r.returncode == 0) == ok, r.stderr.decode()[-300:]
        return r.stdout
    for name, t in cases.items():
        plain = tmp_path / f"{name}.txt"
        plain.write_bytes(t)
        want = run([str(plain)])
        assert want.count(b"\n") > (20 if name == "long_fq" else 300)
        for kind, blob in (("gz", member(t)), ("gz1", member(t, deflate_raw(t, 1))), ("bgzf", bgzf(t, 65280)), ("bgzf_small", bgzf(t, 3001))):
            z = tmp_path / f"{name}.{kind}.gz"
            z.write_bytes(blob)
            assert run(["-j", "4", str(z)]) == want, (name, kind)
            assert run(["-j", "3", str(z)], KU_REGION_KB="64", KU_PGZIP_SPAN_KB="64", KU_TEXT_AHEAD_MB="1") == want, (name, kind)
            assert run(["-j", "7", str(z)], KU_REGION_KB="16", KU_PGZIP_SPAN_KB="16", KU_PGZIP_TEAM="3", KU_BGZF_TEAM="3") == want, (name, kind)
    blob = member(text)
    cut = tmp_path / "cut.gz"
    cut.write_bytes(blob[:len(blob) // 2])
    r = subprocess.run([DUMP, "-j", "4", str(cut)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 65 and b"gzip" in r.stderr
    bz = bgzf(text, 65280)
    cut.write_bytes(bz[:len(bz) // 2 + 11] + bz[len(bz) // 2 + 12:])
    r = subprocess.run([DUMP, "-j", "4", str(cut)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0


def test_damaged_bgzf_block_is_refused_by_the_sequential_reader_too(tmp_path):
    """the BGZF path of the sequential reader (mate pairs -P, -T prefetch) did not look at a block's CRC-32: a
    flipped byte inside a STORED block (which inflates whatever it holds) went through as a wrong base.  Both readers refuse it
    now; a block whose extra field claims more than the block holds is no BGZF block."""
    import struct as st
    from test_seqio import bgzf
    text = FQ[:600_000]
    blob = bytearray(bgzf(text, 65280, level=0))  # level 0: stored blocks
    good = tmp_path / "good.gz"
    good.write_bytes(bytes(blob))
    want = subprocess.run([DUMP, "-T", str(good)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert want.returncode == 0
    assert subprocess.run([DUMP, "-j", "3", str(good)], stdout=subprocess.PIPE).stdout == want.stdout
    at = 18 + 5 + 40000  # inside the first block's payload: member header (12 + 6), stored-block header (5), then the text
    assert blob[at:at + 1] in (b"A", b"C", b"G", b"T", b"I", b"@", b"+", b"\n") or True
    blob[at] ^= 0x02
    bad = tmp_path / "bad.gz"
    bad.write_bytes(bytes(blob))
    for args in (["-T"], ["-j", "3"], []):
        r = subprocess.run([DUMP] + args + [str(bad)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode != 0 and r.stdout != want.stdout, args
    r = subprocess.run([DUMP, "-T", "-P", str(bad), str(good)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0
    # XLEN larger than the block: the file does not qualify as BGZF (zlib's reader takes it and stops at the damage)
    blob = bytearray(bgzf(text, 65280))
    st.pack_into("<H", blob, 10, 60000)
    bad.write_bytes(bytes(blob))
    for args in (["-T"], ["-j", "3"]):
        assert subprocess.run([DUMP] + args + [str(bad)], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode != 0, args


def corrupt(blob, rng):
    b = bytearray(blob)
    mode = int(rng.integers(0, 4))
    if mode == 0:
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1:
        b = b[:int(rng.integers(1, len(b)))]
    elif mode == 2:
        i = int(rng.integers(0, len(b)))
        j = min(len(b), i + int(rng.integers(1, 2000)))
        b[i:j] = rng.integers(0, 256, j - i, dtype=np.uint8).tobytes()
    else:
        i = int(rng.integers(0, len(b)))
        b[i:i] = rng.integers(0, 256, int(rng.integers(1, 50)), dtype=np.uint8).tobytes()
    return bytes(b)


def test_random_damage_never_yields_other_bytes(tmp_path):
    """flipped bits, truncation, overwritten and inserted bytes at random places: the run fails, or (damage in bytes that do not
    matter: the header's mtime, the padding) hands out exactly the original text -- never anything else.  (The same loop ran
    600 times under AddressSanitizer / UBSan and ThreadSanitizer builds of seqio_dump without a report.)"""
    rng = np.random.default_rng(77)
    src = FQ[:1_200_000]
    blob = member(src)
    p = tmp_path / "d.gz"
    refused = 0
    for it in range(24):
        p.write_bytes(corrupt(blob, rng))
        r = inflate(p, 3, [16, 64, None][it % 3])
        assert r.returncode != 0 or r.stdout == src, it
        refused += r.returncode != 0
    assert refused >= 18


def test_one_stream_writer_of_the_bench(tmp_path):
    """scripts/write_one_stream_gz.py (the bench's `e2e.gz` input: chunks deflated side by side, joined by sync flushes into ONE
    deflate stream, CRCs combined) writes what zlib and the team read back"""
    import gzip
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import write_one_stream_gz as w
    assert w._combine(zlib.crc32(FA[:1000]), zlib.crc32(FA[1000:]), len(FA) - 1000) == zlib.crc32(FA)
    old, w.CH = w.CH, 1 << 20
    try:
        src = tmp_path / "r.fq"
        src.write_bytes(FQ)
        out = w.write(str(src), procs=3)
    finally:
        w.CH = old
    blob = open(out, "rb").read()
    assert gzip.decompress(blob) == FQ and blob.count(b"\x1f\x8b\x08") >= 1
    r = inflate(out, 4, 64)
    assert r.returncode == 0 and r.stdout == FQ


def test_gz_output_written_in_parts_is_one_valid_stream(tmp_path):
    """ku_pgzout.h (how the executable writes `-o x.gz`: every formatting helper deflates its own lines, closed with a sync
    flush; the writer joins the parts and combines their CRCs): seqio_dump -G writes the parsed records that way -- zlib reads
    the file back, with parts of 200 kB, of 777 bytes, and for an input without records"""
    import gzip
    src = tmp_path / "r.fq"
    src.write_bytes(FQ[:3_000_000])
    want = subprocess.run([DUMP, str(src)], stdout=subprocess.PIPE, check=True).stdout
    for part in ("200000", "777"):
        out = tmp_path / f"o{part}.gz"
        subprocess.run([DUMP, "-G", str(out), str(src)], check=True, stderr=subprocess.PIPE, env=dict(os.environ, KU_GZ_PART=part))
        blob = out.read_bytes()
        assert zlib_accepts(blob) and gzip.decompress(blob) == want
        r = inflate(out, 3, 64)  # ... and so does the team
        assert r.returncode == 0 and r.stdout == want
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    out = tmp_path / "e.gz"
    subprocess.run([DUMP, "-G", str(out), str(empty)], check=True, stderr=subprocess.PIPE)
    assert gzip.decompress(out.read_bytes()) == b""
