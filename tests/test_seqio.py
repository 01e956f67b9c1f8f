"""Input stage of the classify executable (krakenuniq_amd/csrc/ku_seqio.h) through the parser-only aid
bin/seqio_dump: record semantics of src/seqreader.cpp:26-133 and of scripts/read_merger.pl (mate pairs)."""
import gzip
import os
import subprocess

import numpy as np

from krakenuniq_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "krakenuniq_amd", "bin", "seqio_dump")
G = os.path.join(ROOT, "tests", "golden")


def dump(args, **kw):
    r = subprocess.run([DUMP] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    assert r.returncode == 0, r.stderr.decode()
    recs = [ln.split(b"\t") for ln in r.stdout.split(b"\n")[:-1]]
    return [a.decode() for a, _ in recs], [b for _, b in recs], r.stderr.decode()


def test_fixture_files_parse_like_the_python_reader(tmp_path):
    assert os.path.exists(DUMP), "build with make -C krakenuniq_amd/csrc"
    for fn in ("f1/reads.fq", "f2/edge.fa", "f4/merged.fa", "f8/reads.fq"):
        ids, seqs = synth.read_seqfile(os.path.join(G, fn))
        got_ids, got_seqs, _ = dump([os.path.join(G, fn)])
        assert got_ids == ids and got_seqs == seqs, fn
    # gz input, two files in one run
    gz = tmp_path / "r.fq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(f"{G}/f1/reads.fq", "rb").read())
    a, b, _ = dump([str(gz), f"{G}/f2/edge.fa"])
    i1, s1 = synth.read_seqfile(f"{G}/f1/reads.fq")
    i2, s2 = synth.read_seqfile(f"{G}/f2/edge.fa")
    assert a == i1 + i2 and b == s1 + s2


def test_mate_pairs_merge_like_read_merger(tmp_path):
    """-P: id without /1, seq1 + N + seq2 == the reference's read_merger.pl output committed as f4/merged.fa"""
    ids, seqs = synth.read_seqfile(f"{G}/f4/merged.fa")
    got_ids, got_seqs, _ = dump(["-P", f"{G}/f4/r_1.fq", f"{G}/f4/r_2.fq"])
    assert got_ids == ids and got_seqs == seqs
    # unequal counts: the longer file's tail goes through unpaired (read_merger.pl:104-126)
    r1 = open(f"{G}/f4/r_1.fq").read().split("\n")
    short = tmp_path / "short_2.fq"
    short.write_text("\n".join(open(f"{G}/f4/r_2.fq").read().split("\n")[:4 * 150]) + "\n")
    a, b, _ = dump(["-P", f"{G}/f4/r_1.fq", str(short)])
    assert len(a) == 200 and b[:150] == seqs[:150]
    assert b[150:] == [x.encode() for x in r1[1::4][150:200]]


def test_record_edge_cases(tmp_path):
    p = tmp_path / "x.fa"
    # multi-line FASTA, blank line inside a record, description after the id, no trailing newline, CRLF kept as is
    p.write_bytes(b">a desc one\nACGT\nAC\n\nGT\n>b\n>c\tdesc\nTTTT")
    ids, seqs, _ = dump([str(p)])
    assert ids == ["a", "b", "c"] and seqs == [b"ACGTACGT", b"", b"TTTT"]
    q = tmp_path / "x.fq"
    q.write_bytes(b"@r1 d\nACGT\n+\nIIII\n@r2\nAC\n+r2\nII\n\n@never\nAA\n+\nII\n")
    ids, seqs, _ = dump([str(q)])
    assert ids == ["r1", "r2"] and seqs == [b"ACGT", b"AC"]  # an empty line ends a FASTQ stream (seqreader.cpp:103-106)
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r1\nACGT\n+\nIIII\n@r2\nACGT\nIIII\n")
    ids, seqs, err = dump([str(bad)])
    assert ids == ["r1"] and "malformed fastq file - quality header (IIII)" in err
    # a record longer than the reader's 16 MiB buffer
    big = tmp_path / "big.fa"
    rng = np.random.default_rng(1)
    s = synth.codes_to_ascii(rng.integers(0, 4, 40_000_000, dtype=np.uint8))
    with open(big, "wb") as f:
        f.write(b">big\n")
        for i in range(0, len(s), 80):
            f.write(s[i:i + 80] + b"\n")
        f.write(b">tail\nACGT\n")
    r = subprocess.run([DUMP, str(big)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = r.stdout.split(b"\n")
    assert lines[0] == b"big\t" + s and lines[1] == b"tail\tACGT"


def test_region_parallel_parse_equals_sequential(tmp_path):
    """-j N: N record-aligned regions parsed independently (find_record_start + parse_region) == one sequential pass,
    with quality lines that start with '@' or '+', and the stream still ends at the first malformed record"""
    rng = np.random.default_rng(4)
    recs = []
    for i in range(3000):
        L = int(rng.integers(1, 200))
        seq = synth.codes_to_ascii(rng.integers(0, 4, L, dtype=np.uint8))
        q = bytes(rng.choice(np.frombuffer(b"@+I5#", dtype=np.uint8), L).tolist())
        recs.append(b"@r%d x\n" % i + seq + b"\n+\n" + q + b"\n")
    fq = tmp_path / "tricky.fq"
    fq.write_bytes(b"".join(recs))
    want = dump([str(fq)])[:2]
    for j in (1, 2, 5, 16, 64):
        assert dump(["-j", str(j), str(fq)])[:2] == want, j
    for fn in ("f1/reads.fq", "f2/edge.fa", "f4/merged.fa"):
        assert dump(["-j", "6", os.path.join(G, fn)])[:2] == dump([os.path.join(G, fn)])[:2], fn
    # smaller first regions (what the classify executable does since round 5: KU_REGION_RAMP regions of a quarter, as many of half
    # the size, then full ones): the same records in the same order -- plain, and from a one-stream .gz file whose text grows
    # while it is cut
    import gzip
    gzf = tmp_path / "tricky.fq.gz"
    gzf.write_bytes(gzip.compress(b"".join(recs), 6))
    for ramp in (1, 3, 12, 40):
        env = dict(os.environ, KU_REGION_RAMP=str(ramp), KU_REGION_KB="16")
        for j in (5, 64):
            assert dump(["-j", str(j), str(fq)], env=env)[:2] == want, (ramp, j)
        assert dump(["-j", "4", str(gzf)], env=env)[:2] == want, ramp
    # an empty line after record 1500 ends the stream in both modes
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"".join(recs[:1500]) + b"\n" + b"".join(recs[1500:]))
    assert dump(["-j", "8", str(bad)])[:2] == dump([str(bad)])[:2]
    assert len(dump([str(bad)])[0]) == 1500


def test_pipes_are_read_through_one_handle(tmp_path):
    """inputs such as <(cat library/*.fna) (scripts/build_db.sh:272) are pipes: nothing may be lost between the format
    probe and the parser, with and without the producer thread"""
    ids, seqs = synth.read_seqfile(f"{G}/f1/reads.fq")
    for flags in ([], ["-T"]):
        r = subprocess.run(["bash", "-c", f"{DUMP} {' '.join(flags)} <(cat {G}/f1/reads.fq) <(gzip -c {G}/f4/merged.fa)"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        recs = [ln.split(b"\t") for ln in r.stdout.split(b"\n")[:-1]]
        i4, s4 = synth.read_seqfile(f"{G}/f4/merged.fa")
        assert [a.decode() for a, _ in recs] == ids + i4 and [b for _, b in recs] == seqs + s4


def bgzf(data, block=65280, level=6):
    """BGZF: every block a gzip member with its size in a "BC" extra field (SAM specification 4.1), an empty block at the end"""
    import struct
    import zlib

    def one(b):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(b) + c.flush()
        hdr = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, 18 + len(comp) + 8 - 1)
        return hdr + comp + struct.pack("<II", zlib.crc32(b) & 0xffffffff, len(b))
    return b"".join(one(data[i:i + block]) for i in range(0, len(data), block)) + one(b"")


def test_bgzf_input_is_inflated_by_a_team_and_parses_like_the_text(tmp_path):
    """blocked gzip: the blocks are independent deflate streams (a team of threads inflates them, -T: as the executable reads);
    same records as the plain text, in file order, whatever the block size (records straddle blocks and tasks); a file
    that only starts like BGZF goes through zlib's reader"""
    rng = np.random.default_rng(3)
    n = 30000
    text = b"".join(b"@r%d x\n" % i + bytes(rng.choice(list(b"ACGTN"), size=int(rng.integers(30, 260))).astype(np.uint8)) + b"\n+\n" +
                    b"I" * 5 + b"\n" for i in range(n))
    # (quality lines shorter than the sequence: the parser does not care, the reference's does not either)
    plain = tmp_path / "r.fq"
    plain.write_bytes(text)
    want = dump([str(plain)])[:2]
    for block in (65280, 4093, 700):
        z = tmp_path / f"r{block}.fq.gz"
        z.write_bytes(bgzf(text, block))
        assert dump(["-T", str(z)])[:2] == want, block
        assert dump([str(z)])[:2] == want, block  # without the producer side: zlib reads the members one after the other
    mixed = tmp_path / "mixed.fq.gz"
    mixed.write_bytes(bgzf(text[:len(text) // 2])[:-28] + gzip.compress(text[len(text) // 2:]))
    assert dump(["-T", str(mixed)])[:2] == want


def _golden_records(path):
    """(id, sequence) of the reference's Kraken lines printed with -s (columns 2 and 6)"""
    out = []
    for ln in open(path, "rb").read().split(b"\n")[:-1]:
        c = ln.split(b"\t", 5)
        out.append((c[1].decode(), c[5]))
    return out


def test_end_of_input_rules_equal_the_reference(tmp_path):
    """tests/golden/f12 (outputs of the compiled reference, make_golden.py f12): a FASTA header as the last line without
    a line end is no record unless it is the file's first (seqreader.cpp:37-40,62-71); a work unit without nucleotides ends
    the file and is never printed -- at WORK-UNIT granularity (classify.cpp:510-523); a damaged FASTQ file ends where the
    sequential reader ends, whatever the number of region parsers"""
    import json
    d = f"{G}/f12"
    for case in json.load(open(f"{d}/cases.json")):
        want = _golden_records(f"{d}/{case['output']}")
        path = f"{d}/{case['input']}"
        gz = tmp_path / (case["input"] + ".gz")
        with gzip.open(gz, "wb") as f:
            f.write(open(path, "rb").read())
        for extra, env in (([], None), (["-j", "2"], None), (["-j", "3"], None), (["-j", "5"], None), (["-j", "6"], None), (["-j", "7"], None), (["-j", "11"], None),
                           (["-j", "3"], {"KU_REGION_RAMP": "2"}),
                           (["-T"], None), ("gz", None), ("gz-j", {"KU_REGION_KB": "1"})):
            if extra == "gz":
                args = case["flags"] + [str(gz)]
            elif extra == "gz-j":
                if os.path.getsize(gz) < 20:
                    continue
                args = case["flags"] + ["-j", "3", str(gz)]
            else:
                args = case["flags"] + extra + [path]
            ids, seqs, _ = dump(args, env=dict(os.environ, **env) if env else None)
            assert list(zip(ids, seqs)) == want, (case, extra)
    # the adversarial file really has its regions cut inside records (and parsed again from where the region before stopped)
    r = subprocess.run([DUMP, "-j", "6", "-n", f"{d}/plus_seqs.fq"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KU_SEQIO_DEBUG="1"))
    assert r.stderr.count(b"was cut inside a record") >= 2


def test_fuzz_against_reference():
    """differential fuzzing against the compiled reference's reader (tests/fuzz_seqio.py; build container only)"""
    import pytest
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fuzz_seqio
    if not os.path.exists(fuzz_seqio.REF):
        pytest.skip("oracle/_ref/classify is built in the build container only")
    failures = fuzz_seqio.fuzz(120, seed=20261001)
    assert not failures, failures[:3]


def test_long_stretches_of_empty_records_between_region_parsers(tmp_path):
    """whole regions of empty records: they share a work unit with the nucleotides that follow (printed), or form the file's last,
    nucleotide-free unit (never printed) -- whatever the number of region parsers; the held batches are folded into one
    (UnitGate), so a pool of batches cannot run dry"""
    d = f"{G}/f12"
    head = open(f"{d}/empty_inside.fa", "rb").read().split(b">e1\n")[0]          # 20 reads of 150 nt
    tail = b">t1\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>t2\nTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT\n"
    empties = b"".join(b">x%d\n" % j for j in range(5000))
    for name, data, n_want in (("inside.fa", head + empties + tail, 20 + 5000 + 2), ("behind.fa", head + empties, 20),
                               ("front.fa", empties + tail, 5002), ("all.fa", empties, 0)):
        p = tmp_path / name
        p.write_bytes(data)
        want = dump(["-u", "1500", str(p)])[:2]
        assert len(want[0]) == n_want, name
        for j in ("2", "5", "13"):
            got = dump(["-u", "1500", "-j", j, str(p)])[:2]
            assert got == want, (name, j)
