#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Inputs are produced by this repo's own seeded generator (krakenuniq_amd/synth.py);
expected outputs come from the reference binaries in oracle/_ref/ (classify,
classifyExact, db_sort, count_unique, and oracle/ref_kat.cpp linked against the
reference's objects) plus the reference's scripts/read_merger.pl.  Only data
(inputs + expected outputs) is committed -- no reference source.

Fixtures:
  f1/   tiny DB (k=31, nt=7) + taxDB + 1000 x 150 bp reads.fq + reference outputs
        out.tsv/report.tsv (-t 1), report_exact.tsv (classifyExact),
        out_u1000.tsv/report_u1000.tsv (-u 1000: sketches stay sparse),
        out_chunk.tsv/report_chunk.tsv (-x 70K -t 2), out_quick.tsv (-q -m 2),
        out_chunk_quick.tsv/report_chunk_quick.tsv (-x 70K -t 2 -q -m 2),
        out_c.tsv (-c), report_p0.tsv (-p 0: six columns), database.kdb.counts
  f2/   edge FASTA (short / N / empty / lower-case / multi-line) + outputs
  f4/   paired FASTQs -> read_merger.pl -> merged.fa + outputs
  f7/   legacy KRAKIDX (type 1) index variant of f1 + outputs
  f9/   set_lcas: library FASTA + seqid map -> the reference's database.kdb / counts
  f11/  UID mapping: the reference's set_lcas -I database + map, its classify -I output and report; kat_uid.json:
        known answers of resolve_uids3 and of std::unordered_map's iteration order
  f8/   second database + reads for hierarchical multi-database runs (both orders, quick mode)
  f10/  CRLF inputs (FASTQ, one-line-per-sequence FASTA, multi-line FASTA) of f1 reads + outputs
  f12/  how a file's input ENDS (src/seqreader.cpp:37-40,62-71, src/classify.cpp:510-523): a FASTA header as the last line
        without a line end, trailing empty records that form a work unit of their own / that share one with nucleotides, a FASTQ
        file with a deleted sequence line, quality lines that start with '@'; cases.json lists (input, flags, output)
  kat.json  per-function known-answer vectors from ref_kat
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from krakenuniq_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
MERGER = "/root/reference/scripts/read_merger.pl"
K, NT = 31, 7


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    if r.returncode != 0:
        sys.stderr.write(r.stderr.decode(errors="replace"))
        raise SystemExit(f"command failed: {cmd}")
    return r


def classify(db, args, reads, binary="classify"):
    cmd = [os.path.join(REF, binary), "-d", f"{db}/database.kdb", "-i", f"{db}/database.idx",
           "-a", f"{db}/taxDB"] + args + reads
    return run(cmd)


def make_f1_exact_variants(d=None):
    """classifyExact with the other run modes (it is the same program with every flag, classify.cpp:46-53): quick mode
    (only the scanned prefix of a read is counted) and a chunked run; `make_golden.py f1x` adds them to an existing f1"""
    d = d or os.path.join(HERE, "f1")
    rd = [f"{d}/reads.fq"]
    tmp = f"{d}/_x.tsv"
    classify(d, ["-q", "-m", "2", "-o", tmp, "-r", f"{d}/report_exact_quick.tsv"], rd, "classifyExact")
    assert open(tmp, "rb").read() == open(f"{d}/out_quick.tsv", "rb").read()
    classify(d, ["-x", "70K", "-t", "2", "-o", tmp, "-r", f"{d}/report_exact_chunk.tsv"], rd, "classifyExact")
    assert open(tmp, "rb").read() == open(f"{d}/out_chunk.tsv", "rb").read()
    os.remove(tmp)


def make_f1():
    d = os.path.join(HERE, "f1")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    rng = np.random.default_rng(7)
    tax = synth.small_taxonomy()
    g4 = synth.procedural_genome(7, 4, 3000)
    g5 = synth.mutate(g4, 0.03, rng)            # sibling species: shared k-mers -> LCA = genus 2
    g6 = synth.procedural_genome(7, 6, 3000)
    gp = np.concatenate([g6[2000:2300], synth.procedural_genome(7, 99, 300)])  # plasmid shares 300 bp with S6
    genomes = {4: g4, 5: g5, 6: g6, 1000000001: gp}
    kmers, vals = synth.lca_database(genomes, tax, K)
    # a few DB k-mers whose taxid is absent from taxDB and a zero-valued entry (found, but taxon 0)
    extra = synth.canonical(synth.kmers_forward(synth.procedural_genome(7, 1234, 200), K), K)
    extra = np.setdiff1d(np.unique(extra), kmers)
    kmers = np.concatenate([kmers, extra])
    vals = np.concatenate([vals, np.where(np.arange(len(extra)) % 2 == 0, 777, 0).astype(np.uint32)])
    genomes_for_reads = dict(genomes)
    genomes_for_reads[777] = synth.procedural_genome(7, 1234, 200)
    perm = rng.permutation(len(kmers))
    synth.write_jdb(os.path.join(d, "database.jdb"), kmers[perm], vals[perm], K)
    run([os.path.join(REF, "db_sort"), "-n", str(NT), "-d", f"{d}/database.jdb", "-o", f"{d}/database.kdb",
         "-i", f"{d}/database.idx"])
    os.remove(os.path.join(d, "database.jdb"))
    tax.write(os.path.join(d, "taxDB"))
    # our own writer must reproduce the reference's db_sort byte for byte
    sk, sv, off = synth.sort_db(kmers, vals, K, NT)
    synth.write_db(os.path.join(d, "_mine"), sk, sv, off, K, NT)
    for fn in ("database.kdb", "database.idx"):
        a = open(os.path.join(d, fn), "rb").read()
        b = open(os.path.join(d, "_mine", fn), "rb").read()
        assert a == b, f"synth.write_db differs from reference db_sort for {fn}"
    shutil.rmtree(os.path.join(d, "_mine"))

    reads, src = synth.sample_reads(genomes_for_reads, 1000, 150, rng, frac_random=0.25)
    ids = [f"r{i}_t{t}" for i, t in enumerate(src)]
    synth.write_fastq(os.path.join(d, "reads.fq"), reads, ids)

    rd = [f"{d}/reads.fq"]
    classify(d, ["-o", f"{d}/out.tsv", "-r", f"{d}/report.tsv"], rd)          # writes .counts too
    classify(d, ["-o", f"{d}/out_exact.tsv", "-r", f"{d}/report_exact.tsv"], rd, "classifyExact")
    make_f1_exact_variants(d)
    classify(d, ["-u", "1000", "-o", f"{d}/out_u1000.tsv", "-r", f"{d}/report_u1000.tsv"], rd)
    classify(d, ["-x", "70K", "-t", "2", "-o", f"{d}/out_chunk.tsv", "-r", f"{d}/report_chunk.tsv"], rd)
    classify(d, ["-q", "-m", "2", "-o", f"{d}/out_quick.tsv"], rd)
    # quick mode inside a chunked run: every k-mer is booked and the call is the LAST k-mer's taxon (classify.cpp:700-737)
    classify(d, ["-x", "70K", "-t", "2", "-q", "-m", "2", "-o", f"{d}/out_chunk_quick.tsv", "-r", f"{d}/report_chunk_quick.tsv"], rd)
    classify(d, ["-c", "-o", f"{d}/out_c.tsv"], rd)
    classify(d, ["-s", "-o", f"{d}/out_s.tsv"], rd)
    # -p 0: the six-column report without the k-mer columns (classify.cpp:289,316-323); the Kraken file does not change
    classify(d, ["-p", "0", "-o", f"{d}/out_p0.tsv", "-r", f"{d}/report_p0.tsv"], rd)
    assert open(f"{d}/out.tsv", "rb").read() == open(f"{d}/out_p0.tsv", "rb").read()
    os.remove(f"{d}/out_p0.tsv")
    assert open(f"{d}/out.tsv", "rb").read() == open(f"{d}/out_exact.tsv", "rb").read()
    os.remove(f"{d}/out_exact.tsv")
    return d, genomes_for_reads


def make_f2(f1):
    d = os.path.join(HERE, "f2")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    ids, seqs = synth.read_seqfile(os.path.join(f1, "reads.fq"))
    hit = next(s for i, s in zip(ids, seqs) if i.endswith("_t4") and b"N" not in s)
    recs = [
        ("short", hit[:8]),
        ("exactk", hit[:31]),
        ("kminus1", hit[:30]),
        ("withN", hit[:40] + b"N" + hit[41:79]),
        ("empty", b""),
        ("lower", hit.lower()),
        ("mixed", hit[:75] + hit[75:].lower()),
        ("iupac", hit[:60] + b"R" + hit[61:100] + b"y" + hit[101:]),
        ("allN", b"N" * 64),
        ("long", hit + hit[::-1] + hit),
    ]
    with open(os.path.join(d, "edge.fa"), "wb") as f:
        for rid, s in recs:
            f.write(b">" + rid.encode() + b" some description\n")
            if rid == "long":  # multi-line record
                for i in range(0, len(s), 60):
                    f.write(s[i:i + 60] + b"\n")
            else:
                f.write(s + b"\n")
    classify(f1, ["-o", f"{d}/out.tsv", "-r", f"{d}/report.tsv"], [f"{d}/edge.fa"])
    return d


def make_f4(f1, genomes):
    d = os.path.join(HERE, "f4")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    rng = np.random.default_rng(11)
    r1, s1 = synth.sample_reads(genomes, 200, 150, rng, frac_random=0.25, n_rate=0.003)
    r2 = []
    for a, t in zip(r1, s1):
        if t == 0:
            r2.append(synth.codes_to_ascii(rng.integers(0, 4, 150, dtype=np.uint8)))
        else:
            g = genomes[t]
            s = int(rng.integers(0, max(1, len(g) - 150 + 1)))
            r2.append(synth.codes_to_ascii(synth.revcomp_codes(g[s:s + 150])))
    synth.write_fastq(os.path.join(d, "r_1.fq"), r1, [f"p{i}_t{t}/1" for i, t in enumerate(s1)])
    synth.write_fastq(os.path.join(d, "r_2.fq"), r2, [f"p{i}_t{t}/2" for i, t in enumerate(s1)])
    merged = run(["perl", MERGER, "--check-names", f"{d}/r_1.fq", f"{d}/r_2.fq"]).stdout
    open(os.path.join(d, "merged.fa"), "wb").write(merged)
    classify(f1, ["-o", f"{d}/out.tsv", "-r", f"{d}/report.tsv"], [f"{d}/merged.fa"])
    return d


def make_f7(f1):
    """Same pairs under a legacy type-1 (unscrambled KRAKIDX) index."""
    d = os.path.join(HERE, "f7")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    kmers, vals, _, k, nt, _ = synth.read_db(f1)
    bk = synth.bin_key(kmers, k, nt, idx_type=1)
    order = np.lexsort((kmers, bk))
    counts = np.bincount(bk.astype(np.int64), minlength=4 ** nt)
    off = np.zeros(4 ** nt + 1, dtype=np.uint64)
    np.cumsum(counts, out=off[1:])
    synth.write_db(d, kmers[order], vals[order], off, k, nt)
    raw = bytearray(open(f"{d}/database.idx", "rb").read())
    raw[:7] = b"KRAKIDX"
    open(f"{d}/database.idx", "wb").write(bytes(raw))
    shutil.copy(f"{f1}/taxDB", f"{d}/taxDB")
    classify(d, ["-o", f"{d}/out.tsv", "-r", f"{d}/report.tsv"], [f"{f1}/reads.fq"])
    assert open(f"{d}/out.tsv", "rb").read() == open(f"{f1}/out.tsv", "rb").read()
    os.remove(f"{d}/database.kdb.counts")
    return d


def make_f8(f1, genomes):
    """Second database for hierarchical runs (classify -d A -i A.idx -d B -i B.idx, classify.cpp:928-936): other
    minimizer length, a new genome under genus 3, and a third of f1's species-4 k-mers re-labelled species 6 so
    that the database order decides their taxon."""
    d = os.path.join(HERE, "f8")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    rng = np.random.default_rng(23)
    k1, v1, _, k, _, _ = synth.read_db(f1)
    g_new = synth.procedural_genome(7, 55, 2500)
    new = np.setdiff1d(np.unique(synth.canonical(synth.kmers_forward(g_new, K), K)), k1)
    shared = k1[v1 == 4][::3]
    kmers = np.concatenate([new, shared])
    vals = np.concatenate([np.full(len(new), 3, np.uint32), np.full(len(shared), 6, np.uint32)])
    perm = rng.permutation(len(kmers))
    synth.write_jdb(os.path.join(d, "database.jdb"), kmers[perm], vals[perm], K)
    run([os.path.join(REF, "db_sort"), "-n", "6", "-d", f"{d}/database.jdb", "-o", f"{d}/database.kdb",
         "-i", f"{d}/database.idx"])
    os.remove(os.path.join(d, "database.jdb"))
    src_genomes = {4: genomes[4], 6: genomes[6], 3: g_new}
    reads, src = synth.sample_reads(src_genomes, 300, 150, rng, frac_random=0.2)
    synth.write_fastq(os.path.join(d, "reads.fq"), reads, [f"m{i}_t{t}" for i, t in enumerate(src)])
    a = ["-d", f"{f1}/database.kdb", "-i", f"{f1}/database.idx"]
    b = ["-d", f"{d}/database.kdb", "-i", f"{d}/database.idx"]
    cl = [os.path.join(REF, "classify"), "-a", f"{f1}/taxDB"]
    rd = [f"{d}/reads.fq"]
    run(cl + a + b + ["-o", f"{d}/out.tsv", "-r", f"{d}/report.tsv"] + rd)          # writes f8's .counts too
    run(cl + b + a + ["-o", f"{d}/out_swapped.tsv", "-r", f"{d}/report_swapped.tsv"] + rd)
    run(cl + a + b + ["-q", "-m", "2", "-o", f"{d}/out_quick.tsv"] + rd)
    assert open(f"{d}/out.tsv", "rb").read() != open(f"{d}/out_swapped.tsv", "rb").read()
    return d


def make_f9(f1, genomes):
    """set_lcas (the database build step behind db_sort): f1's k-mers with zeroed values + a small library -> the
    reference's set_lcas output.  The library exercises the ID forms of src/set_lcas.cpp:263-300: plain ID, ID with a
    .N version suffix, kraken:taxid| header, lower case / N bases, an unmapped ID (skipped), a taxid missing from
    taxDB (skipped), an empty record, and a sequence whose k-mers the database does not hold (-x)."""
    d = os.path.join(HERE, "f9")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    kmers, vals, off, k, nt, _ = synth.read_db(f1)
    synth.write_db(d, kmers, np.zeros_like(vals), off, k, nt)
    os.rename(f"{d}/database.kdb", f"{d}/database0.kdb")
    os.remove(f"{d}/database.idx")
    a = synth.codes_to_ascii
    g4, g5, g6, gp = a(genomes[4]), a(genomes[5]), a(genomes[6]), a(genomes[1000000001])
    recs = [
        (b"seqA first genome", g4[:1500].lower() + g4[1500:]),
        (b"seqB.1 versioned id", g5),
        (b"kraken:taxid|6|seqC taxid in the header", g6[:1000] + b"N" + g6[1001:]),
        (b"seqP plasmid", gp),
        (b"unmapped no taxid for this one", g4[:200]),
        (b"seqX taxid 999 is not in taxDB", g5[:200]),
        (b"seqE empty record", b""),
        (b"seqN novel sequence, not in the database", a(synth.procedural_genome(7, 4242, 300))),
    ]
    with open(f"{d}/library.fa", "wb") as f:
        for h, s in recs:
            f.write(b">" + h + b"\n")
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + b"\n")
    with open(f"{d}/seqid2taxid.map", "w") as f:
        f.write("seqA\t4\nseqB\t5\nseqP\t1000000001\nseqX\t999\nseqE\t4\nseqN\t5\nseqA\t6\n")
    run([os.path.join(REF, "set_lcas"), "-M", "-x", "-d", f"{d}/database0.kdb", "-o", f"{d}/database.kdb",
         "-i", f"{f1}/database.idx", "-b", f"{f1}/taxDB", "-m", f"{d}/seqid2taxid.map", "-F", f"{d}/library.fa",
         "-c", f"{d}/database.kdb.counts"])
    # -a -A: new taxids for assemblies (third map column) and sequences; rewrites taxDB, prints the new map
    with open(f"{d}/seqid2taxid_aA.map", "w") as f:
        f.write("seqA\t4\tassembly one\nseqB\t5\tassembly one\nseqP\t1000000001\nseqX\t999\tasm x\nseqE\t4\n"
                "seqN\t5\tassembly two\nseqA\t6\tdup\n")
    for tag, flags in (("a", ["-a"]), ("A", ["-A"]), ("aA", ["-a", "-A"])):
        shutil.copy(f"{f1}/taxDB", f"{d}/taxDB_{tag}")
        r = run([os.path.join(REF, "set_lcas"), "-M", "-x", "-d", f"{d}/database0.kdb", "-o", f"{d}/tmp.kdb",
                 "-i", f"{f1}/database.idx", "-b", f"{d}/taxDB_{tag}", "-m", f"{d}/seqid2taxid_aA.map",
                 "-F", f"{d}/library.fa", "-c", f"{d}/counts_{tag}"] + flags)
        open(f"{d}/map_{tag}.out", "wb").write(r.stdout)
        raw = np.fromfile(f"{d}/tmp.kdb", dtype=np.uint8)
        raw[-12 * len(kmers):].view(synth.PAIR_DT)["val"].astype("<u4").tofile(f"{d}/values_{tag}.u32")
        os.remove(f"{d}/tmp.kdb")
    os.remove(f"{d}/database0.kdb")  # == f1's database.kdb with zeroed values (rebuilt by the test)
    # the genomes are f1's: the LCAs must be f1's values wherever f1's value came from the genomes
    k9, v9, *_ = synth.read_db(d, idx=os.path.join(f1, "database.idx"))
    assert np.array_equal(k9, kmers)
    return d


class Kat:
    def __init__(self):
        self.p = subprocess.Popen([os.path.join(REF, "ref_kat")], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True, bufsize=1)

    def cmd(self, line, nlines=1):
        self.p.stdin.write(line + "\n")
        self.p.stdin.flush()
        return [self.p.stdout.readline().rstrip("\n") for _ in range(nlines)]

    def scan(self, seq):
        self.p.stdin.write(f"SCAN {seq}\n")
        self.p.stdin.flush()
        n = int(self.p.stdout.readline())
        return [self.p.stdout.readline().split() for _ in range(n)]


def make_f10(f1):
    """CRLF line ends: reads 0..59 of f1 as FASTQ, reads 60..119 as FASTA with one and with several lines per sequence"""
    d = os.path.join(HERE, "f10")
    os.makedirs(d, exist_ok=True)
    ids, seqs = synth.read_seqfile(f"{f1}/reads.fq")
    with open(f"{d}/crlf.fq", "wb") as f:
        for i, s in zip(ids[:60], seqs[:60]):
            f.write(b"@" + i.encode() + b"\r\n" + s + b"\r\n+\r\n" + b"I" * len(s) + b"\r\n")
    with open(f"{d}/crlf_oneline.fa", "wb") as f:
        for i, s in zip(ids[60:120], seqs[60:120]):
            f.write(b">" + i.encode() + b" some description\r\n" + s + b"\r\n")
    with open(f"{d}/crlf_multiline.fa", "wb") as f:
        for i, s in zip(ids[60:120], seqs[60:120]):
            f.write(b">" + i.encode() + b" some description\r\n")
            for o in range(0, len(s), 60):
                f.write(s[o:o + 60] + b"\r\n")
    for name in ("crlf.fq", "crlf_oneline.fa", "crlf_multiline.fa"):
        classify(f1, ["-o", f"{d}/{name.rsplit('.', 1)[0]}.out.tsv"], [f"{d}/{name}"])


def make_f12(f1):
    """end-of-input rules of the reference's readers and of process_file's work units (outputs with -s: id AND sequence)"""
    d = os.path.join(HERE, "f12")
    os.makedirs(d, exist_ok=True)
    ids, seqs = synth.read_seqfile(f"{f1}/reads.fq")
    fa = lambda lo, hi: b"".join(b">" + i.encode() + b"\n" + s + b"\n" for i, s in zip(ids[lo:hi], seqs[lo:hi]))
    fq = lambda lo, hi, q=b"I": b"".join(b"@" + i.encode() + b"\n" + s + b"\n+\n" + q * len(s) + b"\n" for i, s in zip(ids[lo:hi], seqs[lo:hi]))
    empties_fa = b">e1\n>e2 with a description\n\n>e3\n"
    files = {
        # the last line is a header without a line end: dropped (the first record of a file would go through: lone_header.fa)
        "hdr_at_eof.fa": fa(0, 30) + b">tail",
        "lone_header.fa": b">only",
        # 20 reads x 150 nt with -u 1500: the units close behind reads 10 and 20 -- the empty records behind them are a unit
        # without nucleotides: never printed; with -u 1400 the same (a unit closes at 1500 >= 1400)
        "empty_unit.fa": fa(0, 20) + empties_fa,
        # 21 reads: the third unit holds read 21 AND the empty records: all printed
        "empty_shared_unit.fa": fa(0, 21) + empties_fa,
        # empty records in the middle of the file share a unit with what follows
        "empty_inside.fa": fa(0, 20) + empties_fa + fa(20, 25),
        "only_empties.fa": empties_fa,
        "empty_unit.fq": fq(0, 20) + b"@e1\n\n+\n\n@e2\n\n+\n\n",
        # record 120 of 200 lost its sequence line: the stream ends there (malformed quality header)
        "deleted_seq_line.fq": fq(0, 120) + b"@" + ids[120].encode() + b"\n+\n" + b"I" * 150 + b"\n" + fq(121, 200),
        # quality lines that start with '@' (and one '+' line carrying the id): every record start must be found all the same
        "at_quals.fq": fq(0, 100, b"@") + fq(100, 200),
        # a deleted quality line AND '@' qualities: what follows the damage looks like records at the wrong line offsets
        "deleted_qual_line.fq": fq(0, 60, b"@") + b"@" + ids[60].encode() + b"\n" + seqs[60] + b"\n+\n" + fq(61, 200, b"@"),
    }
    # sequences that start with '+' (any byte is a base to the reference: an ambiguous one) behind quality lines that start with
    # '@': "a line starting with '@' whose second successor starts with '+'" then holds for every QUALITY line as well -- three of
    # four cuts of a region parser land inside a record
    files["plus_seqs.fq"] = b"".join(b"@" + i.encode() + b"\n+" + s[1:] + b"\n+\n" + b"@" * len(s) + b"\n" for i, s in zip(ids[:200], seqs[:200]))
    cases = []
    for name, data in files.items():
        open(f"{d}/{name}", "wb").write(data)
        for flags in (["-u", "1500"], ["-u", "1400"]) if name.startswith("empty") or name.startswith("only") else ([],):
            out = f"{name}{'.u' + flags[1] if flags else ''}.out.tsv"
            classify(f1, ["-s", "-t", "1", "-o", f"{d}/{out}"] + flags, [f"{d}/{name}"])
            cases.append({"input": name, "flags": flags, "output": out})
    json.dump(cases, open(f"{d}/cases.json", "w"), indent=1)


def make_kat(f1):
    rng = np.random.default_rng(3)
    kat = {"k": K}
    h = Kat()
    h.cmd(f"K {K}")
    h.cmd("IDX 2 7")
    # scanner / canonical
    seqs = ["ACGTTGCAAGGCTTAACCGGTTAGCATCGATCGGATATCGCGNACGTACGTTAGC"]
    for L in (31, 32, 60, 150):
        s = bytearray(synth.codes_to_ascii(rng.integers(0, 4, L, dtype=np.uint8)))
        if L > 40:
            s[int(rng.integers(0, L))] = ord("N")
            s[int(rng.integers(0, L))] = ord("c")
            s[int(rng.integers(0, L))] = ord("t")
        seqs.append(s.decode())
    kat["scan"] = [{"seq": s, "kmers": h.scan(s)} for s in seqs]
    # canonical / revcomp for n in {7, 13, 15, 31}
    vals = [int(x) for x in rng.integers(0, 2 ** 62, 64, dtype=np.uint64)]
    kat["canon"] = []
    for n in (7, 12, 13, 15, 31):
        for v in vals[:16]:
            v &= (1 << (2 * n)) - 1
            kat["canon"].append({"x": f"{v:016x}", "n": n, "rc": h.cmd(f"RC {v:016x} {n}")[0],
                                 "canon": h.cmd(f"CANON {v:016x} {n}")[0]})
    # bin keys of canonical k-mers
    kat["binkey"] = []
    for v in vals:
        c = int(synth.canonical(np.array([v], dtype=np.uint64), K)[0])
        row = {"kmer": f"{c:016x}"}
        for nt in (7, 10, 12, 13, 15):
            row[f"nt{nt}"] = int(h.cmd(f"BINKEY {c:016x} {nt}")[0])
        kat["binkey"].append(row)
    h.cmd("IDX 1 7")
    kat["binkey_type1_nt7"] = [{"kmer": r["kmer"], "bin": int(h.cmd(f"BINKEY1 {r['kmer']}")[0])}
                               for r in kat["binkey"][:16]]
    h.cmd("IDX 2 7")
    # hash
    hv = [0, 1, 42, 0x0123456789ABCDEF, 0xFFFFFFFFFFFFFFFF] + vals[:27]
    kat["hash"] = [{"x": f"{v:016x}", "h": h.cmd(f"HASH {v:016x}")[0]} for v in hv]
    # kmer_query on the f1 database: all DB k-mers are hits; random ones are misses
    h.cmd(f"OPEN {f1}/database.kdb {f1}/database.idx")
    kmers, dbvals, _, _, _, _ = synth.read_db(f1)
    pick = rng.choice(len(kmers), 200, replace=False)
    q = [int(kmers[i]) for i in pick] + [int(synth.canonical(np.array([v], dtype=np.uint64), K)[0]) for v in vals]
    kat["query"] = [{"kmer": f"{v:016x}", "val": int(h.cmd(f"QUERY {v:016x}")[0])} for v in q]
    # lca / resolve_tree on a random tree (ids sparse, incl. >= 1e9, an orphan, a self-parent)
    ids = sorted(set(int(x) for x in rng.integers(2, 5000, 60)))
    parent = {1: 1}
    for i, t in enumerate(ids):
        parent[t] = 1 if i < 4 else ids[int(rng.integers(0, i))]
    parent[1000000005] = ids[10]
    parent[1000000006] = 1000000005
    parent[7001] = 9999      # orphan: parent id has no entry
    parent[7002] = 7002      # self-parent (not root)
    pm = {t: (0 if (p == t or p not in parent) else p) for t, p in parent.items()}  # Parent_map semantics
    h.cmd("PARENT " + " ".join(f"{a}:{b}" for a, b in pm.items()))
    allids = list(pm.keys())
    kat["tree"] = {"parent_map": {str(a): b for a, b in pm.items()}, "lca": [], "resolve": []}
    for _ in range(200):
        a, b = (allids[int(rng.integers(0, len(allids)))] for _ in range(2))
        if rng.random() < 0.1:
            a = 0
        kat["tree"]["lca"].append([a, b, int(h.cmd(f"LCA {a} {b}")[0])])
    fixed = [{4: 3, 5: 3}, {}, {ids[5]: 2, ids[6]: 2, ids[0]: 1}, {7001: 3}, {7002: 1, ids[3]: 1}, {8888: 2},
             {8888: 2, ids[2]: 2}]
    for i in range(200):
        if i < len(fixed):
            hits = fixed[i]
        else:
            n = int(rng.integers(1, 7))
            hits = {allids[int(rng.integers(0, len(allids)))]: int(rng.integers(1, 4)) for _ in range(n)}
        res = int(h.cmd("RESOLVE " + " ".join(f"{a}:{b}" for a, b in hits.items()))[0])
        kat["tree"]["resolve"].append({"hits": {str(a): b for a, b in hits.items()}, "call": res})
    # HLL
    kat["hll"] = []
    sid = 0
    mult = "9E3779B97F4A7C15"
    for n in (1, 10, 100, 1000, 1023, 1024, 1025, 1026, 2000, 10000, 100000, 1000000):
        for sparse in (1, 0):
            for use_n in (0, 1):
                h.cmd(f"HNEW {sid} 12 {sparse}")
                h.cmd(f"HUSEN {sid} {use_n}")
                h.cmd(f"HSEQ {sid} {n} {mult} 0")
                ertl, heule, flaj, nobs, sp, lsz = h.cmd(f"HCARD {sid}")[0].split()
                kat["hll"].append({"n": n, "sparse_start": sparse, "use_n": use_n, "ertl": int(ertl),
                                   "n_observed": int(nobs), "is_sparse": int(sp), "list_size": int(lsz)})
                sid += 1
    # duplicates do not grow the sparse set but the 1025th *insert call* is what switches
    h.cmd(f"HNEW {sid} 12 1"); h.cmd(f"HUSEN {sid} 0")
    h.cmd(f"HSEQ {sid} 1024 {mult} 0"); a = h.cmd(f"HCARD {sid}")[0]
    h.cmd(f"HSEQ {sid} 1 {mult} 5"); b = h.cmd(f"HCARD {sid}")[0]
    kat["hll_dup_switch"] = {"after_1024": a, "after_dup_insert": b}
    sid += 1
    # state dumps + merges (sparse+sparse beyond 1024 stays sparse; dense+sparse; sparse+dense; dense+dense)
    kat["hll_merge"] = []
    for (na, sa, nb, sb) in ((600, 1, 600, 1), (3000, 1, 600, 1), (600, 1, 3000, 1), (3000, 1, 5000, 1),
                             (0, 1, 700, 1), (700, 1, 0, 1), (50, 0, 50, 1)):
        A, B = sid, sid + 1
        sid += 2
        h.cmd(f"HNEW {A} 12 {sa}"); h.cmd(f"HNEW {B} 12 {sb}")
        h.cmd(f"HUSEN {A} 1"); h.cmd(f"HUSEN {B} 1")
        if na: h.cmd(f"HSEQ {A} {na} {mult} 0")
        if nb: h.cmd(f"HSEQ {B} {nb} {mult} {na // 2}")
        h.cmd(f"HMERGE {A} {B}")
        card = h.cmd(f"HCARD {A}")[0].split()
        dump = h.cmd(f"HDUMP {A}")[0].split()
        kat["hll_merge"].append({"na": na, "sa": sa, "nb": nb, "sb": sb, "start_b": na // 2, "ertl": int(card[0]),
                                 "n_observed": int(card[3]), "is_sparse": int(card[4]), "kind": dump[0],
                                 "state": [int(x) for x in dump[1:]]})
    # small dumps of sparse lists / registers for exact state parity
    kat["hll_state"] = []
    for n, sparse in ((300, 1), (1025, 1), (5000, 0)):
        h.cmd(f"HNEW {sid} 12 {sparse}")
        h.cmd(f"HSEQ {sid} {n} {mult} 0")
        dump = h.cmd(f"HDUMP {sid}")[0].split()
        kat["hll_state"].append({"n": n, "sparse_start": sparse, "kind": dump[0], "state": [int(x) for x in dump[1:]]})
        sid += 1
    h.p.stdin.close()
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, separators=(",", ":"))


def make_f11(f1, genomes):
    """UID mapping (classify -I): the reference's set_lcas -I turns f1's k-mers (values zeroed) into a UID
    database -- every k-mer's value names the SET of taxids whose library sequences hold it, uid_to_taxid.map stores the
    sets as {taxid, parent uid} blocks (src/uid_mapping.cpp:32-91) -- and the reference's classify -I resolves reads on it
    (resolve_uids3, :212-274).  The library makes sets of one, two and three taxids."""
    d = os.path.join(HERE, "f11")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    kmers, vals, off, k, nt, _ = synth.read_db(f1)
    synth.write_db(d, kmers, np.zeros_like(vals), off, k, nt)
    os.rename(f"{d}/database.kdb", f"{d}/database0.kdb")
    os.remove(f"{d}/database.idx")
    a = synth.codes_to_ascii
    g4, g5, g6, gp = a(genomes[4]), a(genomes[5]), a(genomes[6]), a(genomes[1000000001])
    recs = [(b"seqA", g4), (b"seqB", g5), (b"seqC", g6), (b"seqP", gp), (b"seqD part of the first genome under another taxid", g4[500:1500]),
            (b"seqE part of the second genome under the genus", g5[1200:2200])]
    with open(f"{d}/library.fa", "wb") as f:
        for h, sq in recs:
            f.write(b">" + h + b"\n")
            for i in range(0, len(sq), 70):
                f.write(sq[i:i + 70] + b"\n")
    with open(f"{d}/seqid2taxid.map", "w") as f:
        f.write("seqA\t4\nseqB\t5\nseqC\t6\nseqP\t1000000001\nseqD\t6\nseqE\t2\n")
    run([os.path.join(REF, "set_lcas"), "-M", "-x", "-d", f"{d}/database0.kdb", "-I", f"{d}/uid_to_taxid.map",
         "-o", f"{d}/uid_database.kdb", "-i", f"{f1}/database.idx", "-b", f"{f1}/taxDB", "-m", f"{d}/seqid2taxid.map",
         "-F", f"{d}/library.fa", "-c", f"{d}/uid_database.kdb.counts"])
    os.remove(f"{d}/database0.kdb")
    for fn in ("library.fa", "seqid2taxid.map"):
        os.remove(f"{d}/{fn}")  # inputs of the reference's build step only
    db = ["-d", f"{d}/uid_database.kdb", "-i", f"{f1}/database.idx", "-a", f"{f1}/taxDB", "-I", f"{d}/uid_to_taxid.map"]
    for tag, extra in (("", []), ("_u1000", ["-u", "1000"])):
        rep = f"{d}/report_uid{tag}.tsv"
        if os.path.exists(rep):
            os.remove(rep)
        run([os.path.join(REF, "classify")] + db + extra + ["-o", f"{d}/out_uid{tag}.tsv", "-r", rep, f"{f1}/reads.fq"])
    if os.path.exists(f"{d}/out_uid_u1000.tsv"):
        assert open(f"{d}/out_uid_u1000.tsv").read() == open(f"{d}/out_uid.tsv").read()
        os.remove(f"{d}/out_uid_u1000.tsv")
    m = np.fromfile(f"{d}/uid_to_taxid.map", dtype="<u4").reshape(-1, 2)
    assert len(m) > 4 and (m[:, 1] != 0).any(), "the fixture must hold UIDs of several taxids"
    return d


def make_kat_uid():
    """known answers of resolve_uids3 and of the container order it depends on (oracle/ref_kat.cpp UIDRESOLVE / UMORDER)"""
    rng = np.random.default_rng(11)
    h = Kat()
    h.cmd(f"K {K}")
    ids = sorted(set(int(x) for x in rng.integers(2, 3000, 50)))
    parent = {1: 1}
    for i, t in enumerate(ids):
        parent[t] = 1 if i < 3 else ids[int(rng.integers(0, i))]
    parent[1000000005] = ids[7]
    parent[7001] = 9999  # orphan
    pm = {t: (0 if (p == t or p not in parent) else p) for t, p in parent.items()}
    h.cmd("PARENT " + " ".join(f"{a}:{b}" for a, b in pm.items()))
    allids = list(pm.keys()) + [4242]  # one taxid the taxonomy does not know
    kat = {"parent_map": {str(a): b for a, b in pm.items()}, "order": [], "maps": []}
    for n in (1, 5, 13, 14, 29, 30, 59, 60, 127, 128, 300, 1200):
        for mode in range(3):
            if mode == 0:
                keys = rng.integers(1, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
            elif mode == 1:
                keys = rng.integers(1, max(n // 2, 3), size=n).astype(np.uint32)
            else:
                keys = (rng.integers(1, 200, size=n) * 13).astype(np.uint32)
            out = h.cmd("UMORDER " + " ".join(map(str, keys.tolist())))[0]
            kat["order"].append({"keys": keys.tolist(), "order": [int(x) for x in out.split()]})
    for n_uid in (6, 40, 400):
        blocks = []
        for u in range(1, n_uid + 1):
            par = 0 if (u == 1 or rng.random() < 0.3) else int(rng.integers(1, u))
            blocks.append([allids[int(rng.integers(0, len(allids)))], par])
        h.cmd("UIDMAP " + " ".join(f"{t}:{p}" for t, p in blocks))
        cases = []
        for c in range(120):
            n = int(rng.choice([1, 2, 3, 4, 6, 10, 30, 120, 400]))
            pool = rng.integers(1, n_uid + 1, size=int(rng.integers(1, min(n, n_uid) + 1)))
            uids = pool[rng.integers(0, len(pool), size=n)].astype(np.uint32)
            call = int(h.cmd("UIDRESOLVE " + " ".join(map(str, uids.tolist())))[0])
            cases.append({"uids": uids.tolist(), "call": call})
        chains = {str(u): [int(x) for x in h.cmd(f"UIDTAXIDS {u}")[0].split()] for u in range(1, min(n_uid, 40) + 1)}
        kat["maps"].append({"blocks": blocks, "chains": chains, "cases": cases})
    h.p.stdin.close()
    with open(os.path.join(HERE, "kat_uid.json"), "w") as f:
        json.dump(kat, f, separators=(",", ":"))


def main():
    if not os.path.exists(os.path.join(REF, "classify")):
        raise SystemExit("build the reference first: make -C oracle ref")
    if sys.argv[1:] == ["f11"]:  # the UID-mapping fixture and known answers without regenerating the others
        rng = np.random.default_rng(7)
        g4 = synth.procedural_genome(7, 4, 3000)
        g5 = synth.mutate(g4, 0.03, rng)
        g6 = synth.procedural_genome(7, 6, 3000)
        gp = np.concatenate([g6[2000:2300], synth.procedural_genome(7, 99, 300)])
        make_f11(os.path.join(HERE, "f1"), {4: g4, 5: g5, 6: g6, 1000000001: gp})
        make_kat_uid()
        return
    if sys.argv[1:] == ["f9"]:  # add the set_lcas fixture without regenerating the others
        rng = np.random.default_rng(7)
        g4 = synth.procedural_genome(7, 4, 3000)
        g5 = synth.mutate(g4, 0.03, rng)
        g6 = synth.procedural_genome(7, 6, 3000)
        gp = np.concatenate([g6[2000:2300], synth.procedural_genome(7, 99, 300)])
        make_f9(os.path.join(HERE, "f1"), {4: g4, 5: g5, 6: g6, 1000000001: gp})
        return
    if sys.argv[1:] == ["f10"]:
        make_f10(os.path.join(HERE, "f1"))
        return
    if sys.argv[1:] == ["f12"]:
        make_f12(os.path.join(HERE, "f1"))
        return
    if sys.argv[1:] == ["f1x"]:
        make_f1_exact_variants()
        return
    if sys.argv[1:] == ["f8"]:  # add the multi-database fixture without regenerating the others
        g4 = synth.procedural_genome(7, 4, 3000)
        make_f8(os.path.join(HERE, "f1"), {4: g4, 6: synth.procedural_genome(7, 6, 3000)})
        return
    f1, genomes = make_f1()
    make_f8(f1, genomes)
    make_f9(f1, genomes)
    make_f2(f1)
    make_f4(f1, genomes)
    make_f7(f1)
    make_f10(f1)
    make_f12(f1)
    make_f11(f1, genomes)
    make_kat(f1)
    make_kat_uid()
    # count_unique known answer (HLL p=12 on the reads' k-mers)
    r = run([os.path.join(REF, "count_unique"), "-k", "31", "-p", "12"], stdin=open(f"{HERE}/f2/edge.fa", "rb"))
    open(os.path.join(HERE, "count_unique_edge.txt"), "wb").write(r.stdout)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
