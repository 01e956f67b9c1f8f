#!/usr/bin/env python3
"""Differential run of the ORACLE against the compiled reference, on the CPU (not collected by pytest; `python
tests/fuzz_oracle.py [cases] [first_seed]`; tests/test_oracle_golden.py runs a bounded slice where oracle/_ref exists):
oracle/ku_oracle.c -- the checker every GPU parity test compares with -- against oracle/_ref/classify on random databases
(nt, taxonomy, sometimes a second database), FASTA / FASTQ files and flags (-q -m, -c, -s, -u): Kraken lines byte for byte,
report row for row (the sparse / dense sketch states per work unit included).  Pins the oracle beyond the committed vectors."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from krakenuniq_amd import synth  # noqa: E402
from oracle import ku_oracle as ko  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "classify")
K = 31


def rows(text):
    return sorted(text.strip("\n").split("\n"))


def small_db(rng, tax, n_gen, glen, nt):
    genomes, base = {}, {}
    for i, tid in enumerate(tax.species):
        par = tax.parent[tid]
        if par not in base:
            base[par] = synth.procedural_genome(int(rng.integers(1, 1 << 30)), i, glen)
        genomes[tid] = synth.mutate(base[par], 0.03, rng)
    kmers, vals = synth.lca_database(genomes, tax, K)
    sk, sv, off = synth.sort_db(kmers, vals, K, nt)
    return genomes, sk, sv, off


def one_case(seed, tmp):
    rng = np.random.default_rng(seed)
    nt = int(rng.choice([6, 9, 10, 11]))
    n_gen = int(rng.integers(2, 7))
    tax = synth.random_taxonomy(n_gen, rng, levels=tuple(int(x) for x in rng.integers(2, 7, size=int(rng.integers(2, 6)))))
    dirs = [os.path.join(tmp, "db_a")]
    shutil.rmtree(dirs[0], ignore_errors=True)
    genomes, sk, sv, off = small_db(rng, tax, n_gen, int(rng.integers(2000, 5000)), nt)
    synth.write_db(dirs[0], sk, sv, off, K, nt)
    taxdb = os.path.join(dirs[0], "taxDB")
    tax.write(taxdb)
    pool = dict(genomes)
    if rng.random() < 0.25:  # a second database behind the first
        dirs.append(os.path.join(tmp, "db_b"))
        shutil.rmtree(dirs[1], ignore_errors=True)
        nt_b = int(rng.choice([6, 9, 10]))
        g_b, sk_b, sv_b, off_b = small_db(rng, tax, n_gen, int(rng.integers(2000, 5000)), nt_b)
        synth.write_db(dirs[1], sk_b, sv_b, off_b, K, nt_b)
        pool = {("a", t): g for t, g in genomes.items()}
        pool.update({("b", t): g for t, g in g_b.items()})
    sp = list(pool)
    weights = rng.pareto(0.7, size=len(sp)) + 0.01
    weights = weights / weights.sum()
    n_reads = int(rng.integers(1, 2500))
    seqs, ids = [], []
    for i in range(n_reads):
        if rng.random() < 0.08:
            s = bytes(rng.choice(np.frombuffer(b"ACGTNacgtnRY", dtype=np.uint8), size=int(rng.integers(1, 220))).tobytes())
        else:
            g = pool[sp[int(rng.choice(len(sp), p=weights))]]
            n = int(rng.integers(K - 2, min(700, len(g) - 1)))
            a = int(rng.integers(0, len(g) - n))
            c = g[a:a + n]
            r = bytearray(synth.codes_to_ascii(c if rng.random() < 0.5 else synth.revcomp_codes(c)))
            for _ in range(int(rng.poisson(n * float(rng.choice([0.0, 0.003, 0.03]))))):
                r[int(rng.integers(0, n))] = ord("ACGTN"[int(rng.integers(0, 5))])
            s = bytes(r)
        seqs.append(s)
        ids.append(f"r{i}" + (" some description" if rng.random() < 0.2 else ""))
    path = os.path.join(tmp, "reads." + ("fq" if rng.random() < 0.5 else "fa"))
    if path.endswith("fq"):
        synth.write_fastq(path, seqs, ids)
    else:
        synth.write_fasta(path, seqs, ids, width=int(rng.choice([0, 0, 60, 70])))
    quick = rng.random() < 0.25
    mh = int(rng.integers(1, 5))
    only_c, pseq = rng.random() < 0.2, rng.random() < 0.2
    unit = int(rng.choice([1, 1500, 7000, 30000, 500000]))
    out, rep = os.path.join(tmp, "out.tsv"), os.path.join(tmp, "rep.tsv")
    for p in (out, rep):
        if os.path.exists(p):
            os.unlink(p)
    cmd = [REF]
    for d in dirs:
        cmd += ["-d", f"{d}/database.kdb", "-i", f"{d}/database.idx"]
    cmd += ["-a", taxdb, "-t", "1", "-u", str(unit), "-o", out, "-r", rep]
    cmd += (["-q", "-m", str(mh)] if quick else []) + (["-c"] if only_c else []) + (["-s"] if pseq else [])
    r = subprocess.run(cmd + [path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    desc = (f"nt {nt} genomes {n_gen} databases {len(dirs)} reads {n_reads} {os.path.basename(path)} -u {unit}"
            f"{' -q -m ' + str(mh) if quick else ''}{' -c' if only_c else ''}{' -s' if pseq else ''}")
    assert r.returncode == 0, (desc, "the reference failed", r.returncode, r.stderr.decode(errors="replace")[-300:])
    # ---- the oracle on the same files
    r_ids, r_seqs = synth.read_seqfile(path)
    dbs = [ko.Db(f"{d}/database.kdb", f"{d}/database.idx") for d in dirs]
    otax = ko.Tax(taxdb)
    run = ko.Run(dbs[0], otax, work_unit_nt=unit, quick=quick, min_hits=mh, extra_dbs=tuple(dbs[1:]))
    res = run.classify(r_seqs)
    got = ko.kraken_lines(r_ids, r_seqs, res, quick=quick, only_classified=only_c, print_seq=pseq)
    assert got.encode() == open(out, "rb").read(), (desc, "Kraken lines")
    want_rep = open(rep).read()
    got_rep = run.report(taxdb, "\n".join(f"{d}/database.kdb.counts" for d in dirs))
    assert rows(got_rep) == rows(want_rep), (desc, "report")
    return desc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    assert os.path.exists(REF), "oracle/_ref/classify is built where /root/reference is present (oracle/Makefile)"
    t0 = time.time()
    bad = 0
    tmp = tempfile.mkdtemp(prefix="ku_fuzz_oracle_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for seed in range(first, first + n):
            try:
                print(f"seed {seed}: ok  ({one_case(seed, tmp)})", flush=True)
            except AssertionError as e:
                bad += 1
                print(f"seed {seed}: MISMATCH {str(e)[:600]}", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
