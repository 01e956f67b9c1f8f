#!/usr/bin/env python3
"""Extended differential run on a GPU box (not collected by pytest; `python tests/fuzz_build_tools.py [cases] [first_seed]`):
the build steps krakenuniq_amd/bin/db_sort and bin/set_lcas against the compiled reference's oracle/_ref/db_sort and
oracle/_ref/set_lcas ON THE SAME FILES -- a shuffled Jellyfish-style k-mer list of random genomes -> sorted database + index
(random minimizer length, with and without -z), then the LCAs from a random multi-FASTA library under a random taxonomy
(sequences of known, unknown and unmapped ids, wrapped lines, lower case, ambiguous bases, k-mers missing from the database
with -x): database.kdb / database.idx byte for byte; then the UID form of the same (set_lcas -I: database and UID map byte for
byte) and classify -I over it with both executables (Kraken file byte for byte, report row for row)."""
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

K = 31
BIN = os.path.join(ROOT, "krakenuniq_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return r.returncode, r.stderr.decode(errors="replace")[-300:]


def one_case(seed, tmp):
    rng = np.random.default_rng(seed)
    nt = int(rng.choice([6, 9, 10, 11, 12]))  # (the index file holds 8 * 4^nt bytes)
    n_gen = int(rng.integers(2, 7))
    tax = synth.random_taxonomy(n_gen, rng, levels=tuple(int(x) for x in rng.integers(2, 6, size=int(rng.integers(2, 5)))))
    genomes = {}
    base = {}
    for i, tid in enumerate(tax.species):
        par = tax.parent[tid]
        if par not in base:
            base[par] = synth.procedural_genome(int(rng.integers(1, 1 << 30)), i, int(rng.integers(1500, 5000)))
        genomes[tid] = synth.mutate(base[par], 0.03, rng)
    kmers = np.unique(np.concatenate([synth.canonical(synth.kmers_forward(g, K), K) for g in genomes.values()]))
    if rng.random() < 0.5:  # the database lacks some of the library's k-mers (-x)
        kmers = kmers[rng.random(len(kmers)) > 0.05]
    vals = rng.integers(1, 1 << 31, len(kmers), dtype=np.uint32)
    perm = rng.permutation(len(kmers))
    jdb = os.path.join(tmp, "in.jdb")
    synth.write_jdb(jdb, kmers[perm], vals[perm], K)
    zero = rng.random() < 0.5
    out = {}
    for who, d in (("ref", REF), ("ours", BIN)):
        kdb, idx = os.path.join(tmp, f"{who}.kdb"), os.path.join(tmp, f"{who}.idx")
        for p in (kdb, idx):
            if os.path.exists(p):
                os.unlink(p)
        rc, err = run([os.path.join(d, "db_sort")] + (["-z"] if zero else []) + ["-t", "1", "-n", str(nt), "-d", jdb, "-o", kdb, "-i", idx])
        assert rc == 0, (who, "db_sort", rc, err)
        out[who] = (open(kdb, "rb").read(), open(idx, "rb").read())
    assert out["ref"][0] == out["ours"][0], "db_sort: database.kdb"
    assert out["ref"][1] == out["ours"][1], "db_sort: database.idx"
    # ---- set_lcas over the (zeroed or not) sorted database
    taxdb = os.path.join(tmp, "taxDB")
    tax.write(taxdb)
    lib, smap = os.path.join(tmp, "library.fa"), os.path.join(tmp, "seqid2taxid.map")
    with open(lib, "wb") as f, open(smap, "w") as m:
        n_seq = 0
        for tid, g in genomes.items():
            for part in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, len(g) // 2))
                b = int(rng.integers(a + K + 5, len(g)))
                s = bytearray(synth.codes_to_ascii(g[a:b]))
                if rng.random() < 0.3:
                    s = bytearray(bytes(s).lower())
                for _ in range(int(rng.poisson(2))):
                    s[int(rng.integers(0, len(s)))] = ord("N")
                sid = f"seq{n_seq}"
                n_seq += 1
                u = rng.random()
                if u < 0.85:
                    m.write(f"{sid}\t{tid}\n")
                elif u < 0.92:
                    m.write(f"{sid}\t{999999}\n")  # a taxid the taxonomy does not hold
                # else: unmapped
                f.write(f">{sid} some description\n".encode())
                w = int(rng.choice([0, 60, 70, 80]))
                if w:
                    for i in range(0, len(s), w):
                        f.write(bytes(s[i:i + w]) + b"\n")
                else:
                    f.write(bytes(s) + b"\n")
        if rng.random() < 0.3:
            m.write(f"seq0\t{tax.species[-1]}\n")  # an id listed twice
    res = {}
    for who, d in (("ref", REF), ("ours", BIN)):
        src_kdb, idx = os.path.join(tmp, f"{who}.kdb"), os.path.join(tmp, f"{who}.idx")
        okdb = os.path.join(tmp, f"{who}_lca.kdb")
        if os.path.exists(okdb):
            os.unlink(okdb)
        rc, err = run([os.path.join(d, "set_lcas"), "-x", "-t", "1", "-d", src_kdb, "-o", okdb, "-i", idx, "-b", taxdb, "-m", smap, "-F", lib])
        res[who] = (rc, open(okdb, "rb").read() if os.path.exists(okdb) else None, err)
    assert res["ref"][0] == res["ours"][0], ("set_lcas exit codes", res["ref"][0], res["ours"][0], res["ref"][2], res["ours"][2])
    if res["ref"][0] == 0:
        assert res["ref"][1] == res["ours"][1], "set_lcas: database.kdb"
    # ---- UID databases (set_lcas -I, src/uid_mapping.cpp): the k-mers' values are ids of taxid SETS, numbered as they come up;
    # then classify -I over the reference's UID database with both executables (resolve_uids3, src/classify.cpp:1067-...)
    uid_note = ""
    if rng.random() < 0.6:
        zdb = os.path.join(tmp, "zero")
        os.makedirs(zdb, exist_ok=True)
        rc, err = run([os.path.join(REF, "db_sort"), "-z", "-t", "1", "-n", str(nt), "-d", jdb, "-o", f"{zdb}/database.kdb", "-i", f"{zdb}/database.idx"])
        assert rc == 0, ("db_sort -z", err)
        ures = {}
        for who, d in (("ref", REF), ("ours", BIN)):
            ukdb, umap = os.path.join(tmp, f"{who}_uid.kdb"), os.path.join(tmp, f"{who}_uid.map")
            for pth in (ukdb, umap):
                if os.path.exists(pth):
                    os.unlink(pth)
            zin = os.path.join(tmp, f"{who}_zero.kdb")  # (its own copy: the reference's set_lcas writes through its mapping of the input)
            shutil.copy(f"{zdb}/database.kdb", zin)
            rc, err = run([os.path.join(d, "set_lcas"), "-x", "-t", "1", "-d", zin, "-I", umap, "-o", ukdb, "-i", f"{zdb}/database.idx",
                           "-b", taxdb, "-m", smap, "-F", lib])
            ures[who] = (rc, open(ukdb, "rb").read() if os.path.exists(ukdb) else None, open(umap, "rb").read() if os.path.exists(umap) else None, err)
        assert ures["ref"][0] == ures["ours"][0], ("set_lcas -I exit codes", ures["ref"][0], ures["ours"][0], ures["ref"][3], ures["ours"][3])
        if ures["ref"][0] == 0:
            assert ures["ref"][1] == ures["ours"][1], "set_lcas -I: database.kdb"
            assert ures["ref"][2] == ures["ours"][2], "set_lcas -I: UID map"
            # reads over the UID database
            reads = os.path.join(tmp, "reads.fa")
            with open(reads, "wb") as f:
                tids = list(genomes)
                for i in range(int(rng.integers(50, 600))):
                    g = genomes[tids[int(rng.integers(0, len(tids)))]]
                    n = int(rng.integers(K - 1, min(400, len(g) - 1)))
                    a = int(rng.integers(0, len(g) - n))
                    r = bytearray(synth.codes_to_ascii(g[a:a + n]))
                    if rng.random() < 0.3:
                        r[int(rng.integers(0, n))] = ord("ACGTN"[int(rng.integers(0, 5))])
                    f.write(f">r{i}\n".encode() + bytes(r) + b"\n")
            outs = {}
            for who, exe in (("ref", os.path.join(REF, "classify")), ("ours", os.path.join(BIN, "classify"))):
                udir = os.path.join(tmp, f"udb_{who}")
                shutil.rmtree(udir, ignore_errors=True)
                os.makedirs(udir)
                shutil.copy(os.path.join(tmp, "ref_uid.kdb"), f"{udir}/database.kdb")
                shutil.copy(f"{zdb}/database.idx", f"{udir}/database.idx")
                o, rp = os.path.join(tmp, f"{who}_uid_out"), os.path.join(tmp, f"{who}_uid_rep")
                for pth in (o, rp):
                    if os.path.exists(pth):
                        os.unlink(pth)
                rc, err = run([exe, "-d", f"{udir}/database.kdb", "-i", f"{udir}/database.idx", "-a", taxdb, "-I", os.path.join(tmp, "ref_uid.map"),
                               "-t", "1", "-o", o, "-r", rp, reads])
                outs[who] = (rc, open(o, "rb").read() if os.path.exists(o) else None,
                             sorted(open(rp).read().strip("\n").split("\n")) if os.path.exists(rp) else None, err)
            assert outs["ref"][0] == outs["ours"][0], ("classify -I exit codes", outs["ref"][0], outs["ours"][0], outs["ref"][3], outs["ours"][3])
            if outs["ref"][0] == 0:
                assert outs["ref"][1] == outs["ours"][1], "classify -I: Kraken file"
                assert outs["ref"][2] == outs["ours"][2], "classify -I: report"
            uid_note = f", UID build + classify -I (exit {outs['ref'][0]})"
        else:
            uid_note = f", set_lcas -I exit {ures['ref'][0]} on both"
    return f"nt {nt} genomes {n_gen} k-mers {len(kmers)}{' -z' if zero else ''} sequences {n_seq} (set_lcas exit {res['ref'][0]}){uid_note}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    for t in ("db_sort", "set_lcas"):
        assert os.path.exists(os.path.join(REF, t)), f"oracle/_ref/{t} is built where /root/reference is present (oracle/Makefile)"
    t0 = time.time()
    bad = 0
    tmp = tempfile.mkdtemp(prefix="ku_fuzz_build_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for seed in range(first, first + n):
            try:
                print(f"seed {seed}: ok  ({one_case(seed, tmp)})", flush=True)
            except AssertionError as e:
                bad += 1
                print(f"seed {seed}: MISMATCH {str(e)[:500]}", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
