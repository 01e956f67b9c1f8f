#!/usr/bin/env python3
"""Extended differential run on a GPU box (not collected by pytest; `python tests/fuzz_cli.py [cases] [first_seed]`): the drop-in
executable krakenuniq_amd/bin/classify against the compiled reference oracle/_ref/classify ON THE SAME FILES AND FLAGS --
random databases (nt, taxonomy), FASTA / FASTQ inputs (one or two files, plain or .gz with one or several members, wrapped
lines, ambiguous bases, short and empty reads), random flags (-q -m, -c, -s, -u, -p, -x, -M, -C, -U, -r), sometimes a second database behind the first, sometimes as classifyExact: the Kraken file, the classified / unclassified read files byte
for byte, the report row for row.  The reference runs with -t 1 (its output order is its input order only then)."""
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
import gpu_common as gc  # noqa: E402

OURS = os.environ.get("KU_FUZZ_CLI_OURS") or os.path.join(ROOT, "krakenuniq_amd", "bin", "classify")  # (the variable: a dry run of the generator, reference against reference)
REF = os.path.join(ROOT, "oracle", "_ref", "classify")


def rows(text):
    return sorted(text.strip("\n").split("\n"))


def one_case(seed, tmp):
    rng = np.random.default_rng(seed)
    k = 31
    nt = int(rng.choice([6, 9, 10, 11, 12]))  # (the index file holds 8 * 4^nt bytes: 134 MB at 12, 8.6 GB at 15)
    n_gen = int(rng.integers(2, 7))
    tax = synth.random_taxonomy(n_gen, rng, levels=tuple(int(x) for x in rng.integers(2, 7, size=int(rng.integers(2, 6)))))
    long_reads = os.environ.get("KU_FUZZ_CLI_BIAS", "") == "long"  # reads of several kbp (the windowed kernel, many runs per read)
    db = gc.random_db(rng, n_genomes=min(n_gen, 4) if long_reads else n_gen, glen=int(rng.integers(12000, 30000)) if long_reads else int(rng.integers(2000, 5000)),
                      k=k, nt=nt, tax=tax)
    dirs = {}
    for who in ("ref", "ours"):  # (each run writes its own database.kdb.counts; the files themselves are shared)
        d = os.path.join(tmp, f"db_{who}")
        shutil.rmtree(d, ignore_errors=True)
        if who == "ref":
            synth.write_db(d, db["kmers"], db["vals"], db["offsets"], k, nt)
            tax.write(os.path.join(d, "taxDB"))
        else:
            os.makedirs(d)
            for fn in ("database.kdb", "database.idx", "taxDB"):
                os.symlink(os.path.join(dirs["ref"], fn), os.path.join(d, fn))
        dirs[who] = d
    # a second database searched behind the first (classify -d A -d B, src/classify.cpp:928-936): other genomes, same taxonomy
    two_dbs = rng.random() < 0.2 and os.environ.get("KU_FUZZ_CLI_BIAS", "") != "chunk"
    genome_pool = dict(db["genomes"])
    if two_dbs:
        nt_b = int(rng.choice([6, 9, 10, 11]))
        db_b = gc.random_db(rng, n_genomes=n_gen, glen=int(rng.integers(2000, 5000)), k=k, nt=nt_b, tax=tax)
        synth.write_db(os.path.join(tmp, "db_b"), db_b["kmers"], db_b["vals"], db_b["offsets"], k, nt_b)
        genome_pool = {("a", t): g for t, g in db["genomes"].items()}
        genome_pool.update({("b", t): g for t, g in db_b["genomes"].items()})
    sp = list(genome_pool)
    weights = rng.pareto(0.7, size=len(sp)) + 0.01
    weights = weights / weights.sum()
    files = []
    for fi in range(int(rng.integers(1, 3))):
        n_reads = int(rng.integers(1, 1500)) if not long_reads else int(rng.integers(1, 300))
        seqs, ids = [], []
        for i in range(n_reads):
            if rng.random() < 0.08:
                s = bytes(rng.choice(np.frombuffer(b"ACGTNacgtnRY", dtype=np.uint8), size=int(rng.integers(0, 220))).tobytes())
            else:
                g = genome_pool[sp[int(rng.choice(len(sp), p=weights))]]
                n = int(rng.integers(k - 2, min(600, len(g) - 1))) if not (long_reads and rng.random() < 0.5) else int(rng.integers(600, len(g) - 1))
                a = int(rng.integers(0, len(g) - n))
                c = g[a:a + n]
                r = bytearray(synth.codes_to_ascii(c if rng.random() < 0.5 else synth.revcomp_codes(c)))
                if rng.random() < 0.25:
                    r[int(rng.integers(0, n))] = ord("N")
                if rng.random() < 0.2:
                    r[int(rng.integers(0, n))] = ord("ACGT"[int(rng.integers(0, 4))])
                if long_reads:  # substitutions all along: the taxon changes every few k-mers
                    for _ in range(int(rng.poisson(n * float(rng.choice([0.0, 0.01, 0.05]))))):
                        r[int(rng.integers(0, n))] = ord("ACGTN"[int(rng.integers(0, 5))])
                s = bytes(r)
            seqs.append(s)
            ids.append(f"f{fi}r{i}" + (" some description" if rng.random() < 0.2 else ""))
        path = os.path.join(tmp, f"reads{fi}." + ("fq" if rng.random() < 0.5 else "fa"))
        if path.endswith("fq"):
            seqs = [s if s else b"A" for s in seqs]  # (an empty FASTQ record ends the reference's reader: tests/fuzz_seqio.py covers those rules)
            synth.write_fastq(path, seqs, ids)
        else:
            seqs = [s if s else b"N" for s in seqs]
            synth.write_fasta(path, seqs, ids, width=int(rng.choice([0, 0, 60, 70])))
        if rng.random() < 0.3:  # the same text as a .gz file (the reference reads it through gzstream, this executable through its inflating team)
            import gzip
            raw = open(path, "rb").read()
            os.unlink(path)
            path += ".gz"
            if rng.random() < 0.5:
                open(path, "wb").write(gzip.compress(raw, compresslevel=int(rng.integers(1, 10))))
            else:  # several members
                cut = sorted(rng.integers(0, len(raw) + 1, size=int(rng.integers(1, 4))).tolist())
                parts = [raw[a:b] for a, b in zip([0] + cut, cut + [len(raw)])]
                open(path, "wb").write(b"".join(gzip.compress(x) for x in parts))
        files.append(path)
    flags = []
    if rng.random() < 0.25:
        flags += ["-q", "-m", str(int(rng.integers(1, 5)))]
    if rng.random() < 0.2:
        flags += ["-c"]
    if rng.random() < 0.2:
        flags += ["-s"]
    if rng.random() < 0.5:
        flags += ["-u", str(int(rng.choice([1, 1500, 7000, 30000, 150000])))]
    if rng.random() < 0.3:
        flags += ["-p", str(int(rng.choice([0, 10, 12, 14, 16])))]
    bias = os.environ.get("KU_FUZZ_CLI_BIAS", "")  # "chunk": every case in the reference's chunk mode, with a report; "devices": this executable with KU_DEVICES=0,0,...
    if (rng.random() < 0.25 and not two_dbs) or bias == "chunk":
        flags += ["-x", str(int(rng.integers(64, 400))) + "K"]
    if rng.random() < 0.3:
        flags += ["-M"]
    want_c, want_u, want_r = rng.random() < 0.3, rng.random() < 0.3, rng.random() < 0.7 or bias == "chunk"
    outs = {}
    exact = rng.random() < 0.15  # classifyExact: the same executable under its other name (exact distinct k-mer counts in the report)
    sfx = "Exact" if exact else ""
    for who, exe, threads in (("ref", REF + sfx, "1"), ("ours", OURS + sfx, "1" if os.environ.get("KU_FUZZ_CLI_OURS") else str(int(rng.integers(1, 9))))):
        d = dirs[who]
        o = {kk: os.path.join(tmp, f"{who}_{kk}") for kk in ("out", "rep", "cls", "ucls")}
        for p in o.values():
            if os.path.exists(p):
                os.unlink(p)
        cmd = [exe, "-d", f"{d}/database.kdb", "-i", f"{d}/database.idx"]
        if two_dbs:
            cmd += ["-d", os.path.join(tmp, "db_b", "database.kdb"), "-i", os.path.join(tmp, "db_b", "database.idx")]
        cmd += ["-a", f"{d}/taxDB", "-t", threads, "-o", o["out"]] + flags
        if want_r:
            cmd += ["-r", o["rep"]]
        if want_c:
            cmd += ["-C", o["cls"]]
        if want_u:
            cmd += ["-U", o["ucls"]]
        env = dict(os.environ)
        # a group of ranks on the one device (minimizer-range shards, or chunks dealt out with -x); classifyExact on a group takes
        # neither quick mode nor a second database (the executable says so and exits 70)
        if who == "ours" and bias == "devices" and not (exact and ("-q" in flags or two_dbs)):
            env["KU_DEVICES"] = ",".join(["0"] * int(rng.integers(2, 6)))
        r = subprocess.run(cmd + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
        outs[who] = (r.returncode, {kk: (open(p, "rb").read() if os.path.exists(p) else None) for kk, p in o.items()}, r.stderr.decode(errors="replace")[-400:])
    (rc_r, f_r, e_r), (rc_o, f_o, e_o) = outs["ref"], outs["ours"]
    desc = f"{'classifyExact ' if exact else ''}{'two databases ' if two_dbs else ''}nt {nt} genomes {n_gen} files {[os.path.basename(f) for f in files]} flags {' '.join(flags)}{' -r' if want_r else ''}{' -C' if want_c else ''}{' -U' if want_u else ''}"
    if rc_r < 0:
        return desc + f" (the reference died of signal {-rc_r}: nothing to compare; this executable exits {rc_o})"
    assert rc_r == rc_o, (desc, "exit codes", rc_r, rc_o, e_r, e_o)
    if rc_r != 0:
        return desc + f" (both exit {rc_r})"
    assert f_r["out"] == f_o["out"], (desc, "Kraken file")
    for kk in ("cls", "ucls"):
        assert f_r[kk] == f_o[kk], (desc, kk)
    if want_r:
        assert (f_r["rep"] is None) == (f_o["rep"] is None), (desc, "report presence")
        if f_r["rep"] is not None:
            assert rows(f_r["rep"].decode()) == rows(f_o["rep"].decode()), (desc, "report")
    return desc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    assert os.path.exists(REF), "oracle/_ref/classify is built where /root/reference is present (oracle/Makefile) and travels with the tree"
    t0 = time.time()
    bad = 0
    tmp = tempfile.mkdtemp(prefix="ku_fuzz_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
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
