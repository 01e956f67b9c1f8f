#!/usr/bin/env python3
"""Extended differential run on a GPU box (not collected by pytest; `python tests/fuzz_gpu_parity.py [cases] [first_seed]`):
random database geometries (k, nt), reads from k - 2 bases to several kbp (the one-pass, the windowed and the flat instances),
ambiguous bases, empty reads, quick mode with random --min-hits, with and without accounting -- calls, per-k-mer codes, hit
counts, per-taxon counts and HLL registers against the oracle, bit for bit; the kernel's own runs (the host-batch call) expanded
again against the per-k-mer array and formatted against the line formatter of the per-k-mer form."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from krakenuniq_amd import capi, synth  # noqa: E402
from oracle import ku_oracle as ko  # noqa: E402
import gpu_common as gc  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([17, 21, 25, 31, 31, 31]))
    nt = int(rng.choice([n for n in (1, 6, 9, 12, 13, 15) if n < k]))
    n_gen = int(rng.integers(2, 9))
    db = gc.random_db(rng, n_genomes=n_gen, glen=int(rng.integers(2000, 12000)), k=k, nt=nt)
    sp = list(db["genomes"])
    shape = rng.choice(["short", "mixed", "long"], p=[0.5, 0.3, 0.2])
    n_reads = int(rng.integers(50, 1500)) if shape != "long" else int(rng.integers(10, 80))
    seqs = []
    for _ in range(n_reads):
        u = rng.random()
        if u < 0.05:
            seqs.append(bytes(rng.choice(np.frombuffer(b"ACGTNacgtn", dtype=np.uint8), size=int(rng.integers(0, 300))).tobytes()))
            continue
        g = db["genomes"][sp[int(rng.integers(0, len(sp)))]]
        hi = {"short": 320, "mixed": 1500, "long": len(g) - 1}[shape]
        n = int(rng.integers(max(1, k - 2), min(hi, len(g) - 1)))
        s = int(rng.integers(0, len(g) - n))
        c = g[s:s + n]
        r = bytearray(synth.codes_to_ascii(c if rng.random() < 0.5 else synth.revcomp_codes(c)))
        for _ in range(int(rng.poisson(n * float(rng.choice([0.0, 0.002, 0.02]))))):
            r[int(rng.integers(0, n))] = ord("N")
        for _ in range(int(rng.poisson(n * 0.01))):
            r[int(rng.integers(0, n))] = ord("ACGT"[int(rng.integers(0, 4))])
        seqs.append(bytes(r))
    kl = (2 * k + 7) // 8
    raw = np.zeros((len(db["kmers"]), kl + 4), dtype=np.uint8)
    raw[:, :kl] = db["kmers"].astype("<u8").view(np.uint8).reshape(-1, 8)[:, :kl]
    raw[:, kl:] = db["vals"].astype("<u4").view(np.uint8).reshape(-1, 4)
    raw = raw.reshape(-1)
    ids, par = db["tax"].arrays()
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=k, offsets=db["offsets"], nt=nt)
    otax = ko.Tax(ids=ids, parents=par)
    cdb = capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=k, offsets=db["offsets"], nt=nt)
    ctax = capi.Tax(ids=ids, parents=par)
    ctx, _, _ = gc.make_ctx(cdb=cdb, ctax=ctax)
    quick = rng.random() < 0.25
    mh = int(rng.integers(1, 6))
    kw = dict(quick=True, min_hits=mh) if quick else {}
    run, res, buf, off, lens, taxa = gc.oracle_flat(odb, otax, seqs, **kw)
    gpu = ctx.classify_batch(buf, off, lens, flags=capi.KU_F_QUICK if quick else 0, min_hits=mh)
    gc.assert_same_classification(gpu, res, taxa, off, lens, k, quick=quick)
    gc.assert_same_counts(ctx.counts(), run)
    if not quick:
        # the host-batch call: the kernel's runs, expanded, are the per-k-mer array; both line formatters agree
        ctx.reset_counts()
        rle = ctx.classify_batch_rle(buf, off, lens)
        assert np.array_equal(rle["calls"], res["calls"]), "rle calls"
        gc.assert_same_counts(ctx.counts(), run)
        names = [f"r{i}" for i in range(len(seqs))]
        a = capi.format_kraken(buf, off, lens, names, k, gpu["calls"], taxa=gpu["taxa"])
        b = capi.format_kraken_rle(buf, off, lens, names, k, rle)
        assert a == b, "formatted lines"
    return f"k {k} nt {nt} genomes {n_gen} reads {n_reads} {shape}{' quick ' + str(mh) if quick else ''}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    bad = 0
    for seed in range(first, first + n):
        try:
            info = one_case(seed)
            print(f"seed {seed}: ok  ({info})", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:300]}", flush=True)
    print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
