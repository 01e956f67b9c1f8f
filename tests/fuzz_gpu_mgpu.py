#!/usr/bin/env python3
"""Extended differential run on a GPU box (not collected by pytest; `python tests/fuzz_gpu_mgpu.py [cases] [first_seed]`): the
multi-GPU driver (ku_mgpu.cpp) with 2-6 ranks on one device -- minimizer-range shards with owner routing (also with queues too
small for the first pass and with rounds cut inside reads), the position-wise exchange, replicas -- on random databases
(k, nt, taxonomy), reads of very different abundance per taxon, random work-unit sizes and batch cuts, the sparse-sketch
emulation on: calls, per-taxon counts and registers, which sketches stayed sparse and their exact sets against the oracle after
ku_mgpu_reduce_state.  What tests/test_gpu_mgpu.py checks on one seed, on many."""
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


def sparse_state_equals_oracle(ctx, run):
    counts = ctx.counts()
    flags, pairs = ctx.sparse_export()
    want = run.counts()
    slot_of = {int(t): s for s, t in enumerate(counts["slot_taxid"])}
    got_sets = {}
    for p in pairs.tolist():
        got_sets.setdefault(p >> 32, set()).add(p & 0xFFFFFFFF)
    n_sparse = n_dense = 0
    for t, c in want.items():
        if not c["n_kmers"]:
            continue
        s = slot_of[t]
        assert bool(flags[s]) == c["sparse"], ("sparse flag", t, c["n_kmers"])
        if c["sparse"]:
            assert got_sets.get(s, set()) == set(c["sketch"].sparse_list().tolist()), ("sparse set", t)
            n_sparse += 1
        else:
            n_dense += 1
    return counts, flags, pairs, n_sparse, n_dense


def one_case(seed):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([21, 25, 31, 31, 31]))
    nt = int(rng.choice([6, 9, 9, 12, 13, 15]))
    n_gen = int(rng.integers(3, 10))
    tax = synth.random_taxonomy(n_gen, rng, levels=tuple(int(x) for x in rng.integers(2, 7, size=int(rng.integers(2, 6)))))
    db = gc.random_db(rng, n_genomes=n_gen, glen=int(rng.integers(2500, 9000)), k=k, nt=nt, tax=tax)
    sp = list(db["genomes"])
    weights = rng.pareto(0.7, size=len(sp)) + 0.01
    weights = weights / weights.sum()
    n_reads = int(rng.integers(500, 7000))
    seqs = []
    for _ in range(n_reads):
        u = rng.random()
        if u < 0.08:
            seqs.append(bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=int(rng.integers(0, 220))).tobytes()))
            continue
        g = db["genomes"][sp[int(rng.choice(len(sp), p=weights))]]
        n = int(rng.integers(k - 2, min(420, len(g) - 1)))
        s = int(rng.integers(0, len(g) - n))
        c = g[s:s + n]
        r = bytearray(synth.codes_to_ascii(c if rng.random() < 0.5 else synth.revcomp_codes(c)))
        if n and rng.random() < 0.25:
            r[int(rng.integers(0, n))] = ord("N")
        if n and rng.random() < 0.2:  # a substitution: taxa change along the read
            r[int(rng.integers(0, n))] = ord("ACGT"[int(rng.integers(0, 4))])
        seqs.append(bytes(r))
    unit = int(rng.choice([1500, 7000, 30000, 150000, 500000]))
    buf, off, lens = ko.pack_reads(seqs)
    ids, par = tax.arrays()
    ctax, otax = capi.Tax(ids=ids, parents=par), ko.Tax(ids=ids, parents=par)
    kl = (2 * k + 7) // 8
    raw = np.zeros((len(db["kmers"]), kl + 4), dtype=np.uint8)
    raw[:, :kl] = db["kmers"].astype("<u8").view(np.uint8).reshape(-1, 8)[:, :kl]
    raw[:, kl:] = db["vals"].astype("<u4").view(np.uint8).reshape(-1, 4)
    raw = raw.reshape(-1)
    cdb = capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=k, offsets=db["offsets"], nt=nt)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=k, offsets=db["offsets"], nt=nt)
    world = int(rng.integers(2, 7))
    mode = str(rng.choice(["route", "route", "route_tight", "route_rounds", "slots", "replicas"]))
    env = {}
    if mode == "route_tight":
        env["KU_ROUTE_CAP"] = str(int(rng.integers(200, 5000)))
    if mode == "route_rounds":
        env["KU_ROUTE_ROUND"] = str(int(rng.integers(20000, 400000)) | 1)
    if mode == "slots":
        env["KU_MGPU_EXCHANGE"] = "slots"
    os.environ.update(env)
    try:
        mg = capi.Mgpu([0] * world, flags=capi.KU_MGPU_REPLICAS if mode == "replicas" else 0)
        mg.load(cdb, ctax)
        mg.enable_sparse(unit)
        n_b = int(rng.integers(1, 8))
        cuts = [0] + sorted(set(rng.integers(1, max(2, n_reads), size=n_b - 1).tolist())) + [n_reads] if n_b > 1 else [0, n_reads]
        calls = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            lo = int(off[a])
            hi = int(off[b]) if b < len(off) else len(buf)
            calls.append(mg.classify_batch_rle(buf[lo:hi], off[a:b] - lo, lens[a:b])["calls"].copy())
        mg.reduce_state()
        run = ko.Run(odb, otax, work_unit_nt=unit)
        res = run.classify(seqs)
        assert np.array_equal(np.concatenate(calls) if calls else np.zeros(0, np.uint32), res["calls"]), "calls"
        counts, flags, pairs, n_sparse, n_dense = sparse_state_equals_oracle(mg.ctx(0), run)  # (the sets are folded into rank 0, include/krakenuniq_amd.h)
        gc.assert_same_counts(counts, run)
        routed = mg.uses_routing()
        mg.close()
    finally:
        for kk in env:
            os.environ.pop(kk, None)
    return (f"k {k} nt {nt} genomes {n_gen} reads {n_reads} unit {unit} batches {len(cuts) - 1} ranks {world} {mode}"
            f"{' (routed)' if routed else ''} sparse {n_sparse} dense {n_dense}")


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
            print(f"seed {seed}: MISMATCH {e}", flush=True)
    print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
