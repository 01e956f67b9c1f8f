#!/usr/bin/env python3
"""Extended differential run on a GPU box (not collected by pytest; `python tests/fuzz_gpu_step_device.py [cases] [first_seed]`):
ku_mgpu_step_device -- the call bench.py's sharded leg times -- with 2-8 ranks on one device against ONE context that holds the
whole database: random minimizer lengths, read shapes (150 bp, mate pairs, reads of a few kbp: the windowed ROUTE instance),
batch sizes, several steps in a row (the state accumulates), rounds cut by position, queues too small for the first pass, the
position-wise exchange.  Calls and per-k-mer codes of every rank's slice, and the merged per-taxon state after
ku_mgpu_reduce_state, bit for bit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
from krakenuniq_amd import capi, synth_torch  # noqa: E402

K = 31


def same_counts(a, b):
    return all(np.array_equal(a[kk], b[kk]) for kk in ("slot_taxid", "n_kmers", "registers", "node_taxid", "n_reads"))


def one_case(seed):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    NT = int(rng.choice([9, 11, 13]))
    W = int(rng.integers(2, 9))
    n_species = int(rng.integers(20, 120))
    glen = int(rng.integers(20_000, 70_000))
    dbseed = int(rng.integers(1, 1000))
    shape = str(rng.choice(["short", "short", "pairs", "long"]))
    mode = str(rng.choice(["route", "route", "route_tight", "route_rounds", "slots"]))
    env = {}
    if mode == "route_tight":
        env["KU_ROUTE_CAP"] = str(int(rng.integers(2_000, 200_000)))
    if mode == "route_rounds":
        env["KU_ROUTE_ROUND"] = str(int(rng.integers(200_000, 4_000_000)) | 1)
    if mode == "slots":
        env["KU_MGPU_EXCHANGE"] = "slots"
    os.environ.update(env)
    try:
        db = synth_torch.BenchDb(dev, n_species=n_species, genome_len=glen, k=K, nt=NT, seed=dbseed)
        ids, par = db.tax.arrays()
        ctax = capi.Tax(ids=ids, parents=par)
        ctx = capi.Ctx(0)
        ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
        ctx.set_taxonomy(ctax)
        offs = db.offsets
        bounds = [0] + [int(torch.searchsorted(offs, offs[-1] * q // W).item()) for q in range(1, W)] + [4 ** NT]
        mg = capi.Mgpu([0] * W)
        shards = []
        for r in range(W):
            sh = synth_torch.BenchDb(dev, n_species=n_species, genome_len=glen, k=K, nt=NT, seed=dbseed, bin_lo=bounds[r], bin_hi=bounds[r + 1])
            mg.ctx(r).adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, bounds[r], bounds[r + 1])
            shards.append(sh)
        mg.set_taxonomy(ctax)
        steps = int(rng.integers(1, 4))
        for step in range(steps):
            if shape == "pairs":
                N, L = int(rng.integers(2_000, 60_000)), 301
                seqs, off, lens = db.sample_pairs(N, 150, seed=int(rng.integers(1, 10_000)))
            elif shape == "long":
                L = int(rng.integers(600, 4000))
                N = int(rng.integers(200, 6_000))
                seqs, off, lens, _ = db.sample_reads(N, L, seed=int(rng.integers(1, 10_000)))
            else:
                L = int(rng.choice([75, 100, 150, 151]))
                N = int(rng.integers(1_000, 250_000))
                seqs, off, lens, _ = db.sample_reads(N, L, seed=int(rng.integers(1, 10_000)))
            seqs = seqs.reshape(-1)
            nb = seqs.numel()
            stride = nb // N
            taxa1 = torch.zeros(nb, dtype=torch.int32, device=dev)
            calls1 = torch.zeros(N, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.classify_batch_device(seqs.data_ptr(), nb, off.data_ptr(), lens.data_ptr(), N, calls1.data_ptr(), taxa1.data_ptr(), max_read_len=L)
            ctx.synchronize()
            # read slices of uneven size (the bounds the caller hands in)
            cuts = sorted(rng.integers(0, N + 1, size=W - 1).tolist()) if rng.random() < 0.5 else [N * r // W for r in range(1, W)]
            rb = [0] + cuts + [N]
            pb = [x * stride for x in rb]
            bufs = []
            for r in range(W):
                bufs.append({"seqs": seqs if r == 0 else torch.zeros(nb + 16, dtype=torch.uint8, device=dev),
                             "off": off if r == 0 else torch.zeros(N, dtype=torch.int64, device=dev),
                             "len": lens if r == 0 else torch.zeros(N, dtype=torch.int32, device=dev),
                             "calls": torch.zeros(N, dtype=torch.int32, device=dev), "taxa": torch.zeros(nb + 16, dtype=torch.int32, device=dev)})
            torch.cuda.synchronize()
            mg.step_device([{"d_seqs": b["seqs"].data_ptr(), "d_seq_off": b["off"].data_ptr(), "d_seq_len": b["len"].data_ptr(),
                             "d_calls": b["calls"].data_ptr(), "d_taxa": b["taxa"].data_ptr()} for b in bufs], nb, N, rb, pb, max_read_len=L)
            for r in range(W):
                mg.ctx(r).synchronize()
            nk = L - K + 1
            for r in range(W):
                lo, hi = rb[r], rb[r + 1]
                assert torch.equal(bufs[r]["calls"][lo:hi], calls1[lo:hi]), ("calls", step, r)
                assert torch.equal(bufs[r]["taxa"][:nb].view(N, stride)[lo:hi, :nk], taxa1.view(N, stride)[lo:hi, :nk]), ("codes", step, r)
            del bufs, taxa1, calls1, seqs, off, lens
        want = ctx.counts()
        mg.reduce_state()
        assert same_counts(mg.ctx(int(rng.integers(0, W))).counts(), want), "merged state"
        routed = mg.uses_routing()
        mg.close()
        del ctx, shards, db
        torch.cuda.empty_cache()
    finally:
        for kk in env:
            os.environ.pop(kk, None)
    return f"nt {NT} ranks {W} species {n_species} x {glen} {shape} steps {steps} {mode}{' (routed)' if routed else ''}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    bad = 0
    for seed in range(first, first + n):
        try:
            print(f"seed {seed}: ok  ({one_case(seed)})", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:300]}", flush=True)
    print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
