#!/usr/bin/env python3
"""Time of the device stage of a `classify -r` run without the executable around it (gpurun): the bench database, N reads
in batches of 64 Mi nt through ku_classify_batch_rle with the sparse-sketch emulation on (-u 500000), pinned buffers.
    [KU_LIB=variant.so] python scripts/sparse_probe.py [n_reads]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from krakenuniq_amd import capi, synth_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
db = synth_torch.BenchDb(dev, n_species=2000, genome_len=310_000, k=31, nt=13, seed=7)
db.kmers = db.vals = None
ctx = capi.Ctx(0)
ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), 31, 13, 2, keep=db)
ids_t, par_t = db.tax.arrays()
ctx.set_taxonomy(capi.Tax(ids=ids_t, parents=par_t))
L, stride = 150, 151
d_seqs, _, _, _ = db.sample_reads(n, L, seed=1)
hb = d_seqs.cpu().pin_memory().numpy().reshape(-1)
per = (64 << 20) // stride  # reads per batch (KU_BATCH_NT of the executable)
pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()
off = pin(per, torch.int64).view(np.uint64)
lens = pin(per, torch.int32).view(np.uint32)
off[:] = np.arange(per, dtype=np.uint64) * stride
lens[:] = L
obuf = {"calls": pin(per, torch.int32).view(np.uint32), "hits": pin(per, torch.int32).view(np.uint32),
        "run_cnt": pin(per, torch.int32).view(np.uint32), "run_off": pin(per, torch.int64).view(np.uint64),
        "runs": pin((per * 8 + (1 << 20), 2), torch.int32).view(np.uint32)}
for rep in range(2):
    for sparse in (False, True):
        ctx.reset_counts()
        if sparse:
            ctx.enable_sparse(500000, 31)
        else:
            ctx.disable_sparse()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in range(0, n, per):
            b = min(n, a + per)
            ctx.classify_batch_rle(hb[a * stride:b * stride], off[:b - a], lens[:b - a], out=obuf)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print(f"{'with' if sparse else 'without'} the emulation: {n} reads in {(n + per - 1) // per} batches: {dt * 1e3:.1f} ms", flush=True)
