#!/usr/bin/env python3
"""Where does ku_classify_batch_rle spend its time?  (run through gpurun)
    python scripts/rle_pipeline_probe.py [n_reads] [variant ...]
Times the host-buffer entry point on pinned buffers for the variants default / no_overlap (KU_NO_H2D_OVERLAP) /
old (KU_NO_FUSED_RLE: per-k-mer array + RLE kernel); with one variant named it runs only that (for rocprofv3)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from krakenuniq_amd import capi, synth_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
variants = sys.argv[2:] or ["default", "no_overlap", "old", "default"]
dev = torch.device("cuda:0")
db = synth_torch.BenchDb(dev, n_species=int(os.environ.get("PROBE_SPECIES", "400")), genome_len=310_000, k=31, nt=13, seed=7)
db.kmers = db.vals = None
ctx = capi.Ctx(0)
ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), 31, 13, 2, keep=db)
ids_t, par_t = db.tax.arrays()
ctx.set_taxonomy(capi.Tax(ids=ids_t, parents=par_t))
L = 150
stride = L + 1
d_seqs, _, _, _ = db.sample_reads(n, L, seed=1)
hb = d_seqs.cpu().pin_memory().numpy()
pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()
off, lens = pin(n, torch.int64).view(np.uint64), pin(n, torch.int32).view(np.uint32)
off[:] = np.arange(n, dtype=np.uint64) * stride
lens[:] = L
obuf = {"calls": pin(n, torch.int32).view(np.uint32), "hits": pin(n, torch.int32).view(np.uint32),
        "run_cnt": pin(n, torch.int32).view(np.uint32), "run_off": pin(n, torch.int64).view(np.uint64),
        "runs": pin((n * 8 + (1 << 22), 2), torch.int32).view(np.uint32)}
torch.cuda.synchronize()
ENV = {"default": {}, "no_overlap": {"KU_NO_H2D_OVERLAP": "1"}, "old": {"KU_NO_FUSED_RLE": "1"}}
for v in variants:
    for k_ in ("KU_NO_H2D_OVERLAP", "KU_NO_FUSED_RLE"):
        os.environ.pop(k_, None)
    os.environ.update(ENV[v])
    ts = []
    for rep in range(4):
        ctx.reset_counts()
        t0 = time.perf_counter()
        r = ctx.classify_batch_rle(hb, off, lens, out=obuf)
        ts.append(time.perf_counter() - t0)
    print(f"{v:12s} {n} reads: " + " ".join(f"{t * 1e3:.2f}" for t in ts) + f" ms  -> {n / min(ts) / 1e6:.1f} Mreads/s; extent {len(r['runs'])} "
          f"runs {int(r['run_cnt'].sum())}", flush=True)
# the pieces alone
t0 = time.perf_counter()
d = torch.from_numpy(hb).to(dev, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D of the {hb.nbytes >> 20} MiB read buffer alone: {(time.perf_counter() - t0) * 1e3:.2f} ms")
