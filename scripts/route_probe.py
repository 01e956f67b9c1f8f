#!/usr/bin/env python3
"""Device work of one sharded step with N = 8 ownership, owner routing against the position-wise exchange.  (gpurun)
    python scripts/route_probe.py [route|slots] [n_reads] [ranks]
Eight ranks of one ku_mgpu group share the one device of the box (the exchange runs through device copies behind the same
interface), each holding one minimizer-range shard of the 8 GB bench database.  The wall time of a step is then the SUM of
the eight ranks' device work (plus the copies); on eight devices each rank does an eighth of it side by side.  Under
rocprofv3 --kernel-trace --stats the per-kernel totals say where it goes (scripts/summarize_profile.py)."""
import os
import sys
import time

mode = sys.argv[1] if len(sys.argv) > 1 else "route"
if mode == "slots":
    os.environ["KU_MGPU_EXCHANGE"] = "slots"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from krakenuniq_amd import capi, synth_torch

N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K, NT, L = 31, 13, 150
dev = torch.device("cuda:0")
geo = dict(n_species=int(os.environ.get("PROBE_SPECIES", "2000")), genome_len=310_000, k=K, nt=NT, seed=7)
probe = synth_torch.BenchDb(dev, n_species=32, genome_len=50_000, k=K, nt=NT, seed=7)
bounds = synth_torch.quantile_bin_bounds(synth_torch.bin_key(probe.kmers[torch.randperm(probe.n_pairs, device=dev)[:1_000_000]], K, NT),
                                         4 ** NT, W)
del probe
mg = capi.Mgpu([0] * W)
keep = []
for r in range(W):
    sh = synth_torch.BenchDb(dev, bin_lo=int(bounds[r]), bin_hi=int(bounds[r + 1]), **geo)
    sh.kmers = sh.vals = None
    mg.ctx(r).adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, int(bounds[r]), int(bounds[r + 1]))
    keep.append(sh)
ids, par = keep[0].tax.arrays()
mg.set_taxonomy(capi.Tax(ids=ids, parents=par))
for sh in keep:
    sh.pairs = None
torch.cuda.empty_cache()
assert mg.uses_routing() == (mode == "route")
seqs, off, lens, _ = keep[0].sample_reads(N, L, seed=1)  # (a shard keeps the genomes; only its database is cut)
seqs = seqs.reshape(-1)
nb = seqs.numel()
stride = L + 1
rb = [N * r // W for r in range(W + 1)]
pb = [x * stride for x in rb]
bufs = []
for r in range(W):
    bufs.append({"seqs": seqs if r == 0 else torch.zeros(nb + 16, dtype=torch.uint8, device=dev),
                 "off": off if r == 0 else torch.zeros(N, dtype=torch.int64, device=dev),
                 "len": lens if r == 0 else torch.zeros(N, dtype=torch.int32, device=dev),
                 "calls": torch.zeros(N, dtype=torch.int32, device=dev), "taxa": torch.zeros(nb + 16, dtype=torch.int32, device=dev)})
args = [{"d_seqs": b["seqs"].data_ptr(), "d_seq_off": b["off"].data_ptr(), "d_seq_len": b["len"].data_ptr(),
         "d_calls": b["calls"].data_ptr(), "d_taxa": b["taxa"].data_ptr()} for b in bufs]
torch.cuda.synchronize()
ts = []
for rep in range(4):
    t0 = time.perf_counter()
    mg.step_device(args, nb, N, rb, pb, max_read_len=L)
    for r in range(W):
        mg.ctx(r).synchronize()
    ts.append(time.perf_counter() - t0)
classified = sum(int((bufs[r]["calls"][rb[r]:rb[r + 1]] != 0).sum()) for r in range(W))
print(f"{mode}: {W} ranks on one device, {N} reads per step: " + " ".join(f"{t * 1e3:.1f}" for t in ts) +
      f" ms per step (sum over the ranks) -> {min(ts) * 1e3 / W:.2f} ms per rank; classified {classified}", flush=True)
mg.close()
