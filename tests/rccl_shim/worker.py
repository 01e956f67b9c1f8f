#!/usr/bin/env python3
"""One rank of a multi-PROCESS ku_mgpu group on a single GPU (tests/test_gpu_rccl_shim.py starts `world` of these with
KU_RCCL_LIB pointing at libku_rccl_shim.so): the driver's RCCL code paths -- ncclCommInitRank, grouped send / receive
all-to-alls, the scatter of the read slices, the all-gathers of values and routing counts, the all-reduce of the state --
run between real peers.  Every rank checks its own slice against one context that holds the whole database.
    worker.py <rank> <world> <mode: route|slots|reduce|replicas> <scratch dir>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank, W, mode, scratch = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
if mode in ("slots", "reduce"):
    os.environ["KU_MGPU_EXCHANGE"] = mode
import numpy as np
import torch

from krakenuniq_amd import capi, synth_torch

K, NT, L, N = 31, 9, 150, 60_000
dev = torch.device("cuda:0")
geo = dict(n_species=40, genome_len=20_000, k=K, nt=NT, seed=5)
db = synth_torch.BenchDb(dev, **geo)
ids, par = db.tax.arrays()
ctax = capi.Tax(ids=ids, parents=par)
batches = [db.sample_reads(N, L, seed=21 + i) for i in range(2)]
stride = L + 1
nb = N * stride
nk = L - K + 1
pairs_again = db.pairs.clone() if mode == "replicas" else None  # (set_taxonomy turns an adopted array's taxids into slot ids)
# ---- the answer: one context, whole database
ctx = capi.Ctx(0)
ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
ctx.set_taxonomy(ctax)
want = []
for seqs, off, lens, _ in batches:
    taxa1 = torch.zeros(nb, dtype=torch.int32, device=dev)
    calls1 = torch.zeros(N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.classify_batch_device(seqs.data_ptr(), nb, off.data_ptr(), lens.data_ptr(), N, calls1.data_ptr(), taxa1.data_ptr(), max_read_len=L)
    ctx.synchronize()
    want.append((calls1, taxa1))
want_counts = ctx.counts()
# ---- the communicator's id: rank 0 makes it, the others find it in the scratch directory
uid_path = os.path.join(scratch, "uid.bin")
if rank == 0:
    uid = capi.mgpu_unique_id()
    uid.tofile(uid_path + ".tmp")
    os.rename(uid_path + ".tmp", uid_path)
else:
    t0 = time.time()
    while not os.path.exists(uid_path):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.01)
    uid = np.fromfile(uid_path, dtype=np.uint8)


def same_counts(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("slot_taxid", "n_kmers", "registers", "node_taxid", "n_reads"))


replicas = mode == "replicas"
mg = capi.Mgpu([0], first_rank=rank, world=W, unique_id=uid, flags=capi.KU_MGPU_REPLICAS if replicas else 0)
assert mg.uses_rccl()
if replicas:
    mg.ctx(0).adopt_db(pairs_again.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=pairs_again)
    mg.set_taxonomy(ctax)  # (all-gather of the ranks' value lists)
    r0, r1 = N * rank // W, N * (rank + 1) // W
    for seqs, off, lens, _ in batches:  # every rank its own reads, the state all-reduced at the end
        taxa = torch.zeros(nb, dtype=torch.int32, device=dev)
        calls = torch.zeros(N, dtype=torch.int32, device=dev)
        n = r1 - r0
        mg.ctx(0).classify_batch_device(seqs.data_ptr(), nb, off[r0:].data_ptr(), lens[r0:].data_ptr(), n, calls.data_ptr(), taxa.data_ptr(), max_read_len=L)
        mg.ctx(0).synchronize()
    mg.reduce_state()
    assert same_counts(mg.ctx(0).counts(), want_counts)
else:
    offs = db.offsets
    bounds = [0] + [int(torch.searchsorted(offs, offs[-1] * q // W).item()) for q in range(1, W)] + [4 ** NT]
    sh = synth_torch.BenchDb(dev, bin_lo=bounds[rank], bin_hi=bounds[rank + 1], **geo)
    mg.ctx(0).adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, bounds[rank], bounds[rank + 1])
    mg.set_taxonomy(ctax)
    assert mg.uses_routing() == (mode == "route")
    rb = [N * r // W for r in range(W + 1)]
    pb = [x * stride for x in rb]
    buf = {"seqs": torch.zeros(nb + 16, dtype=torch.uint8, device=dev), "off": torch.zeros(N, dtype=torch.int64, device=dev),
           "len": torch.zeros(N, dtype=torch.int32, device=dev), "calls": torch.zeros(N, dtype=torch.int32, device=dev),
           "taxa": torch.zeros(nb + 16, dtype=torch.int32, device=dev)}
    for (seqs, off, lens, _), (calls1, taxa1) in zip(batches, want):
        if rank == 0:  # rank 0 holds the batch: the slices are scattered (routing) or the batch broadcast (position-wise)
            buf["seqs"][:nb] = seqs.reshape(-1)
            buf["off"][:] = off
            buf["len"][:] = lens
        torch.cuda.synchronize()
        mg.step_device([{"d_seqs": buf["seqs"].data_ptr(), "d_seq_off": buf["off"].data_ptr(), "d_seq_len": buf["len"].data_ptr(),
                         "d_calls": buf["calls"].data_ptr(), "d_taxa": buf["taxa"].data_ptr()}], nb, N, rb, pb, max_read_len=L)
        mg.ctx(0).synchronize()
        lo, hi = rb[rank], rb[rank + 1]
        assert torch.equal(buf["calls"][lo:hi], calls1[lo:hi]), "calls of the slice differ"
        assert torch.equal(buf["taxa"][:nb].view(N, stride)[lo:hi, :nk], taxa1.view(N, stride)[lo:hi, :nk]), "per-k-mer codes of the slice differ"
    mg.reduce_state()
    assert same_counts(mg.ctx(0).counts(), want_counts), "the reduced per-taxon state differs"
mg.close()
print(f"rank {rank} of {W} ({mode}): ok", flush=True)
