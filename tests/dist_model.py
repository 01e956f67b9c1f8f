"""torch.distributed model of the multi-GPU exchange (backend "gloo" in the CPU tests).

TEST INFRASTRUCTURE (it lives under tests/ for that reason).  The product path is the C++ driver
krakenuniq_amd/csrc/ku_mgpu.cpp (RCCL through the C ABI: ku_mgpu_*), which bench.py and the classify executable use.
This module states the same exchange protocol on plain torch tensors so that the world-size-2 CPU tests
(tests/test_dist_gloo.py) can check it -- slice plan, "non-zero wins" max-merge, owner-computes accounting, end-of-run
state merge -- without a GPU.

The reference shards its database only *in time* (--preload-size chunk mode,
src/krakendb.cpp:411-526, src/classify.cpp:566-791) and merges per-k-mer taxa with
"non-zero wins" (src/classify.cpp:445-452).  Here the same minimizer-range shards
live on different GPUs at once:

  replicas  every rank holds the whole DB and classifies its own reads; the only
            exchange is the end-of-run merge of the per-taxon state
            (HLL registers: MAX, n_kmers / n_reads: SUM)  == taxon_counts[t] += local[t]
            (src/classify.cpp:541-544) across ranks.
  sharded   rank r holds the bins [bounds[r], bounds[r+1]); every rank scans the same
            read batch and looks up only the k-mers it owns (is_minimizer_in_chunk);
            per-k-mer slot ids are merged by MAX (exactly one rank is non-zero per
            k-mer, ambiguous k-mers are 0xFFFFFFFF = -1 on every rank) -- as a
            reduce-scatter over read-aligned slices done with point-to-point transfers
            (exchange_slices_max; merge_taxa_max is the plain all-reduce form) -- and
            each rank then resolves its own slice of the reads.  HLL / n_kmers are
            owner-computes (the bin owner also accounts the misses), merged as above.
            Round 3 (the default of the C++ driver): OWNER ROUTING -- a rank scans only its own slice of
            the reads and sends each unambiguous k-mer to the rank that owns its bin; the slot comes back
            (alltoallv / route_lookup).  Same results, an eighth of the scan and a third of the bytes.

All functions take plain torch tensors so the same code runs under gloo on CPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def read_slice(n_reads: int, rank: int, world_size: int):
    """contiguous slice of a broadcast batch that `rank` resolves"""
    per = (n_reads + world_size - 1) // world_size
    lo = min(rank * per, n_reads)
    return lo, min(lo + per, n_reads)


def allgather_values(local_values: np.ndarray, device) -> np.ndarray:
    """union of the distinct DB taxids of all shards, ascending (so every rank numbers slots alike)"""
    rank, ws = world()
    if ws == 1:
        return np.asarray(local_values, dtype=np.uint32)
    n = torch.tensor([len(local_values)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(max(m, 1), dtype=torch.int64, device=device)
    pad[:len(local_values)] = torch.from_numpy(np.asarray(local_values, dtype=np.int64)).to(device)
    parts = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    vals = np.concatenate([p[:int(s.item())].cpu().numpy() for p, s in zip(parts, sizes)])
    return np.unique(vals).astype(np.uint32)


def merge_taxa_max(taxa_i32: torch.Tensor) -> torch.Tensor:
    """in-place MAX all-reduce of per-k-mer slot ids viewed as int32 (non-zero wins; KU_AMBIG == -1 everywhere)"""
    _, ws = world()
    if ws > 1:
        dist.all_reduce(taxa_i32, op=dist.ReduceOp.MAX)
    return taxa_i32


def exchange_slices_max(taxa_i32: torch.Tensor, pos) -> torch.Tensor:
    """the sharded step's reduce-scatter as the C++ driver does it over xGMI (comm_reduce_slices_max, ku_mgpu.cpp): an
    all-to-all of point-to-point transfers -- slice q = [pos[q], pos[q+1]) of every rank's array goes straight to rank q,
    which folds the world-1 slices it receives into its own with a max.  In place; afterwards this rank's slice holds
    the merged slots (the other slices keep the local values)."""
    rank, ws = world()
    if ws == 1:
        return taxa_i32
    lo, hi = int(pos[rank]), int(pos[rank + 1])
    stage = [torch.empty(hi - lo, dtype=taxa_i32.dtype) for _ in range(ws - 1)]
    ops, k = [], 0
    for q in range(ws):
        if q == rank:
            continue
        a, b = int(pos[q]), int(pos[q + 1])
        if b > a:
            ops.append(dist.P2POp(dist.isend, taxa_i32[a:b].contiguous(), q))
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, stage[k], q))
        k += 1
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for st in stage:
        if hi > lo:
            torch.maximum(taxa_i32[lo:hi], st, out=taxa_i32[lo:hi])
    return taxa_i32


def alltoallv(send: list, dtype=torch.int64) -> list:
    """the routed step's variable all-to-all as the C++ driver does it (comm_allgather_u64 + comm_alltoallv, ku_mgpu.cpp):
    the per-owner counts are all-gathered so that every rank knows how much it gets from whom, then queue q of every rank
    goes straight to rank q by point-to-point transfers.  send[q]: 1-D tensor for rank q; returns recv[q] from rank q."""
    rank, ws = world()
    if ws == 1:
        return [send[0].clone()]
    cnt = torch.tensor([len(s) for s in send], dtype=torch.int64)
    table = [torch.zeros_like(cnt) for _ in range(ws)]
    dist.all_gather(table, cnt)
    recv = [torch.empty(int(table[q][rank]), dtype=dtype) for q in range(ws)]
    recv[rank] = send[rank].clone()
    ops = []
    for q in range(ws):
        if q == rank:
            continue
        if len(send[q]):
            ops.append(dist.P2POp(dist.isend, send[q].contiguous(), q))
        if len(recv[q]):
            ops.append(dist.P2POp(dist.irecv, recv[q], q))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv


def route_lookup(kmers_i64: torch.Tensor, owner: torch.Tensor, probe) -> torch.Tensor:
    """owner routing of one rank's slice (rank_step_routed, ku_mgpu.cpp): the unambiguous k-mers of the slice (kmers_i64, in
    scan order) go to the rank that owns their bin (owner[i]), probe(k-mers received) -> slots runs THERE (with the
    owner's accounting as a side effect), the slots come back and are scattered into scan order."""
    _, ws = world()
    order = [torch.nonzero(owner == q).flatten() for q in range(ws)]
    got = alltoallv([kmers_i64[ix] for ix in order])
    back = alltoallv([probe(g).to(torch.int64) for g in got])
    out = torch.zeros(len(kmers_i64), dtype=torch.int64)
    for q in range(ws):
        out[order[q]] = back[q]
    return out


def reduce_state(registers_u8: torch.Tensor, n_kmers_i64: torch.Tensor, n_reads_i64: torch.Tensor):
    """end-of-run merge of the per-taxon state across ranks, IN PLACE (as ku_mgpu_reduce_state): registers MAX,
    counters SUM; returns the three tensors"""
    _, ws = world()
    r, k, n = registers_u8, n_kmers_i64, n_reads_i64
    if ws > 1:
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return r, k, n
