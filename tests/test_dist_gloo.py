"""world_size-2 `gloo` test of the multi-GPU orchestration (tests/dist_model.py) on CPU.

The collectives are exercised with the CPU oracle standing in for the kernels: rank r produces exactly what
ku_lookup_device would produce for minimizer shard r (slots of the k-mers whose bin it owns, 0 elsewhere, -1
for ambiguous k-mers; owner-computes HLL registers / n_kmers including the misses) and the merged result must
equal the unsharded oracle run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from krakenuniq_amd import capi, synth, synth_torch
import dist_model as kdist
from oracle import ku_oracle as ko

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f1")
K, NT = 31, 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        ids, seqs = synth.read_seqfile(f"{GOLDEN}/reads.fq")
        seqs = seqs[:300]
        odb, otax = ko.Db(f"{GOLDEN}/database.kdb", f"{GOLDEN}/database.idx"), ko.Tax(f"{GOLDEN}/taxDB")
        run = ko.Run(odb, otax)
        res = run.classify(seqs)
        kmers, vals, off, *_ = synth.read_db(GOLDEN)
        bounds = capi.Db(f"{GOLDEN}/database.kdb", f"{GOLDEN}/database.idx").shard_plan(ws)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        # distinct values of this shard -> union over ranks
        mine = np.unique(vals[int(off[lo]):int(off[hi])])
        allv = kdist.allgather_values(mine[mine != 0], torch.device("cpu"))
        assert (allv == np.unique(vals[vals != 0])).all()
        slot_of = {int(t): i + 1 for i, t in enumerate(allv)}
        slot_of[0] = 0
        # per-k-mer slots this rank would write + its owner-computes state
        n_slots = len(allv) + 1
        regs = np.zeros((n_slots, 4096), dtype=np.uint8)
        n_kmers = np.zeros(n_slots, dtype=np.int64)
        sketches = {}
        local, full = [], []
        for i, s in enumerate(seqs):
            fwd, amb = ko.scan(s, K)
            a, n = int(res["taxa_off"][i]), int(res["n_slots"][i])
            canon = synth.canonical(fwd, K)
            bins = synth.bin_key(canon, K, NT)
            own = (bins >= lo) & (bins < hi) & (amb == 0)
            t = res["taxa"][a:a + n]
            sl = np.array([slot_of[int(x)] for x in t], dtype=np.int32)
            full.append(np.where(amb != 0, -1, sl).astype(np.int32))
            local.append(np.where(amb != 0, -1, np.where(own, sl, 0)).astype(np.int32))
            for j in np.nonzero(own)[0]:
                h = sketches.setdefault(int(sl[j]), ko.Hll(12, False))
                h.insert(int(canon[j]))
                n_kmers[int(sl[j])] += 1
        for s, h in sketches.items():
            regs[s] = h.registers()
        want_all = np.concatenate(full)
        r0, r1 = kdist.read_slice(len(seqs), rank, ws)  # the reads this rank resolves (and, routed, the only ones it scans)
        # the exchange as the C++ driver does it: all-to-all of read-aligned slices + max-merge of the received ones
        starts = np.concatenate([[0], np.cumsum([len(x) for x in local])]).astype(np.int64)
        pos = [int(starts[kdist.read_slice(len(seqs), q, ws)[0]]) for q in range(ws)] + [int(starts[-1])]
        mine_after = kdist.exchange_slices_max(torch.from_numpy(np.concatenate(local).copy()), pos).numpy()
        assert (mine_after[pos[rank]:pos[rank + 1]] == want_all[pos[rank]:pos[rank + 1]]).all()
        # ... and as one all-reduce (round 1's form)
        merged = kdist.merge_taxa_max(torch.from_numpy(np.concatenate(local)))
        assert (merged.numpy() == want_all).all()
        # ... and by owner routing (round 3): this rank scans only ITS reads, k-mers travel to their owners, slots return
        table = {int(km): slot_of[int(v)] for km, v in zip(kmers[int(off[lo]):int(off[hi])], vals[int(off[lo]):int(off[hi])])}
        r_regs, r_nk, r_sk = np.zeros_like(regs), np.zeros_like(n_kmers), {}

        def probe(got):
            out = torch.zeros(len(got), dtype=torch.int64)
            for j, km in enumerate(got.numpy().view(np.uint64)):
                s = table.get(int(km), 0)
                out[j] = s
                r_sk.setdefault(s, ko.Hll(12, False)).insert(int(km))
                r_nk[s] += 1
            return out

        my_kmers, my_owner, my_want, my_amb = [], [], [], []
        for i in range(r0, r1):
            fwd, amb = ko.scan(seqs[i], K)
            canon = synth.canonical(fwd, K)
            bins = synth.bin_key(canon, K, NT)
            my_kmers.append(canon[amb == 0])
            my_owner.append(np.searchsorted(np.asarray(bounds, dtype=np.uint64), bins[amb == 0], side="right") - 1)
            my_want.append(full[i][amb == 0])
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        slots = kdist.route_lookup(torch.from_numpy(cat(my_kmers, np.uint64).view(np.int64)), torch.from_numpy(cat(my_owner, np.int64)), probe)
        assert (slots.numpy() == cat(my_want, np.int64)).all()
        for s, h in r_sk.items():
            r_regs[s] = h.registers()
        assert (r_regs == regs).all() and (r_nk == n_kmers).all()  # the owner booked exactly what the position-wise scan books
        node_ids = sorted({int(c) for c in res["calls"]})
        n_reads = np.zeros(len(node_ids), dtype=np.int64)
        for c in res["calls"][r0:r1]:
            n_reads[node_ids.index(int(c))] += 1
        R, Kc, Nr = kdist.reduce_state(torch.from_numpy(regs), torch.from_numpy(n_kmers), torch.from_numpy(n_reads))
        want = run.counts()
        for t, c in want.items():
            if c["n_kmers"]:
                s = slot_of[t]
                assert int(Kc[s]) == c["n_kmers"]
                assert (R[s].numpy() == c["sketch"].registers()).all()
            assert int(Nr[node_ids.index(t)]) == c["n_reads"] if t in node_ids else c["n_reads"] == 0
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ws", [2, 3])
def test_sharded_merge(ws):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(ws, port, ret), nprocs=ws, join=True)
    assert dict(ret) == {r: "ok" for r in range(ws)}


def _replica_worker(rank, ws, port, ret):
    """replicas mode: every rank classifies its own reads; reduce_state (registers MAX, counters SUM) must give the
    per-taxon state of one run over all reads (taxon_counts[t] += local[t], classify.cpp:541-544)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        ids, seqs = synth.read_seqfile(f"{GOLDEN}/reads.fq")
        seqs = seqs[:400]
        odb, otax = ko.Db(f"{GOLDEN}/database.kdb", f"{GOLDEN}/database.idx"), ko.Tax(f"{GOLDEN}/taxDB")
        whole = ko.Run(odb, otax)
        whole.classify(seqs)
        want = whole.counts()
        taxids = sorted(want)  # same universe on every rank
        mine = ko.Run(odb, otax)
        mine.classify(seqs[rank::ws])  # interleaved shares
        c = mine.counts()
        regs = np.stack([c[t]["sketch"].registers() if t in c else np.zeros(4096, np.uint8) for t in taxids])
        nk = np.array([c[t]["n_kmers"] if t in c else 0 for t in taxids], dtype=np.int64)
        nr = np.array([c[t]["n_reads"] if t in c else 0 for t in taxids], dtype=np.int64)
        R, Kc, Nr = kdist.reduce_state(torch.from_numpy(regs), torch.from_numpy(nk), torch.from_numpy(nr))
        for i, t in enumerate(taxids):
            assert int(Kc[i]) == want[t]["n_kmers"] and int(Nr[i]) == want[t]["n_reads"]
            assert (R[i].numpy() == want[t]["sketch"].registers()).all()
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_replicas_state_merge_world2():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_replica_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_read_slice_and_bounds():
    for n, ws in ((10, 3), (0, 2), (7, 8), (1000, 4)):
        cover = []
        for r in range(ws):
            lo, hi = kdist.read_slice(n, r, ws)
            cover += list(range(lo, hi))
        assert cover == list(range(n))
    b = synth_torch.quantile_bin_bounds(torch.arange(0, 1000), 4 ** 7, 4)
    assert b[0] == 0 and b[-1] == 4 ** 7 and (np.diff(b.astype(np.int64)) > 0).all()
