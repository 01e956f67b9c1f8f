"""-m gpu, the configs[2] shape on one MI355X: a minimizer-range shard of 36 GB (3 G pairs, nt = 15) + the rest of that
database as a second shard, SHARDED lookup kernels, per-k-mer slots merged with max ("non-zero wins",
classify.cpp:445-452), resolve on the merge -- calls, per-k-mer codes, HLL registers, n_kmers and n_reads against the
oracle on a 200 k-read sample, followed by the configs[3] / configs[4] shapes on the same standard-geometry shards (40 k mate
pairs 2 x 150 + N and 600 reads of 10 kbp: the sharded lookup kernel at nt = 15 + the one-wave and the block-per-read
resolve kernels on the merged slots).  The oracle cannot hold that database on the host; it runs against the sub-database of
every pair whose k-mer occurs in the sampled reads: a k-mer outside it is a miss in both, so its answers for these
reads are those of the full database (the GPU side always searches the full shards).  Own module: the build needs
most of the 288 GB."""
import gc

import numpy as np
import pytest

from krakenuniq_amd import capi, synth_torch
from oracle import ku_oracle as ko
from test_gpu_fullsize import K, L, assert_counts_equal_oracle, host_cores, oracle_db_from_device

pytestmark = pytest.mark.gpu


def test_configs2_shape_36GB_nt15_shard_merge_vs_oracle_sample():
    import torch
    dev = torch.device("cuda:0")
    gc.collect()
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info()
    assert free_b > 230 * 2 ** 30, f"only {free_b >> 30} GiB of HBM free: something of an earlier test module is still resident"
    NT2, S2, G2 = 15, 12_000, 310_000
    n_bins = 4 ** NT2
    split = int(0.815 * n_bins)  # the scrambled bin keys are spread evenly: ~81.5 % of the pairs in the first shard
    n_reads, n_sample = 2_000_000, 200_000
    w = {"torch": torch, "dev": dev}
    merged, ctxs, counts = None, [], []
    seqs = off = lens = None
    sub_k, sub_v = [], []
    ctax = otax = values = None
    # the slot table has to cover both shards before either is finalised: all distinct values = all species and their
    # ancestors that occur as LCA values -- take every taxid of the taxonomy that can be a database value
    for si, (lo, hi) in enumerate(((0, split), (split, n_bins))):
        sh = synth_torch.BenchDb(dev, n_species=S2, genome_len=G2, k=K, nt=NT2, seed=9, bin_lo=lo, bin_hi=hi)
        if si == 0:
            assert sh.n_pairs * 12 >= 36 * 10 ** 9, sh.n_pairs  # the 36 GB shard of the 300 GB layout
            ids, par = sh.tax.arrays()
            ctax = capi.Tax(ids=ids, parents=par)
            otax = ko.Tax(ids=ids, parents=par)
            values = np.unique(np.asarray(ids, dtype=np.uint32))
            values = values[values != 0]
            seqs, off, lens, _ = sh.sample_reads(n_reads, L, seed=5)
            seqs = seqs.reshape(-1)
            # canonical k-mers of the sampled reads (N -> A: a superset of the unambiguous ones is fine)
            rows = seqs.view(n_reads, L + 1)[:n_sample, :L]
            codes = ((rows >> 1) ^ (rows >> 2)) & 3
            q = [torch.unique(synth_torch.canonical(synth_torch.kmers_of_rows(codes, K).reshape(-1), K))]
            # configs[3] / configs[4] shapes against the same shards
            ps, po, pl = sh.sample_pairs(40_000, L, seed=6)
            ls, lo_, ll, _ = sh.sample_reads(600, 10_000, seed=8)
            extra = [("pairs", ps, po, pl, 40_000, 2 * L + 1), ("long", ls, lo_, ll, 600, 10_000)]
            for _, es, _, _, en, eL in extra:
                r2 = es.view(en, eL + 1)[:, :eL]
                c2 = ((r2 >> 1) ^ (r2 >> 2)) & 3
                q.append(torch.unique(synth_torch.canonical(synth_torch.kmers_of_rows(c2, K).reshape(-1), K)))
            q = torch.unique(torch.cat(q))
            w["q"] = q
            w["extra"] = extra
            w["extra_merged"] = [None] * len(extra)
        # pairs of this shard whose k-mer occurs in the sample
        q = w["q"]
        for c0 in range(0, sh.n_pairs, 1 << 28):  # torch's index kernels stop at 2^31 elements
            kc, vc = sh.kmers[c0:c0 + (1 << 28)], sh.vals[c0:c0 + (1 << 28)]
            pos = torch.searchsorted(q, kc).clamp_(max=q.numel() - 1)
            hit = q[pos] == kc
            sub_k.append(kc[hit])
            sub_v.append(vc[hit])
            del pos, hit
        sh.kmers = sh.vals = sh.genomes = None
        torch.cuda.empty_cache()
        c = capi.Ctx(0)
        c.adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT2, 2, lo, hi, keep=sh)
        c.set_taxonomy(ctax, values)
        assert c.db_layout()["hash"]
        sh.pairs = None  # the probe table replaced the pairs
        torch.cuda.empty_cache()
        ss = seqs[:n_sample * (L + 1)]
        t = torch.zeros(ss.numel(), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        c.lookup_device(ss.data_ptr(), ss.numel(), t.data_ptr(), flags=capi.KU_F_KEEP_SLOTS)
        c.synchronize()
        merged = t if merged is None else torch.maximum(merged, t)
        for ei, (_, es, _, _, _, _) in enumerate(w["extra"]):  # booked into the same per-taxon state as the 150 bp sample
            te = torch.zeros(es.numel(), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            c.lookup_device(es.data_ptr(), es.numel(), te.data_ptr(), flags=capi.KU_F_KEEP_SLOTS)
            c.synchronize()
            w["extra_merged"][ei] = te if w["extra_merged"][ei] is None else torch.maximum(w["extra_merged"][ei], te)
        counts.append(c.counts())
        # the full 2 M-read batch through the shard as well (size, not compared with the oracle)
        big = torch.zeros(seqs.numel(), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        c.lookup_device(seqs.data_ptr(), seqs.numel(), big.data_ptr(), flags=capi.KU_F_KEEP_SLOTS | capi.KU_F_NO_COUNTS)
        c.synchronize()
        assert torch.equal(big[:ss.numel()], t)
        del big
        if si == 0:
            ctxs.append(c)  # resolves the merge below; its table stays resident
        else:
            c.close()
        torch.cuda.empty_cache()
    calls = torch.zeros(n_sample, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctxs[0].reset_counts()
    ctxs[0].resolve_device(seqs.data_ptr(), off[:n_sample].data_ptr(), lens[:n_sample].data_ptr(), n_sample,
                           calls.data_ptr(), merged.data_ptr(), max_read_len=L)
    ctxs[0].synchronize()
    extra_calls = []
    for (_, es, eo, el, en, eL), em in zip(w["extra"], w["extra_merged"]):
        ce = torch.zeros(en, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctxs[0].resolve_device(es.data_ptr(), eo.data_ptr(), el.data_ptr(), en, ce.data_ptr(), em.data_ptr(), max_read_len=eL)
        ctxs[0].synchronize()
        extra_calls.append(ce)
    n_reads_state = ctxs[0].counts()
    # oracle on the sub-database: a bin lives in exactly one shard and the shards were taken in bin order, so the
    # concatenation is in KrakenDB order already; only the index is rebuilt
    km = torch.cat(sub_k)
    vv = torch.cat(sub_v)
    b = synth_torch.bin_key(km, K, NT2)
    assert bool((b[1:] >= b[:-1]).all())
    offs = torch.zeros(n_bins + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(b, minlength=n_bins), 0, out=offs[1:])
    del b
    odb, keep = oracle_db_from_device(torch, km, vv, offs, K, NT2)
    del offs
    torch.cuda.empty_cache()
    run = ko.Run(odb, otax, threads=host_cores())
    ss = seqs[:n_sample * (L + 1)]
    res = run.classify_packed(ss.cpu().numpy(), off[:n_sample].cpu().numpy().astype(np.uint64),
                              lens[:n_sample].cpu().numpy().astype(np.uint32))
    nk = L - K + 1
    assert np.array_equal(calls.cpu().numpy().view(np.uint32), res["calls"])
    want = res["taxa"].reshape(n_sample, nk).copy()
    want[res["ambig"].reshape(n_sample, nk) != 0] = capi.KU_AMBIG
    got = merged.view(n_sample, L + 1)[:, :nk].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    for (name, es, eo, el, en, eL), em, ce in zip(w["extra"], w["extra_merged"], extra_calls):
        re_ = run.classify_packed(es.cpu().numpy(), eo.cpu().numpy().astype(np.uint64), el.cpu().numpy().astype(np.uint32))
        nke = eL - K + 1
        assert np.array_equal(ce.cpu().numpy().view(np.uint32), re_["calls"]), name
        we = re_["taxa"].reshape(en, nke).copy()
        we[re_["ambig"].reshape(en, nke) != 0] = capi.KU_AMBIG
        assert np.array_equal(em.view(en, eL + 1)[:, :nke].cpu().numpy().view(np.uint32), we), name
        assert int((ce != 0).sum()) > en // 2
    tot = dict(counts[0])
    tot["registers"] = np.maximum(counts[0]["registers"], counts[1]["registers"])
    tot["n_kmers"] = counts[0]["n_kmers"] + counts[1]["n_kmers"]
    tot["n_reads"] = n_reads_state["n_reads"]
    assert_counts_equal_oracle(tot, run)
    assert int((calls != 0).sum()) > n_sample // 2
    ctxs[0].close()
