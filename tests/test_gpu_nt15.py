"""-m gpu: the fused wave-per-read kernel on the STANDARD database geometry (k = 31, minimizer nt = 15) and on nt = 14.

Round 2 left three instances of ku_classify_short_kernel without an oracle comparison: the windowed
one at nt = 15 (<2,true,31,15,true>: mate pairs, long reads), the three-k-mers-per-lane one at nt = 15
(<3,true,31,15,false>: reads of 129-192 k-mers) and the generic windowed instance at nt >= 14 (30-bit minimizer values, 26-bit
order keys: the tie detector and the exact fallback decide the anchor more often than at nt <= 13).  A whole database with a
4^15-bin index (8.6 GB of offsets) is built in HBM by synth_torch; calls, per-k-mer codes, HLL registers, n_kmers and n_reads
are compared bit for bit with the CPU oracle for 150 bp, 200 bp, 2 x 150 + N and 1 kbp / 10 kbp reads, with ambiguous bases,
low-complexity inserts (equal minimizer keys inside a window) and the staged kernels on the same reads.
"""
import gc

import numpy as np
import pytest

from krakenuniq_amd import capi, synth_torch
from oracle import ku_oracle as ko
from test_gpu_fullsize import K, assert_counts_equal_oracle, host_cores, oracle_db_from_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[15, 14])
def geo(request):
    import torch
    nt = request.param
    dev = torch.device("cuda:0")
    db = synth_torch.BenchDb(dev, n_species=96, genome_len=60_000, k=K, nt=nt, seed=31 + nt)
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    otax = ko.Tax(ids=ids, parents=par)
    odb, keep = oracle_db_from_device(torch, db.kmers, db.vals, db.offsets, K, nt)
    db.kmers = db.vals = None
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, nt, 2, keep=db)
    ctx.set_taxonomy(ctax)
    assert ctx.db_layout()["hash"]
    w = {"torch": torch, "dev": dev, "db": db, "ctx": ctx, "odb": odb, "otax": otax, "keep": keep, "nt": nt,
         "cores": host_cores()}
    yield w
    ctx.close()
    w.clear()
    del db, odb, keep
    gc.collect()
    torch.cuda.empty_cache()


def low_complexity(torch, buf, n, stride, L, seed):
    """overwrite a stretch of some reads with short tandem repeats: equal canonical m-mers within one minimizer window
    (the packed window minimum sees ties and the wave falls back to the exact anchor scan)"""
    g = torch.Generator(device=buf.device)
    g.manual_seed(seed)
    rows = buf.view(n, stride)
    pick = torch.nonzero(torch.rand(n, generator=g, device=buf.device) < 0.08).flatten()
    units = [b"A", b"AT", b"ACG", b"AACC", b"GATTACA", b"TTTTTTTTTTTTTTTG"]
    for i, r in enumerate(pick.tolist()[:400]):
        u = units[i % len(units)]
        run = min(L, 40 + 13 * (i % 9))
        s = (i * 37) % max(1, L - run)
        rep = (u * (run // len(u) + 1))[:run]
        rows[r, s:s + run] = torch.tensor(list(rep), dtype=torch.uint8, device=buf.device)


def run_both(w, seqs, off, lens, n, L, monkeypatch=None):
    """fused path vs the oracle (everything), then the staged kernels on the same reads vs the fused results"""
    torch, ctx = w["torch"], w["ctx"]
    stride = L + 1
    taxa = torch.zeros(seqs.numel(), dtype=torch.int32, device=w["dev"])
    calls = torch.zeros(n, dtype=torch.int32, device=w["dev"])
    torch.cuda.synchronize()
    ctx.reset_counts()
    ctx.classify_batch_device(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), n, calls.data_ptr(),
                              taxa.data_ptr(), max_read_len=L)
    ctx.synchronize()
    run = ko.Run(w["odb"], w["otax"], threads=w["cores"])
    res = run.classify_packed(seqs.cpu().numpy(), off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32))
    nk = L - K + 1
    assert np.array_equal(calls.cpu().numpy().view(np.uint32), res["calls"])
    want = res["taxa"].reshape(n, nk).copy()
    want[res["ambig"].reshape(n, nk) != 0] = capi.KU_AMBIG
    got = taxa.view(n, stride)[:, :nk].cpu().numpy().view(np.uint32)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, (len(bad), bad[:8])
    counts = ctx.counts()
    assert_counts_equal_oracle(counts, run)
    assert int((calls != 0).sum()) > n // 3
    # the same reads with the kernel's own run-length encoded output (ku_classify_batch_device_rle): expanded == taxa[]
    cap = ctx.device_rle_runs_cap(seqs.numel(), n, L)
    runs = torch.zeros((cap, 2), dtype=torch.int32, device=w["dev"])
    roff = torch.zeros(n, dtype=torch.int64, device=w["dev"])
    rcnt = torch.zeros(n, dtype=torch.int32, device=w["dev"])
    nruns = torch.zeros(1, dtype=torch.int64, device=w["dev"])
    c3 = torch.zeros_like(calls)
    torch.cuda.synchronize()
    ctx.reset_counts()
    ctx.classify_batch_device_rle(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), n, c3.data_ptr(), runs.data_ptr(), cap,
                                  roff.data_ptr(), rcnt.data_ptr(), nruns.data_ptr(), max_read_len=L)
    ctx.synchronize()
    assert int(nruns.item()) <= cap and torch.equal(c3, calls)
    flat = synth_torch.expand_runs(runs, roff, rcnt, torch.full((n,), nk, dtype=torch.int64, device=w["dev"]))
    assert torch.equal(flat.view(n, nk), taxa.view(n, stride)[:, :nk])
    c = ctx.counts()
    assert all(np.array_equal(c[key], counts[key]) for key in ("n_kmers", "registers", "n_reads"))
    if monkeypatch is not None:
        monkeypatch.setenv("KU_NO_FUSED", "1")
        t2 = torch.zeros_like(taxa)
        c2 = torch.zeros_like(calls)
        ctx.reset_counts()
        ctx.classify_batch_device(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), n, c2.data_ptr(),
                                  t2.data_ptr(), max_read_len=L)
        ctx.synchronize()
        monkeypatch.delenv("KU_NO_FUSED")
        assert torch.equal(c2, calls)
        assert torch.equal(t2.view(n, stride)[:, :nk], taxa.view(n, stride)[:, :nk])
        c = ctx.counts()
        assert all(np.array_equal(c[key], counts[key]) for key in ("n_kmers", "registers", "n_reads"))


@pytest.mark.parametrize("L", [150, 200, 222])  # 2 k-mers per lane; 3 per lane (129-192 k-mers)
def test_one_pass_instances(geo, L, monkeypatch):
    w = geo
    n = 30_000
    seqs, off, lens, _ = w["db"].sample_reads(n, L, seed=100 + L, n_rate=0.003)
    low_complexity(w["torch"], seqs, n, L + 1, L, seed=L)
    run_both(w, seqs, off, lens, n, L, monkeypatch)


def test_windowed_mate_pairs(geo, monkeypatch):
    w = geo
    n = 30_000
    seqs, off, lens = w["db"].sample_pairs(n, 150, seed=7, n_rate=0.003)
    L = 301
    low_complexity(w["torch"], seqs, n, L + 1, L, seed=3)
    run_both(w, seqs, off, lens, n, L, monkeypatch)


@pytest.mark.parametrize("L,n", [(1000, 3000), (10_000, 400)])
def test_windowed_long_reads(geo, L, n, monkeypatch):
    w = geo
    seqs, off, lens, _ = w["db"].sample_reads(n, L, seed=900 + L, n_rate=0.002)
    low_complexity(w["torch"], seqs, n, L + 1, L, seed=L)
    run_both(w, seqs, off, lens, n, L, monkeypatch)


def test_configs34_shape_on_an_11GB_nt15_database_vs_oracle_sample():
    """the windowed nt = 15 instance at size: a whole 11 GB (0.93 G pairs) + 8.6 GB index database, 2 M mate pairs and
    20 k x 10 kbp reads through the fused kernel; a sample of each against the oracle run on the same database"""
    import torch
    dev = torch.device("cuda:0")
    gc.collect()
    torch.cuda.empty_cache()
    db = synth_torch.BenchDb(dev, n_species=3000, genome_len=310_000, k=K, nt=15, seed=15)
    assert db.n_pairs * 12 > 10 * 10 ** 9
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    otax = ko.Tax(ids=ids, parents=par)
    odb, keep = oracle_db_from_device(torch, db.kmers, db.vals, db.offsets, K, 15)
    db.kmers = db.vals = None
    torch.cuda.empty_cache()
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, 15, 2, keep=db)
    ctx.set_taxonomy(ctax)
    assert ctx.db_layout()["hash"]
    w = {"torch": torch, "dev": dev, "db": db, "ctx": ctx, "odb": odb, "otax": otax, "cores": host_cores()}
    try:
        for kind, n_all, n_s in (("pairs", 2_000_000, 100_000), ("long", 20_000, 1_500)):
            if kind == "pairs":
                seqs, off, lens = db.sample_pairs(n_all, 150, seed=41)
                L = 301
            else:
                L = 10_000
                seqs, off, lens, _ = db.sample_reads(n_all, L, seed=43)
            stride = L + 1
            taxa = torch.zeros(seqs.numel(), dtype=torch.int32, device=dev)
            calls = torch.zeros(n_all, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.reset_counts()
            ctx.classify_batch_device(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), n_all, calls.data_ptr(),
                                      taxa.data_ptr(), max_read_len=L)
            ctx.synchronize()
            assert int((calls != 0).sum()) > n_all // 2
            # the sample alone, fresh state, against the oracle; its results do not depend on the rest of the batch
            s = seqs[:n_s * stride]
            t_s = torch.zeros(s.numel(), dtype=torch.int32, device=dev)
            c_s = torch.zeros(n_s, dtype=torch.int32, device=dev)
            ctx.reset_counts()
            ctx.classify_batch_device(s.data_ptr(), s.numel(), off.data_ptr(), lens.data_ptr(), n_s, c_s.data_ptr(), t_s.data_ptr(),
                                      max_read_len=L)
            ctx.synchronize()
            assert torch.equal(c_s, calls[:n_s]) and torch.equal(t_s, taxa[:n_s * stride])
            run = ko.Run(odb, otax, threads=w["cores"])
            res = run.classify_packed(s.cpu().numpy(), off[:n_s].cpu().numpy().astype(np.uint64), lens[:n_s].cpu().numpy().astype(np.uint32))
            nk = L - K + 1
            assert np.array_equal(c_s.cpu().numpy().view(np.uint32), res["calls"])
            want = res["taxa"].reshape(n_s, nk).copy()
            want[res["ambig"].reshape(n_s, nk) != 0] = capi.KU_AMBIG
            assert np.array_equal(t_s.view(n_s, stride)[:, :nk].cpu().numpy().view(np.uint32), want)
            assert_counts_equal_oracle(ctx.counts(), run)
            del taxa, calls, seqs, t_s, c_s
            torch.cuda.empty_cache()
    finally:
        ctx.close()
        del db, odb, keep
        gc.collect()
        torch.cuda.empty_cache()
