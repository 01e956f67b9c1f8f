"""-m gpu, size-independent properties on a database and batch three orders of magnitude above the fixtures (20 M pairs
built in HBM, 1 M reads): independent routes through the library must agree bit for bit, without an oracle.
  fused wave-per-read kernel == flat lookup + resolve kernels;
  two minimizer-range shards merged with max == the whole database;
  reverse-complemented reads get the same calls and mirrored per-k-mer codes, and leave the same per-taxon state;
  classifying the batch in two halves leaves the same per-taxon state as classifying it at once."""
import numpy as np
import pytest

from krakenuniq_amd import capi, synth_torch

pytestmark = pytest.mark.gpu
K, NT, L, N = 31, 11, 150, 1_000_000


@pytest.fixture(scope="module")
def world():
    import torch
    dev = torch.device("cuda:0")
    db = synth_torch.BenchDb(dev, n_species=200, genome_len=100_000, k=K, nt=NT, seed=3)
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
    values = ctx.db_values()
    ctx.set_taxonomy(ctax)
    seqs, off, lens, _ = db.sample_reads(N, L, seed=5)
    torch.cuda.synchronize()
    return {"torch": torch, "dev": dev, "db": db, "ctax": ctax, "ctx": ctx, "values": values, "seqs": seqs.reshape(-1),
            "off": off, "lens": lens}


def run(w, ctx, seqs=None, n=None, first=0):
    torch = w["torch"]
    seqs = w["seqs"] if seqs is None else seqs
    n = N if n is None else n
    stride = L + 1
    s = seqs[first * stride:(first + n) * stride]
    taxa = torch.zeros(s.numel(), dtype=torch.int32, device=w["dev"])
    calls = torch.zeros(n, dtype=torch.int32, device=w["dev"])
    torch.cuda.synchronize()  # torch produced the inputs on its stream; the library works on the context's own
    ctx.classify_batch_device(s.data_ptr(), s.numel(), w["off"][:n].data_ptr(), w["lens"][:n].data_ptr(), n,
                              calls.data_ptr(), taxa.data_ptr(), max_read_len=L)
    ctx.synchronize()
    return calls, taxa


def same_counts(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("slot_taxid", "n_kmers", "registers", "node_taxid", "n_reads"))


def test_fused_and_staged_paths_agree(world, monkeypatch):
    w, torch = world, world["torch"]
    ctx = w["ctx"]
    ctx.reset_counts()
    calls_f, taxa_f = run(w, ctx)
    counts_f = ctx.counts()
    assert int((calls_f != 0).sum()) > N // 2  # the workload is not degenerate
    monkeypatch.setenv("KU_NO_FUSED", "1")
    ctx.reset_counts()
    calls_s, taxa_s = run(w, ctx)
    n_k = L - K + 1
    m = (torch.arange(taxa_f.numel(), device=w["dev"]) % (L + 1)) < n_k
    assert torch.equal(calls_f, calls_s) and torch.equal(taxa_f[m], taxa_s[m])
    assert same_counts(counts_f, ctx.counts())
    w["calls"], w["taxa"], w["counts"], w["mask"] = calls_f, taxa_f, counts_f, m


def test_two_shards_merge_to_the_whole(world):
    w, torch = world, world["torch"]
    db = w["db"]
    # split at the median bin of the resident pairs
    offs = db.offsets.cpu().numpy()
    mid = int(np.searchsorted(offs, offs[-1] // 2))
    merged, parts = None, []
    for lo, hi in ((0, mid), (mid, 4 ** NT)):
        sh = synth_torch.BenchDb(w["dev"], n_species=200, genome_len=100_000, k=K, nt=NT, seed=3, bin_lo=lo, bin_hi=hi)
        c = capi.Ctx(0)
        c.adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, lo, hi, keep=sh)
        c.set_taxonomy(w["ctax"], w["values"])
        t = torch.zeros(w["seqs"].numel(), dtype=torch.int32, device=w["dev"])
        torch.cuda.synchronize()
        c.lookup_device(w["seqs"].data_ptr(), w["seqs"].numel(), t.data_ptr(), flags=capi.KU_F_KEEP_SLOTS)
        c.synchronize()
        merged = t if merged is None else torch.maximum(merged, t)  # KU_AMBIG == -1 on both, else one side is 0
        parts.append(c)
    calls = torch.zeros(N, dtype=torch.int32, device=w["dev"])
    torch.cuda.synchronize()  # the merge ran on torch's stream
    parts[0].resolve_device(w["seqs"].data_ptr(), w["off"].data_ptr(), w["lens"].data_ptr(), N, calls.data_ptr(),
                            merged.data_ptr(), max_read_len=L)
    parts[0].synchronize()
    assert torch.equal(calls, w["calls"]) and torch.equal(merged[w["mask"]], w["taxa"][w["mask"]])
    cs = [c.counts() for c in parts]
    tot = dict(cs[0])
    tot["registers"] = np.maximum(cs[0]["registers"], cs[1]["registers"])
    tot["n_kmers"] = cs[0]["n_kmers"] + cs[1]["n_kmers"]
    tot["n_reads"] = cs[0]["n_reads"] + cs[1]["n_reads"]
    assert same_counts(tot, w["counts"])


def test_reverse_complement_invariance(world):
    w, torch = world, world["torch"]
    rows = w["seqs"].view(N, L + 1)
    comp = torch.arange(256, dtype=torch.uint8, device=w["dev"])
    for a, b in ((65, 84), (67, 71), (71, 67), (84, 65)):
        comp[a] = b
    rc = rows.clone()
    rc[:, :L] = comp[rows[:, :L].flip(1).long()]
    ctx = w["ctx"]
    ctx.reset_counts()
    calls, taxa = run(w, ctx, seqs=rc.reshape(-1))
    n_k = L - K + 1
    assert torch.equal(calls, w["calls"])
    assert torch.equal(taxa.view(N, L + 1)[:, :n_k].flip(1), w["taxa"].view(N, L + 1)[:, :n_k])
    assert same_counts(ctx.counts(), w["counts"])  # the canonical k-mers, hence the sketches, are the same set


def test_batches_accumulate(world):
    w = world
    ctx = w["ctx"]
    ctx.reset_counts()
    run(w, ctx, n=N // 2)
    run(w, ctx, n=N - N // 2, first=N // 2)
    assert same_counts(ctx.counts(), w["counts"])
