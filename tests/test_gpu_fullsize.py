"""-m gpu, BASELINE.json sizes on one MI355X.

configs[1] at full size: the 8 GB MiniKraken-style database (k = 31, nt = 13, ~0.61 G pairs, 2000 species; the bench
database) and 10 M x 150 bp reads.
  * a 1 M-read sample classified in a FRESH context is compared with the CPU oracle (oracle/ku_oracle.c, all host cores)
    run against the same database: calls, per-k-mer codes, HLL registers, n_kmers, n_reads -- bit for bit;
  * the whole 10 M batch goes through the size-independent properties: fused == staged kernels, two minimizer-range
    shards merged with max == the whole database, reverse-complement invariance, batch accumulation.
Single-GPU proxies of the 8-GPU configurations, each against the oracle on a sample:
  * configs[3] shape: 5 M mate pairs 2 x 150 joined with 'N' (scripts/read_merger.pl:187-191);
  * configs[4] shape: 100 k reads x 10 kbp;
  * configs[2] shape (36 GB nt = 15 shard): tests/test_gpu_shard36.py (its own module: it needs most of the HBM).
"""
import os

import numpy as np
import pytest

from krakenuniq_amd import capi, synth, synth_torch
from oracle import ku_oracle as ko

pytestmark = pytest.mark.gpu
K = 31


def host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def oracle_db_from_device(torch, kmers, vals, offsets, k, nt):
    """host KrakenDB (12-byte pairs + offsets) from bin-ordered device k-mers / taxids"""
    n = kmers.numel()
    pairs = torch.empty((n, 3), dtype=torch.int32, device=kmers.device)
    pairs[:, 0] = ((kmers << 32) >> 32).to(torch.int32)
    pairs[:, 1] = (kmers >> 32).to(torch.int32)
    pairs[:, 2] = ((vals << 32) >> 32).to(torch.int32)
    hp = pairs.cpu().numpy().view(np.uint8).reshape(-1)
    del pairs
    ho = offsets.cpu().numpy().view(np.uint64)
    return ko.Db(pairs=hp, key_ct=n, k=k, offsets=ho, nt=nt), (hp, ho)


def same_counts(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("slot_taxid", "n_kmers", "registers", "node_taxid", "n_reads"))


def assert_counts_equal_oracle(counts, run):
    """per-taxon state of a context vs the oracle's run: n_reads, n_kmers, dense HLL registers, bit for bit"""
    want = run.counts()
    got_r = {int(t): int(n) for t, n in zip(counts["node_taxid"], counts["n_reads"]) if n}
    assert got_r == {t: c["n_reads"] for t, c in want.items() if c["n_reads"]}
    got_k = {int(t): int(n) for t, n in zip(counts["slot_taxid"], counts["n_kmers"]) if n}
    assert got_k == {t: c["n_kmers"] for t, c in want.items() if c["n_kmers"]}
    idx = {int(t): i for i, t in enumerate(counts["slot_taxid"])}
    n_checked = 0
    for t, c in want.items():
        if c["n_kmers"]:
            assert (counts["registers"][idx[t]] == c["sketch"].registers()).all(), t
            n_checked += 1
    assert n_checked > 0
    unused = counts["n_kmers"] == 0
    assert not counts["registers"][unused].any()


def gpu_classify(w, ctx, seqs, off, lens, n, max_len):
    torch = w["torch"]
    taxa = torch.zeros(seqs.numel(), dtype=torch.int32, device=w["dev"])
    calls = torch.zeros(n, dtype=torch.int32, device=w["dev"])
    torch.cuda.synchronize()  # torch produced the inputs on its stream; the library works on the context's own
    ctx.classify_batch_device(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), n, calls.data_ptr(),
                              taxa.data_ptr(), max_read_len=max_len)
    ctx.synchronize()
    return calls, taxa


def compare_with_oracle(w, odb, otax, ctx, seqs, n, L, cores):
    """n fixed-length reads (stride L + 1): GPU results of a fresh run of `ctx` vs the oracle on the same reads"""
    torch = w["torch"]
    stride = L + 1
    s = seqs[:n * stride]
    off = torch.arange(n, device=w["dev"], dtype=torch.int64) * stride
    lens = torch.full((n,), L, dtype=torch.int32, device=w["dev"])
    ctx.reset_counts()
    calls, taxa = gpu_classify(w, ctx, s, off, lens, n, L)
    run = ko.Run(odb, otax, threads=cores)
    host = s.cpu().numpy()
    res = run.classify_packed(host, off.cpu().numpy().astype(np.uint64), lens.cpu().numpy().astype(np.uint32))
    nk = L - K + 1
    assert np.array_equal(calls.cpu().numpy().view(np.uint32), res["calls"])
    want = res["taxa"].reshape(n, nk).copy()
    want[res["ambig"].reshape(n, nk) != 0] = capi.KU_AMBIG
    got = taxa.view(n, stride)[:, :nk].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    assert_counts_equal_oracle(ctx.counts(), run)
    return calls, taxa


# ---------------------------------------------------------------------------- configs[1]: 8 GB database
NT, L, N = 13, 150, 10_000_000
SPECIES, GLEN = 2000, 310_000


@pytest.fixture(scope="module")
def world():
    import torch
    dev = torch.device("cuda:0")
    db = synth_torch.BenchDb(dev, n_species=SPECIES, genome_len=GLEN, k=K, nt=NT, seed=7)
    assert db.n_pairs > 600_000_000  # the 8 GB configuration (7.3 GB of pairs + 0.54 GB of index)
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    otax = ko.Tax(ids=ids, parents=par)
    odb, keep = oracle_db_from_device(torch, db.kmers, db.vals, db.offsets, K, NT)
    db.kmers = db.vals = None
    torch.cuda.empty_cache()
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
    values = ctx.db_values()
    ctx.set_taxonomy(ctax)
    assert ctx.db_layout()["hash"]
    seqs, off, lens, _ = db.sample_reads(N, L, seed=1)
    torch.cuda.synchronize()
    w = {"torch": torch, "dev": dev, "db": db, "ctax": ctax, "otax": otax, "odb": odb, "keep": keep, "ctx": ctx,
         "values": values, "seqs": seqs.reshape(-1), "off": off, "lens": lens, "cores": host_cores()}
    yield w
    ctx.close()  # the 44 GB probe table and the batch go back before the next module starts
    w.clear()
    del db, seqs, off, lens
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_configs1_sample_of_1M_reads_vs_oracle(world):
    w = world
    compare_with_oracle(w, w["odb"], w["otax"], w["ctx"], w["seqs"], 1_000_000, L, w["cores"])


def test_configs1_full_batch_fused_equals_staged(world, monkeypatch):
    w, torch = world, world["torch"]
    ctx = w["ctx"]
    ctx.reset_counts()
    calls_f, taxa_f = gpu_classify(w, ctx, w["seqs"], w["off"], w["lens"], N, L)
    counts_f = ctx.counts()
    assert int((calls_f != 0).sum()) > N // 2
    monkeypatch.setenv("KU_NO_FUSED", "1")
    ctx.reset_counts()
    calls_s, taxa_s = gpu_classify(w, ctx, w["seqs"], w["off"], w["lens"], N, L)
    nk = L - K + 1
    assert torch.equal(calls_f, calls_s)
    assert torch.equal(taxa_f.view(N, L + 1)[:, :nk], taxa_s.view(N, L + 1)[:, :nk])
    assert same_counts(counts_f, ctx.counts())
    del taxa_s, calls_s
    w["calls"], w["taxa"], w["counts"] = calls_f, taxa_f, counts_f


def test_configs1_full_batch_runs_from_the_kernel_equal_the_per_kmer_array(world):
    """ku_classify_batch_device_rle (what bench.py times): device buffers in, the fused kernel's own run-length encoded
    codes out -- expanded again they are the per-k-mer array of ku_classify_batch_device for all 10 M reads; calls and the
    per-taxon state are the same; a run array that is too small is reported through the run total, calls stay right"""
    torch, ctx, dev = world["torch"], world["ctx"], world["dev"]
    seqs, off, lens = world["seqs"], world["off"], world["lens"]
    stride, nk = L + 1, L - K + 1
    ctx.reset_counts()
    calls, taxa = gpu_classify(world, ctx, seqs, off, lens, N, L)
    want_counts = ctx.counts()
    cap = ctx.device_rle_runs_cap(seqs.numel(), N, L)
    runs = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
    roff = torch.zeros(N, dtype=torch.int64, device=dev)
    rcnt = torch.zeros(N, dtype=torch.int32, device=dev)
    nruns = torch.zeros(1, dtype=torch.int64, device=dev)
    calls2 = torch.zeros(N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.reset_counts()
    ctx.classify_batch_device_rle(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), N, calls2.data_ptr(), runs.data_ptr(), cap,
                                  roff.data_ptr(), rcnt.data_ptr(), nruns.data_ptr(), max_read_len=L)
    ctx.synchronize()
    assert 0 < int(nruns.item()) <= cap
    assert torch.equal(calls2, calls)
    flat = synth_torch.expand_runs(runs, roff, rcnt, torch.full((N,), nk, dtype=torch.int64, device=dev))
    assert torch.equal(flat.view(N, nk), taxa.view(N, stride)[:, :nk])
    assert int(rcnt.sum().item()) < 4 * N  # (a few runs per read: 224 MB instead of 6 GB)
    assert same_counts(ctx.counts(), want_counts)
    del flat, taxa
    small = 50_000  # far too small for 10 M reads
    ctx.classify_batch_device_rle(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), N, calls2.data_ptr(), runs.data_ptr(), small,
                                  roff.data_ptr(), rcnt.data_ptr(), nruns.data_ptr(), max_read_len=L, flags=capi.KU_F_NO_COUNTS)
    ctx.synchronize()
    assert int(nruns.item()) > small and torch.equal(calls2, calls)
    with pytest.raises(capi.KuError):  # the longest read must be named
        ctx.classify_batch_device_rle(seqs.data_ptr(), seqs.numel(), off.data_ptr(), lens.data_ptr(), N, calls2.data_ptr(), runs.data_ptr(), cap,
                                      roff.data_ptr(), rcnt.data_ptr(), nruns.data_ptr(), max_read_len=0)
    ctx.reset_counts()


def test_configs1_full_batch_batches_accumulate(world):
    w = world
    ctx = w["ctx"]
    ctx.reset_counts()
    h = N // 2
    stride = L + 1
    c0, t0 = gpu_classify(w, ctx, w["seqs"][:h * stride], w["off"][:h], w["lens"][:h], h, L)
    c1, t1 = gpu_classify(w, ctx, w["seqs"][h * stride:], w["off"][:N - h], w["lens"][:N - h], N - h, L)
    assert same_counts(ctx.counts(), w["counts"])
    torch = w["torch"]
    assert torch.equal(torch.cat([c0, c1]), w["calls"])


def test_configs1_full_batch_reverse_complement_invariance(world):
    w, torch = world, world["torch"]
    rows = w["seqs"].view(N, L + 1)
    comp = torch.arange(256, dtype=torch.uint8, device=w["dev"])
    for a, b in ((65, 84), (67, 71), (71, 67), (84, 65)):
        comp[a] = b
    rc = rows.clone()
    rc[:, :L] = comp[rows[:, :L].flip(1).long()]
    ctx = w["ctx"]
    ctx.reset_counts()
    calls, taxa = gpu_classify(w, ctx, rc.reshape(-1), w["off"], w["lens"], N, L)
    nk = L - K + 1
    assert torch.equal(calls, w["calls"])
    assert torch.equal(taxa.view(N, L + 1)[:, :nk].flip(1), w["taxa"].view(N, L + 1)[:, :nk])
    assert same_counts(ctx.counts(), w["counts"])


def test_configs1_full_batch_two_shards_merge_to_the_whole(world):
    w, torch = world, world["torch"]
    offs = w["db"].offsets
    mid = int(torch.searchsorted(offs, offs[-1] // 2).item())
    merged, parts = None, []
    for lo, hi in ((0, mid), (mid, 4 ** NT)):
        sh = synth_torch.BenchDb(w["dev"], n_species=SPECIES, genome_len=GLEN, k=K, nt=NT, seed=7, bin_lo=lo, bin_hi=hi)
        sh.kmers = sh.vals = sh.genomes = None
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
    torch.cuda.synchronize()
    parts[0].resolve_device(w["seqs"].data_ptr(), w["off"].data_ptr(), w["lens"].data_ptr(), N, calls.data_ptr(),
                            merged.data_ptr(), max_read_len=L)
    parts[0].synchronize()
    nk = L - K + 1
    assert torch.equal(calls, w["calls"])
    assert torch.equal(merged.view(N, L + 1)[:, :nk], w["taxa"].view(N, L + 1)[:, :nk])
    cs = [c.counts() for c in parts]
    tot = dict(cs[0])
    tot["registers"] = np.maximum(cs[0]["registers"], cs[1]["registers"])
    tot["n_kmers"] = cs[0]["n_kmers"] + cs[1]["n_kmers"]
    tot["n_reads"] = cs[0]["n_reads"] + cs[1]["n_reads"]
    assert same_counts(tot, w["counts"])
    for c in parts:
        c.close()
    w.pop("taxa", None)
    w.pop("calls", None)
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------- configs[3] shape: mate pairs
def test_configs3_shape_5M_pairs_vs_oracle_sample(world):
    w, torch = world, world["torch"]
    db, dev = w["db"], w["dev"]
    NP = 5_000_000
    m1, _, _, _ = db.sample_reads(NP, L, seed=11)
    m2, _, _, _ = db.sample_reads(NP, L, seed=1011)
    LM = 2 * L + 1
    merged = torch.empty((NP, LM + 1), dtype=torch.uint8, device=dev)
    merged[:, :L] = m1.view(NP, L + 1)[:, :L]
    merged[:, L] = 78  # 'N'
    merged[:, L + 1:LM] = m2.view(NP, L + 1)[:, :L]
    merged[:, LM] = 10
    del m1, m2
    seqs = merged.reshape(-1)
    off = torch.arange(NP, device=dev, dtype=torch.int64) * (LM + 1)
    lens = torch.full((NP,), LM, dtype=torch.int32, device=dev)
    ctx = w["ctx"]
    ctx.reset_counts()
    calls_all, taxa_all = gpu_classify(w, ctx, seqs, off, lens, NP, LM)
    ns = 250_000
    calls_s, taxa_s = compare_with_oracle(w, w["odb"], w["otax"], ctx, seqs, ns, LM, w["cores"])
    # the sample's results do not depend on what else is in the batch
    assert torch.equal(calls_s, calls_all[:ns]) and torch.equal(taxa_s, taxa_all[:ns * (LM + 1)])
    # every pair carries the 31 ambiguous k-mers that span the joining N
    nk = LM - K + 1
    assert bool((taxa_all.view(NP, LM + 1)[:, L - K + 1:L + 1] == -1).all())
    assert int((calls_all != 0).sum()) > NP // 2 and nk == 271


# ---------------------------------------------------------------------------- configs[4] shape: long reads
def test_configs4_shape_100k_reads_of_10kbp_vs_oracle_sample(world):
    w, torch = world, world["torch"]
    db = w["db"]
    NR, LL = 100_000, 10_000
    seqs, off, lens, _ = db.sample_reads(NR, LL, seed=21)
    seqs = seqs.reshape(-1)
    ctx = w["ctx"]
    ctx.reset_counts()
    calls_all, taxa_all = gpu_classify(w, ctx, seqs, off, lens, NR, LL)
    ns = 3_000
    calls_s, taxa_s = compare_with_oracle(w, w["odb"], w["otax"], ctx, seqs, ns, LL, w["cores"])
    assert torch.equal(calls_s, calls_all[:ns]) and torch.equal(taxa_s, taxa_all[:ns * (LL + 1)])
    assert int((calls_all != 0).sum()) > NR // 2


# ---------------------------------------------------------------------------- configs[1]: the report at BASELINE size
def sparse_pairs_of_oracle(run, slot_of):
    """(slot << 32 | encoded hash) of every taxon whose oracle sketch stayed sparse, ascending; and the sparse flags"""
    parts, sparse_slots, dense_slots = [], [], []
    for t, c in run.counts().items():
        if not c["n_kmers"]:
            continue
        s = slot_of[t]
        if c["sparse"]:
            sparse_slots.append(s)
            parts.append((np.uint64(s) << np.uint64(32)) | c["sketch"].sparse_list().astype(np.uint64))
        else:
            dense_slots.append(s)
    allp = np.sort(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.uint64)
    return allp, sparse_slots, dense_slots


@pytest.mark.parametrize("unit,n,glog2", [(500000, 1_000_000, 20), (60000, 300_000, 0)])
def test_configs1_report_with_sparse_sketches_vs_oracle(world, tmp_path, unit, n, glog2):
    """ku_ctx_enable_sparse + ku_ctx_report on the 8 GB database against the oracle's report (reference sketch semantics:
    sparse sets per work unit, dense switch at 1024): row for row, plus every taxon's sparse / dense state and encoded
    set.  Ten species are over-represented so that their sketches turn dense while ~2000 others stay sparse; the run-wide
    set starts at 2^20 cells (first case) and has to move to larger tables several times (ku_sparse_rehash_kernel)."""
    w, torch = world, world["torch"]
    db, ctx, dev = w["db"], w["ctx"], w["dev"]
    hot = torch.arange(0, 10, device=dev, dtype=torch.int64) * 37 % db.n_species
    n_hot = n // 6
    s_u, _, _, _ = db.sample_reads(n - n_hot, L, seed=77)
    s_h, _, _, _ = db.sample_reads(n_hot, L, seed=78, species=hot)
    rows = torch.cat([s_u.view(-1, L + 1), s_h.view(-1, L + 1)])
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    rows = rows[torch.randperm(n, generator=g, device=dev)]
    host = rows.reshape(-1).cpu().numpy()
    del rows, s_u, s_h
    off = np.arange(n, dtype=np.uint64) * (L + 1)
    lens = np.full(n, L, dtype=np.uint32)
    ctx.enable_sparse(unit, glog2)
    cuts = [0, n // 7, n // 2 + 13, n]  # uneven host batches: work units straddle them
    calls = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r = ctx.classify_batch_rle(host[a * (L + 1):b * (L + 1)], off[:b - a], lens[:b - a])
        calls.append(r["calls"])
    assert ctx.sparse_state() == 1
    run = ko.Run(w["odb"], w["otax"], work_unit_nt=unit, threads=w["cores"])
    res = run.classify_packed(host, off, lens, want_taxa=False)
    assert np.array_equal(np.concatenate(calls), res["calls"])
    counts = ctx.counts()
    assert_counts_equal_oracle(counts, run)
    flags, pairs = ctx.sparse_export()
    slot_of = {int(t): s for s, t in enumerate(counts["slot_taxid"])}
    want_pairs, sparse_slots, dense_slots = sparse_pairs_of_oracle(run, slot_of)
    assert len(dense_slots) >= 5 and len(sparse_slots) > 1000, (len(dense_slots), len(sparse_slots))
    assert all(flags[s] == 1 for s in sparse_slots) and all(flags[s] == 0 for s in dense_slots)
    assert np.array_equal(np.sort(pairs), want_pairs)
    if glog2:
        assert len(pairs) > 8 * (1 << glog2)  # the set outgrew its first table several times over
    taxdb = str(tmp_path / "taxDB")
    db.tax.write(taxdb)
    rtax = capi.Tax(taxdb)
    got = sorted(ctx.report(rtax).strip("\n").split("\n"))
    want = sorted(run.report(taxdb).strip("\n").split("\n"))
    if got != want and n <= 300_000:  # which side of the roll-up differs: the host's from the exported state, or the device's
        host = sorted(capi.report_sparse(rtax, counts, flags, pairs, []).strip("\n").split("\n"))
        dh = [(a, b) for a, b in zip(host, want) if a != b]
        dd = [(a, b) for a, b in zip(got, want) if a != b]
        raise AssertionError(f"host roll-up vs oracle: {len(dh)} rows differ {dh[:3]}; device roll-up vs oracle: {len(dd)} rows differ {dd[:3]}")
    assert got == want, [(a, b) for a, b in zip(got, want) if a != b][:5]
    ctx.disable_sparse()  # the shared context goes on without the emulation's tables
    ctx.reset_counts()
