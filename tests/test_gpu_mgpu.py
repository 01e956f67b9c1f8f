"""-m gpu: the C++ multi-GPU driver (ku_mgpu, krakenuniq_amd/csrc/ku_mgpu.cpp) through the C ABI.

A 1-GPU box runs several ranks on its one device (same-process exchange: device copies + merge kernels); boxes with
more devices also take the RCCL exchange.  Every variant has to reproduce the single-context results: the reference's
Kraken output byte for byte, and -- after ku_mgpu_reduce_state -- the per-taxon state (HLL registers, n_kmers,
n_reads) bit for bit, i.e. k-mers are accounted exactly once (by the rank that owns their minimizer bin, misses under
taxon 0 included)."""
import os
import subprocess

import numpy as np
import pytest

from krakenuniq_amd import capi, synth, synth_torch
from oracle import ku_oracle as ko
import gpu_common as gc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F1 = os.path.join(ROOT, "tests", "golden", "f1")
BIN = os.path.join(ROOT, "krakenuniq_amd", "bin", "classify")
K = 31


def device_lists():
    n = capi.lib().ku_device_count()
    out = [[0, 0], [0, 0, 0]]
    if n >= 2:
        out.append(list(range(min(n, 4))))
    return out


@pytest.fixture(scope="module")
def f1():
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    ctx, cdb, ctax = gc.make_ctx(F1)
    rle = ctx.classify_batch_rle(buf, off, lens)
    want_counts = ctx.counts()
    return {"ids": ids, "seqs": seqs, "buf": buf, "off": off, "lens": lens, "counts": want_counts, "cdb": cdb,
            "ctax": ctax, "text": open(f"{F1}/out.tsv").read()}


def same_counts(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("slot_taxid", "n_kmers", "registers", "node_taxid", "n_reads"))


@pytest.mark.parametrize("devices", device_lists())
@pytest.mark.parametrize("flags,exchange", [(0, "route"), (0, "route_tight"), (0, "slots"), (capi.KU_MGPU_REPLICAS, None)])
def test_group_reproduces_reference_output_and_state(f1, devices, flags, exchange, monkeypatch):
    if exchange == "route_tight":  # queues far too small for their totals: the second scanning pass, sized by the first
        monkeypatch.setenv("KU_ROUTE_CAP", "128")
        exchange = "route"
    if exchange == "slots":
        monkeypatch.setenv("KU_MGPU_EXCHANGE", "slots")
    mg = capi.Mgpu(devices, flags=flags)
    mg.load(f1["cdb"], f1["ctax"])
    assert mg.uses_routing() == (exchange == "route")
    # two batches: the state accumulates in the ranks' contexts across batches
    n = len(f1["lens"])
    h = n // 3
    cut = int(f1["off"][h])
    a = mg.classify_batch_rle(f1["buf"][:cut], f1["off"][:h], f1["lens"][:h])
    b = mg.classify_batch_rle(f1["buf"][cut:], f1["off"][h:] - cut, f1["lens"][h:])
    text = capi.format_kraken_rle(f1["buf"][:cut], f1["off"][:h], f1["lens"][:h], f1["ids"][:h], K, a)
    text += capi.format_kraken_rle(f1["buf"][cut:], f1["off"][h:] - cut, f1["lens"][h:], f1["ids"][h:], K, b)
    assert text == f1["text"]
    mg.reduce_state()
    for i in range(len(devices)):  # every rank ends up with the whole run's state
        assert same_counts(mg.ctx(i).counts(), f1["counts"]), i
    if not flags:
        t, c = mg.count_taxons()
        want = dict(tuple(map(int, ln.split("\t"))) for ln in open(f"{F1}/database.kdb.counts").read().split("\n") if ln)
        assert dict(zip(t.tolist(), c.tolist())) == want
    mg.close()


@pytest.mark.parametrize("devices", device_lists()[:2])
def test_group_quick_mode(f1, devices):
    mg = capi.Mgpu(devices)
    mg.load(f1["cdb"], f1["ctax"])
    q = mg.classify_batch_rle(f1["buf"], f1["off"], f1["lens"], flags=capi.KU_F_QUICK, min_hits=2)
    assert capi.format_kraken_rle(f1["buf"], f1["off"], f1["lens"], f1["ids"], K, q, flags=capi.KU_P_QUICK) == \
        open(f"{F1}/out_quick.tsv").read()
    mg.reduce_state()
    ctx, _, _ = gc.make_ctx(F1)
    ctx.classify_batch_rle(f1["buf"], f1["off"], f1["lens"], flags=capi.KU_F_QUICK, min_hits=2)
    assert same_counts(mg.ctx(0).counts(), ctx.counts())
    with pytest.raises(capi.KuError) as e:  # a second reduce would add the counters of every rank once more
        mg.reduce_state()
    assert e.value.status == -6
    assert same_counts(mg.ctx(0).counts(), ctx.counts())
    mg.close()


def test_group_of_one_rank_through_rccl(f1, monkeypatch):
    """the RCCL calls themselves (broadcast, grouped reduce, all-reduce, all-gather) on a world of one rank"""
    monkeypatch.setenv("KU_MGPU_FORCE_RCCL", "1")
    for uid in (None, capi.mgpu_unique_id()):  # ncclCommInitAll and ncclCommInitRank
        mg = capi.Mgpu([0], unique_id=uid)
        assert mg.uses_rccl()
        mg.load(f1["cdb"], f1["ctax"])
        rle = mg.classify_batch_rle(f1["buf"], f1["off"], f1["lens"])
        assert capi.format_kraken_rle(f1["buf"], f1["off"], f1["lens"], f1["ids"], K, rle) == f1["text"]
        mg.reduce_state()
        assert same_counts(mg.ctx(0).counts(), f1["counts"])
        mg.close()


def test_empty_and_tiny_batches(f1):
    mg = capi.Mgpu([0, 0, 0, 0])
    mg.load(f1["cdb"], f1["ctax"])
    e = mg.classify_batch_rle(b"", np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    assert len(e["calls"]) == 0 and len(e["runs"]) == 0
    # fewer reads than ranks: some slices are empty
    cut = int(f1["off"][2])
    r = mg.classify_batch_rle(f1["buf"][:cut], f1["off"][:2], f1["lens"][:2])
    assert capi.format_kraken_rle(f1["buf"][:cut], f1["off"][:2], f1["lens"][:2], f1["ids"][:2], K, r) == \
        "".join(f1["text"].splitlines(True)[:2])
    mg.close()


def rows(text):
    return sorted(text.strip("\n").split("\n"))


@pytest.mark.parametrize("unit,report", [(500000, "report.tsv"), (1000, "report_u1000.tsv")])
@pytest.mark.parametrize("devices,flags,exchange", [([0, 0, 0], 0, "route"), ([0, 0, 0], 0, "slots"), ([0, 0], capi.KU_MGPU_REPLICAS, None),
                                                   ([0, 0, 0, 0], capi.KU_MGPU_REPLICAS, None)])
def test_group_report_equals_the_reference(f1, devices, flags, exchange, unit, report, monkeypatch):
    if exchange == "slots":
        monkeypatch.setenv("KU_MGPU_EXCHANGE", "slots")
    """the HyperLogLog++ sparse-mode emulation over a group: host batches are cut at work-unit boundaries, every rank runs
    the emulation on whole units, the open unit moves on to rank 0, ku_mgpu_reduce_state folds the ranks' states: the
    report of rank 0's context equals the reference's row for row -- like one GPU's"""
    mg = capi.Mgpu(devices, flags=flags)
    mg.load(f1["cdb"], f1["ctax"])
    mg.enable_sparse(unit)
    n = len(f1["lens"])
    cuts = [0, n // 5, n // 5 + 3, n // 2, n]
    text = ""
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo = int(f1["off"][a])
        hi = int(f1["off"][b]) if b < n else len(f1["buf"])
        r = mg.classify_batch_rle(f1["buf"][lo:hi], f1["off"][a:b] - lo, f1["lens"][a:b])
        text += capi.format_kraken_rle(f1["buf"][lo:hi], f1["off"][a:b] - lo, f1["lens"][a:b], f1["ids"][a:b], K, r)
    assert text == f1["text"]
    assert mg.sparse_state() == 1
    mg.reduce_state()
    assert same_counts(mg.ctx(0).counts(), f1["counts"])
    assert rows(mg.ctx(0).report(f1["ctax"], [f"{F1}/database.kdb.counts"])) == rows(open(f"{F1}/{report}").read())
    mg.close()


@pytest.mark.parametrize("flags,exchange", [(0, "route"), (0, "slots"), (capi.KU_MGPU_REPLICAS, None)])
def test_group_goes_on_when_a_rank_runs_out_of_memory_for_the_emulation(f1, flags, exchange, monkeypatch):
    """a ceiling of 2^11 cells for the run-wide set (test hook) makes ranks give the emulation up in mid-run -- also the rank
    that holds the open work unit: the later batches must not stumble over a context without tables (the open unit used to be
    moved into / out of it: KU_ESTATE took the run down).  Classification is untouched, the state says 2, the report is the
    dense-register one"""
    if exchange == "slots":
        monkeypatch.setenv("KU_MGPU_EXCHANGE", "slots")
    monkeypatch.setenv("KU_SPARSE_MAX_LOG2", "11")
    mg = capi.Mgpu([0, 0, 0], flags=flags)
    mg.load(f1["cdb"], f1["ctax"])
    mg.enable_sparse(1000, 10)
    n = len(f1["lens"])
    cuts = [0, n // 7, n // 3, n // 3 + 5, n // 2, n]
    text = ""
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo = int(f1["off"][a])
        hi = int(f1["off"][b]) if b < n else len(f1["buf"])
        r = mg.classify_batch_rle(f1["buf"][lo:hi], f1["off"][a:b] - lo, f1["lens"][a:b])
        text += capi.format_kraken_rle(f1["buf"][lo:hi], f1["off"][a:b] - lo, f1["lens"][a:b], f1["ids"][a:b], K, r)
    assert text == f1["text"]
    assert mg.sparse_state() == 2
    mg.reduce_state()
    assert mg.sparse_state() == 2
    assert same_counts(mg.ctx(0).counts(), f1["counts"])
    paths = [f"{F1}/database.kdb.counts"]
    assert mg.ctx(0).report(f1["ctax"], paths) == capi.report(f1["ctax"], mg.ctx(0).counts(), paths)
    mg.close()


def test_group_sparse_state_equals_the_oracle_on_a_random_database():
    """taxa of very different abundance over three ranks, sharded and replicas: which sketches switch to dense and the exact
    sets of the ones that do not (as tests/test_gpu_sparse.py checks for one GPU)"""
    from test_gpu_sparse import assert_sparse_state_equals_oracle
    rng = np.random.default_rng(4)
    db = gc.random_db(rng, n_genomes=8, glen=6000, k=K, nt=9)
    weights = np.array([200, 60, 20, 8, 3, 1, 1, 0.3])
    weights = weights / weights.sum()
    sp = list(db["genomes"])
    seqs = []
    for _ in range(5000):
        g = db["genomes"][sp[int(rng.choice(len(sp), p=weights))]]
        L = int(rng.integers(60, 260))
        s0 = int(rng.integers(0, len(g) - L))
        seqs.append(synth.codes_to_ascii(g[s0:s0 + L]))
    buf, off, lens = ko.pack_reads(seqs)
    ids, par = db["tax"].arrays()
    raw = db["pairs"].view(np.uint8).reshape(-1)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=9)
    otax = ko.Tax(ids=ids, parents=par)
    unit = 30000
    run = ko.Run(odb, otax, work_unit_nt=unit)
    run.classify(seqs)
    for flags in (0, capi.KU_MGPU_REPLICAS):
        cdb = capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=9)
        ctax = capi.Tax(ids=ids, parents=par)
        mg = capi.Mgpu([0, 0, 0], flags=flags)
        mg.load(cdb, ctax)
        mg.enable_sparse(unit)
        n = len(seqs)
        cuts = [0, 700, 2100, 2150, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            lo = int(off[a])
            hi = int(off[b]) if b < n else len(buf)
            mg.classify_batch_rle(buf[lo:hi], off[a:b] - lo, lens[a:b])
        mg.reduce_state()
        counts, flg, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(mg.ctx(0), run)
        gc.assert_same_counts(counts, run)
        assert n_sparse > 0 and n_dense > 0
        mg.close()


def test_group_exact_counts_equal_the_reference(f1):
    """classifyExact over a sharded group: a k-mer goes into the set of the rank that owns its minimizer bin, the distinct
    counts add up: the reference's report_exact.tsv"""
    mg = capi.Mgpu([0, 0, 0])
    mg.load(f1["cdb"], f1["ctax"])
    mg.enable_exact(20)
    n = len(f1["lens"])
    h = n // 3
    cut = int(f1["off"][h])
    mg.classify_batch_rle(f1["buf"][:cut], f1["off"][:h], f1["lens"][:h])
    mg.classify_batch_rle(f1["buf"][cut:], f1["off"][h:] - cut, f1["lens"][h:])
    mg.reduce_state()
    assert same_counts(mg.ctx(0).counts(), f1["counts"])
    assert rows(mg.ctx(0).report(f1["ctax"], [f"{F1}/database.kdb.counts"])) == rows(open(f"{F1}/report_exact.tsv").read())
    mg.close()
    with pytest.raises(capi.KuError):  # replicas would see a k-mer on several ranks
        m2 = capi.Mgpu([0, 0], flags=capi.KU_MGPU_REPLICAS)
        m2.load(f1["cdb"], f1["ctax"])
        try:
            m2.enable_exact(20)
        finally:
            m2.close()


def test_group_of_replicas_with_two_databases(f1):
    """-d A -d B on several GPUs: every rank holds both databases, the first one with the k-mer wins (classify.cpp:928-936)"""
    g = os.path.join(ROOT, "tests", "golden")
    d8 = os.path.join(g, "f8")
    ids, seqs = synth.read_seqfile(f"{d8}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    for order, dirs in (("", [F1, d8]), ("_swapped", [d8, F1])):
        cdbs = [capi.Db(f"{x}/database.kdb", f"{x}/database.idx") for x in dirs]
        mg = capi.Mgpu([0, 0, 0], flags=capi.KU_MGPU_REPLICAS)
        mg.load_dbs(cdbs, f1["ctax"])
        mg.enable_sparse()
        rle = mg.classify_batch_rle(buf, off, lens)
        assert capi.format_kraken_rle(buf, off, lens, ids, K, rle) == open(f"{d8}/out{order}.tsv").read()
        mg.reduce_state()
        got = mg.ctx(0).report(f1["ctax"], [f"{x}/database.kdb.counts" for x in dirs])
        assert rows(got) == rows(open(f"{d8}/report{order}.tsv").read())
        mg.close()
    with pytest.raises(capi.KuError) as e:  # shards: a later database could not know what another rank found
        m2 = capi.Mgpu([0, 0])
        try:
            m2.load_dbs([capi.Db(f"{x}/database.kdb", f"{x}/database.idx") for x in (F1, d8)], f1["ctax"])
        finally:
            m2.close()
    assert e.value.status == -7


def test_cli_with_several_ranks(tmp_path):
    """KU_DEVICES: the executable's outputs -- Kraken file AND report, sparse sketches included -- equal the one-GPU run's,
    i.e. the reference's"""
    db = tmp_path / "db"
    db.mkdir()
    for fn in ("database.kdb", "database.idx", "taxDB"):
        (db / fn).write_bytes(open(f"{F1}/{fn}", "rb").read())
    args = ["-d", f"{db}/database.kdb", "-i", f"{db}/database.idx", "-a", f"{db}/taxDB", "-t", "4"]
    outs = {}
    for name, env in (("one", {}), ("sharded", {"KU_DEVICES": "0,0,0"}), ("replicas", {"KU_DEVICES": "0,0", "KU_MGPU_MODE": "replicas"})):
        out, rep = tmp_path / f"{name}.tsv", tmp_path / f"{name}.report"
        r = subprocess.run([BIN] + args + ["-o", str(out), "-r", str(rep), f"{F1}/reads.fq"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env={**os.environ, "KU_BATCH_NT": "65536", **env})  # several batches
        assert r.returncode == 0, r.stderr.decode()
        if name != "one":
            assert b"GPU ranks" in r.stderr
        outs[name] = (out.read_bytes(), rows(rep.read_text()))
        (db / "database.kdb.counts").unlink()  # regenerated by every run: the group sums it over the shards
    assert outs["one"][0] == open(f"{F1}/out.tsv", "rb").read()
    assert outs["one"][1] == rows(open(f"{F1}/report.tsv").read())
    for name in ("sharded", "replicas"):
        assert outs[name] == outs["one"], name
    # classifyExact on the sharded group
    exact = os.path.join(ROOT, "krakenuniq_amd", "bin", "classifyExact")
    out, rep = tmp_path / "exact.tsv", tmp_path / "exact.report"
    r = subprocess.run([exact] + args + ["-o", str(out), "-r", str(rep), f"{F1}/reads.fq"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "KU_DEVICES": "0,0,0", "KU_EXACT_LOG2": "20"})
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == open(f"{F1}/out.tsv", "rb").read()
    assert rows(rep.read_text()) == rows(open(f"{F1}/report_exact.tsv").read())


@pytest.mark.parametrize("exchange", ["route", "route_tight", "route_rounds", "slots"])
def test_device_step_matches_single_context(exchange, monkeypatch):
    """ku_mgpu_step_device (the bench path) with three ranks on one device: shards adopted from device memory, batch
    scattered (owner routing) or broadcast (position-wise exchange) from rank 0, slices resolved per rank == one context
    holding the whole database"""
    if exchange == "route_tight":
        monkeypatch.setenv("KU_ROUTE_CAP", "100000")  # (records; less than a queue gets: the second pass, sized by the first)
        exchange = "route"
    if exchange == "route_rounds":
        monkeypatch.setenv("KU_ROUTE_ROUND", "3000017")  # a slice of 15.1 M positions in six rounds, cut inside reads
        exchange = "route"
    if exchange == "slots":
        monkeypatch.setenv("KU_MGPU_EXCHANGE", "slots")
    import torch
    dev = torch.device("cuda:0")
    NT, L, N, W = 11, 150, 300_000, 3
    db = synth_torch.BenchDb(dev, n_species=100, genome_len=60_000, k=K, nt=NT, seed=3)
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    seqs, off, lens, _ = db.sample_reads(N, L, seed=5)
    seqs = seqs.reshape(-1)
    nb = seqs.numel()
    # reference: one context
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
    ctx.set_taxonomy(ctax)
    taxa1 = torch.zeros(nb, dtype=torch.int32, device=dev)
    calls1 = torch.zeros(N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.classify_batch_device(seqs.data_ptr(), nb, off.data_ptr(), lens.data_ptr(), N, calls1.data_ptr(), taxa1.data_ptr(),
                              max_read_len=L)
    ctx.synchronize()
    want = ctx.counts()
    # group: shard bounds at pair-count quantiles
    offs = db.offsets
    bounds = [0] + [int(torch.searchsorted(offs, offs[-1] * q // W).item()) for q in range(1, W)] + [4 ** NT]
    mg = capi.Mgpu([0] * W)
    shards, bufs = [], []
    for r in range(W):
        sh = synth_torch.BenchDb(dev, n_species=100, genome_len=60_000, k=K, nt=NT, seed=3, bin_lo=bounds[r], bin_hi=bounds[r + 1])
        mg.ctx(r).adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, bounds[r], bounds[r + 1])
        shards.append(sh)
    mg.set_taxonomy(ctax)
    assert mg.uses_routing() == (exchange == "route")
    stride = L + 1
    rb = [N * r // W for r in range(W + 1)]
    pb = [x * stride for x in rb]
    for r in range(W):
        b = {"seqs": seqs if r == 0 else torch.zeros(nb + 16, dtype=torch.uint8, device=dev),
             "off": off if r == 0 else torch.zeros(N, dtype=torch.int64, device=dev),
             "len": lens if r == 0 else torch.zeros(N, dtype=torch.int32, device=dev),
             "calls": torch.zeros(N, dtype=torch.int32, device=dev), "taxa": torch.zeros(nb + 16, dtype=torch.int32, device=dev)}
        bufs.append(b)
    torch.cuda.synchronize()
    mg.step_device([{"d_seqs": b["seqs"].data_ptr(), "d_seq_off": b["off"].data_ptr(), "d_seq_len": b["len"].data_ptr(),
                     "d_calls": b["calls"].data_ptr(), "d_taxa": b["taxa"].data_ptr()} for b in bufs],
                   nb, N, rb, pb, max_read_len=L)
    for r in range(W):
        mg.ctx(r).synchronize()
    nk = L - K + 1
    for r in range(W):
        lo, hi = rb[r], rb[r + 1]
        assert torch.equal(bufs[r]["calls"][lo:hi], calls1[lo:hi])
        assert torch.equal(bufs[r]["taxa"][:nb].view(N, stride)[lo:hi, :nk], taxa1.view(N, stride)[lo:hi, :nk])
    mg.reduce_state()
    assert same_counts(mg.ctx(1).counts(), want)
    mg.close()


@pytest.mark.parametrize("shape", ["pairs", "long", "long_rounds"])
def test_routed_step_with_windowed_reads_matches_single_context(shape, monkeypatch):
    """reads beyond 128 k-mers through the routed step: the resolve stage is then the WINDOWED `ROUTE` instance of the fused
    kernel (tickets -> slots window by window, hit counts across windows).  Mate pairs 2 x 150 + N and 3 kbp reads (with
    ambiguous bases), three ranks on one device, against one context that holds the whole database (whose windowed
    kernel the oracle tests pin); `long_rounds`: the same in several rounds cut at read boundaries"""
    import torch
    if shape == "long_rounds":
        monkeypatch.setenv("KU_ROUTE_ROUND", "2000003")
    dev = torch.device("cuda:0")
    NT, W = 11, 3
    db = synth_torch.BenchDb(dev, n_species=100, genome_len=60_000, k=K, nt=NT, seed=3)
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    if shape == "pairs":
        N, L = 60_000, 301
        seqs, off, lens = db.sample_pairs(N, 150, seed=9)
    else:
        N, L = 6_000, 3000
        seqs, off, lens, _ = db.sample_reads(N, L, seed=9)
    seqs = seqs.reshape(-1)
    nb = seqs.numel()
    stride = nb // N
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
    ctx.set_taxonomy(ctax)
    taxa1 = torch.zeros(nb, dtype=torch.int32, device=dev)
    calls1 = torch.zeros(N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.classify_batch_device(seqs.data_ptr(), nb, off.data_ptr(), lens.data_ptr(), N, calls1.data_ptr(), taxa1.data_ptr(), max_read_len=L)
    ctx.synchronize()
    want = ctx.counts()
    offs = db.offsets
    bounds = [0] + [int(torch.searchsorted(offs, offs[-1] * q // W).item()) for q in range(1, W)] + [4 ** NT]
    mg = capi.Mgpu([0] * W)
    shards = []
    for r in range(W):
        sh = synth_torch.BenchDb(dev, n_species=100, genome_len=60_000, k=K, nt=NT, seed=3, bin_lo=bounds[r], bin_hi=bounds[r + 1])
        mg.ctx(r).adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, bounds[r], bounds[r + 1])
        shards.append(sh)
    mg.set_taxonomy(ctax)
    assert mg.uses_routing()
    rb = [N * r // W for r in range(W + 1)]
    pb = [x * stride for x in rb]
    bufs = []
    for r in range(W):
        bufs.append({"seqs": seqs if r == 0 else torch.zeros(nb + 16, dtype=torch.uint8, device=dev),
                     "off": off if r == 0 else torch.zeros(N, dtype=torch.int64, device=dev),
                     "len": lens if r == 0 else torch.zeros(N, dtype=torch.int32, device=dev),
                     "calls": torch.zeros(N, dtype=torch.int32, device=dev), "taxa": torch.zeros(nb + 16, dtype=torch.int32, device=dev)})
    torch.cuda.synchronize()
    mg.step_device([{"d_seqs": b["seqs"].data_ptr(), "d_seq_off": b["off"].data_ptr(), "d_seq_len": b["len"].data_ptr(),
                     "d_calls": b["calls"].data_ptr(), "d_taxa": b["taxa"].data_ptr()} for b in bufs], nb, N, rb, pb, max_read_len=L)
    for r in range(W):
        mg.ctx(r).synchronize()
    nk = L - K + 1
    for r in range(W):
        lo, hi = rb[r], rb[r + 1]
        assert torch.equal(bufs[r]["calls"][lo:hi], calls1[lo:hi]), r
        assert torch.equal(bufs[r]["taxa"][:nb].view(N, stride)[lo:hi, :nk], taxa1.view(N, stride)[lo:hi, :nk]), r
    mg.reduce_state()
    assert same_counts(mg.ctx(2).counts(), want)
    mg.close()


def test_routed_rounds_are_cut_by_position_when_read_lengths_are_skewed(monkeypatch):
    """rounds of equal READ counts are not rounds of equal size when contigs are mixed with short reads -- a round
    could outgrow what a ticket can number.  3 kbp reads first, 150 bp reads behind them, rounds of 2 M positions: the
    rounds are then cut by position (here by bisection over the device array: the batch is device-resident)"""
    import torch
    monkeypatch.setenv("KU_ROUTE_ROUND", "2000003")
    dev = torch.device("cuda:0")
    NT, W = 11, 2
    db = synth_torch.BenchDb(dev, n_species=100, genome_len=60_000, k=K, nt=NT, seed=3)
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    N1, L1, N2, L2 = 3_000, 3000, 40_000, 150
    s1, o1, l1, _ = db.sample_reads(N1, L1, seed=9)
    s2, o2, l2, _ = db.sample_reads(N2, L2, seed=10)
    seqs = torch.cat([s1.reshape(-1), s2.reshape(-1)])
    nb1 = s1.numel()
    off = torch.cat([o1, o2 + nb1])
    lens = torch.cat([l1, l2])
    N, nb = N1 + N2, seqs.numel()
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
    ctx.set_taxonomy(ctax)
    taxa1 = torch.zeros(nb, dtype=torch.int32, device=dev)
    calls1 = torch.zeros(N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.classify_batch_device(seqs.data_ptr(), nb, off.data_ptr(), lens.data_ptr(), N, calls1.data_ptr(), taxa1.data_ptr(), max_read_len=L1)
    ctx.synchronize()
    want = ctx.counts()
    offs = db.offsets
    bounds = [0] + [int(torch.searchsorted(offs, offs[-1] * q // W).item()) for q in range(1, W)] + [4 ** NT]
    mg = capi.Mgpu([0] * W)
    shards = []
    for r in range(W):
        sh = synth_torch.BenchDb(dev, n_species=100, genome_len=60_000, k=K, nt=NT, seed=3, bin_lo=bounds[r], bin_hi=bounds[r + 1])
        mg.ctx(r).adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, bounds[r], bounds[r + 1])
        shards.append(sh)
    mg.set_taxonomy(ctax)
    assert mg.uses_routing()
    rb = [0, N1 + 2_000, N]  # rank 0: every contig and a few short reads (9.3 M positions), rank 1: short reads only (5.7 M)
    off_h = off.cpu().numpy()
    pb = [0, int(off_h[rb[1]]), nb]
    bufs = []
    for r in range(W):
        bufs.append({"seqs": torch.cat([seqs, torch.zeros(16, dtype=torch.uint8, device=dev)]) if r == 0 else torch.zeros(nb + 16, dtype=torch.uint8, device=dev),
                     "off": off if r == 0 else torch.zeros(N, dtype=torch.int64, device=dev),
                     "len": lens if r == 0 else torch.zeros(N, dtype=torch.int32, device=dev),
                     "calls": torch.zeros(N, dtype=torch.int32, device=dev), "taxa": torch.zeros(nb + 16, dtype=torch.int32, device=dev)})
    torch.cuda.synchronize()
    mg.step_device([{"d_seqs": b["seqs"].data_ptr(), "d_seq_off": b["off"].data_ptr(), "d_seq_len": b["len"].data_ptr(),
                     "d_calls": b["calls"].data_ptr(), "d_taxa": b["taxa"].data_ptr()} for b in bufs], nb, N, rb, pb, max_read_len=L1)
    for r in range(W):
        mg.ctx(r).synchronize()
    valid = torch.zeros(nb, dtype=torch.bool, device=dev)  # positions that carry a k-mer code
    pos = torch.arange(nb, device=dev)
    rid = torch.searchsorted(off, pos, right=True) - 1
    valid = (pos - off[rid]) < (lens[rid].to(torch.int64) - K + 1)
    for r in range(W):
        lo, hi = rb[r], rb[r + 1]
        assert torch.equal(bufs[r]["calls"][lo:hi], calls1[lo:hi]), r
        m = valid & (pos >= pb[r]) & (pos < pb[r + 1])
        assert torch.equal(bufs[r]["taxa"][:nb][m], taxa1[m]), r
    mg.reduce_state()
    assert same_counts(mg.ctx(1).counts(), want)
    mg.close()


def test_routed_step_of_eight_ranks_on_the_bench_database_matches_one_context():
    """owner routing at size: the 8 GB bench database in eight minimizer-range shards, eight ranks on
    the one device, 2 M reads per step -- calls, per-k-mer codes and the reduced per-taxon state equal one context that
    holds the whole database; two steps, so the queues, their chunk cursors and the owners' counters are reused"""
    import torch
    dev = torch.device("cuda:0")
    NT, L, N, W = 13, 150, 2_000_000, 8
    geo = dict(n_species=2000, genome_len=310_000, k=K, nt=NT, seed=7)
    db = synth_torch.BenchDb(dev, **geo)
    ids, par = db.tax.arrays()
    ctax = capi.Tax(ids=ids, parents=par)
    bins = synth_torch.bin_key(db.kmers[torch.randperm(db.n_pairs, device=dev)[:1_000_000]], K, NT)
    bounds = [int(x) for x in synth_torch.quantile_bin_bounds(bins, 4 ** NT, W)]
    db.kmers = db.vals = None
    batches = [db.sample_reads(N, L, seed=11 + i) for i in range(2)]
    ctx = capi.Ctx(0)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), K, NT, 2, keep=db)
    ctx.set_taxonomy(ctax)
    stride = L + 1
    nb = N * stride
    want = []
    for seqs, off, lens, _ in batches:
        taxa1 = torch.zeros(nb, dtype=torch.int32, device=dev)
        calls1 = torch.zeros(N, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.classify_batch_device(seqs.data_ptr(), nb, off.data_ptr(), lens.data_ptr(), N, calls1.data_ptr(), taxa1.data_ptr(), max_read_len=L)
        ctx.synchronize()
        want.append((calls1, taxa1))
    want_counts = ctx.counts()
    ctx.close()
    db.pairs = None
    torch.cuda.empty_cache()
    mg = capi.Mgpu([0] * W)
    shards = []
    for r in range(W):
        sh = synth_torch.BenchDb(dev, bin_lo=bounds[r], bin_hi=bounds[r + 1], **geo)
        sh.kmers = sh.vals = None
        mg.ctx(r).adopt_db(sh.pairs.data_ptr(), sh.n_pairs, sh.offsets.data_ptr(), K, NT, 2, bounds[r], bounds[r + 1])
        shards.append(sh)
    mg.set_taxonomy(ctax)
    assert mg.uses_routing()
    rb = [N * r // W for r in range(W + 1)]
    pb = [x * stride for x in rb]
    nk = L - K + 1
    bufs = [{"seqs": torch.zeros(nb + 16, dtype=torch.uint8, device=dev), "off": torch.zeros(N, dtype=torch.int64, device=dev),
             "len": torch.zeros(N, dtype=torch.int32, device=dev), "calls": torch.zeros(N, dtype=torch.int32, device=dev),
             "taxa": torch.zeros(nb + 16, dtype=torch.int32, device=dev)} for r in range(W)]
    for (seqs, off, lens, _), (calls1, taxa1) in zip(batches, want):
        bufs[0]["seqs"][:nb] = seqs.reshape(-1)
        bufs[0]["off"][:] = off
        bufs[0]["len"][:] = lens
        torch.cuda.synchronize()
        mg.step_device([{"d_seqs": b["seqs"].data_ptr(), "d_seq_off": b["off"].data_ptr(), "d_seq_len": b["len"].data_ptr(),
                         "d_calls": b["calls"].data_ptr(), "d_taxa": b["taxa"].data_ptr()} for b in bufs], nb, N, rb, pb, max_read_len=L)
        for r in range(W):
            mg.ctx(r).synchronize()
        for r in range(W):
            lo, hi = rb[r], rb[r + 1]
            assert torch.equal(bufs[r]["calls"][lo:hi], calls1[lo:hi]), r
            assert torch.equal(bufs[r]["taxa"][:nb].view(N, stride)[lo:hi, :nk], taxa1.view(N, stride)[lo:hi, :nk]), r
    mg.reduce_state()
    assert same_counts(mg.ctx(3).counts(), want_counts)
    mg.close()
