"""-m gpu: the report from the device-resident state (ku_ctx_report): clade roll-up on the GPU (byte-wise
maximum of the members' HLL registers per clade -> register histogram; union of the members' encoded hashes for clades
that stayed sparse) == the host roll-up over the exported state, character for character, in all three sketch modes
(dense registers, sparse-mode emulation, classifyExact), and == the reference's report files."""
import os

import numpy as np
import pytest

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko
import gpu_common as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["tables", "bitmaps"], autouse=True)
def union_form(request, monkeypatch):
    """clade unions of the sparse sketches in ku_ctx_report: per-clade hash tables, or (test hook: for every clade with
    several members, not only the big ones) bitmaps over the 2^25 indices merged level by level"""
    if request.param == "bitmaps":
        monkeypatch.setenv("KU_ROLLUP_BITMAP_MIN", "1")
    return request.param
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
F1 = os.path.join(G, "f1")
K = 31


def rows(text):
    return sorted(text.strip("\n").split("\n"))


def f1_reads():
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    return ko.pack_reads(seqs)


@pytest.mark.parametrize("unit,golden", [(500000, "report.tsv"), (1000, "report_u1000.tsv"), (0, None)])
def test_sparse_mode_report(unit, golden):
    buf, off, lens = f1_reads()
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(unit)
    ctx.classify_batch_rle(buf, off, lens)
    paths = [f"{F1}/database.kdb.counts"]
    got = ctx.report(ctax, paths)
    flags, pairs = ctx.sparse_export()
    assert got == capi.report_sparse(ctax, ctx.counts(), flags, pairs, paths)
    if golden:
        assert rows(got) == rows(open(f"{F1}/{golden}").read())
    assert got.count("\n") > 5


def test_dense_register_report_and_accumulation():
    buf, off, lens = f1_reads()
    ctx, cdb, ctax = gc.make_ctx(F1)
    paths = [f"{F1}/database.kdb.counts"]
    assert ctx.report(ctax, paths) == ""  # no reads yet: no report
    ctx.classify_batch_rle(buf, off, lens)
    got = ctx.report(ctax, paths)
    assert got == capi.report(ctax, ctx.counts(), paths)
    assert got == capi.report(ctax, ctx.counts(), paths[0])
    ctx.classify_batch_rle(buf, off, lens)  # the state keeps accumulating after a report
    assert ctx.report(ctax, paths) == capi.report(ctax, ctx.counts(), paths)
    assert ctx.report(ctax, []) == capi.report(ctax, ctx.counts(), [])  # no genome sizes: cov = NA


def test_exact_counting_report():
    buf, off, lens = f1_reads()
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_exact(16)
    ctx.classify_batch(buf, off, lens)
    paths = [f"{F1}/database.kdb.counts"]
    got = ctx.report(ctax, paths)
    assert got == capi.report_exact(ctax, ctx.counts(), ctx.exact_counts(), paths)
    assert rows(got) == rows(open(f"{F1}/report_exact.tsv").read())


@pytest.mark.parametrize("order", ["", "_swapped"])
def test_hierarchical_two_database_report(order):
    d8 = os.path.join(G, "f8")
    dirs = [F1, d8] if order == "" else [d8, F1]
    cdbs = [capi.Db(f"{x}/database.kdb", f"{x}/database.idx") for x in dirs]
    ctax = capi.Tax(f"{F1}/taxDB")
    ctx = capi.Ctx(0)
    ctx.load_db(cdbs[0])
    ctx.add_db(cdbs[1])
    ctx.set_taxonomy(ctax)
    ctx.enable_sparse()
    ids, seqs = synth.read_seqfile(f"{d8}/reads.fq")
    ctx.classify_batch_rle(*ko.pack_reads(seqs))
    got = ctx.report(ctax, [f"{x}/database.kdb.counts" for x in dirs])
    assert rows(got) == rows(open(f"{d8}/report{order}.tsv").read())


@pytest.mark.parametrize("unit,seed,levels", [(30000, 1, (2, 3, 4, 5)), (7000, 3, (2, 2, 3, 3, 4, 6, 8)), (None, 5, (3, 5, 8))])
def test_random_taxonomy_mixed_clades(unit, seed, levels):
    """taxa of very different abundance under a random taxonomy: clades that are dense through one member, clades whose
    members all stayed sparse (union of sets, also beyond 1024 entries), taxa counted through reads only, database
    values missing from the taxDB"""
    rng = np.random.default_rng(seed)
    tax = synth.random_taxonomy(8, rng, levels=levels)
    db = gc.random_db(rng, n_genomes=8, glen=6000, k=K, nt=9, tax=tax)
    weights = np.array([200, 60, 20, 8, 3, 1, 1, 0.3])
    weights = weights / weights.sum()
    sp = list(db["genomes"])
    seqs = []
    for _ in range(5000):
        if rng.random() < 0.1:
            seqs.append(bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=int(rng.integers(20, 200))).tobytes()))
            continue
        g = db["genomes"][sp[int(rng.choice(len(sp), p=weights))]]
        n = int(rng.integers(60, 260))
        s = int(rng.integers(0, len(g) - n))
        seqs.append(bytes(synth.codes_to_ascii(g[s:s + n])))
    buf, off, lens = ko.pack_reads(seqs)
    ids, par = tax.arrays()
    # drop one species from the taxDB: its counts exist but the report has "no entry" for it
    keep = ids != sp[3]
    ctax_full = capi.Tax(ids=ids, parents=par)
    ctax_cut = capi.Tax(ids=ids[keep], parents=par[keep])
    raw = db["pairs"].view(np.uint8).reshape(-1)
    cdb = capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=9)
    ctx = capi.Ctx(0)
    ctx.load_db(cdb)
    ctx.set_taxonomy(ctax_full)
    if unit is not None:
        ctx.enable_sparse(unit)
    cut = len(seqs) // 3
    lo = int(off[cut])
    ctx.classify_batch_rle(buf[:lo], off[:cut], lens[:cut])
    ctx.classify_batch_rle(buf[lo:], off[cut:] - lo, lens[cut:])
    counts = ctx.counts()
    for t in (ctax_full, ctax_cut):
        got = ctx.report(t, [])
        if unit is not None:
            flags, pairs = ctx.sparse_export()
            want = capi.report_sparse(t, counts, flags, pairs, [])
            assert 0 < flags[counts["n_kmers"] > 0].sum() < (counts["n_kmers"] > 0).sum()  # both kinds of sketches
        else:
            want = capi.report(t, counts, [])
        assert got == want
        assert got.count("\n") > 8
