"""-m gpu: HyperLogLog++ sparse-mode emulation (ku_ctx_enable_sparse / ku_sparse_export / ku_report_sparse, SURVEY 8a
A13/A14): the report equals the reference's row for row -- no estimator allowance -- and, stronger, every taxon's
sparse / dense state and its set of encoded hashes equal the oracle's (which is pinned to the reference's reports)."""
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


def split_points(n, k, seed):
    rng = np.random.default_rng(seed)
    cuts = sorted(set(rng.integers(1, n, size=k - 1).tolist()))
    return [0] + cuts + [n]


def classify_in_batches(ctx, buf, off, lens, cuts, **kw):
    """host batches [cuts[i], cuts[i+1]) of the reads, in order"""
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo = int(off[a])
        hi = int(off[b]) if b < len(off) else len(buf)
        out.append(ctx.classify_batch_rle(buf[lo:hi], off[a:b] - lo, lens[a:b], **kw))
    return out


def assert_sparse_state_equals_oracle(ctx, run):
    counts = ctx.counts()
    flags, pairs = ctx.sparse_export()
    want = run.counts()
    slot_of = {int(t): s for s, t in enumerate(counts["slot_taxid"])}
    got_sets = {}
    for p in pairs.tolist():
        got_sets.setdefault(p >> 32, set()).add(p & 0xFFFFFFFF)
    n_sparse = n_dense = 0
    for t, c in want.items():
        if not c["n_kmers"]:
            continue
        s = slot_of[t]
        assert bool(flags[s]) == c["sparse"], (t, c["n_kmers"])
        if c["sparse"]:
            assert got_sets.get(s, set()) == set(c["sketch"].sparse_list().tolist()), t
            n_sparse += 1
        else:
            n_dense += 1
    return counts, flags, pairs, n_sparse, n_dense


@pytest.mark.parametrize("fixture,reads,unit,report,n_batches", [
    ("f1", "f1/reads.fq", 500000, "report.tsv", 1), ("f1", "f1/reads.fq", 500000, "report.tsv", 5),
    ("f1", "f1/reads.fq", 1000, "report_u1000.tsv", 1), ("f1", "f1/reads.fq", 1000, "report_u1000.tsv", 7),
    ("f2", "f2/edge.fa", 500000, "report.tsv", 1), ("f4", "f4/merged.fa", 500000, "report.tsv", 3)])
def test_report_equals_the_reference(fixture, reads, unit, report, n_batches):
    ids, seqs = synth.read_seqfile(os.path.join(G, reads))
    buf, off, lens = ko.pack_reads(seqs)
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(unit)
    classify_in_batches(ctx, buf, off, lens, split_points(len(seqs), n_batches, 11) if n_batches > 1 else [0, len(seqs)])
    run = ko.Run(ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB"), work_unit_nt=unit)
    run.classify(seqs)
    counts, flags, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(ctx, run)
    gc.assert_same_counts(counts, run)
    got = capi.report_sparse(ctax, counts, flags, pairs, [f"{F1}/database.kdb.counts"])
    assert rows(got) == rows(open(os.path.join(G, fixture, report)).read())
    if fixture == "f1":
        assert (n_sparse > 0 and n_dense > 0) if unit == 500000 else n_dense == 0


def test_run_wide_set_grows_on_demand():
    """the (slot, encoding) set starts with 2^10 cells and takes the run's tens of thousands of entries by moving to
    larger tables between passes; slots that turned dense meanwhile are dropped on the way"""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(1000, 10)
    classify_in_batches(ctx, buf, off, lens, split_points(len(seqs), 9, 4))
    run = ko.Run(ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB"), work_unit_nt=1000)
    run.classify(seqs)
    counts, flags, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(ctx, run)
    assert len(pairs) > 4096
    assert rows(ctx.report(ctax, [f"{F1}/database.kdb.counts"])) == rows(open(f"{F1}/report_u1000.tsv").read())
    ctx.reset_counts()  # a second run in the grown table
    classify_in_batches(ctx, buf, off, lens, [0, len(seqs)])
    assert rows(ctx.report(ctax, [f"{F1}/database.kdb.counts"])) == rows(open(f"{F1}/report_u1000.tsv").read())


def test_out_of_memory_switches_the_emulation_off_and_the_run_goes_on(monkeypatch):
    """no room for the run-wide set (test hook: a ceiling of 2^11 cells): classification is untouched, the state says so,
    the report is the dense-register one"""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    ctx, cdb, ctax = gc.make_ctx(F1)
    monkeypatch.setenv("KU_SPARSE_MAX_LOG2", "11")
    ctx.enable_sparse(1000, 10)
    assert ctx.sparse_state() == 1
    rle = classify_in_batches(ctx, buf, off, lens, split_points(len(seqs), 5, 8))
    assert ctx.sparse_state() == 2
    text = ""
    cuts = split_points(len(seqs), 5, 8)
    for (a, b), r in zip(zip(cuts[:-1], cuts[1:]), rle):
        lo = int(off[a])
        hi = int(off[b]) if b < len(off) else len(buf)
        text += capi.format_kraken_rle(buf[lo:hi], off[a:b] - lo, lens[a:b], ids[a:b], K, r)
    assert text == open(f"{F1}/out.tsv").read()
    paths = [f"{F1}/database.kdb.counts"]
    assert ctx.report(ctax, paths) == capi.report(ctax, ctx.counts(), paths)
    with pytest.raises(capi.KuError):
        ctx.sparse_export()


def test_whole_run_as_one_unit_equals_the_chunk_mode_report():
    """-x: the reference inserts into the global sketches directly (classify.cpp:719): work_unit_nt = 0"""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    cdb = capi.Db(f"{F1}/database.kdb", f"{F1}/database.idx")
    ctax = capi.Tax(f"{F1}/taxDB")
    bounds = cdb.chunk_plan(70 << 10)
    bounds[-1] = cdb.info.n_bins
    ctx = capi.Ctx(0)
    ctx.load_db(cdb, int(bounds[0]), int(bounds[1]))
    ctx.set_taxonomy(ctax, cdb.values())
    ctx.enable_sparse(0)
    cuts = split_points(len(seqs), 4, 5)
    batches = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo, hi = int(off[a]), (int(off[b]) if b < len(off) else len(buf))
        batches.append(ctx.batch(buf[lo:hi], off[a:b] - lo, lens[a:b]))
    for c in range(len(bounds) - 1):
        if c:
            ctx.swap_shard(cdb, int(bounds[c]), int(bounds[c + 1]))
        for bt in batches:
            bt.lookup()
    for bt in batches:
        bt.finish()
    counts = ctx.counts()
    flags, pairs = ctx.sparse_export()
    got = capi.report_sparse(ctax, counts, flags, pairs, [f"{F1}/database.kdb.counts"])
    assert rows(got) == rows(open(f"{F1}/report_chunk.tsv").read())


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
    buf, off, lens = ko.pack_reads(seqs)
    rle = ctx.classify_batch_rle(buf, off, lens)
    assert capi.format_kraken_rle(buf, off, lens, ids, K, rle) == open(f"{d8}/out{order}.tsv").read()
    counts = ctx.counts()
    flags, pairs = ctx.sparse_export()
    got = capi.report_sparse(ctax, counts, flags, pairs, [f"{x}/database.kdb.counts" for x in dirs])
    assert rows(got) == rows(open(f"{d8}/report{order}.tsv").read())


def test_quick_mode_books_the_scanned_prefix_only():
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(20000)
    classify_in_batches(ctx, buf, off, lens, split_points(len(seqs), 3, 2), flags=capi.KU_F_QUICK, min_hits=2)
    run = ko.Run(ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB"), work_unit_nt=20000, quick=True,
                 min_hits=2)
    run.classify(seqs)
    counts, flags, pairs, *_ = assert_sparse_state_equals_oracle(ctx, run)
    assert rows(capi.report_sparse(ctax, counts, flags, pairs, [f"{F1}/database.kdb.counts"])) == \
        rows(run.report(f"{F1}/taxDB", f"{F1}/database.kdb.counts"))


def test_files_close_their_last_unit():
    """work units do not span input files: ku_sparse_close_unit between them"""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    a, b = seqs[:400], seqs[400:]
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(40000)
    odb, otax = ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB")
    run = ko.Run(odb, otax, work_unit_nt=40000)
    for part in (a, b):
        buf, off, lens = ko.pack_reads(part)
        ctx.sparse_close_unit()
        ctx.classify_batch_rle(buf, off, lens)
        run.classify(part)  # one call = one input file: its last, partial unit closes
    assert_sparse_state_equals_oracle(ctx, run)


def classify_in_batches_by_path(ctx, buf, off, lens, cuts, path, monkeypatch):
    """path: "fast" (the fused kernel's sparse fast path, the default), "staged" (KU_NO_SPARSE_FAST: lookup + emulation
    pass + resolve kernels) or "mixed" (the batches alternate: both keep the carried-over work unit in the same form)"""
    out = []
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        staged = path == "staged" or (path == "mixed" and i % 2 == 1)
        if staged:
            monkeypatch.setenv("KU_NO_SPARSE_FAST", "1")
        else:
            monkeypatch.delenv("KU_NO_SPARSE_FAST", raising=False)
        lo = int(off[a])
        hi = int(off[b]) if b < len(off) else len(buf)
        out.append(ctx.classify_batch_rle(buf[lo:hi], off[a:b] - lo, lens[a:b]))
    monkeypatch.delenv("KU_NO_SPARSE_FAST", raising=False)
    return out


@pytest.mark.parametrize("path", ["fast", "staged", "mixed"])
@pytest.mark.parametrize("extra", [0, 1, 2])
def test_switch_to_dense_at_exactly_1024_entries(extra, path, monkeypatch):
    """hyperloglogplus.cpp:496-498: the size test precedes the insert.  One work unit gives a taxon exactly 1024 distinct
    k-mers: with nothing behind them the sketch stays sparse (1024 entries); one more insert -- a duplicate -- switches it.
    extra = 0: 1024 inserts; 1: + a read with one (known) k-mer; 2: 1023 distinct k-mers + two duplicates (1025 inserts,
    still sparse).  The fast path counts inserts (>= 1025 makes the unit a candidate), the exact pass decides."""
    rng = np.random.default_rng(5)
    db = gc.random_db(rng, n_genomes=4, glen=5000, k=K, nt=9)
    tax = db["tax"]
    sp = list(db["genomes"])
    g = db["genomes"][sp[0]]
    n_first = 1023 if extra == 2 else 1024
    seqs = [synth.codes_to_ascii(g[100:100 + n_first + K - 1])]
    if extra == 1:
        seqs.append(synth.codes_to_ascii(g[300:300 + K]))
    if extra == 2:
        seqs.append(synth.codes_to_ascii(g[300:300 + K + 1]))
    seqs += [synth.codes_to_ascii(db["genomes"][sp[1]][50:200])]  # another taxon in the same unit
    buf, off, lens = ko.pack_reads(seqs)
    ids, par = tax.arrays()
    ctax, otax = capi.Tax(ids=ids, parents=par), ko.Tax(ids=ids, parents=par)
    raw = db["pairs"].view(np.uint8).reshape(-1)
    cdb = capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=9)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=9)
    ctx = capi.Ctx(0)
    ctx.load_db(cdb)
    ctx.set_taxonomy(ctax)
    ctx.enable_sparse(500000)
    classify_in_batches_by_path(ctx, buf, off, lens, [0, 1, len(seqs)] if len(seqs) > 2 else [0, len(seqs)], path, monkeypatch)
    run = ko.Run(odb, otax, work_unit_nt=500000)
    res = run.classify(seqs)
    counts, flags, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(ctx, run)
    # the read's k-mers all carry one taxon and are distinct as encodings (else the scenario does not probe the edge)
    first = res["taxa"][:n_first]
    if len(set(first.tolist())) == 1 and first[0] != 0:
        t = int(first[0])
        w = run.counts()[t]
        if extra == 1:
            assert not w["sparse"]
        elif len(w["sketch"].sparse_list()) >= 1023:
            assert w["sparse"]


@pytest.mark.parametrize("path", ["fast", "staged", "mixed"])
@pytest.mark.parametrize("unit,seed", [(30000, 1), (150000, 2), (7000, 3)])
def test_random_database_mixed_sparse_and_dense_taxa(unit, seed, path, monkeypatch):
    """taxa of very different abundance, several work units, uneven batches: which sketches switch to dense and the
    exact sets of the ones that do not"""
    rng = np.random.default_rng(seed)
    db = gc.random_db(rng, n_genomes=8, glen=6000, k=K, nt=9)
    tax = db["tax"]
    weights = np.array([200, 60, 20, 8, 3, 1, 1, 0.3])
    weights = weights / weights.sum()
    sp = list(db["genomes"])
    seqs = []
    for _ in range(6000):
        if rng.random() < 0.1:
            seqs.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(80, 200))).tobytes()))
            continue
        g = db["genomes"][sp[int(rng.choice(len(sp), p=weights))]]  # 2-bit codes
        n = int(rng.integers(60, 260))
        s = int(rng.integers(0, len(g) - n))
        c = g[s:s + n]
        r = bytearray(synth.codes_to_ascii(c if rng.random() < 0.5 else synth.revcomp_codes(c)))
        if rng.random() < 0.3:
            r[int(rng.integers(0, n))] = ord("N")
        seqs.append(bytes(r))
    buf, off, lens = ko.pack_reads(seqs)
    ids, par = tax.arrays()
    ctax, otax = capi.Tax(ids=ids, parents=par), ko.Tax(ids=ids, parents=par)
    raw = db["pairs"].view(np.uint8).reshape(-1)
    cdb = capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=9)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=9)
    ctx = capi.Ctx(0)
    ctx.load_db(cdb)
    ctx.set_taxonomy(ctax)
    ctx.enable_sparse(unit)
    rle = classify_in_batches_by_path(ctx, buf, off, lens, split_points(len(seqs), 6, seed), path, monkeypatch)
    run = ko.Run(odb, otax, work_unit_nt=unit)
    res = run.classify(seqs)
    assert np.array_equal(np.concatenate([r["calls"] for r in rle]), res["calls"])
    counts, flags, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(ctx, run)
    gc.assert_same_counts(counts, run)
    assert n_sparse > 0 and n_dense > 0
