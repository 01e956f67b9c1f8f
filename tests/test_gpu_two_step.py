"""-m gpu: ku_classify_batch_rle in two steps (ku_classify_batch_rle_enqueue / _finish, round 5): up to four batches in
flight on a context, the upload of the next and the copies back of the previous under the kernels of the current one.

Checked: the same calls, runs, per-taxon state and Kraken text as the one-step call and as the reference's files, whatever
the interleaving; with the sparse-sketch emulation on, every taxon's sparse / dense state and encoded set equal the oracle's
and the report equals the reference's row for row -- the open work unit now travels between batches as its reads + insert
counts (tail form) and is evaluated when it closes; batches of the staged paths in between go one at a time
and hand the open unit over in the staged form; what cannot overlap is refused with KU_ESTATE while a batch is in flight."""
import os

import numpy as np
import pytest

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko
import gpu_common as gc
from test_gpu_sparse import assert_sparse_state_equals_oracle, rows, split_points

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
F1 = os.path.join(G, "f1")
K = 31


def batches_of(buf, off, lens, cuts):
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo = int(off[a])
        hi = int(off[b]) if b < len(off) else len(buf)
        yield a, b, buf[lo:hi], off[a:b] - lo, lens[a:b]


def run_two_step(ctx, buf, off, lens, cuts, depth=3, **kw):
    """every batch through enqueue / finish with up to `depth` in flight; results in batch order"""
    flying, out = [], []
    for i, (a, b, bb, bo, bl) in enumerate(batches_of(buf, off, lens, cuts)):
        if len(flying) >= depth:
            out.append(ctx.rle_finish(flying.pop(0)))
        # where the runs go: nowhere (ku_fetch_runs brings them), a buffer that is too small (the same), one that holds them
        # (they come with the calls)
        flying.append(ctx.rle_enqueue(bb, bo, bl, runs_cap=(0, 3, 1 << 16)[i % 3], **kw))
        assert ctx.rle_in_flight() == len(flying)
    while flying:
        out.append(ctx.rle_finish(flying.pop(0)))
    assert ctx.rle_in_flight() == 0
    return out


def kraken_text(buf, off, lens, ids, cuts, results):
    text = ""
    for (a, b, bb, bo, bl), r in zip(batches_of(buf, off, lens, cuts), results):
        text += capi.format_kraken_rle(bb, bo, bl, ids[a:b], K, r)
    return text


@pytest.mark.parametrize("n_batches,depth", [(1, 2), (6, 2), (6, 1), (17, 4), (17, 3)])
def test_two_step_equals_the_reference_files(n_batches, depth):
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    cuts = split_points(len(seqs), n_batches, 5) if n_batches > 1 else [0, len(seqs)]
    ctx, cdb, ctax = gc.make_ctx(F1)
    res = run_two_step(ctx, buf, off, lens, cuts, depth)
    assert kraken_text(buf, off, lens, ids, cuts, res) == open(f"{F1}/out.tsv").read()
    run = ko.Run(ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB"))
    run.classify(seqs)
    gc.assert_same_counts(ctx.counts(), run)
    # the one-step call on a second context: the same state, bit for bit
    ctx2, _, _ = gc.make_ctx(F1)
    for a, b, bb, bo, bl in batches_of(buf, off, lens, cuts):
        ctx2.classify_batch_rle(bb, bo, bl)
    c1, c2 = ctx.counts(), ctx2.counts()
    for key in ("n_kmers", "registers", "n_reads"):
        assert np.array_equal(c1[key], c2[key]), key


@pytest.mark.parametrize("unit,report,n_batches", [(500000, "report.tsv", 1), (500000, "report.tsv", 7), (1000, "report_u1000.tsv", 1),
                                                   (1000, "report_u1000.tsv", 9), (20000, None, 13), (7000, None, 40)])
def test_two_step_with_the_sparse_emulation(unit, report, n_batches):
    """work units of 1000 nt close several times per batch, units of 500000 nt stay open for the whole run (evaluated from
    the tail when the run ends), 7000 / 20000 nt straddle most batch borders"""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    cuts = split_points(len(seqs), n_batches, 23) if n_batches > 1 else [0, len(seqs)]
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(unit)
    res = run_two_step(ctx, buf, off, lens, cuts)
    assert kraken_text(buf, off, lens, ids, cuts, res) == open(f"{F1}/out.tsv").read()
    run = ko.Run(ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB"), work_unit_nt=unit)
    run.classify(seqs)
    text = ctx.report(ctax, [f"{F1}/database.kdb.counts"])  # (closes the last unit; the roll-up reads the table's marks)
    counts, flags, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(ctx, run)
    gc.assert_same_counts(counts, run)
    assert rows(text) == rows(capi.report_sparse(ctax, counts, flags, pairs, [f"{F1}/database.kdb.counts"]))
    if report:
        assert rows(text) == rows(open(os.path.join(F1, report)).read())
    assert rows(ctx.report(ctax, [f"{F1}/database.kdb.counts"], flags=1)) == rows(open(f"{F1}/report_p0.tsv").read())


def test_staged_batches_between_two_step_batches_hand_the_open_unit_over(monkeypatch):
    """batches that cannot take the fused kernel's fast path (here: KU_NO_SPARSE_FAST, as the mixed cases of test_gpu_sparse) go
    one at a time -- KU_ESTATE while another is in flight -- and leave / pick up the open unit in the staged form"""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    cuts = split_points(len(seqs), 10, 77)
    unit = 9000
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(unit)
    run = ko.Run(ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB"), work_unit_nt=unit)
    run.classify(seqs)
    flying, results = [], []
    for i, (a, b, bb, bo, bl) in enumerate(batches_of(buf, off, lens, cuts)):
        staged = i in (3, 4, 7)
        if staged:
            monkeypatch.setenv("KU_NO_SPARSE_FAST", "1")
            if flying:
                with pytest.raises(capi.KuError):
                    ctx.rle_enqueue(bb, bo, bl)
        else:
            monkeypatch.delenv("KU_NO_SPARSE_FAST", raising=False)
        if staged or len(flying) >= 2:
            while flying and (staged or len(flying) >= 2):
                results.append(ctx.rle_finish(flying.pop(0)))
        try:
            flying.append(ctx.rle_enqueue(bb, bo, bl))
        except capi.KuError:  # (the batch behind a staged one meets an open unit in the staged form: one at a time)
            while flying:
                results.append(ctx.rle_finish(flying.pop(0)))
            flying.append(ctx.rle_enqueue(bb, bo, bl))
    monkeypatch.delenv("KU_NO_SPARSE_FAST", raising=False)
    with pytest.raises(capi.KuError):
        ctx.report(ctax, [f"{F1}/database.kdb.counts"])  # batches in flight
    while flying:
        results.append(ctx.rle_finish(flying.pop(0)))
    assert kraken_text(buf, off, lens, ids, cuts, results) == open(f"{F1}/out.tsv").read()
    counts, flags, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(ctx, run)
    gc.assert_same_counts(counts, run)


def test_what_cannot_be_in_flight_is_refused():
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs[:300])
    ctx, cdb, ctax = gc.make_ctx(F1)
    with pytest.raises(capi.KuError):
        ctx.rle_finish({})  # nothing in flight
    jobs = [ctx.rle_enqueue(buf, off, lens) for _ in range(4)]  # KU_RLE_MAX_IN_FLIGHT
    with pytest.raises(capi.KuError):
        ctx.rle_enqueue(buf, off, lens)  # four in flight already
    with pytest.raises(capi.KuError):
        ctx.classify_batch_rle(buf, off, lens)
    with pytest.raises(capi.KuError):
        ctx.reset_counts()
    res = [ctx.rle_finish(j) for j in jobs]
    r1, r2 = res[0], res[3]
    assert all(np.array_equal(r1["calls"], r["calls"]) for r in res)
    want = ctx.classify_batch_rle(buf, off, lens)
    assert np.array_equal(want["calls"], r1["calls"])
    assert capi.format_kraken_rle(buf, off, lens, ids[:300], K, r2) == capi.format_kraken_rle(buf, off, lens, ids[:300], K, want)


def test_reserve_and_its_warm_up_change_nothing(monkeypatch):
    """ku_classify_batch_rle_reserve sends synthetic batches through the path (count-less) so that the first real batch finds
    the runtime set up: the run's state -- counts, registers, the emulation's sets, the table's marks -- is as without it; and the
    flags the emulation takes again at _finish (the state may have moved on since the kernels ran) only ever spare work:
    KU_NO_REFLAG=1 gives the same result"""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    cuts = split_points(len(seqs), 11, 3)
    states = []
    for variant in ("plain", "reserve", "no_reflag"):
        if variant == "no_reflag":
            monkeypatch.setenv("KU_NO_REFLAG", "1")
        ctx, cdb, ctax = gc.make_ctx(F1)
        ctx.enable_sparse(1000)
        if variant != "plain":
            ctx.rle_reserve(len(buf) // 4, len(seqs) // 4, int(lens.max()), 4)
            c0 = ctx.counts()
            assert not c0["n_reads"].any() and not c0["n_kmers"].any() and not c0["registers"].any()
        res = run_two_step(ctx, buf, off, lens, cuts, depth=3)
        assert kraken_text(buf, off, lens, ids, cuts, res) == open(f"{F1}/out.tsv").read()
        text = ctx.report(ctax, [f"{F1}/database.kdb.counts"])
        assert rows(text) == rows(open(f"{F1}/report_u1000.tsv").read())
        c = ctx.counts()
        states.append((text, c["n_reads"].copy(), c["n_kmers"].copy(), c["registers"].copy()))
    for t, a, b, r in states[1:]:
        assert t == states[0][0] and np.array_equal(a, states[0][1]) and np.array_equal(b, states[0][2]) and np.array_equal(r, states[0][3])


@pytest.mark.parametrize("unit,report,n_batches", [(1000, "report_u1000.tsv", 9), (500000, "report.tsv", 7), (7000, None, 12)])
def test_run_array_overflow_with_the_emulation_on(unit, report, n_batches, monkeypatch):
    """a batch whose runs outgrow the run array (KU_RUNS_CAP forces it) is redone through the per-k-mer array --
    and the sparse-sketch emulation, on by default under `classify -r`, evaluates its flagged work units from THAT redo's runs
    (round 5 ended the run with KU_EUNSUP there).  Batches in flight; state and report equal the oracle's / the reference's."""
    ids, seqs = synth.read_seqfile(f"{F1}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    cuts = split_points(len(seqs), n_batches, 23)
    monkeypatch.setenv("KU_RUNS_CAP", "40")
    ctx, cdb, ctax = gc.make_ctx(F1)
    ctx.enable_sparse(unit)
    res = run_two_step(ctx, buf, off, lens, cuts)
    assert kraken_text(buf, off, lens, ids, cuts, res) == open(f"{F1}/out.tsv").read()
    run = ko.Run(ko.Db(f"{F1}/database.kdb", f"{F1}/database.idx"), ko.Tax(f"{F1}/taxDB"), work_unit_nt=unit)
    run.classify(seqs)
    text = ctx.report(ctax, [f"{F1}/database.kdb.counts"])
    counts, flags, pairs, n_sparse, n_dense = assert_sparse_state_equals_oracle(ctx, run)
    gc.assert_same_counts(counts, run)
    assert ctx.sparse_state() == 1  # (not given up)
    if report:
        assert rows(text) == rows(open(os.path.join(F1, report)).read())
