"""-m gpu: run-length encoded output straight from the fused kernel (ku_short.hip, OUT = 1; ku_classify_batch_rle).

The wave that classifies a read also finds the run starts of its per-k-mer codes and writes {code, start} pairs into a
chunk of the run array it claimed -- no per-k-mer array, no second kernel.  Checked here: the decoded runs equal the
per-k-mer codes of ku_classify_batch (and through it the oracle's) for one-pass and windowed reads, across window
edges and skipped ambiguous stretches, for reads whose runs outgrow a chunk several times over (the read's earlier runs
move along), for a run array that is too small (the batch is redone through the per-k-mer array), and for batches uploaded
in several segments on the copy stream; the per-taxon state equals that of the per-k-mer path; the Kraken text is
byte-identical."""
import numpy as np
import pytest

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko

from gpu_common import assert_same_classification, assert_same_counts, oracle_flat, random_db
from test_gpu_windowed import K, build, with_n

pytestmark = pytest.mark.gpu


def decode_runs(r, off, lens, n_bytes, k=K):
    """runs -> the array parallel to the read buffer that ku_classify_batch returns"""
    taxa = np.zeros(max(n_bytes, 1), dtype=np.uint32)
    runs, ro, rc = r["runs"], r["run_off"], r["run_cnt"]
    for i in range(len(lens)):
        n = int(lens[i]) - k + 1
        if n <= 0:
            assert rc[i] == 0
            continue
        c = int(rc[i])
        assert c >= 1
        mine = runs[int(ro[i]):int(ro[i]) + c]
        starts = mine[:, 1].astype(np.int64)
        assert starts[0] == 0 and (np.diff(starts) > 0).all() and starts[-1] < n, (i, starts[:5], n)
        assert (mine[1:, 0] != mine[:-1, 0]).all(), i  # neighbouring runs differ in their code
        ends = np.append(starts[1:], n)
        o = int(off[i])
        for (code, st), en in zip(mine.tolist(), ends.tolist()):
            taxa[o + st:o + en] = code
    return taxa


def check_rle_against_per_kmer(ctx, buf, off, lens, ids=None):
    ctx.reset_counts()
    ref = ctx.classify_batch(buf, off, lens)
    counts_ref = ctx.counts()
    ctx.reset_counts()
    r = ctx.classify_batch_rle(buf, off, lens)
    assert np.array_equal(r["calls"], ref["calls"])
    assert not r["hits"].any()
    got = decode_runs(r, off, lens, len(buf))
    m = np.zeros(len(buf), dtype=bool)
    for o, l in zip(off.tolist(), lens.tolist()):
        if l >= K:
            m[o:o + l - K + 1] = True
    assert np.array_equal(got[m], ref["taxa"][:len(buf)][m])
    c = ctx.counts()
    for key in ("n_kmers", "registers", "n_reads"):
        assert np.array_equal(c[key], counts_ref[key]), key
    if ids is not None:
        assert capi.format_kraken_rle(buf, off, lens, ids, K, r) == capi.format_kraken(buf, off, lens, ids, K, ref["calls"], taxa=ref["taxa"])
    return r, ref


@pytest.mark.parametrize("nt", [13, 10])
def test_mixed_lengths_window_edges_and_ambiguous_stretches(nt):
    rng = np.random.default_rng(100 + nt)
    db = random_db(rng, n_genomes=8, glen=9000, nt=nt)
    odb, otax, ctx = build(db, nt)
    g = list(db["genomes"].values())
    reads = []
    for L in (31, 60, 150, 158, 159, 200, 222, 223, 224, 250, 285, 286, 287, 301, 415, 1000, 2600):
        rs, _ = synth.sample_reads(db["genomes"], 10, L, rng, n_rate=0.003)
        reads += rs
    for i in range(40):  # chimeras: several taxa per read, runs that cross window edges
        a, b, c = (g[int(rng.integers(0, len(g)))] for _ in range(3))
        reads.append(synth.codes_to_ascii(np.concatenate([a[:400], b[100:350], c[200:900]])))
    for i in range(40):  # mate pairs joined by N
        a = g[int(rng.integers(0, len(g)))]
        s = int(rng.integers(0, len(a) - 500))
        reads.append(synth.codes_to_ascii(a[s:s + 150]) + b"N" + synth.codes_to_ascii(synth.revcomp_codes(a[s + 250:s + 400])))
    base, _ = synth.sample_reads(db["genomes"], 1, 700, rng)
    for p in list(range(120, 165)) + list(range(250, 295)) + [0, 30, 31, 668, 669, 699]:
        reads.append(with_n(base[0], [p]))
    for p, run in ((100, 40), (127, 31), (128, 30), (140, 200), (157, 2), (158, 1), (380, 129), (0, 700), (600, 100), (97, 31), (98, 31), (96, 64)):
        reads.append(with_n(base[0], range(p, p + run)))
    reads += [b"", b"ACGT", base[0][:30], base[0][:31], b"N" * 300, b"N" * 31]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    ids = [f"r{i}" for i in range(len(reads))]
    run, res, buf, off, lens, taxa = oracle_flat(odb, otax, reads)
    r, ref = check_rle_against_per_kmer(ctx, buf, off, lens, ids)
    assert_same_classification(ref, res, taxa, off, lens, K)
    # short reads alone: the one-pass instances (2 and 3 k-mers per lane)
    for lim in (158, 222):
        sub = [x for x in reads if len(x) <= lim]
        run2, res2, buf2, off2, lens2, taxa2 = oracle_flat(odb, otax, sub)
        r2, ref2 = check_rle_against_per_kmer(ctx, buf2, off2, lens2)
        assert_same_classification(ref2, res2, taxa2, off2, lens2, K)
        assert_same_counts(ctx.counts(), run2)


def many_taxa_db(rng, every):
    """a genome whose k-mers change taxon every `every` positions over a pool of 700 taxa (as test_gpu_windowed)"""
    tax = synth.Taxonomy()
    tax.add(1, 1, "root", "root")
    ids = [1]
    for i in range(700):
        t = 10 + 7 * i
        tax.add(t, ids[int(rng.integers(max(0, len(ids) - 40), len(ids)))], f"n{t}", "no rank")
        ids.append(t)
    genome = synth.procedural_genome(int(rng.integers(1, 1 << 30)), 1, 30000)
    fwd = synth.canonical(synth.kmers_forward(genome, K), K)
    kmers = np.unique(fwd)
    pool = np.array(ids, dtype=np.uint32)
    seg = pool[rng.integers(0, len(pool), len(fwd) // every + 1)]
    val_of = {}
    for i, km in enumerate(fwd.tolist()):
        val_of.setdefault(km, int(seg[i // every]))
    vals = np.array([val_of[int(x)] for x in kmers.tolist()], dtype=np.uint32)
    sk, sv, off = synth.sort_db(kmers, vals, K, 10)
    raw = synth.pack_pairs(sk, sv).view(np.uint8)
    tids, tpar = tax.arrays()
    ctx = capi.Ctx(0)
    cdb, ctax = capi.Db(pairs=raw, key_ct=len(sk), k=K, offsets=off, nt=10), capi.Tax(ids=tids, parents=tpar)
    ctx.load_db(cdb)
    ctx.set_taxonomy(ctax)
    ctx._keep += [cdb, ctax, raw, off]
    return ctx, genome


@pytest.mark.parametrize("read_len,every", [(3000, 3), (9000, 2), (700, 5)])
def test_reads_whose_runs_outgrow_the_wave_chunk(read_len, every, monkeypatch):
    """thousands of runs per read: the windowed instance claims larger and larger chunks and takes the read's earlier runs
    along; then the same batch with a run array that is too small: redone through the per-k-mer array"""
    rng = np.random.default_rng(read_len)
    ctx, genome = many_taxa_db(rng, every)
    reads, _ = synth.sample_reads({1: genome}, 200, read_len, rng, frac_random=0.05)
    reads += [r[:150] for r in reads[:100]] + [r[:400] for r in reads[:50]]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    buf, off, lens = ko.pack_reads(reads)
    r, ref = check_rle_against_per_kmer(ctx, buf, off, lens, [f"q{i}" for i in range(len(reads))])
    assert int(r["run_cnt"].max()) > (256 if read_len >= 3000 else 100)
    monkeypatch.setenv("KU_RUNS_CAP", "5000")
    r2, _ = check_rle_against_per_kmer(ctx, buf, off, lens)
    assert int(r2["run_cnt"].sum()) == int(r["run_cnt"].sum())


def test_batch_uploaded_in_segments(monkeypatch):
    """a batch of more than 16 MB goes up in segments on the copy stream while the first ones are classified: same
    results as in one piece"""
    rng = np.random.default_rng(77)
    db = random_db(rng, n_genomes=8, glen=9000, nt=13)
    odb, otax, ctx = build(db, 13)
    few, _ = synth.sample_reads(db["genomes"], 3000, 150, rng, n_rate=0.002)
    few += [b"", b"ACGT", few[0][:40]]
    reads = few * 45  # ~20 MB
    buf, off, lens = ko.pack_reads(reads)
    assert len(buf) > (16 << 20)
    ctx.reset_counts()
    r = ctx.classify_batch_rle(buf, off, lens)
    c = ctx.counts()
    monkeypatch.setenv("KU_NO_H2D_OVERLAP", "1")
    ctx.reset_counts()
    one = ctx.classify_batch_rle(buf, off, lens)
    c1 = ctx.counts()
    monkeypatch.delenv("KU_NO_H2D_OVERLAP")
    assert np.array_equal(r["calls"], one["calls"]) and np.array_equal(r["run_cnt"], one["run_cnt"])
    for key in ("n_kmers", "registers", "n_reads"):
        assert np.array_equal(c[key], c1[key]), key
    # the first copy of the read set against the oracle, every other copy against the first
    n1 = len(few)
    run, res, buf1, off1, lens1, taxa1 = oracle_flat(odb, otax, few)
    got = decode_runs({"runs": r["runs"], "run_off": r["run_off"][:n1], "run_cnt": r["run_cnt"][:n1]}, off1, lens1, len(buf1))
    m = np.zeros(len(buf1), dtype=bool)
    for o, l in zip(off1.tolist(), lens1.tolist()):
        if l >= K:
            m[o:o + l - K + 1] = True
    assert np.array_equal(got[m], taxa1[:len(buf1)][m])
    assert np.array_equal(r["calls"][:n1], res["calls"])
    calls = r["calls"].reshape(45, n1)
    assert (calls == calls[0]).all()
    cnt = r["run_cnt"].reshape(45, n1)
    assert (cnt == cnt[0]).all()
    # the unpipelined, per-k-mer-array path (KU_NO_FUSED_RLE) agrees as well
    monkeypatch.setenv("KU_NO_FUSED_RLE", "1")
    ctx.reset_counts()
    old = ctx.classify_batch_rle(buf, off, lens)
    assert np.array_equal(old["calls"], r["calls"]) and np.array_equal(old["run_cnt"], r["run_cnt"])
    c2 = ctx.counts()
    for key in ("n_kmers", "registers", "n_reads"):
        assert np.array_equal(c[key], c2[key]), key


def test_no_counts_and_empty_batches():
    rng = np.random.default_rng(3)
    db = random_db(rng, n_genomes=4, glen=4000, nt=13)
    odb, otax, ctx = build(db, 13)
    reads, _ = synth.sample_reads(db["genomes"], 300, 150, rng)
    buf, off, lens = ko.pack_reads(reads)
    ctx.reset_counts()
    ref = ctx.classify_batch_rle(buf, off, lens)
    ctx.reset_counts()
    r = ctx.classify_batch_rle(buf, off, lens, flags=capi.KU_F_NO_COUNTS)
    assert np.array_equal(r["calls"], ref["calls"]) and np.array_equal(r["run_cnt"], ref["run_cnt"])
    assert int(ctx.counts()["n_kmers"].sum()) == 0 and int(ctx.counts()["n_reads"].sum()) == 0
    e = ctx.classify_batch_rle(b"", np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    assert len(e["calls"]) == 0 and len(e["runs"]) == 0
    short = [b"ACGT", b"", b"NNNN"]
    bs, os_, ls = ko.pack_reads(short)
    z = ctx.classify_batch_rle(bs, os_, ls)
    assert not z["calls"].any() and not z["run_cnt"].any()
