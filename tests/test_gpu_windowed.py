"""-m gpu: the fused kernel's windowed variant (ku_short.hip, WIN = true): reads longer than 192 k-mers (mate pairs, long
reads up to 65535 k-mers) go through the wave-per-read kernel in windows of 128 k-mer positions, hit counts
accumulated across the windows.  Against the oracle bit for bit, and against the flat lookup + resolve kernels
(KU_NO_WINDOWED=1); the window placement (skipping k-mers known to be ambiguous), the single-taxon / LDS-table /
global spill-table stages of the hit accumulation and the accounting-free instance are each driven on purpose."""
import numpy as np
import pytest

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko

from gpu_common import assert_same_classification, assert_same_counts, make_ctx, oracle_flat, random_db

pytestmark = pytest.mark.gpu
K = 31


def build(db, nt):
    ids, par = db["tax"].arrays()
    raw = db["pairs"].view(np.uint8)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=nt)
    otax = ko.Tax(ids=ids, parents=par)
    ctx, _, _ = make_ctx(cdb=capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=K, offsets=db["offsets"], nt=nt),
                         ctax=capi.Tax(ids=ids, parents=par))
    return odb, otax, ctx


def with_n(read, positions):
    r = bytearray(read)
    for p in positions:
        if 0 <= p < len(r):
            r[p] = ord("N")
    return bytes(r)


@pytest.mark.parametrize("nt", [13, 10])  # compile-time geometry (k = 31, nt = 13) and the generic instance
def test_mixed_lengths_and_ambiguous_bases_around_the_window_edges(nt, monkeypatch):
    rng = np.random.default_rng(nt)
    db = random_db(rng, n_genomes=8, glen=9000, nt=nt)
    odb, otax, ctx = build(db, nt)
    g = list(db["genomes"].values())
    reads = []
    for L in (223, 224, 250, 285, 286, 287, 301, 415, 1000, 2600, 6000):
        rs, _ = synth.sample_reads(db["genomes"], 12, L, rng, n_rate=0.002)
        reads += rs
    # chimeras: several taxa per read, the second taxon first met in a later window
    for i in range(40):
        a, b, c = (g[int(rng.integers(0, len(g)))] for _ in range(3))
        reads.append(synth.codes_to_ascii(np.concatenate([a[:400], b[100:350], c[200:900]])))
    # mate pairs 2 x 150 joined by N (read_merger.pl): two windows, the k-mers across the joint are skipped
    for i in range(60):
        a = g[int(rng.integers(0, len(g)))]
        s = int(rng.integers(0, len(a) - 500))
        reads.append(synth.codes_to_ascii(a[s:s + 150]) + b"N" + synth.codes_to_ascii(synth.revcomp_codes(a[s + 250:s + 400])))
    # single N / runs of N at every base around the first two window edges, at the read's end, and long runs
    base, _ = synth.sample_reads(db["genomes"], 1, 700, rng)
    for p in list(range(120, 165)) + list(range(250, 295)) + [0, 30, 31, 668, 669, 699]:
        reads.append(with_n(base[0], [p]))
    for p, run in ((100, 40), (127, 31), (128, 30), (140, 200), (157, 2), (158, 1), (380, 129), (0, 700), (600, 100)):
        reads.append(with_n(base[0], range(p, p + run)))
    reads += [b"", b"ACGT", base[0][:30], base[0][:31], b"N" * 300]
    run, res, buf, off, lens, taxa = oracle_flat(odb, otax, reads)
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, K)
    assert_same_counts(ctx.counts(), run)
    # the flat lookup + resolve kernels on the same batch
    monkeypatch.setenv("KU_NO_WINDOWED", "1")
    ctx.reset_counts()
    staged = ctx.classify_batch(buf, off, lens)
    assert np.array_equal(staged["calls"], gpu["calls"])
    assert_same_counts(ctx.counts(), run)
    monkeypatch.delenv("KU_NO_WINDOWED")
    # no accounting (KU_F_NO_COUNTS): same calls and codes, state untouched
    ctx.reset_counts()
    nc = ctx.classify_batch(buf, off, lens, flags=capi.KU_F_NO_COUNTS)
    assert_same_classification(nc, res, taxa, off, lens, K)
    assert int(ctx.counts()["n_kmers"].sum()) == 0 and int(ctx.counts()["n_reads"].sum()) == 0


@pytest.mark.parametrize("read_len,every", [(3000, 3), (9000, 2), (1500, 40)])
def test_many_distinct_taxa_per_read_spill_to_the_global_table(read_len, every):
    """a taxon change every few k-mers over a pool of 700 taxa: reads meet far more distinct taxa than the wave's LDS
    table takes (96 + a window's worth), the counts move to the wave's spill region; equal scores and LCA folds included"""
    rng = np.random.default_rng(read_len)
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
    odb = ko.Db(pairs=raw, key_ct=len(sk), k=K, offsets=off, nt=10)
    otax = ko.Tax(ids=tids, parents=tpar)
    ctx, _, _ = make_ctx(cdb=capi.Db(pairs=raw, key_ct=len(sk), k=K, offsets=off, nt=10),
                         ctax=capi.Tax(ids=tids, parents=tpar))
    reads, _ = synth.sample_reads({1: genome}, 300, read_len, rng, frac_random=0.05)
    reads += [r[:400] for r in reads[:50]]  # short neighbours: the spill region is left clean for the next read
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    run, res, buf, off_r, lens, taxa = oracle_flat(odb, otax, reads)
    distinct = [len(set(res["taxa"][int(res["taxa_off"][i]):int(res["taxa_off"][i]) + int(res["n_slots"][i])].tolist()) - {0})
                for i in range(len(reads))]
    if every < 40:
        assert max(distinct) > 300
    for rep in range(2):  # the second pass finds the spill regions used
        gpu = ctx.classify_batch(buf, off_r, lens)
        assert_same_classification(gpu, res, taxa, off_r, lens, K)
    ctx.reset_counts()
    ctx.classify_batch(buf, off_r, lens)
    assert_same_counts(ctx.counts(), run)


def test_reads_beyond_the_windowed_limit_take_the_staged_kernels():
    rng = np.random.default_rng(9)
    db = random_db(rng, n_genomes=4, glen=80000, nt=10)
    odb, otax, ctx = build(db, 10)
    g = list(db["genomes"].values())
    reads = [synth.codes_to_ascii(g[0][:65535 + K - 1]), synth.codes_to_ascii(g[1][:65536 + K - 1]), synth.codes_to_ascii(g[2][:500])]
    for sub in (reads[:1] + reads[2:], reads):  # 65535 k-mers: windowed; 65536: flat lookup + resolve
        run, res, buf, off, lens, taxa = oracle_flat(odb, otax, sub)
        ctx.reset_counts()
        gpu = ctx.classify_batch(buf, off, lens)
        assert_same_classification(gpu, res, taxa, off, lens, K)
        assert_same_counts(ctx.counts(), run)
