"""The torch (on-device) DB generator used by bench.py against the numpy generator
(synth.py, itself pinned to the reference's db_sort output).  Runs on CPU tensors."""
import numpy as np
import torch

from krakenuniq_amd import synth, synth_torch as st


def test_bit_tricks_match_numpy():
    rng = np.random.default_rng(0)
    x = rng.integers(0, 2 ** 62, 4096, dtype=np.uint64)
    tx = torch.from_numpy(x.astype(np.int64))
    for n in (31, 25, 15, 13, 7):
        m = np.uint64((1 << (2 * n)) - 1)
        want = synth.canonical(x & m, n)
        got = st.canonical(tx & int(m), n).numpy().astype(np.uint64)
        assert (got == want).all(), n
    c = synth.canonical(x, 31)
    tc = torch.from_numpy(c.astype(np.int64))
    for nt in (7, 13, 15):
        assert (st.bin_key(tc, 31, nt).numpy().astype(np.uint64) == synth.bin_key(c, 31, nt)).all()
    assert (st.bin_key(tc, 31, 9, idx_type=1).numpy().astype(np.uint64) == synth.bin_key(c, 31, 9, 1)).all()
    a = rng.integers(0, 2 ** 63, 1000, dtype=np.uint64)
    assert (st.splitmix64(torch.from_numpy(a.astype(np.int64))).numpy().astype(np.uint64) == synth.splitmix64(a)).all()


def test_bench_db_equals_numpy_build():
    db = st.BenchDb(torch.device("cpu"), n_species=24, genome_len=1500, k=31, nt=8, seed=3, shared_frac=0.3,
                    species_chunk=7)
    genomes = {t: db.genomes[i].numpy() for i, t in enumerate(db.tax.species)}
    kmers, vals = synth.lca_database(genomes, db.tax, 31)
    sk, sv, off = synth.sort_db(kmers, vals, 31, 8)
    assert db.n_pairs == len(sk)
    assert (db.kmers.numpy().astype(np.uint64) == sk).all()
    assert (db.vals.numpy().astype(np.uint32) == sv).all()
    assert (db.offsets.numpy().astype(np.uint64) == off).all()
    # LCA values above species exist (siblings share the genus segment)
    sp = set(db.tax.species)
    assert any(int(v) not in sp for v in sv)
    # 12-byte records as on disk
    want = synth.pack_pairs(sk, sv).tobytes()
    assert db.pairs.numpy().tobytes() == want


def test_sharded_build_is_a_slice_of_the_full_build():
    full = st.BenchDb(torch.device("cpu"), n_species=12, genome_len=1200, k=31, nt=6, seed=5, shared_frac=0.2)
    lo, hi = 1000, 3000
    part = st.BenchDb(torch.device("cpu"), n_species=12, genome_len=1200, k=31, nt=6, seed=5, shared_frac=0.2,
                      bin_lo=lo, bin_hi=hi)
    a, b = int(full.offsets[lo]), int(full.offsets[hi])
    assert part.n_pairs == b - a
    assert (part.pairs == full.pairs[a:b]).all()
    assert (part.offsets + a == full.offsets[lo:hi + 1]).all()


def test_reads_layout():
    db = st.BenchDb(torch.device("cpu"), n_species=6, genome_len=2000, k=31, nt=6, seed=1)
    buf, off, ln, src = db.sample_reads(500, 150, seed=2)
    b = buf.numpy().reshape(500, 151)
    assert (b[:, 150] == 10).all() and set(np.unique(b[:, :150])) <= set(b"ACGTN")
    assert (off.numpy() == np.arange(500) * 151).all() and (ln.numpy() == 150).all()
    assert 0.1 < float((src == 0).float().mean()) < 0.3
