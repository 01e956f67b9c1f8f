"""Shared helpers for the -m gpu parity tests (HIP path vs the CPU oracle)."""
import os

import numpy as np

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko


def oracle_flat(odb, otax, seqs, **kw):
    """Run the oracle and lay its per-k-mer codes out like the C ABI does (parallel to the sequence buffer)."""
    run = ko.Run(odb, otax, **kw)
    res = run.classify(seqs)
    buf, off, lens = ko.pack_reads(seqs)
    taxa = np.zeros(max(len(buf), 1), dtype=np.uint32)
    for i in range(len(seqs)):
        a, n = int(res["taxa_off"][i]), int(res["n_slots"][i])
        t = res["taxa"][a:a + n].copy()
        t[res["ambig"][a:a + n] != 0] = capi.KU_AMBIG
        taxa[int(off[i]):int(off[i]) + n] = t
    return run, res, buf, off, lens, taxa


def valid_mask(off, lens, k, n_bytes):
    """positions of the taxa[] array that carry a k-mer code"""
    m = np.zeros(n_bytes, dtype=bool)
    for o, l in zip(off.tolist(), lens.tolist()):
        if l >= k:
            m[o:o + l - k + 1] = True
    return m


def assert_same_classification(gpu, res, taxa_want, off, lens, k, quick=False):
    assert (gpu["calls"] == res["calls"]).all(), np.nonzero(gpu["calls"] != res["calls"])[0][:10]
    if quick:
        assert (gpu["hits"] == res["hits"]).all()
        return
    m = valid_mask(off, lens, k, len(gpu["taxa"]))
    bad = np.nonzero(gpu["taxa"][m] != taxa_want[:len(m)][m])[0]
    assert len(bad) == 0, (len(bad), bad[:10])


def assert_same_counts(ctx_counts, run):
    """n_kmers / n_reads / HLL registers per taxon: bit-exact against the oracle (dense view of its sketches)."""
    want = run.counts()
    got_k = {int(t): (int(n), r) for t, n, r in zip(ctx_counts["slot_taxid"], ctx_counts["n_kmers"],
                                                    ctx_counts["registers"]) if n}
    got_r = {int(t): int(n) for t, n in zip(ctx_counts["node_taxid"], ctx_counts["n_reads"]) if n}
    assert got_r == {t: c["n_reads"] for t, c in want.items() if c["n_reads"]}
    assert {t: v[0] for t, v in got_k.items()} == {t: c["n_kmers"] for t, c in want.items() if c["n_kmers"]}
    for t, (n, regs) in got_k.items():
        assert (regs == want[t]["sketch"].registers()).all(), t
    # registers of slots that saw no k-mer stay zero
    for n, r in zip(ctx_counts["n_kmers"], ctx_counts["registers"]):
        if n == 0:
            assert not r.any()


def make_ctx(db_dir=None, cdb=None, ctax=None, shard=None, all_values=None):
    cdb = cdb or capi.Db(f"{db_dir}/database.kdb", f"{db_dir}/database.idx")
    ctax = ctax or capi.Tax(f"{db_dir}/taxDB")
    ctx = capi.Ctx(0)
    if shard:
        ctx.load_db(cdb, shard[0], shard[1])
    else:
        ctx.load_db(cdb)
    ctx.set_taxonomy(ctax, all_values)
    ctx._keep += [cdb, ctax]
    return ctx, cdb, ctax


def random_db(rng, n_genomes=6, glen=4000, k=31, nt=9, tax=None):
    """Small synthetic DB with shared k-mers between siblings; returns dict with numpy arrays + genomes."""
    tax = tax or synth.random_taxonomy(n_genomes, rng, levels=(2, 3, 4, 5))
    genomes = {}
    base = {}
    for i, tid in enumerate(tax.species):
        par = tax.parent[tid]
        if par not in base:
            base[par] = synth.procedural_genome(int(rng.integers(1, 1 << 30)), i, glen)
        genomes[tid] = synth.mutate(base[par], 0.03, rng)
    kmers, vals = synth.lca_database(genomes, tax, k)
    sk, sv, off = synth.sort_db(kmers, vals, k, nt)
    pairs = synth.pack_pairs(sk, sv)
    return {"tax": tax, "genomes": genomes, "kmers": sk, "vals": sv, "offsets": off, "pairs": pairs, "k": k, "nt": nt}
