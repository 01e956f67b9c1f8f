"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and against the
golden outputs captured from the compiled reference.  Bit-exact for everything
integer: calls, per-k-mer codes, hit strings, n_kmers, n_reads, HLL registers."""
import os

import numpy as np
import pytest

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko

from gpu_common import (assert_same_classification, assert_same_counts, make_ctx, oracle_flat, random_db,
                        valid_mask)

pytestmark = pytest.mark.gpu
K = 31


@pytest.fixture(scope="module")
def f1(golden):
    d = os.path.join(golden, "f1")
    ctx, cdb, ctax = make_ctx(d)
    return {"dir": d, "ctx": ctx, "cdb": cdb, "ctax": ctax,
            "odb": ko.Db(f"{d}/database.kdb", f"{d}/database.idx"), "otax": ko.Tax(f"{d}/taxDB")}


def rows(text):
    return sorted(text.strip("\n").split("\n"))


def test_loaded_native_library():
    assert capi.lib().ku_device_count() >= 1
    assert os.path.exists(capi.LIB_PATH)


def test_f1_matches_reference_output_and_oracle_state(f1):
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(f"{d}/reads.fq")
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs)
    ctx = f1["ctx"]
    ctx.reset_counts()
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, K)
    # the reference's own output file, byte for byte
    assert capi.format_kraken(buf, off, lens, ids, K, gpu["calls"], taxa=gpu["taxa"]) == open(f"{d}/out.tsv").read()
    counts = ctx.counts()
    assert_same_counts(counts, run)
    # report: identical to the oracle run with dense sketches; reference's report within 3 sigma on kmers
    ko.set_hll_sparse(False)
    try:
        run_d = ko.Run(f1["odb"], f1["otax"])
        run_d.classify(seqs)
        want = run_d.report(f"{d}/taxDB", f"{d}/database.kdb.counts")
    finally:
        ko.set_hll_sparse(True)
    got = capi.report(f1["ctax"], counts, f"{d}/database.kdb.counts")
    assert got == want
    ref = {ln.split("\t")[6]: ln.split("\t") for ln in open(f"{d}/report.tsv").read().strip().split("\n")}
    for ln in got.strip().split("\n"):
        f = ln.split("\t")
        r = ref[f[6]]
        assert f[:3] == r[:3] and f[6:] == r[6:]
        if f[3] != "kmers":
            assert abs(int(f[3]) - int(r[3])) <= max(2, 3 * 0.01625 * int(r[3]))


def test_f1_counts_accumulate_over_batches(f1):
    """two batches == one batch (the global taxon_counts merge, classify.cpp:541-544)"""
    ids, seqs = synth.read_seqfile(f"{f1['dir']}/reads.fq")
    ctx = f1["ctx"]
    ctx.reset_counts()
    for part in (seqs[:300], seqs[300:]):
        buf, off, lens = ko.pack_reads(part)
        ctx.classify_batch(buf, off, lens)
    run = ko.Run(f1["odb"], f1["otax"])
    run.classify(seqs)
    assert_same_counts(ctx.counts(), run)


def test_f1_quick_mode(f1):
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(f"{d}/reads.fq")
    for mh in (1, 2, 5):
        run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs, quick=True, min_hits=mh)
        ctx = f1["ctx"]
        ctx.reset_counts()
        gpu = ctx.classify_batch(buf, off, lens, flags=capi.KU_F_QUICK, min_hits=mh)
        assert_same_classification(gpu, res, taxa, off, lens, K, quick=True)
        assert_same_counts(ctx.counts(), run)
        if mh == 2:
            assert capi.format_kraken(buf, off, lens, ids, K, gpu["calls"], hits=gpu["hits"],
                                      flags=capi.KU_P_QUICK) == open(f"{d}/out_quick.tsv").read()


def test_f2_edge_reads(golden, f1):
    d = os.path.join(golden, "f2")
    ids, seqs = synth.read_seqfile(f"{d}/edge.fa")
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs)
    ctx = f1["ctx"]
    ctx.reset_counts()
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, K)
    assert capi.format_kraken(buf, off, lens, ids, K, gpu["calls"], taxa=gpu["taxa"]) == open(f"{d}/out.tsv").read()
    assert_same_counts(ctx.counts(), run)


def test_f4_paired_merged(golden, f1):
    d = os.path.join(golden, "f4")
    ids, seqs = synth.read_seqfile(f"{d}/merged.fa")
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs)
    ctx = f1["ctx"]
    ctx.reset_counts()
    gpu = ctx.classify_batch(buf, off, lens)
    assert capi.format_kraken(buf, off, lens, ids, K, gpu["calls"], taxa=gpu["taxa"]) == open(f"{d}/out.tsv").read()
    assert_same_counts(ctx.counts(), run)


def test_f7_legacy_unscrambled_index(golden, f1):
    d = os.path.join(golden, "f7")
    ctx, _, _ = make_ctx(d)
    ids, seqs = synth.read_seqfile(f"{f1['dir']}/reads.fq")
    buf, off, lens = ko.pack_reads(seqs)
    gpu = ctx.classify_batch(buf, off, lens)
    assert capi.format_kraken(buf, off, lens, ids, K, gpu["calls"], taxa=gpu["taxa"]) == open(f"{d}/out.tsv").read()


def test_count_taxons_equals_counts_file(f1):
    t, c = f1["ctx"].count_taxons()
    want = open(f"{f1['dir']}/database.kdb.counts").read()
    assert "".join(f"{a}\t{b}\n" for a, b in zip(t.tolist(), c.tolist())) == want


def test_db_values(f1):
    _, vals, *_ = synth.read_db(f1["dir"])
    want = np.unique(vals[vals != 0])
    assert (f1["ctx"].db_values() == want).all()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_sharded_lookup_merges_to_unsharded(f1, n_shards):
    """Minimizer-range shards (the 8-GPU layout, here one after another on one GPU): per-k-mer slots merged with
    max (exactly one shard is non-zero, classify.cpp:447), resolve once, owner-computes counts summed."""
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(f"{d}/reads.fq")
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs)
    import torch
    dev = torch.device("cuda:0")
    bounds = f1["cdb"].shard_plan(n_shards)
    all_values = f1["ctx"].db_values()
    t_seq = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    merged = None
    ctxs = []
    for s in range(n_shards):
        ctx, _, _ = make_ctx(cdb=f1["cdb"], ctax=f1["ctax"], shard=(int(bounds[s]), int(bounds[s + 1])),
                             all_values=all_values)
        t_taxa = torch.zeros(len(buf), dtype=torch.int32, device=dev)
        ctx.lookup_device(t_seq.data_ptr(), len(buf), t_taxa.data_ptr(), flags=capi.KU_F_KEEP_SLOTS)
        ctx.synchronize()
        u = t_taxa.view(torch.int32).cpu().numpy().view(np.uint32)
        merged = u if merged is None else np.maximum(merged, u)
        ctxs.append(ctx)
    # exactly one shard may be non-zero per k-mer -> max == the owner's value; resolve on shard 0's context
    t_m = torch.from_numpy(merged.view(np.int32)).to(dev)
    t_calls = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    ctxs[0].resolve_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), len(lens), t_calls.data_ptr(),
                           t_m.data_ptr(), max_read_len=int(lens.max()))
    ctxs[0].synchronize()
    gpu = {"calls": t_calls.cpu().numpy().view(np.uint32), "taxa": t_m.cpu().numpy().view(np.uint32)}
    assert_same_classification(gpu, res, taxa, off, lens, K)
    # reduce the per-shard state: max on registers, sum on counters
    cs = [c.counts() for c in ctxs]
    tot = dict(cs[0])
    tot["registers"] = np.maximum.reduce([c["registers"] for c in cs])
    tot["n_kmers"] = np.sum([c["n_kmers"] for c in cs], axis=0)
    tot["n_reads"] = np.sum([c["n_reads"] for c in cs], axis=0)
    assert_same_counts(tot, run)


def test_device_api_with_torch_buffers(f1):
    import torch
    dev = torch.device("cuda:0")
    ids, seqs = synth.read_seqfile(f"{f1['dir']}/reads.fq")
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs)
    t_seq = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    t_taxa = torch.zeros(len(buf), dtype=torch.int32, device=dev)
    t_calls = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    ctx = f1["ctx"]
    ctx.reset_counts()
    stream = torch.cuda.current_stream().cuda_stream
    ctx.classify_batch_device(t_seq.data_ptr(), len(buf), t_off.data_ptr(), t_len.data_ptr(), len(lens),
                              t_calls.data_ptr(), t_taxa.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    gpu = {"calls": t_calls.cpu().numpy().view(np.uint32), "taxa": t_taxa.cpu().numpy().view(np.uint32)}
    assert_same_classification(gpu, res, taxa, off, lens, K)
    assert_same_counts(ctx.counts(), run)


@pytest.mark.parametrize("k,nt", [(31, 9), (31, 13), (31, 15), (25, 6), (21, 12), (31, 1)])
def test_random_db_geometries(k, nt):
    """other (k, nt): key_len < 8 repack path (k=21, 25), nt=15 (30-bit minimizers), degenerate nt=1"""
    rng = np.random.default_rng(100 * k + nt)
    db = random_db(rng, k=k, nt=nt, glen=2500)
    raw = np.zeros(len(db["kmers"]) * ((2 * k + 7) // 8 + 4), dtype=np.uint8)
    kl = (2 * k + 7) // 8
    rec = raw.reshape(-1, kl + 4)
    kb = db["kmers"].astype("<u8").view(np.uint8).reshape(-1, 8)
    rec[:, :kl] = kb[:, :kl]
    rec[:, kl:] = db["vals"].astype("<u4").view(np.uint8).reshape(-1, 4)
    ids, par = db["tax"].arrays()
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=k, offsets=db["offsets"], nt=nt)
    otax = ko.Tax(ids=ids, parents=par)
    cdb = capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=k, offsets=db["offsets"], nt=nt)
    ctax = capi.Tax(ids=ids, parents=par)
    ctx, _, _ = make_ctx(cdb=cdb, ctax=ctax)
    reads, _ = synth.sample_reads(db["genomes"], 400, 150, rng, n_rate=0.002)
    reads += [b"", b"ACGT", reads[0][:k], reads[1][:k - 1]]
    run, res, buf, off, lens, taxa = oracle_flat(odb, otax, reads)
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, k)
    assert_same_counts(ctx.counts(), run)


def test_long_and_huge_reads(f1):
    """10 kbp reads (block-per-read resolve, LDS table) and one > 12288-k-mer read (global-memory table)"""
    rng = np.random.default_rng(5)
    db = random_db(rng, n_genomes=12, glen=30000, nt=10)
    ids, par = db["tax"].arrays()
    raw = db["pairs"].view(np.uint8)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=31, offsets=db["offsets"], nt=10)
    otax = ko.Tax(ids=ids, parents=par)
    ctx, _, _ = make_ctx(cdb=capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=31, offsets=db["offsets"], nt=10),
                         ctax=capi.Tax(ids=ids, parents=par))
    reads, _ = synth.sample_reads(db["genomes"], 24, 10000, rng, frac_random=0.1)
    # chimeras touch many taxa; one read longer than 12288 k-mers
    g = list(db["genomes"].values())
    chim = b"".join(synth.codes_to_ascii(x[i * 700:(i + 1) * 700]) for i, x in enumerate(g))
    reads += [chim, chim + reads[0] + chim[::-1], reads[1][:400], reads[2][:150], b"ACGTN" * 10]
    run, res, buf, off, lens, taxa = oracle_flat(odb, otax, reads)
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, K)
    assert_same_counts(ctx.counts(), run)


def _decode_runs(rle, off, lens, n_bytes, k=K):
    """runs -> flat per-k-mer codes laid out like ku_classify_batch's taxa (positions past n_kmers left 0)"""
    taxa = np.zeros(n_bytes, dtype=np.uint32)
    for i, (o, L) in enumerate(zip(off.tolist(), lens.tolist())):
        n = max(L - k + 1, 0)
        a, c = int(rle["run_off"][i]), int(rle["run_cnt"][i])
        rr = rle["runs"][a:a + c]
        assert (c == 0) == (n == 0)
        if c:
            assert rr[0, 1] == 0 and np.all(np.diff(rr[:, 1].astype(np.int64)) > 0) and rr[-1, 1] < n
            assert np.all(rr[1:, 0] != rr[:-1, 0])  # maximal runs
            taxa[o:o + n] = np.repeat(rr[:, 0], np.diff(np.r_[rr[:, 1], n]))
    return taxa


@pytest.mark.parametrize("fixture,reads", [("f1", "f1/reads.fq"), ("f2", "f2/edge.fa"), ("f4", "f4/merged.fa")])
def test_rle_output_path_matches_reference_output(golden, f1, fixture, reads):
    """ku_classify_batch_rle + ku_fetch_runs + ku_format_kraken_rle: the reference's Kraken file byte for byte, and
    the same per-taxon state as the per-k-mer path"""
    ids, seqs = synth.read_seqfile(os.path.join(golden, reads))
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs)
    ctx = f1["ctx"]
    ctx.reset_counts()
    rle = ctx.classify_batch_rle(buf, off, lens)
    assert int(rle["run_cnt"].sum()) <= len(rle["runs"])  # the run array has unused entries between the waves' chunks
    assert capi.format_kraken_rle(buf, off, lens, ids, K, rle) == open(os.path.join(golden, fixture, "out.tsv")).read()
    dec = _decode_runs(rle, off, lens, len(taxa))
    m = valid_mask(off, lens, K, len(taxa))
    assert np.array_equal(dec[m], taxa[m]) and np.array_equal(rle["calls"], res["calls"])
    assert_same_counts(ctx.counts(), run)
    if fixture == "f1":
        q = ctx.classify_batch_rle(buf, off, lens, flags=capi.KU_F_QUICK | capi.KU_F_NO_COUNTS, min_hits=2)
        assert len(q["runs"]) == 0
        assert capi.format_kraken_rle(buf, off, lens, ids, K, q, flags=capi.KU_P_QUICK) == open(f"{f1['dir']}/out_quick.tsv").read()
    ctx.reset_counts()


def test_rle_long_reads_and_empty_batch(f1):
    """reads spanning many 64-k-mer strips, runs crossing strip boundaries, reads without k-mers, empty batch"""
    rng = np.random.default_rng(11)
    ids, seqs = synth.read_seqfile(f"{f1['dir']}/reads.fq")
    long_reads = [b"".join(seqs[i:i + 40]) for i in range(0, 400, 40)] + [b"ACGT" * 5, b"", seqs[0][:31], seqs[1][:30] + b"N" + seqs[1][31:95]]
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], long_reads)
    ctx = f1["ctx"]
    ctx.reset_counts()
    rle = ctx.classify_batch_rle(buf, off, lens)
    dec = _decode_runs(rle, off, lens, len(taxa))
    m = valid_mask(off, lens, K, len(taxa))
    assert np.array_equal(dec[m], taxa[m]) and np.array_equal(rle["calls"], res["calls"])
    names = [f"r{i}" for i in range(len(long_reads))]
    assert capi.format_kraken_rle(buf, off, lens, names, K, rle) == capi.format_kraken(buf, off, lens, names, K, res["calls"], taxa=taxa)
    empty = ctx.classify_batch_rle(b"", np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    assert len(empty["runs"]) == 0 and len(empty["calls"]) == 0
    ctx.reset_counts()


@pytest.mark.parametrize("order,layout", [("", "hash"), ("_swapped", "hash"), ("", "sorted")])
def test_f8_hierarchical_multi_db(golden, f1, monkeypatch, order, layout):
    """classify -d A -d B (classify.cpp:928-936): one lookup pass per database, first hit wins, accounted once"""
    if layout == "sorted":
        monkeypatch.setenv("KU_LAYOUT", "sorted")
    d8, d1 = os.path.join(golden, "f8"), f1["dir"]
    dirs = [d1, d8] if order == "" else [d8, d1]
    cdbs = [capi.Db(f"{x}/database.kdb", f"{x}/database.idx") for x in dirs]
    odbs = [ko.Db(f"{x}/database.kdb", f"{x}/database.idx") for x in dirs]
    ctx = capi.Ctx(0)
    ctx.load_db(cdbs[0])
    ctx.add_db(cdbs[1])
    with pytest.raises(capi.KuError):  # k differs (classify.cpp:199-208)
        rng = np.random.default_rng(3)
        other = random_db(rng, n_genomes=2, glen=300, k=25, nt=6)
        raw = np.zeros((len(other["kmers"]), 7 + 4), dtype=np.uint8)
        raw[:, :7] = other["kmers"].astype("<u8").view(np.uint8).reshape(-1, 8)[:, :7]
        raw[:, 7:] = other["vals"].astype("<u4").view(np.uint8).reshape(-1, 4)
        ctx.add_db(capi.Db(pairs=raw.reshape(-1), key_ct=len(other["kmers"]), k=25, offsets=other["offsets"], nt=6))
    ctx.set_taxonomy(f1["ctax"])
    with pytest.raises(capi.KuError):  # too late
        ctx.add_db(cdbs[1])
    ids, seqs = synth.read_seqfile(f"{d8}/reads.fq")
    run, res, buf, off, lens, taxa = oracle_flat(odbs[0], f1["otax"], seqs, extra_dbs=odbs[1:])
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, K)
    assert capi.format_kraken(buf, off, lens, ids, K, gpu["calls"], taxa=gpu["taxa"]) == open(f"{d8}/out{order}.tsv").read()
    counts = ctx.counts()
    assert_same_counts(counts, run)
    # the run-length encoded output path goes through the same passes
    ctx.reset_counts()
    rle = ctx.classify_batch_rle(buf, off, lens)
    assert capi.format_kraken_rle(buf, off, lens, ids, K, rle) == open(f"{d8}/out{order}.tsv").read()
    assert_same_counts(ctx.counts(), run)
    # one counts file per database, genome sizes add up (report identical to the dense-sketch oracle)
    cpaths = [f"{x}/database.kdb.counts" for x in dirs]
    for i, x in enumerate(dirs):
        t, c = ctx.count_taxons(i)
        assert "".join(f"{a}\t{b}\n" for a, b in zip(t.tolist(), c.tolist())) == open(f"{x}/database.kdb.counts").read()
    ko.set_hll_sparse(False)
    try:
        run_d = ko.Run(odbs[0], f1["otax"], extra_dbs=odbs[1:])
        run_d.classify(seqs)
        want = run_d.report(f"{d1}/taxDB", "\n".join(cpaths))
    finally:
        ko.set_hll_sparse(True)
    assert capi.report(f1["ctax"], counts, cpaths) == want
    if order == "":
        ctx.reset_counts()
        runq, resq, *_ = oracle_flat(odbs[0], f1["otax"], seqs, quick=True, min_hits=2, extra_dbs=odbs[1:])
        q = ctx.classify_batch(buf, off, lens, flags=capi.KU_F_QUICK, min_hits=2)
        assert capi.format_kraken(buf, off, lens, ids, K, q["calls"], hits=q["hits"], flags=capi.KU_P_QUICK) == open(f"{d8}/out_quick.tsv").read()
        assert_same_counts(ctx.counts(), runq)


@pytest.mark.parametrize("budget,layout,quick", [(70 << 10, "hash", False), (30 << 10, "hash", False),
                                                 (70 << 10, "sorted", False), (50 << 10, "hash", True)])
def test_out_of_core_chunked_run(golden, f1, monkeypatch, budget, layout, quick):
    """classify -x SIZE (krakendb.cpp:411-526, classify.cpp:566-791): the database streamed through HBM chunk by chunk
    over device-resident read batches == the run with the whole database resident == the reference's -x output"""
    if layout == "sorted":
        monkeypatch.setenv("KU_LAYOUT", "sorted")
    d = f1["dir"]
    cdb = capi.Db(f"{d}/database.kdb", f"{d}/database.idx")
    bounds = cdb.chunk_plan(budget).tolist()
    bounds[-1] = cdb.info.n_bins  # the bins behind the last chunk hold no pairs
    assert len(bounds) - 1 >= 2
    ctx = capi.Ctx(0)
    ctx.load_db(cdb, bounds[0], bounds[1])
    ctx.set_taxonomy(f1["ctax"], cdb.values())
    ids, seqs = synth.read_seqfile(f"{d}/reads.fq")
    parts = [(0, 400), (400, 401), (401, 1000)]  # several resident batches, one of a single read
    packed = [ko.pack_reads(seqs[a:b]) for a, b in parts]
    batches = [ctx.batch(*p) for p in packed]
    flags = capi.KU_F_QUICK if quick else 0
    for c in range(len(bounds) - 1):
        if c:
            ctx.swap_shard(cdb, bounds[c], bounds[c + 1])
        for b in batches:
            b.lookup(flags=flags, min_hits=2)
    text = ""
    calls = []
    for (a, e), p, b in zip(parts, packed, batches):
        res = b.finish(flags=flags, min_hits=2)
        calls.append(res["calls"])
        text += capi.format_kraken_rle(p[0], p[1], p[2], ids[a:e], K, res, flags=capi.KU_P_QUICK if quick else 0)
        with pytest.raises(capi.KuError):  # a finished batch takes no further passes
            b.lookup()
        b.close()
    if quick:
        # the reference's chunked run has its own quick mode (classify.cpp:686-737): every k-mer booked, "Q:n" capped at
        # min_hits, call = taxon of the read's last unambiguous k-mer -> its own golden files; per-taxon state = the
        # full (non-quick) accounting except for n_reads
        assert text == open(f"{d}/out_chunk_quick.tsv").read()
        run, res, *_ = oracle_flat(f1["odb"], f1["otax"], seqs)
        got, want = ctx.counts(), run.counts()
        for t, nk, regs in zip(got["slot_taxid"], got["n_kmers"], got["registers"]):
            if nk:
                assert int(nk) == want[int(t)]["n_kmers"] and (regs == want[int(t)]["sketch"].registers()).all()
        called = {int(t): int(n) for t, n in zip(got["node_taxid"], got["n_reads"]) if n}
        ref_calls = [int(ln.split("\t")[2]) for ln in open(f"{d}/out_chunk_quick.tsv").read().strip().split("\n")]
        assert called == {t: ref_calls.count(t) for t in set(ref_calls)}
    else:
        run, res, *_ = oracle_flat(f1["odb"], f1["otax"], seqs)
        assert np.array_equal(np.concatenate(calls), res["calls"])
        assert text == open(f"{d}/out_chunk.tsv").read()
        assert_same_counts(ctx.counts(), run)
    # a slot table that does not cover the next shard is refused
    d8 = os.path.join(golden, "f8")
    ctx2 = capi.Ctx(0)
    ctx2.load_db(capi.Db(f"{d8}/database.kdb", f"{d8}/database.idx"))
    ctx2.set_taxonomy(f1["ctax"])  # slots for taxids 3 and 6 only
    with pytest.raises(capi.KuError):
        ctx2.swap_shard(cdb, bounds[0], bounds[1])


def test_exact_counting_reproduces_classify_exact(golden, f1):
    """classifyExact (EXACT_COUNTING build, classify.cpp:46-53): distinct k-mers counted exactly -> the reference's
    report_exact.tsv row for row; the sketches are filled as usual next to it"""
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(f"{d}/reads.fq")
    run, res, buf, off, lens, taxa = oracle_flat(f1["odb"], f1["otax"], seqs)
    ctx, _, _ = make_ctx(d)
    ctx.enable_exact(16)
    half = 500
    for a, b in ((0, half), (half, 1000)):  # the set persists across batches
        p = ko.pack_reads(seqs[a:b])
        gpu = ctx.classify_batch(*p)
        assert np.array_equal(gpu["calls"], res["calls"][a:b])
    counts = ctx.counts()
    assert_same_counts(counts, run)
    uniq = ctx.exact_counts()
    got = capi.report_exact(f1["ctax"], counts, uniq, [f"{d}/database.kdb.counts"])
    assert rows(got) == rows(open(f"{d}/report_exact.tsv").read())
    assert (uniq <= counts["n_kmers"]).all() and uniq.sum() > 50000
    with pytest.raises(capi.KuError):  # a whole classification only
        ctx.classify_batch(*ko.pack_reads(seqs[:10]), flags=capi.KU_F_NO_COUNTS)
    ctx.reset_counts()
    assert not ctx.exact_counts().any()
    # quick mode (-q -m 2): only the scanned prefix of a read is counted -> the reference's classifyExact -q report
    for a, b in ((0, half), (half, 1000)):
        gpu = ctx.classify_batch(*ko.pack_reads(seqs[a:b]), flags=capi.KU_F_QUICK, min_hits=2)
    got = capi.report_exact(f1["ctax"], ctx.counts(), ctx.exact_counts(), [f"{d}/database.kdb.counts"])
    assert rows(got) == rows(open(f"{d}/report_exact_quick.tsv").read())
    ctx.reset_counts()
    # a set that is too small is reported, not silently wrong
    ctx.enable_exact(10)
    ctx.classify_batch(*ko.pack_reads(seqs))
    with pytest.raises(capi.KuError):
        ctx.exact_counts()


def test_revcomp_invariance_property(f1):
    """size-independent property: a read and its reverse complement get the same call and mirrored hit list"""
    ids, seqs = synth.read_seqfile(f"{f1['dir']}/reads.fq")
    comp = bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN")
    rc = [s.translate(comp)[::-1] for s in seqs]
    ctx = f1["ctx"]
    b1, o1, l1 = ko.pack_reads(seqs)
    b2, o2, l2 = ko.pack_reads(rc)
    g1 = ctx.classify_batch(b1, o1, l1)
    g2 = ctx.classify_batch(b2, o2, l2)
    assert (g1["calls"] == g2["calls"]).all()
    for o, l in zip(o1.tolist(), l1.tolist()):
        if l >= K:
            n = l - K + 1
            assert (g1["taxa"][o:o + n] == g2["taxa"][o:o + n][::-1]).all()


def test_errors_are_reported_not_swallowed(f1):
    ctx = capi.Ctx(0)
    buf, off, lens = ko.pack_reads([b"ACGT" * 20])
    with pytest.raises(capi.KuError) as e:
        ctx.classify_batch(buf, off, lens)
    assert e.value.status == -6
    ctx.load_db(f1["cdb"])
    with pytest.raises(capi.KuError):
        ctx.classify_batch(buf, off, lens)  # taxonomy missing
    ctx.set_taxonomy(f1["ctax"])
    bad_len = lens.copy()
    bad_len[0] = 1000
    with pytest.raises(capi.KuError) as e:
        ctx.classify_batch(buf, off, bad_len)
    assert e.value.status == -1


def test_hash_table_spilled_buckets(monkeypatch):
    """load factor 0.95: many buckets spill into the following lines; every DB k-mer must still be found"""
    monkeypatch.setenv("KU_LOAD_FACTOR", "0.9")
    rng = np.random.default_rng(77)
    db = random_db(rng, n_genomes=8, glen=6000, nt=8)
    ids, par = db["tax"].arrays()
    raw = db["pairs"].view(np.uint8)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=31, offsets=db["offsets"], nt=8)
    otax = ko.Tax(ids=ids, parents=par)
    ctx, _, _ = make_ctx(cdb=capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=31, offsets=db["offsets"], nt=8),
                         ctax=capi.Tax(ids=ids, parents=par))
    reads = [synth.codes_to_ascii(g) for g in db["genomes"].values()]  # whole genomes: every k-mer is a DB hit
    reads += [synth.codes_to_ascii(synth.revcomp_codes(g)) for g in db["genomes"].values()]  # ... on both strands
    reads += synth.sample_reads(db["genomes"], 300, 150, rng)[0]
    run, res, buf, off, lens, taxa = oracle_flat(odb, otax, reads)
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, K)
    assert_same_counts(ctx.counts(), run)
    t, c = ctx.count_taxons()
    assert int(c.sum()) == len(db["kmers"])


@pytest.mark.parametrize("read_len,no_fused", [(100, False), (150, True), (200, False), (222, False), (223, False)])
def test_fused_short_read_kernel_and_staged_path_agree(monkeypatch, read_len, no_fused):
    """<= 128 k-mers -> fused<2>, <= 192 -> fused<3>, longer or KU_NO_FUSED -> flat lookup + resolve kernels"""
    if no_fused:
        monkeypatch.setenv("KU_NO_FUSED", "1")
    rng = np.random.default_rng(read_len)
    db = random_db(rng, n_genomes=8, glen=5000, nt=13)
    ids, par = db["tax"].arrays()
    raw = db["pairs"].view(np.uint8)
    odb = ko.Db(pairs=raw, key_ct=len(db["kmers"]), k=31, offsets=db["offsets"], nt=13)
    otax = ko.Tax(ids=ids, parents=par)
    ctx, _, _ = make_ctx(cdb=capi.Db(pairs=raw, key_ct=len(db["kmers"]), k=31, offsets=db["offsets"], nt=13),
                         ctax=capi.Tax(ids=ids, parents=par))
    reads, _ = synth.sample_reads(db["genomes"], 600, read_len, rng, n_rate=0.004)
    # chimeric reads give several distinct taxa per read (the LDS-table resolve path), plus degenerate lengths
    g = list(db["genomes"].values())
    for i in range(60):
        a, b, c = (g[int(rng.integers(0, len(g)))] for _ in range(3))
        reads.append(synth.codes_to_ascii(np.concatenate([a[:read_len // 3], b[100:100 + read_len // 3], c[200:200 + read_len // 3]])))
    reads += [b"", b"ACGT", reads[0][:31], reads[1][:30], b"N" * 50]
    run, res, buf, off, lens, taxa = oracle_flat(odb, otax, reads)
    gpu = ctx.classify_batch(buf, off, lens)
    assert_same_classification(gpu, res, taxa, off, lens, K)
    assert_same_counts(ctx.counts(), run)


@pytest.mark.parametrize("read_len", [150, 301, 1000])
def test_adversarial_taxonomy_and_ties(read_len):
    """DB values scattered over an irregular taxonomy (deep chain, orphans, a self-parent, taxids missing from
    taxDB, pseudo-taxids >= 1e9): reads see many distinct taxa, equal root-path scores and LCA folds -- every
    resolve path (ballot fast path, wave/block LDS table) against the oracle."""
    rng = np.random.default_rng(1000 + read_len)
    tax = synth.Taxonomy()
    tax.add(1, 1, "root", "root")
    ids = [1]
    for i in range(60):                       # random tree
        t = 10 + 3 * i
        tax.add(t, ids[int(rng.integers(0, len(ids)))], f"n{t}", "no rank")
        ids.append(t)
    chain = ids[-1]
    for i in range(45):                       # deep chain below the last node
        t = 1000 + i
        tax.add(t, chain, f"c{t}", "no rank")
        chain = t
        ids.append(t)
    tax.add(5000, 4242, "orphan", "species")      # parent id has no entry
    tax.add(5001, 5001, "selfparent", "species")  # handled as "no parent" (taxdb.hpp:421)
    tax.add(1000000007, ids[5], "pseudo", "sequence")
    ids += [5000, 5001, 1000000007]
    absent = [777, 888]                       # DB values that taxDB does not know
    genome = synth.procedural_genome(int(rng.integers(1, 1 << 30)), 1, 20000)
    kmers = np.unique(synth.canonical(synth.kmers_forward(genome, K), K))
    # taxon changes every ~40 k-mers along the genome order of first appearance -> reads span several taxa
    fwd = synth.canonical(synth.kmers_forward(genome, K), K)
    pool = np.array(ids + absent, dtype=np.uint32)
    seg = pool[rng.integers(0, len(pool), len(fwd) // 40 + 1)]
    val_of = {}
    for i, km in enumerate(fwd.tolist()):
        val_of.setdefault(km, int(seg[i // 40]))
    vals = np.array([val_of[int(x)] for x in kmers.tolist()], dtype=np.uint32)
    sk, sv, off = synth.sort_db(kmers, vals, K, 10)
    raw = synth.pack_pairs(sk, sv).view(np.uint8)
    tids, tpar = tax.arrays()
    odb = ko.Db(pairs=raw, key_ct=len(sk), k=K, offsets=off, nt=10)
    otax = ko.Tax(ids=tids, parents=tpar)
    ctx, _, _ = make_ctx(cdb=capi.Db(pairs=raw, key_ct=len(sk), k=K, offsets=off, nt=10),
                         ctax=capi.Tax(ids=tids, parents=tpar))
    reads, _ = synth.sample_reads({1: genome}, 500, read_len, rng, frac_random=0.1)
    run, res, buf, off_r, lens, taxa = oracle_flat(odb, otax, reads)
    gpu = ctx.classify_batch(buf, off_r, lens)
    assert_same_classification(gpu, res, taxa, off_r, lens, K)
    assert_same_counts(ctx.counts(), run)
    # the scenario really exercises ties / multi-taxon reads
    multi = sum(1 for i in range(len(reads))
                if len(set(res["taxa"][int(res["taxa_off"][i]):int(res["taxa_off"][i]) + int(res["n_slots"][i])].tolist()) - {0}) > 2)
    assert multi > len(reads) // 4
