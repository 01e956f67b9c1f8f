"""CPU-only checks of the product library (no kernel launches): it loads, exports
every symbol include/krakenuniq_amd.h declares, and its host-side logic (format
parsers, shard planning, taxonomy, HLL estimator, Kraken line formatting, report)
agrees with the golden vectors captured from the reference."""
import json
import os
import re

import numpy as np
import pytest

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def f1(golden):
    d = os.path.join(golden, "f1")
    return {"dir": d, "db": capi.Db(f"{d}/database.kdb", f"{d}/database.idx"), "tax": capi.Tax(f"{d}/taxDB")}


def rows(text):
    return sorted(text.strip("\n").split("\n"))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "krakenuniq_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ku_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    assert L.ku_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    if capi.lib().ku_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.KuError) as e:
        capi.Ctx(0)
    assert e.value.status == -5  # KU_EHIP
    # the database build steps are device code too: no host stand-in
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f1")
    with pytest.raises(capi.KuError) as e:
        capi.db_sort_files(f"{d}/database.kdb", "/tmp/ku_never.kdb", "/tmp/ku_never.idx", 7)
    assert e.value.status == -5 and not os.path.exists("/tmp/ku_never.kdb")
    with pytest.raises(capi.KuError) as e:
        capi.SetLcas(capi.Db(f"{d}/database.kdb", f"{d}/database.idx"), capi.Tax(f"{d}/taxDB"))
    assert e.value.status == -5


def test_db_open_and_errors(f1, tmp_path):
    i = f1["db"].info
    assert (i.k, i.nt, i.idx_type, i.key_len, i.n_bins) == (31, 7, 2, 8, 4 ** 7)
    kmers, vals, off, *_ = synth.read_db(f1["dir"])
    assert i.key_ct == len(kmers)
    with pytest.raises(capi.KuError) as e:
        capi.Db(str(tmp_path / "nope.kdb"), f"{f1['dir']}/database.idx")
    assert e.value.status == -3
    bad = tmp_path / "bad.kdb"
    bad.write_bytes(b"NOTJELLY" + b"\0" * 2000)
    with pytest.raises(capi.KuError) as e:
        capi.Db(str(bad), f"{f1['dir']}/database.idx")
    assert e.value.status == -2
    badidx = tmp_path / "bad.idx"
    badidx.write_bytes(b"KRAKXXX\x07" + b"\0" * 64)
    with pytest.raises(capi.KuError) as e:
        capi.Db(f"{f1['dir']}/database.kdb", str(badidx))
    assert e.value.status == -2
    # a pair count whose byte size wraps around 2^64 to something small must not pass the truncation test
    img = bytearray(open(f"{f1['dir']}/database.kdb", "rb").read())
    img[48:56] = ((1 << 64) // 12 + 1).to_bytes(8, "little")
    wrap = tmp_path / "wrap.kdb"
    wrap.write_bytes(bytes(img))
    with pytest.raises(capi.KuError) as e:
        capi.Db(str(wrap), f"{f1['dir']}/database.idx")
    assert e.value.status == -2 and "truncated" in str(e.value)


def test_shard_plan_balanced_and_contiguous(f1):
    _, _, off, *_ = synth.read_db(f1["dir"])
    nb = 4 ** 7
    cost = lambda b: 8 * int(b) + 12 * int(off[int(b)])
    for n in (1, 2, 3, 8):
        b = f1["db"].shard_plan(n)
        assert b[0] == 0 and b[-1] == nb and (np.diff(b.astype(np.int64)) >= 0).all()
        sizes = [cost(b[i + 1]) - cost(b[i]) for i in range(n)]
        assert sum(sizes) == cost(nb)
        assert max(sizes) - min(sizes) <= 2 * (8 + 12 * int(np.diff(off.astype(np.int64)).max()))


def test_chunk_plan_matches_reference_rule(f1):
    """prepare_chunking (krakendb.cpp:463-522): every chunk's idx slice + pair slice + 8 fits the budget,
    adding one more bin would not, chunks without pairs are dropped."""
    _, _, off, *_ = synth.read_db(f1["dir"])
    budget = 70 * 1024
    b = f1["db"].chunk_plan(budget)
    assert b[0] == 0 and b[-1] <= 4 ** 7 and off[int(b[-1])] == off[-1]  # trailing pair-less bins are dropped
    lo = 0
    for hi in b[1:]:
        hi = int(hi)
        # chunks with zero pairs were merged away: recompute the reference's greedy walk
        assert off[hi] > off[lo]
        lo = hi
    # greedy walk restated
    pos, bounds = 0, [0]
    while pos < 4 ** 7:
        nxt = pos
        while nxt < 4 ** 7 and (nxt + 1 - pos) * 8 + (int(off[nxt + 1]) - int(off[pos])) * 12 + 8 <= budget:
            nxt += 1
        assert nxt > pos
        pos = nxt
        if int(off[pos]) != int(off[bounds[-1]]):
            bounds.append(pos)
    assert bounds == [int(x) for x in b]
    with pytest.raises(capi.KuError):
        f1["db"].chunk_plan(16)


def test_db_values_host_scan(golden, f1):
    """ku_db_values == the taxids of database.kdb.counts minus 0 (whole-database slot universe of a chunked run)"""
    for name in ("f1", "f8"):
        d = os.path.join(golden, name)
        db = capi.Db(f"{d}/database.kdb", f"{d}/database.idx")
        want = sorted(int(ln.split()[0]) for ln in open(f"{d}/database.kdb.counts") if ln.strip() and int(ln.split()[0]))
        assert db.values().tolist() == want
    # a larger wrapped database exercises the threaded scan
    rng = np.random.default_rng(3)
    n = 1 << 21
    pairs = np.zeros((n, 3), dtype=np.uint32)
    pairs[:, 0] = np.arange(n, dtype=np.uint32)
    vals = rng.choice(np.array([0, 7, 9, 1000000001, 4294967295, 12345], dtype=np.uint64), n).astype(np.uint32)
    pairs[:, 2] = vals
    off = np.zeros(4 ** 3 + 1, dtype=np.uint64)
    off[1:] = n
    big = capi.Db(pairs=pairs.reshape(-1).view(np.uint8), key_ct=n, k=31, offsets=off, nt=3)
    assert big.values().tolist() == [7, 9, 12345, 1000000001, 4294967295]


def test_taxonomy_parent_map(f1):
    tax, otax = f1["tax"], ko.Tax(f"{f1['dir']}/taxDB")
    for t in (0, 1, 2, 3, 4, 5, 6, 777, 1000000001, 12345):
        assert tax.parent(t) == otax.parent(t)
    kat = json.load(open(os.path.join(os.path.dirname(f1["dir"]), "kat.json")))
    pm = {int(a): b for a, b in kat["tree"]["parent_map"].items()}
    ids = np.array(list(pm.keys()), dtype=np.uint32)
    par = np.array([pm[int(t)] if pm[int(t)] else int(t) for t in ids], dtype=np.uint32)
    t2 = capi.Tax(ids=ids, parents=par)
    for t, p in pm.items():
        assert t2.parent(t) == p


def test_hll_estimator_matches_reference(golden):
    kat = json.load(open(os.path.join(golden, "kat.json")))
    MULT = 0x9E3779B97F4A7C15
    for c in kat["hll"]:
        if c["sparse_start"] == 0 and c["n"] <= 100000:
            h = ko.Hll(12, False)
            h.insert_seq(c["n"], MULT)
            got = capi.hll_cardinality(h.registers(), c["n"] if c["use_n"] else 1 << 62)
            assert got == c["ertl"], c
    for c in kat["hll_merge"] + kat["hll_state"]:
        if c["kind"] == "D":
            regs = np.array(c["state"], dtype=np.uint8)
            want = ko.lib().ko_ertl_from_registers(regs.ctypes.data_as(ko.u8p), 12, 1 << 62, 0)
            assert capi.hll_cardinality(regs, 1 << 62) == want
    assert capi.hll_cardinality(np.zeros(4096, dtype=np.uint8), 0) == 0


def _oracle_flat(f1dir, seqs, **kw):
    """Oracle results re-laid out as the C ABI lays them out (taxa parallel to the sequence buffer)."""
    db, tax = ko.Db(f"{f1dir}/database.kdb", f"{f1dir}/database.idx"), ko.Tax(f"{f1dir}/taxDB")
    run = ko.Run(db, tax, **kw)
    res = run.classify(seqs)
    buf, off, lens = ko.pack_reads(seqs)
    taxa = np.zeros(len(buf), dtype=np.uint32)
    for i in range(len(seqs)):
        a, n = int(res["taxa_off"][i]), int(res["n_slots"][i])
        t = res["taxa"][a:a + n].copy()
        t[res["ambig"][a:a + n] != 0] = capi.KU_AMBIG
        taxa[int(off[i]):int(off[i]) + n] = t
    return run, res, buf, off, lens, taxa


def test_format_kraken_reproduces_reference_output(f1):
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(f"{d}/reads.fq")
    run, res, buf, off, lens, taxa = _oracle_flat(d, seqs)
    got = capi.format_kraken(buf, off, lens, ids, 31, res["calls"], taxa=taxa)
    assert got == open(f"{d}/out.tsv").read()
    assert capi.format_kraken(buf, off, lens, ids, 31, res["calls"], taxa=taxa,
                              flags=capi.KU_P_ONLY_CLASSIFIED) == open(f"{d}/out_c.tsv").read()
    assert capi.format_kraken(buf, off, lens, ids, 31, res["calls"], taxa=taxa,
                              flags=capi.KU_P_SEQUENCE) == open(f"{d}/out_s.tsv").read()
    _, resq, *_ = _oracle_flat(d, seqs, quick=True, min_hits=2)
    assert capi.format_kraken(buf, off, lens, ids, 31, resq["calls"], hits=resq["hits"],
                              flags=capi.KU_P_QUICK) == open(f"{d}/out_quick.tsv").read()


def test_format_kraken_edge_reads(golden, f1):
    d = os.path.join(golden, "f2")
    ids, seqs = synth.read_seqfile(f"{d}/edge.fa")
    _, res, buf, off, lens, taxa = _oracle_flat(f1["dir"], seqs)
    assert capi.format_kraken(buf, off, lens, ids, 31, res["calls"], taxa=taxa) == open(f"{d}/out.tsv").read()
    assert capi.hitlist_string(np.zeros(0, dtype=np.uint32)) == "0:0"


def _rle(taxa, off, lens, calls, k=31, shuffle=False):
    """numpy model of ku_classify_batch_rle's output (runs in arbitrary per-read order when shuffle is set)."""
    per_read = []
    for o, L in zip(off.tolist(), lens.tolist()):
        n = max(L - k + 1, 0)
        t = taxa[o:o + n]
        starts = np.flatnonzero(np.r_[True, t[1:] != t[:-1]]) if n else np.zeros(0, dtype=np.int64)
        per_read.append(np.stack([t[starts], starts.astype(np.uint32)], axis=1).astype(np.uint32).reshape(-1, 2))
    order = list(range(len(per_read)))
    if shuffle:
        order = order[::-1]
    run_off, pos = np.zeros(len(per_read), dtype=np.uint64), 0
    for i in order:
        run_off[i] = pos
        pos += len(per_read[i])
    runs = np.concatenate([per_read[i] for i in order]) if per_read else np.zeros((0, 2), np.uint32)
    return {"calls": calls, "hits": np.zeros(len(lens), np.uint32), "runs": runs, "run_off": run_off,
            "run_cnt": np.array([len(x) for x in per_read], dtype=np.uint32)}


@pytest.mark.parametrize("fixture,reads", [("f1", "f1/reads.fq"), ("f2", "f2/edge.fa"), ("f4", "f4/merged.fa")])
def test_format_kraken_rle_reproduces_reference_output(golden, f1, fixture, reads):
    ids, seqs = synth.read_seqfile(os.path.join(golden, reads))
    _, res, buf, off, lens, taxa = _oracle_flat(f1["dir"], seqs)
    d = os.path.join(golden, fixture)
    for shuffle in (False, True):
        rle = _rle(taxa, off, lens, res["calls"], shuffle=shuffle)
        assert capi.format_kraken_rle(buf, off, lens, ids, 31, rle) == open(f"{d}/out.tsv").read()
    if fixture == "f1":
        assert capi.format_kraken_rle(buf, off, lens, ids, 31, rle, flags=capi.KU_P_ONLY_CLASSIFIED) == open(f"{d}/out_c.tsv").read()
        assert capi.format_kraken_rle(buf, off, lens, ids, 31, rle, flags=capi.KU_P_SEQUENCE) == open(f"{d}/out_s.tsv").read()
        _, resq, *_ = _oracle_flat(d, seqs, quick=True, min_hits=2)
        q = dict(rle, calls=resq["calls"], hits=resq["hits"])
        assert capi.format_kraken_rle(buf, off, lens, ids, 31, q, flags=capi.KU_P_QUICK) == open(f"{d}/out_quick.tsv").read()


def test_integers_of_every_width_in_the_hit_list():
    """taxids and run lengths are written two digits at a time: every digit count up to 2^32 - 2 (2^32 - 1 is KU_AMBIG, 'A')"""
    vals = [0, 1, 9, 10, 11, 99, 100, 101, 999, 1000, 9999, 10000, 65535, 99999, 100000, 999999, 1000000, 9999999, 10000000,
            99999999, 100000000, 999999999, 1000000000, 2147483647, 2147483648, 4294967294]
    rng = np.random.default_rng(3)
    vals += [int(x) for x in rng.integers(0, 2 ** 32 - 1, 200)]
    taxa, want = [], []
    for i, v in enumerate(vals):
        run = 1 + (i * 7) % 23
        taxa += [v] * run
        if want and want[-1][0] == v:
            want[-1][1] += run
        else:
            want.append([v, run])
    assert capi.hitlist_string(np.array(taxa, dtype=np.uint32)) == " ".join(f"{v}:{c}" for v, c in want)


def test_format_kraken_rle_equals_raw_on_random_codes():
    """randomised: any code sequence (taxids incl. > 2^31, 0, KU_AMBIG, runs crossing word sizes) formats the same from
    raw codes and from runs; -c / -s / quick flags included; both agree with the oracle's hitlist_string"""
    rng = np.random.default_rng(12)
    k = 31
    seqs, codes = [], []
    alphabet = np.array([0, 0, 5, 5, 77, 4000000000, capi.KU_AMBIG, 1, 1000000001], dtype=np.uint64)
    for i in range(400):
        L = int(rng.integers(0, 400)) if i % 7 else int(rng.integers(0, 31))
        seqs.append(synth.codes_to_ascii(rng.integers(0, 4, L, dtype=np.uint8)))
        n = max(L - k + 1, 0)
        run_lens = rng.geometric(0.2, size=n + 1)
        c = np.repeat(rng.choice(alphabet, size=n + 1), run_lens)[:n].astype(np.uint32)
        codes.append(c)
    buf, off, lens = ko.pack_reads(seqs)
    taxa = np.zeros(len(buf), dtype=np.uint32)
    for o, c in zip(off.tolist(), codes):
        taxa[o:o + len(c)] = c
    calls = rng.choice(np.array([0, 5, 4000000000], dtype=np.uint64), size=len(seqs)).astype(np.uint32)
    hits = rng.integers(0, 9, len(seqs)).astype(np.uint32)
    ids = [f"read{i}" for i in range(len(seqs))]
    rle = dict(_rle(taxa, off, lens, calls, shuffle=True), hits=hits)
    for flags in (0, capi.KU_P_ONLY_CLASSIFIED, capi.KU_P_SEQUENCE, capi.KU_P_QUICK, capi.KU_P_SEQUENCE | capi.KU_P_ONLY_CLASSIFIED):
        a = capi.format_kraken(buf, off, lens, ids, k, calls, taxa=taxa, hits=hits, flags=flags)
        assert a == capi.format_kraken_rle(buf, off, lens, ids, k, rle, flags=flags), flags
    lines = capi.format_kraken(buf, off, lens, ids, k, calls, taxa=taxa).split("\n")
    for i in (0, 5, 123, 399):
        c = codes[i]
        want = ko.hitlist(np.where(c == capi.KU_AMBIG, 0, c).astype(np.uint32), (c == capi.KU_AMBIG).astype(np.uint8))
        assert lines[i].split("\t")[4] == want and capi.hitlist_string(c) == want


def _counts_from_oracle(run):
    """Oracle per-taxon state -> the arrays ku_counts_export would deliver (dense registers)."""
    c = run.counts()
    taxids = sorted(c)
    slot_taxid = np.array(sorted(set([0] + [t for t in taxids if c[t]["n_kmers"]])), dtype=np.uint32)
    n_kmers = np.array([c[t]["n_kmers"] if t in c else 0 for t in slot_taxid.tolist()], dtype=np.uint64)
    regs = np.stack([c[t]["sketch"].registers() if t in c else np.zeros(4096, np.uint8) for t in slot_taxid.tolist()])
    node_taxid = np.array(taxids, dtype=np.uint32)
    n_reads = np.array([c[t]["n_reads"] for t in taxids], dtype=np.uint64)
    return {"slot_taxid": slot_taxid, "n_kmers": n_kmers, "registers": regs, "node_taxid": node_taxid,
            "n_reads": n_reads}


@pytest.mark.parametrize("fixture,reads", [("f1", "f1/reads.fq"), ("f2", "f2/edge.fa"), ("f4", "f4/merged.fa")])
def test_report_equals_dense_oracle_and_tracks_reference(golden, f1, fixture, reads):
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(os.path.join(golden, reads))
    ko.set_hll_sparse(False)
    try:
        run, *_ = _oracle_flat(d, seqs)
        want_dense = run.report(f"{d}/taxDB", f"{d}/database.kdb.counts")
        counts = _counts_from_oracle(run)
    finally:
        ko.set_hll_sparse(True)
    got = capi.report(f1["tax"], counts, f"{d}/database.kdb.counts")
    assert got == want_dense  # bit-exact against the oracle run with dense-from-start sketches
    # against the reference's own report: every column but kmers/dup/cov identical, kmers within 3 sigma
    ref = open(os.path.join(golden, fixture, "report.tsv")).read()
    g, r = rows(got), rows(ref)
    assert len(g) == len(r)
    key = lambda ln: ln.split("\t")[6]
    for a, b in zip(sorted(g, key=key), sorted(r, key=key)):
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[:3] == fb[:3] and fa[6:] == fb[6:], (a, b)
        if fa[3] != "kmers":
            assert abs(int(fa[3]) - int(fb[3])) <= max(2, 3 * 0.01625 * int(fb[3])), (a, b)


def _sparse_from_oracle(run, counts):
    """the oracle's sparse / dense state per slot -> (slot_is_sparse, pairs) as ku_sparse_export would deliver them"""
    c = run.counts()
    flags = np.zeros(len(counts["slot_taxid"]), dtype=np.uint8)
    pairs = []
    for s, t in enumerate(counts["slot_taxid"].tolist()):
        if t in c and c[t]["n_kmers"] and c[t]["sparse"]:
            flags[s] = 1
            pairs += [(s << 32) | int(e) for e in c[t]["sketch"].sparse_list()]
    return flags, np.array(pairs, dtype=np.uint64)


@pytest.mark.parametrize("fixture,reads,unit,report", [("f1", "f1/reads.fq", 500000, "report.tsv"),
                                                        ("f1", "f1/reads.fq", 1000, "report_u1000.tsv"),
                                                        ("f2", "f2/edge.fa", 500000, "report.tsv"),
                                                        ("f4", "f4/merged.fa", 500000, "report.tsv")])
def test_report_with_sparse_sketches_equals_the_reference(golden, f1, fixture, reads, unit, report):
    """ku_report_sparse fed with the oracle's per-taxon state (dense registers of every slot, which sketches stayed
    sparse and their encoded hashes) reproduces the reference's report row for row -- no estimator allowance"""
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(os.path.join(golden, reads))
    run = ko.Run(ko.Db(f"{d}/database.kdb", f"{d}/database.idx"), ko.Tax(f"{d}/taxDB"), work_unit_nt=unit)
    run.classify(seqs)
    counts = _counts_from_oracle(run)
    flags, pairs = _sparse_from_oracle(run, counts)
    got = capi.report_sparse(f1["tax"], counts, flags, pairs, [f"{d}/database.kdb.counts"])
    assert rows(got) == rows(open(os.path.join(golden, fixture, report)).read())
    assert flags.any() or unit == 500000


def test_sparse_estimator_known_answers():
    """the default-constructed (sparse) sketch of the reference on x * 0x9E3779B97F4A7C15"""
    for n, want in ((1, 1), (10, 10), (100, 100), (1000, 1000), (1023, 1023), (1024, 1024)):
        h = ko.Hll(12, True)
        h.insert_seq(n, 0x9E3779B97F4A7C15)
        assert h.is_sparse
        assert capi.hll_cardinality_sparse(h.sparse_list(), 1 << 40) == want == h.cardinality(False)


def test_report_rows_formats_caller_side_roll_ups(golden, f1):
    """ku_report_rows: the text from per-entry clade summaries laid out by ku_tax_ids -- here the roll-up is done in
    Python over the oracle's per-taxon state and must give the reference's report (rows, order inside a parent, number
    formats, genome sizes from database.kdb.counts incl. the last line applied twice)"""
    d = f1["dir"]
    ids, seqs = synth.read_seqfile(f"{d}/reads.fq")
    run = ko.Run(ko.Db(f"{d}/database.kdb", f"{d}/database.idx"), ko.Tax(f"{d}/taxDB"))
    run.classify(seqs)
    counts = run.counts()
    tax = f1["tax"]
    rows = tax.ids()
    assert len(rows) == 8 and set(rows.tolist()) == {0, 1, 2, 3, 4, 5, 6, 1000000001}
    parent = {int(t): int(capi.lib().ku_tax_parent(tax.h, int(t))) for t in rows if t}
    members = {int(t): [] for t in rows}
    for t, c in counts.items():
        if t not in members:
            continue
        q = t
        while True:
            members[q].append(t)
            if q == 0 or parent.get(q, 0) == 0:
                break
            q = parent[q]
    present = np.zeros(len(rows), np.uint8)
    c_reads, t_reads, c_kmers, c_uniq = (np.zeros(len(rows), np.uint64) for _ in range(4))
    for i, t in enumerate(rows.tolist()):
        ms = members[t]
        if not ms:
            continue
        present[i] = 1
        c_reads[i] = sum(counts[m]["n_reads"] for m in ms)
        c_kmers[i] = sum(counts[m]["n_kmers"] for m in ms)
        t_reads[i] = counts[t]["n_reads"] if t in counts else 0
        sk = ko.Hll(12, True)  # taxon_counts[ancestor] += counts, from a default-constructed entry (taxdb.hpp:928-973)
        for m in ms:
            sk.merge(counts[m]["sketch"])
        c_uniq[i] = sk.cardinality()
    got = capi.report_rows(tax, present, c_reads, t_reads, c_kmers, c_uniq, [f"{d}/database.kdb.counts"])
    assert got == open(f"{d}/report.tsv").read()
    with pytest.raises(capi.KuError):  # one element per entry
        capi.report_rows(tax, present[:-1], c_reads[:-1], t_reads[:-1], c_kmers[:-1], c_uniq[:-1])


def test_resolve_uids_matches_the_reference_known_answers(golden):
    """ku_resolve_uids (host C++: resolve_uids3 with the reference's containers) on the known answers of the reference's
    own function (tests/golden/kat_uid.json), from run-length encoded codes as the device delivers them; one thread and a
    team"""
    import json
    kat = json.load(open(os.path.join(golden, "kat_uid.json")))
    pm = {int(a): b for a, b in kat["parent_map"].items()}
    ids = np.array(sorted(pm), dtype=np.uint32)
    par = np.array([pm[int(t)] if pm[int(t)] else int(t) for t in ids], dtype=np.uint32)
    par[ids == 1] = 1
    tax = capi.Tax(ids=ids, parents=par)
    K = 31
    for m in kat["maps"]:
        umap = capi.UidMap(blocks=np.array(m["blocks"], dtype=np.uint32))
        assert len(umap) == len(m["blocks"])
        runs, roff, rcnt, lens, want = [], [], [], [], []
        for rep in range(40):  # 4800 reads: the threaded path
            for case in m["cases"]:
                u = case["uids"] if rep % 2 == 0 else [0] + case["uids"] + [0xFFFFFFFF, 0]  # misses / ambiguous k-mers around
                roff.append(len(runs))
                n = 0
                for i, x in enumerate(u):
                    if i == 0 or x != u[i - 1]:
                        runs.append((x, i))
                        n += 1
                rcnt.append(n)
                lens.append(len(u) + K - 1)
                want.append(case["call"])
        rle = {"runs": np.array(runs, dtype=np.uint32), "run_off": np.array(roff, dtype=np.uint64), "run_cnt": np.array(rcnt, dtype=np.uint32)}
        for thr in (1, 4):
            got = capi.resolve_uids(tax, umap, rle, lens, K, n_threads=thr)
            assert got.tolist() == want
        bad = {"runs": np.array([(len(m["blocks"]) + 5, 0)], dtype=np.uint32), "run_off": np.zeros(1, np.uint64), "run_cnt": np.ones(1, np.uint32)}
        with pytest.raises(capi.KuError) as e:
            capi.resolve_uids(tax, umap, bad, [K], K)
        assert e.value.status == -2
