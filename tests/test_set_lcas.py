"""set_lcas on the GPU (ku_setlcas_*, bin/set_lcas) against the reference's set_lcas output (tests/golden/f9) and, for
-T / -R, against a sequential Python restatement of src/set_lcas.cpp:429-476 on top of the oracle's lca()."""
import os
import subprocess

import numpy as np
import pytest

from krakenuniq_amd import capi, synth
from oracle import ku_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "krakenuniq_amd", "bin", "set_lcas")
K = 31


def test_cli_usage_without_gpu():
    assert os.path.exists(BIN), "build with make -C krakenuniq_amd/csrc"
    r = subprocess.run([BIN], stderr=subprocess.PIPE)
    assert r.returncode == 64 and b"Usage: set_lcas" in r.stderr
    assert subprocess.run([BIN, "-d", "a", "-i", "b", "-b", "c"], stderr=subprocess.PIPE).returncode == 64  # no -f / -F -m
    assert subprocess.run([BIN, "-h"], stderr=subprocess.PIPE).returncode == 0


@pytest.mark.parametrize("tag,flags", [("a", ["-a"]), ("A", ["-A"]), ("aA", ["-a", "-A"])])
def test_new_taxids_map_and_taxdb_host_logic(tmp_path, tag, flags):
    """-a / -A (src/set_lcas.cpp:169-237,321-330): new taxids for sequences / assemblies, printed map and rewritten
    taxDB byte-identical to the reference's -- host logic, checked here through the KU_SETLCAS_DRY hook (no device)"""
    (tmp_path / "taxDB").write_bytes(open(f"{G}/f1/taxDB", "rb").read())
    r = subprocess.run([BIN, "-x", "-d", "unused.kdb", "-i", "unused.idx", "-b", str(tmp_path / "taxDB"),
                        "-m", f"{G}/f9/seqid2taxid_aA.map", "-F", f"{G}/f9/library.fa"] + flags,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KU_SETLCAS_DRY="1"))
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == open(f"{G}/f9/map_{tag}.out", "rb").read()
    assert (tmp_path / "taxDB").read_bytes() == open(f"{G}/f9/taxDB_{tag}", "rb").read()
    if "-A" in flags:
        assert b"parent taxon 999 not in database" in r.stderr


def _zeroed_db(tmp_path):
    kmers, vals, off, k, nt, _ = synth.read_db(f"{G}/f1")
    synth.write_db(str(tmp_path), kmers, np.zeros_like(vals), off, k, nt)
    return kmers, vals


@pytest.mark.gpu
def test_reproduces_the_reference_database(tmp_path):
    _zeroed_db(tmp_path)
    args = ["-M", "-x", "-t", "2", "-v", "-d", str(tmp_path / "database.kdb"), "-o", str(tmp_path / "out.kdb"),
            "-i", str(tmp_path / "database.idx"), "-b", f"{G}/f1/taxDB", "-m", f"{G}/f9/seqid2taxid.map",
            "-F", f"{G}/f9/library.fa", "-c", str(tmp_path / "counts")]
    r = subprocess.run([BIN] + args, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert (tmp_path / "out.kdb").read_bytes() == open(f"{G}/f9/database.kdb", "rb").read()
    assert (tmp_path / "counts").read_text() == open(f"{G}/f9/database.kdb.counts").read()
    err = r.stderr.decode()
    assert "Didn't find taxonomy ID mapping for sequence unmapped" in err and "taxonomy ID 999 is not in taxonomy database" in err
    assert "Finished processing 5 sequences" in err
    # the input is untouched when -o names another file; without -x the novel sequence is an error (EX_DATAERR)
    assert not synth.read_db(str(tmp_path))[1].any()
    assert subprocess.run([BIN] + [a for a in args if a != "-x"], stderr=subprocess.PIPE).returncode == 65
    # in place (no -o), then -R on one genome: its k-mers go back to zero, the others stay
    assert subprocess.run([BIN] + [a for a in args if a not in ("-o", str(tmp_path / "out.kdb"))], stderr=subprocess.PIPE).returncode == 0
    assert (tmp_path / "database.kdb").read_bytes() == open(f"{G}/f9/database.kdb", "rb").read()
    lib = open(f"{G}/f9/library.fa").read().split(">")[1]
    (tmp_path / "one.fa").write_text(">" + lib)
    (tmp_path / "files.map").write_text(f"{tmp_path}/one.fa 4\n")
    assert subprocess.run([BIN, "-R", "-x", "-d", str(tmp_path / "database.kdb"), "-i", str(tmp_path / "database.idx"),
                           "-b", f"{G}/f1/taxDB", "-f", str(tmp_path / "files.map")], stderr=subprocess.PIPE).returncode == 0
    kmers, got, *_ = synth.read_db(str(tmp_path))
    _, want, *_ = synth.read_db(f"{G}/f9", idx=f"{G}/f1/database.idx")
    seq = "".join(lib.split("\n")[1:]).upper().encode()
    in_a = np.isin(kmers, synth.canonical(synth.kmers_forward(synth.ascii_to_codes(seq), K), K))
    assert in_a.sum() > 1000 and not got[in_a].any() and np.array_equal(got[~in_a], want[~in_a])


def _sequential_model(kmers, init, otax, seqs, force_contaminant=False, reset=False):
    """src/set_lcas.cpp:429-476, one k-mer after the other"""
    pos = {int(x): i for i, x in enumerate(kmers.tolist())}
    vals = init.astype(np.uint32).copy()
    for seq, taxid in seqs:
        codes = synth.ascii_to_codes(seq)
        for km in synth.canonical(synth.kmers_forward(codes, K), K).tolist():
            i = pos.get(int(km))
            if i is None:
                continue
            if reset:
                vals[i] = 0
            elif not force_contaminant:
                vals[i] = otax.lca(taxid, int(vals[i]))
            elif vals[i] in (32630, 81077):
                pass
            elif taxid in (32630, 81077):
                vals[i] = taxid
            else:
                vals[i] = otax.lca(taxid, int(vals[i]))
    return vals


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, capi.KU_SL_FORCE_CONTAMINANT])
def test_contaminants_and_preset_values_against_the_sequential_model(tmp_path, flags):
    rng = np.random.default_rng(9)
    tax = synth.small_taxonomy()
    tax.add(32630, 1, "synthetic construct", "species")
    tax.add(81077, 1, "artificial sequences", "species")
    base = synth.procedural_genome(11, 1, 2400)
    a, cc = synth.codes_to_ascii, np.concatenate
    seqs = [(a(base[:1500]), 4), (a(base[700:1700]), 32630), (a(cc([base[1000:2400], base[:200]])), 5),
            (a(base[1500:1900]), 81077), (a(base[1600:2000]), 32630), (a(synth.mutate(base, 0.02, rng)[:900]), 6),
            (a(base[300:330]), 4)]  # the last one: shorter than k, contributes nothing
    kmers = np.unique(np.concatenate([synth.canonical(synth.kmers_forward(synth.ascii_to_codes(s), K), K) for s, _ in seqs]))
    init = np.zeros(len(kmers), dtype=np.uint32)
    init[::7] = 5           # pre-set values take part in the fold
    init[3::11] = 81077     # ... and contaminant values stay under -T
    sk, sv, off = synth.sort_db(kmers, init, K, 8)
    synth.write_db(str(tmp_path), sk, sv, off, K, 8)
    tax.write(str(tmp_path / "taxDB"))
    cdb, ctax = capi.Db(str(tmp_path / "database.kdb"), str(tmp_path / "database.idx")), capi.Tax(str(tmp_path / "taxDB"))
    sl = capi.SetLcas(cdb, ctax, flags=flags)
    for s, t in seqs:
        sl.add(s, t)
    got, missing = sl.finish()
    want = _sequential_model(sk, sv, ko.Tax(str(tmp_path / "taxDB")), seqs, force_contaminant=bool(flags))
    assert missing == 0 and np.array_equal(got, want)
    if flags:
        assert (got == 32630).sum() > 500 and (got == 81077).sum() > 100
    with pytest.raises(capi.KuError):
        sl.add(seqs[0][0], 424242)  # neither in the taxonomy nor a database value
    sl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,flags", [("a", ["-a"]), ("A", ["-A"]), ("aA", ["-a", "-A"])])
def test_new_taxids_end_to_end(tmp_path, tag, flags):
    """-a / -A through the device: the values of every k-mer, the counts file, the new taxDB and the printed map equal
    the reference's"""
    _zeroed_db(tmp_path)
    (tmp_path / "taxDB").write_bytes(open(f"{G}/f1/taxDB", "rb").read())
    r = subprocess.run([BIN, "-M", "-x", "-d", str(tmp_path / "database.kdb"), "-o", str(tmp_path / "out.kdb"),
                        "-i", str(tmp_path / "database.idx"), "-b", str(tmp_path / "taxDB"), "-m", f"{G}/f9/seqid2taxid_aA.map",
                        "-F", f"{G}/f9/library.fa", "-c", str(tmp_path / "counts")] + flags, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    _, got, *_ = synth.read_db(str(tmp_path), kdb="out.kdb")
    assert np.array_equal(got, np.fromfile(f"{G}/f9/values_{tag}.u32", dtype="<u4"))
    assert (tmp_path / "counts").read_text() == open(f"{G}/f9/counts_{tag}").read()
    assert r.stdout == open(f"{G}/f9/map_{tag}.out", "rb").read()
    assert (tmp_path / "taxDB").read_bytes() == open(f"{G}/f9/taxDB_{tag}", "rb").read()


@pytest.mark.gpu
def test_uid_database_equals_the_reference_build(tmp_path):
    """set_lcas -I (src/set_lcas.cpp:451-455, src/uid_mapping.cpp:32-91): the value of a k-mer is the id of the SET of taxids
    whose library sequences hold it, sets numbered in the order they first come up.  tests/golden/f11 holds what the
    reference's set_lcas -I built from f1's k-mers (values zeroed) and a library of six sequences -- sets of one, two and
    three taxids -- whose inputs tests/golden/make_golden.py:make_f11 derives from fixed seeds; rebuilt here the same way.
    Database, UID map and counts byte for byte."""
    _zeroed_db(tmp_path)
    rng = np.random.default_rng(7)
    g4 = synth.procedural_genome(7, 4, 3000)
    g5 = synth.mutate(g4, 0.03, rng)
    g6 = synth.procedural_genome(7, 6, 3000)
    gp = np.concatenate([g6[2000:2300], synth.procedural_genome(7, 99, 300)])
    a = synth.codes_to_ascii
    g4, g5, g6, gp = a(g4), a(g5), a(g6), a(gp)
    recs = [(b"seqA", g4), (b"seqB", g5), (b"seqC", g6), (b"seqP", gp), (b"seqD part of the first genome under another taxid", g4[500:1500]),
            (b"seqE part of the second genome under the genus", g5[1200:2200])]
    with open(tmp_path / "library.fa", "wb") as f:
        for h, sq in recs:
            f.write(b">" + h + b"\n")
            for i in range(0, len(sq), 70):
                f.write(sq[i:i + 70] + b"\n")
    (tmp_path / "seqid2taxid.map").write_text("seqA\t4\nseqB\t5\nseqC\t6\nseqP\t1000000001\nseqD\t6\nseqE\t2\n")
    r = subprocess.run([BIN, "-M", "-x", "-d", str(tmp_path / "database.kdb"), "-I", str(tmp_path / "uid.map"), "-o", str(tmp_path / "uid.kdb"),
                        "-i", str(tmp_path / "database.idx"), "-b", f"{G}/f1/taxDB", "-m", str(tmp_path / "seqid2taxid.map"),
                        "-F", str(tmp_path / "library.fa"), "-c", str(tmp_path / "uid.counts")], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert (tmp_path / "uid.map").read_bytes() == open(f"{G}/f11/uid_to_taxid.map", "rb").read()
    assert (tmp_path / "uid.kdb").read_bytes() == open(f"{G}/f11/uid_database.kdb", "rb").read()
    assert (tmp_path / "uid.counts").read_text() == open(f"{G}/f11/uid_database.kdb.counts").read()
    # a second run over the result: its values are UIDs of a map this run does not have -- the reference exits there too
    r2 = subprocess.run([BIN, "-x", "-d", str(tmp_path / "uid.kdb"), "-I", str(tmp_path / "uid2.map"), "-o", str(tmp_path / "uid2.kdb"),
                         "-i", str(tmp_path / "database.idx"), "-b", f"{G}/f1/taxDB", "-m", str(tmp_path / "seqid2taxid.map"),
                         "-F", str(tmp_path / "library.fa")], stderr=subprocess.PIPE)
    assert r2.returncode != 0 and b"greater than UID vector size" in r2.stderr
