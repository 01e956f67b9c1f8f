"""db_sort on the GPU (ku_db_sort_files, bin/db_sort) against the files the reference's db_sort wrote
(tests/golden/f1, f8: byte-identical database.kdb / database.idx) and against the numpy model for other geometries."""
import os
import subprocess

import numpy as np
import pytest

from krakenuniq_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "krakenuniq_amd", "bin", "db_sort")


def test_cli_usage_without_gpu():
    assert os.path.exists(BIN), "build with make -C krakenuniq_amd/csrc"
    r = subprocess.run([BIN], stderr=subprocess.PIPE)
    assert r.returncode == 64 and b"Usage: db_sort" in r.stderr
    assert subprocess.run([BIN, "-n", "40", "-d", "a", "-o", "b", "-i", "c"], stderr=subprocess.PIPE).returncode == 64
    assert subprocess.run([BIN, "-h"], stderr=subprocess.PIPE).returncode == 0


def _shuffled_jdb(path, kmers, vals, k, seed):
    perm = np.random.default_rng(seed).permutation(len(kmers))
    synth.write_jdb(path, kmers[perm], vals[perm], k)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["f1", "f8"])
def test_reproduces_the_reference_files(tmp_path, fixture):
    d = os.path.join(G, fixture)
    kmers, vals, _, k, nt, _ = synth.read_db(d)
    _shuffled_jdb(str(tmp_path / "in.jdb"), kmers, vals, k, 5)
    capi.db_sort_files(str(tmp_path / "in.jdb"), str(tmp_path / "out.kdb"), str(tmp_path / "out.idx"), nt)
    assert (tmp_path / "out.kdb").read_bytes() == open(f"{d}/database.kdb", "rb").read()
    assert (tmp_path / "out.idx").read_bytes() == open(f"{d}/database.idx", "rb").read()
    # the executable, with -z: same order, values zeroed (src/db_sort.cpp:103-104)
    r = subprocess.run([BIN, "-z", "-M", "-t", "2", "-n", str(nt), "-d", str(tmp_path / "in.jdb"), "-o", str(tmp_path / "z.kdb"),
                        "-i", str(tmp_path / "z.idx")], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert (tmp_path / "z.idx").read_bytes() == open(f"{d}/database.idx", "rb").read()
    zk, zv, _, _, _, _ = synth.read_db(str(tmp_path), kdb="z.kdb", idx="z.idx")
    assert np.array_equal(zk, kmers) and not zv.any()


@pytest.mark.gpu
@pytest.mark.parametrize("nt", [1, 9, 13, 15])
def test_other_bin_key_lengths_match_the_numpy_model(tmp_path, nt):
    rng = np.random.default_rng(nt)
    g = synth.procedural_genome(3, nt, 60000)
    kmers = np.unique(synth.canonical(synth.kmers_forward(g, 31), 31))
    vals = rng.integers(1, 1 << 31, len(kmers), dtype=np.uint32)
    _shuffled_jdb(str(tmp_path / "in.jdb"), kmers, vals, 31, 1)
    capi.db_sort_files(str(tmp_path / "in.jdb"), str(tmp_path / "database.kdb"), str(tmp_path / "database.idx"), nt)
    sk, sv, off = synth.sort_db(kmers, vals, 31, nt)
    synth.write_db(str(tmp_path / "want"), sk, sv, off, 31, nt)
    for fn in ("database.kdb", "database.idx"):
        assert (tmp_path / fn).read_bytes() == (tmp_path / "want" / fn).read_bytes(), fn
    # the result is a database the classify path opens
    db = capi.Db(str(tmp_path / "database.kdb"), str(tmp_path / "database.idx"))
    assert db.info.key_ct == len(kmers) and db.info.nt == nt


@pytest.mark.gpu
def test_errors(tmp_path):
    (tmp_path / "bad").write_bytes(b"not a jellyfish file" * 10)
    with pytest.raises(capi.KuError):
        capi.db_sort_files(str(tmp_path / "bad"), str(tmp_path / "o.kdb"), str(tmp_path / "o.idx"), 7)
    with pytest.raises(capi.KuError):
        capi.db_sort_files(str(tmp_path / "missing"), str(tmp_path / "o.kdb"), str(tmp_path / "o.idx"), 7)
    with pytest.raises(capi.KuError):
        capi.db_sort_files(f"{G}/f1/database.kdb", str(tmp_path / "o.kdb"), str(tmp_path / "o.idx"), 16)
