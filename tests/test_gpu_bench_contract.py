"""-m gpu: bench.py's contract with the driver on a small workload -- exactly one JSON line on stdout, the driver's
fields, `roofline` measured in the run and `cpu_baseline` from the compiled reference (or the oracle) with its parity
flag, the host-buffer and end-to-end legs."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--reads", "200000",
                        "--species", "60", "--genome-len", "60000", *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_shape_line():
    d = run_bench("--cpu-sample", "20000")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    out = d["config"]["output"]  # the kernel's own runs of one batch, expanded again, are the per-k-mer array
    assert out["expanded_runs_equal_the_per_kmer_array"] is True and out["run_array_extent_used"] > 0 and out["runs_per_read"] >= 1
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["kernel_ms"] > 0
    assert rf["traffic"] is None or rf["traffic"] > 0  # counter bytes only for the profiled workload and kernel source
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["parity_vs_reference_on_sample"] is True
    assert d["device_pipeline"]["calls_match_device_run"] is True and d["device_pipeline"]["value"] > 0
    assert d["e2e"]["calls_match_device_run"] is True and d["e2e"]["value"] > 0
    gz = d["e2e"]["gz"]  # the same reads as one gzip stream through the gzip team and the region parsers; zlib's reader beside it
    assert gz["calls_match_device_run"] is True and gz["value"] > 0 and gz["gz_bytes"] > 0 and gz["seconds_with_zlib_reader"] > 0


@pytest.mark.parametrize("shape", [("--paired",), ("--read-len", "3000", "--reads", "20000"), ("--nt", "15"), ("--output", "runs")])
def test_other_shapes_run(shape):
    d = run_bench("--cpu-sample", "0", "--no-extras", *shape)
    assert d["value"] > 0 and d["roofline"]["kernel"].startswith("ku_classify_short_kernel")
    assert d["config"]["output"]["expanded_runs_equal_the_per_kmer_array"] is True


def run_bench_world(n, *extra):
    """`bench.py --gpus n` on a box with ONE GPU: every rank on cuda:0 (KU_BENCH_ONE_DEVICE), the C++ driver bound to the test
    stand-in for RCCL (tests/rccl_shim), which moves the ranks' messages between the processes"""
    shim = os.path.join(ROOT, "tests", "rccl_shim", "libku_rccl_shim.so")
    assert os.path.exists(shim)
    env = dict(os.environ, KU_BENCH_ONE_DEVICE="1", KU_RCCL_LIB=shim, KU_LIB=os.path.join(ROOT, "tests", "rccl_shim", "libkrakenuniq_amd_testhooks.so"), KU_SHIM_TIMEOUT="120")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--reads", "120000",
                        "--species", "60", "--genome-len", "60000", "--cpu-sample", "0", *extra], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_two_ranks_sharded_line_with_real_peers():
    """the N > 1 sharded line (owner routing over the send / receive pairs of two processes): every read resolved once, the
    rank's own stage roofline and the wire figures present"""
    d = run_bench_world(2, "--mode", "sharded", "--nt", "11", "--no-extras")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["every_read_resolved_once"] is True and d["config"]["exchange"] == "RCCL"
    rf = d["roofline"]
    assert rf["stage_ms_measured"]["owner_ms"] > 0 and rf["kmers_received"] > 0 and 0 < rf["frac"] < 1
    assert d["wire"]["exchange"].startswith("owner routing") and d["wire"]["records_in_bytes_per_read"] > 0


def test_two_ranks_replicas_line_with_real_peers():
    d = run_bench_world(2, "--mode", "replicas", "--no-extras")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["merged_read_count_ok"] is True


def test_two_ranks_plain_line_has_the_sharded_layout_as_its_headline():
    """`bench.py --gpus 2` without --mode / --config: both legs run, the line is the sharded layout's (configs[2] scaled to the
    world -- here at a size two ranks on one GPU hold), the replicas are the second leg"""
    shim = os.path.join(ROOT, "tests", "rccl_shim", "libku_rccl_shim.so")
    env = dict(os.environ, KU_BENCH_ONE_DEVICE="1", KU_RCCL_LIB=shim, KU_LIB=os.path.join(ROOT, "tests", "rccl_shim", "libkrakenuniq_amd_testhooks.so"), KU_SHIM_TIMEOUT="120", KU_BENCH_SHARD_SPECIES="40",
               KU_BENCH_SHARD_GENOME_LEN="50000", KU_BENCH_SHARD_READS="100000", KU_BENCH_SHARD_NT="11")  # (nt = 15: 8.6 GB of index per rank, minutes)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "100000",
                        "--species", "60", "--genome-len", "60000", "--cpu-sample", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert d["config"]["parallelism"] == "sharded2" and d["config"]["nt"] == 11 and d["config"]["every_read_resolved_once"] is True
    assert d["roofline"]["stage_ms_measured"]["owner_ms"] > 0 and d["wire"]["exchange"].startswith("owner routing")
    rep = d["replicas"]
    assert rep["scaling"] == "weak" and rep["value"] > 0 and rep["config"]["merged_read_count_ok"] is True


def test_plain_gpus_2_rehearsal_at_the_presets_shape():
    """the driver's own `python bench.py --gpus 2 --steps K --warmup W` flow, end to end, at the preset's geometry
    (nt = 15: the 8.6 GB index per world, standard k) with its sizes divided by KU_BENCH_SCALE_DIV -- nothing else overridden: the
    pre-flight lines, both legs, the sharded layout as the headline, inside the run's own budget.  Two processes on one device
    through the asynchronous stand-in for librccl (the only thing a 1-GPU box cannot do is RCCL between two devices)."""
    shim = os.path.join(ROOT, "tests", "rccl_shim", "libku_rccl_shim.so")
    env = dict(os.environ, KU_BENCH_ONE_DEVICE="1", KU_RCCL_LIB=shim, KU_LIB=os.path.join(ROOT, "tests", "rccl_shim", "libkrakenuniq_amd_testhooks.so"),
               KU_SHIM_TIMEOUT="300", KU_BENCH_SCALE_DIV="100", KU_BENCH_BUDGET_S="1200")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reads", "200000",
                        "--species", "100", "--cpu-sample", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=1300, env=env)
    wall = time.time() - t0
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert "error" not in d, d.get("error")
    # the pre-flight lines of both legs, from both ranks, before the builds
    pf = [json.loads(ln.split("[bench preflight] ", 1)[1]) for ln in err.split("\n") if "[bench preflight] " in ln]
    assert sorted((p["leg"], p["rank"]) for p in pf) == [("replicas", 0), ("replicas", 1), ("sharded", 0), ("sharded", 1)]
    assert all(p["hbm_free_gb"] > 0 and p["planned_table_gb"] > 0 and p["budget_s"] == 1200 for p in pf)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    c = d["config"]
    assert c["parallelism"] == "sharded2" and c["nt"] == 15 and c["taxa"] == 240 and c["reads_per_step"] == 100000
    assert c["every_read_resolved_once"] is True and c["preflight"]["leg"] == "sharded" and c["preflight"]["leg_limit_s"] > 0
    assert d["replicas"]["value"] > 0 and d["replicas"]["config"]["merged_read_count_ok"] is True
    assert wall < 1200, wall


def test_a_rank_failing_in_the_sharded_leg_ends_the_run_with_the_replicas_line():
    """one rank raising in the second leg (an allocation failure, say) leaves the others inside an exchange; the run must not sit
    there until its limit: the failing rank says so in the rendezvous store, and rank 0 prints the line it has at once"""
    shim = os.path.join(ROOT, "tests", "rccl_shim", "libku_rccl_shim.so")
    env = dict(os.environ, KU_BENCH_ONE_DEVICE="1", KU_RCCL_LIB=shim, KU_LIB=os.path.join(ROOT, "tests", "rccl_shim", "libkrakenuniq_amd_testhooks.so"),
               KU_SHIM_TIMEOUT="300", KU_BENCH_SCALE_DIV="100", KU_BENCH_BUDGET_S="1200", KU_BENCH_TEST_FAIL_SHARDED_RANK="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reads", "200000",
                        "--species", "100", "--cpu-sample", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=1300, env=env)
    wall = time.time() - t0
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["scaling"] == "weak" and d["value"] > 0 and d["config"]["merged_read_count_ok"] is True
    assert d["sharded"]["value"] is None and "rank 1 failed in the sharded leg" in d["sharded"]["error"]
    assert "[bench] rank 1 failed in the sharded leg" in err
    assert wall < 400, wall
