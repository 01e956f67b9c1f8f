"""The multi-GPU driver's RCCL paths with REAL PEERS on a 1-GPU box: `world` processes, one rank each, all on cuda:0,
bound (KU_RCCL_LIB, honoured by the test build of the library only) to tests/rccl_shim/libku_rccl_shim.so instead of RCCL, which refuses two ranks on one device.  The shim
is test infrastructure: the same entry points between processes through files in /dev/shm.  What runs here is the product's
own code: ncclCommInitRank per process, comm_scatter_slices / comm_alltoallv / comm_allgather_u64_dev (owner routing),
comm_broadcast + the grouped send / receive all-to-all or the grouped ncclReduce (position-wise exchange), the all-gather of
the ranks' value lists, the all-reduce of the per-taxon state -- none of which had a peer before."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "rccl_shim", "libku_rccl_shim.so")
# the product library does not look at KU_RCCL_LIB / KU_TEST_DROP_STREAM_JOIN: the test build of it does (-DKU_TEST_HOOKS, the same
# objects with ku_mgpu.cpp compiled once more; tests/rccl_shim/Makefile)
HOOKS_LIB = os.path.join(ROOT, "tests", "rccl_shim", "libkrakenuniq_amd_testhooks.so")
WORKER = os.path.join(ROOT, "tests", "rccl_shim", "worker.py")
pytestmark = pytest.mark.gpu


def run_world(world, mode, extra_env=None, expect_failure=False):
    assert os.path.exists(SHIM), "tests/rccl_shim/libku_rccl_shim.so is not built (make -C tests/rccl_shim)"
    scratch = tempfile.mkdtemp(prefix="ku_shim_test_")
    assert os.path.exists(HOOKS_LIB), "tests/rccl_shim/libkrakenuniq_amd_testhooks.so is not built (make -C tests/rccl_shim)"
    env = dict(os.environ, KU_LIB=HOOKS_LIB, KU_RCCL_LIB=SHIM, KU_SHIM_TIMEOUT="90", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), mode, scratch], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            try:
                outs.append(p.communicate(timeout=400)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                outs.append(p.communicate()[0])
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
        for d in glob.glob("/dev/shm/ku_shim_*"):
            shutil.rmtree(d, ignore_errors=True)
    if expect_failure:
        assert any(p.returncode != 0 for p in procs), "every rank passed although the step was broken on purpose"
        return outs
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {world} ({mode}) failed:\n{o[-3000:]}"
        assert f"rank {r} of {world} ({mode}): ok" in o
    return outs


@pytest.mark.parametrize("world,mode", [(2, "route"), (4, "route"), (2, "slots"), (3, "reduce"), (2, "replicas")])
def test_processes_with_real_peers_match_one_context(world, mode):
    """the stand-in's operations are asynchronous since round 5 (enqueued on the caller's stream, completed by host functions
    while the stream stands still): the default here"""
    run_world(world, mode)


JITTER = {"KU_SHIM_JITTER_US": "3000"}


@pytest.mark.parametrize("world,mode", [(3, "route"), (2, "slots"), (4, "slots"), (2, "reduce"), (4, "reduce"), (3, "replicas"), (4, "replicas")])
def test_peers_that_run_ahead_or_lag_behind(world, mode):
    """every exchange delayed by a random time, another one on every rank (KU_SHIM_JITTER_US), before its sends and behind its
    receives: ranks that are rounds apart, messages that wait for their receiver and receivers that wait for their message"""
    run_world(world, mode, JITTER)


def test_the_synchronous_stand_in_of_round_4_still_agrees():
    run_world(2, "route", {"KU_SHIM_SYNC": "1"})


def test_a_missing_stream_dependency_is_found():
    """rank_step_routed on two streams, the join of the second stream into the caller's left out on purpose
    (KU_TEST_DROP_STREAM_JOIN): behind an asynchronous exchange that lags, the caller reads results that are not there yet --
    the step must FAIL.  (What the stand-in is for: the first run on several physical GPUs must not be the first time such a
    bug can show.)"""
    env = {"KU_ROUTE_ROUND": "1500000", "KU_ROUTE_TWO_STREAMS": "1", "KU_SHIM_JITTER_US": "2000", "KU_SHIM_LAG_OTHER_STREAMS_US": "150000"}
    run_world(2, "route", env)  # with the join: fine, however the peers lag
    run_world(2, "route", dict(env, KU_TEST_DROP_STREAM_JOIN="1"), expect_failure=True)


def test_routed_rounds_on_two_streams_with_real_peers():
    """several rounds per step (two buffer sets, two streams) through the send / receive pairs"""
    run_world(2, "route", {"KU_ROUTE_ROUND": "1500000", "KU_ROUTE_TWO_STREAMS": "1"})
    run_world(2, "route", {"KU_ROUTE_ROUND": "1500000"})  # (the default over RCCL: the rounds on one stream)
    run_world(3, "route", dict(JITTER, KU_ROUTE_ROUND="1500000", KU_ROUTE_TWO_STREAMS="1"))
    run_world(4, "route", dict(JITTER, KU_ROUTE_ROUND="1500000"))
