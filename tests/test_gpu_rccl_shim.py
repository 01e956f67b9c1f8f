"""The multi-GPU driver's RCCL paths with REAL PEERS on a 1-GPU box: `world` processes, one rank each, all on cuda:0,
bound (KU_RCCL_LIB) to tests/rccl_shim/libku_rccl_shim.so instead of RCCL, which refuses two ranks on one device.  The shim
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
WORKER = os.path.join(ROOT, "tests", "rccl_shim", "worker.py")
pytestmark = pytest.mark.gpu


def run_world(world, mode, extra_env=None):
    assert os.path.exists(SHIM), "tests/rccl_shim/libku_rccl_shim.so is not built (make -C tests/rccl_shim)"
    scratch = tempfile.mkdtemp(prefix="ku_shim_test_")
    env = dict(os.environ, KU_RCCL_LIB=SHIM, KU_SHIM_TIMEOUT="90", **(extra_env or {}))
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
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {world} ({mode}) failed:\n{o[-3000:]}"
        assert f"rank {r} of {world} ({mode}): ok" in o


@pytest.mark.parametrize("world,mode", [(2, "route"), (4, "route"), (2, "slots"), (3, "reduce"), (2, "replicas")])
def test_processes_with_real_peers_match_one_context(world, mode):
    run_world(world, mode)


def test_routed_rounds_on_two_streams_with_real_peers():
    """several rounds per step (two buffer sets, two streams) through the send / receive pairs"""
    run_world(2, "route", {"KU_ROUTE_ROUND": "1500000", "KU_ROUTE_TWO_STREAMS": "1"})
    run_world(2, "route", {"KU_ROUTE_ROUND": "1500000"})  # (the default over RCCL: the rounds on one stream)
