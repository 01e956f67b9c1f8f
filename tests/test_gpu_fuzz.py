"""-m gpu: a bounded slice of the extended differential runs (tests/fuzz_gpu_parity.py, fuzz_gpu_sparse.py, fuzz_gpu_mgpu.py,
fuzz_gpu_step_device.py, fuzz_cli.py, fuzz_build_tools.py -- the long runs are logged under profiles/r06_fuzz_*.log): random databases, read shapes, batch cuts, ranks and
flag sets against the oracle, and the executable against the compiled reference on the same files."""
import os
import shutil
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", range(9000, 9012))
def test_flat_windowed_and_quick_paths_against_the_oracle(seed):
    import fuzz_gpu_parity
    fuzz_gpu_parity.one_case(seed)


@pytest.mark.parametrize("seed", range(9100, 9112))
def test_two_step_call_with_batches_in_flight_and_the_emulation_against_the_oracle(seed):
    import fuzz_gpu_sparse
    fuzz_gpu_sparse.one_case(seed)


@pytest.mark.parametrize("seed", range(9200, 9210))
def test_groups_of_ranks_against_the_oracle(seed):
    import fuzz_gpu_mgpu
    fuzz_gpu_mgpu.one_case(seed)


@pytest.mark.parametrize("seed", range(9400, 9440))
def test_routed_device_step_against_one_context(seed):
    import fuzz_gpu_step_device
    fuzz_gpu_step_device.one_case(seed)


def test_executable_against_the_compiled_reference_on_random_inputs_and_flags():
    import fuzz_cli
    if not os.path.exists(fuzz_cli.REF):
        pytest.skip("oracle/_ref/classify is not built here")
    tmp = tempfile.mkdtemp(prefix="ku_fuzz_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        compared = 0
        for seed in range(9300, 9316):
            compared += "reference died" not in fuzz_cli.one_case(seed, tmp)
        assert compared >= 8
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_build_tools_against_the_compiled_reference_on_random_inputs():
    import fuzz_build_tools
    if not os.path.exists(os.path.join(fuzz_build_tools.REF, "db_sort")):
        pytest.skip("oracle/_ref/db_sort is not built here")
    tmp = tempfile.mkdtemp(prefix="ku_fuzz_build_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for seed in range(9500, 9508):
            fuzz_build_tools.one_case(seed, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
