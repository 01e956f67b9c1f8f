"""-m gpu: the out-of-core run (-x SIZE) at a size that needs it: a ~9 GB database on disk streamed through
the GPU in (four) minimizer-range chunks (src/krakendb.cpp:463-522) over 3 M reads that do not fit the device budget at once
(two super-batches, so the chunks cycle twice and chunk 0 is prefetched again under the last pass) -- Kraken file,
report and database.kdb.counts equal to the run with everything resident.  (scripts/ooc_check.py is the same check at
33 GB / 50 M reads; its last log is kept under profiles/.)"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from krakenuniq_amd import synth_torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "krakenuniq_amd", "bin", "classify")


def test_chunked_run_equals_resident_run_at_9GB():
    import torch
    tmp = "/dev/shm/ku_ooc_test" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 16e9 else "/tmp/ku_ooc_test"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    try:
        dev = torch.device("cuda:0")
        db = synth_torch.BenchDb(dev, n_species=2500, genome_len=310_000, k=31, nt=13, seed=7)
        assert db.n_pairs * 12 > 8.5e9
        db.kmers = db.vals = None
        db.write_files(tmp)
        n, L = 3_000_000, 150
        s, _, _, _ = db.sample_reads(n, L, seed=100)
        rows = s.view(n, L + 1).cpu().numpy()
        rec = np.empty((n, 12 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord(">")
        rec[:, 1] = ord("r")
        idx = np.arange(n, dtype=np.int64)
        for d in range(9):
            rec[:, 2 + d] = 48 + (idx // 10 ** (8 - d)) % 10
        rec[:, 11] = 10
        rec[:, 12:12 + L + 1] = rows
        with open(f"{tmp}/reads.fa", "wb") as f:
            f.write(rec.tobytes())
        del db, s, rows, rec
        torch.cuda.empty_cache()
        base = [BIN, "-d", f"{tmp}/database.kdb", "-i", f"{tmp}/database.idx", "-a", f"{tmp}/taxDB", "-t", "8"]
        env = dict(os.environ, KU_NO_SPARSE="1")  # (the reference's -x mode keeps one sketch per taxon for the whole run)
        runs = {"resident": ([], {}), "chunked": (["-x", "2500M"], {"KU_SUPERBATCH_BYTES": str(1200 << 20)})}
        errs = {}
        for name, (extra, e) in runs.items():
            r = subprocess.run(base + extra + ["-o", f"{tmp}/{name}.tsv", "-r", f"{tmp}/{name}.rep", f"{tmp}/reads.fa"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env={**env, **e})
            errs[name] = r.stderr.decode(errors="replace").replace("\r", "\n")
            assert r.returncode == 0, errs[name][-800:]
            os.rename(f"{tmp}/database.kdb.counts", f"{tmp}/{name}.counts")
        import re
        m = re.search(r"in (\d+) chunks of at most", errs["chunked"])
        assert m and int(m.group(1)) >= 3, errs["chunked"][-600:]
        m2 = re.search(r"(\d+) passes over the (\d+) database chunks", errs["chunked"])  # the input did not fit at once
        assert m2 and int(m2.group(1)) >= 2 and m2.group(2) == m.group(1), errs["chunked"][-600:]
        assert subprocess.run(["cmp", "-s", f"{tmp}/resident.tsv", f"{tmp}/chunked.tsv"]).returncode == 0
        assert open(f"{tmp}/resident.rep").read() == open(f"{tmp}/chunked.rep").read()
        assert open(f"{tmp}/resident.counts").read() == open(f"{tmp}/chunked.counts").read()
        assert os.path.getsize(f"{tmp}/resident.tsv") > 100e6
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
