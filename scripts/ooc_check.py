#!/usr/bin/env python3
"""Out-of-core run at scale (run through gpurun): a >= 32 GB synthetic database on disk (/dev/shm), 50 M x 150 bp reads,
the classify executable once with everything resident and once with -x 8G and a device budget that forces several
passes over the chunks; the Kraken outputs must be byte-identical and the reports equal.
    python scripts/ooc_check.py [n_species] [n_reads_million]"""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from krakenuniq_amd import synth_torch

n_species = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
n_m = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tmp = "/dev/shm/ku_ooc"
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(tmp)
dev = torch.device("cuda:0")
t0 = time.time()
db = synth_torch.BenchDb(dev, n_species=n_species, genome_len=310_000, k=31, nt=13, seed=7)
print(f"database: {db.n_pairs} pairs = {db.n_pairs * 12 / 1e9:.1f} GB, built in {time.time() - t0:.0f}s", flush=True)
db.kmers = db.vals = None
t0 = time.time()
db.write_files(tmp)
print(f"files written in {time.time() - t0:.0f}s", flush=True)
L = 150
with open(f"{tmp}/reads.fa", "wb") as f:
    for c in range(n_m // 10 if n_m >= 10 else 1):
        n = 10_000_000 if n_m >= 10 else n_m * 1_000_000
        s, _, _, _ = db.sample_reads(n, L, seed=100 + c)
        rows = s.view(n, L + 1).cpu().numpy()
        rec = np.empty((n, 12 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r")
        idx = np.arange(n, dtype=np.int64) + c * 10_000_000
        for d in range(9):
            rec[:, 2 + d] = 48 + (idx // 10 ** (8 - d)) % 10
        rec[:, 11] = 10
        rec[:, 12:12 + L + 1] = rows
        f.write(rec.tobytes())
        del s, rows, rec
del db
torch.cuda.empty_cache()
cli = f"{ROOT}/krakenuniq_amd/bin/classify"
base = [cli, "-d", f"{tmp}/database.kdb", "-i", f"{tmp}/database.idx", "-a", f"{tmp}/taxDB", "-t", "16"]
env = dict(os.environ, KU_NO_SPARSE="1", KU_CLI_TIMES="1")
runs = {"resident": ([], {}), "chunked": (["-x", "8G"], {"KU_SUPERBATCH_BYTES": str(20 << 30)})}
for name, (extra, e) in runs.items():
    t0 = time.time()
    r = subprocess.run(base + extra + ["-o", f"{tmp}/{name}.tsv", "-r", f"{tmp}/{name}.rep", f"{tmp}/reads.fa"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env={**env, **e})
    err = r.stderr.decode(errors="replace").replace("\r", "\n")
    keep = [l for l in err.split("\n") if any(w in l for w in ("processed in", "passes over", "chunks of", "stage busy"))]
    print(f"{name}: rc={r.returncode} wall {time.time() - t0:.0f}s | " + " | ".join(keep), flush=True)
    if r.returncode:
        print(err[-600:])
        sys.exit(1)
    if os.path.exists(f"{tmp}/database.kdb.counts"):
        os.rename(f"{tmp}/database.kdb.counts", f"{tmp}/{name}.counts")
same_out = subprocess.run(["cmp", "-s", f"{tmp}/resident.tsv", f"{tmp}/chunked.tsv"]).returncode == 0
same_rep = open(f"{tmp}/resident.rep").read() == open(f"{tmp}/chunked.rep").read()
same_cnt = open(f"{tmp}/resident.counts").read() == open(f"{tmp}/chunked.counts").read()
print(f"kraken outputs identical: {same_out}; reports identical: {same_rep}; database.kdb.counts identical: {same_cnt}; "
      f"output size {os.path.getsize(tmp + '/resident.tsv') / 1e9:.2f} GB")
shutil.rmtree(tmp, ignore_errors=True)
sys.exit(0 if same_out and same_rep and same_cnt else 1)
