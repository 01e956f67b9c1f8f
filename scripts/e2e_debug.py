#!/usr/bin/env python3
"""debug aid (run through gpurun): the bench database on disk, a FASTQ file, the classify executable; prints rc + stderr tail"""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from krakenuniq_amd import synth_torch
import bench
n_species = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
tmp = "/dev/shm/ku_e2e"
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(tmp)
dev = torch.device("cuda:0")
db = synth_torch.BenchDb(dev, n_species=n_species, genome_len=310_000, k=31, nt=13, seed=7)
db.kmers = db.vals = None
db.write_files(tmp)
s, _, _, _ = db.sample_reads(n, 150, seed=1)
bench.write_fastq(f"{tmp}/reads.fq", s.view(n, 151).cpu().numpy(), 150)
del db, s
torch.cuda.empty_cache()
env = dict(os.environ)
extra = []
threads = "16"
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1)
    if k.startswith("SWEEP_"):
        continue
    if k == "REPORT":
        extra = ["-r", f"{tmp}/report.tsv"]
    elif k == "THREADS":
        threads = v
    else:
        env[k] = v
cmd = [f"{ROOT}/krakenuniq_amd/bin/classify", "-d", f"{tmp}/database.kdb", "-i", f"{tmp}/database.idx", "-a", f"{tmp}/taxDB",
       "-t", threads, "-o", f"{tmp}/out.tsv"] + extra + [f"{tmp}/reads.fq"]
sweep = [kv for kv in sys.argv[3:] if kv.startswith("SWEEP_")]
if sweep:
    k, vals = sweep[0][6:].split("=", 1)
    for v in vals.split(","):
        e2 = dict(env); e2[k] = v; e2["KU_CLI_TIMES"] = "1"
        c2 = list(cmd)
        if k == "CPULIST" and v != "all":  # SWEEP_CPULIST=all,0-63: taskset
            c2 = ["taskset", "-c", v.replace("+", ",")] + c2
        if k == "THREADS":  # SWEEP_THREADS=6,8,12: the -t value
            c2[c2.index("-t") + 1] = v
        r = subprocess.run(c2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e2)
        err = r.stderr.decode(errors="replace").replace("\r", "\n").split("\n")
        print(k, v, "rc", r.returncode, " | ".join(l.strip() for l in err if "processed in" in l or "stage busy" in l or "ku_classify_batch_rle over" in l))
    shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(0)
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
err = r.stderr.decode(errors="replace").replace("\r", "\n").split("\n")
print("rc", r.returncode)
print("\n".join(l[-300:] for l in err[-8:]))
if extra and os.path.exists(f"{tmp}/report.tsv"):
    print("".join(open(f"{tmp}/report.tsv").readlines()[:6]))
shutil.rmtree(tmp, ignore_errors=True)
