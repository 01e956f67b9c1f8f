"""One sweep of host-side knobs of the `classify` executable on 10 M x 150 bp FASTQ reads (tiny f1 database: the host
pipeline is what is measured): every configuration twice, interleaved.   python scripts/e2e_sweep.py [n_million]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krakenuniq_amd import synth
n_m = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ids, seqs = synth.read_seqfile(f"{ROOT}/tests/golden/f1/reads.fq")
block = b"".join(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in enumerate(seqs))
path = "/dev/shm/ku_sweep.fq"
with open(path, "wb") as f:
    for rep in range(n_m * 1000):
        f.write(block)
db = f"{ROOT}/tests/golden/f1"
configs = [("batch 32 Mi nt", "16", {"KU_BATCH_NT": str(32 << 20)}),
           ("batch 24 Mi nt", "16", {"KU_BATCH_NT": str(24 << 20)}),
           ("batch 16 Mi nt", "16", {"KU_BATCH_NT": str(16 << 20)}),
           ("batch 12 Mi nt", "16", {"KU_BATCH_NT": str(12 << 20)}),
           ("batch 8 Mi nt", "16", {"KU_BATCH_NT": str(8 << 20)}),
           ("batch 16 Mi nt, -t 32", "32", {"KU_BATCH_NT": str(16 << 20)})]
if os.environ.get("KU_SWEEP") == "knobs":  # the first sweep of round 4 (profiles/r04_e2e_sweep.log, upper half)
    configs = [("default (-t 16)", "16", {}),
               ("KU_MALLOPT=1", "16", {"KU_MALLOPT": "1"}),
               ("batch 128 Mi nt", "16", {"KU_BATCH_NT": str(128 << 20)}),
               ("batch 128 Mi nt, KU_MALLOPT=1", "16", {"KU_BATCH_NT": str(128 << 20), "KU_MALLOPT": "1"}),
               ("batch 32 Mi nt", "16", {"KU_BATCH_NT": str(32 << 20)}),
               ("-t 32", "32", {}),
               ("-t 8", "8", {}),
               ("batch 128 Mi nt, 12 parsers", "16", {"KU_BATCH_NT": str(128 << 20), "KU_PARSE_TEAM": "12"})]
res = {c[0]: [] for c in configs}
busy = {}
for rep in range(2):
    for name, thr, env in configs:
        r = subprocess.run([f"{ROOT}/krakenuniq_amd/bin/classify", "-d", f"{db}/database.kdb", "-i", f"{db}/database.idx", "-a", f"{db}/taxDB",
                            "-t", thr, "-o", "/dev/shm/ku_sweep.tsv", path], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL,
                           env=dict(os.environ, KU_CLI_TIMES="1", **env))
        err = r.stderr.decode().replace("\r", "\n")
        m = re.search(r"processed in ([\d.]+)s", err)
        res[name].append(float(m.group(1)) if m and r.returncode == 0 else None)
        b = re.search(r"stage busy seconds: (.*)", err)
        busy[name] = b.group(1) if b else err[-200:]
for name, _, _ in configs:
    print(f"{name:34s} {res[name]}  | {busy[name]}", flush=True)
for f in (path, "/dev/shm/ku_sweep.tsv"):
    if os.path.exists(f):
        os.remove(f)
