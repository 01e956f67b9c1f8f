"""End-to-end rate of the drop-in `classify` executable (host pipeline included): f1 reads replicated
to N reads in /dev/shm, tiny DB.  python scripts/cli_e2e.py [n_million] [threads]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krakenuniq_amd import synth
n_m = int(sys.argv[1]) if len(sys.argv) > 1 else 2
thr = sys.argv[2] if len(sys.argv) > 2 else "8"
ids, seqs = synth.read_seqfile(f"{ROOT}/tests/golden/f1/reads.fq")
path = "/dev/shm/ku_big.fq"
with open(path, "wb") as f:
    for rep in range(n_m * 1000):
        f.write(b"".join(b"@r%d_%d\n" % (rep, i) + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in enumerate(seqs)))
db = f"{ROOT}/tests/golden/f1"
for out in ("/dev/shm/ku_out.tsv", "off"):
    t = time.time()
    r = subprocess.run([f"{ROOT}/krakenuniq_amd/bin/classify", "-d", f"{db}/database.kdb", "-i", f"{db}/database.idx", "-a", f"{db}/taxDB",
                        "-t", thr, "-o", out, path], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, env=dict(os.environ, KU_CLI_TIMES="1"))
    line = [l for l in r.stderr.decode().replace("\r", "\n").split("\n") if "processed in" in l or "stage busy" in l]
    print(f"-o {out}: wall {time.time() - t:.2f}s rc={r.returncode}", " | ".join(line) if line else r.stderr.decode()[-300:])
for team in ("12", "16"):  # the parser team's size (default: 8)
    t = time.time()
    r = subprocess.run([f"{ROOT}/krakenuniq_amd/bin/classify", "-d", f"{db}/database.kdb", "-i", f"{db}/database.idx", "-a", f"{db}/taxDB",
                        "-t", thr, "-o", "/dev/shm/ku_out2.tsv", path], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL,
                       env=dict(os.environ, KU_CLI_TIMES="1", KU_PARSE_TEAM=team))
    line = [l for l in r.stderr.decode().replace("\r", "\n").split("\n") if "processed in" in l or "stage busy" in l]
    print(f"KU_PARSE_TEAM={team}: wall {time.time() - t:.2f}s rc={r.returncode}", " | ".join(line) if line else r.stderr.decode()[-300:], flush=True)
# the same reads as one gzip stream: zlib's single inflate against the gzip team (ku_pgzip.h)
# (written the way pigz writes: chunks deflated side by side, each closed with a sync flush, concatenated into ONE deflate
#  stream inside one gzip member -- no member boundaries, no index; `gzip -6` of 3 GB alone would take a minute)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import write_one_stream_gz
t = time.time()
write_one_stream_gz.write(path)
print(f"{path}.gz: {os.path.getsize(path + '.gz')} bytes, one deflate stream (level 6), written in {time.time() - t:.1f}s", flush=True)
for label, env in (("zlib (one inflate, one parser)", {"KU_NO_PGZIP": "1"}), ("gzip team, one parser", {"KU_NO_GZ_REGIONS": "1"}),
                   ("gzip team + parser team", {}), ("gzip team of 16 + parser team", {"KU_PGZIP_TEAM": "16"})):
    t = time.time()
    r = subprocess.run([f"{ROOT}/krakenuniq_amd/bin/classify", "-d", f"{db}/database.kdb", "-i", f"{db}/database.idx", "-a", f"{db}/taxDB",
                        "-t", thr, "-o", "/dev/shm/ku_out_gz.tsv", path + ".gz"], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL,
                       env=dict(os.environ, KU_CLI_TIMES="1", **env))
    line = [l for l in r.stderr.decode().replace("\r", "\n").split("\n") if "processed in" in l or "stage busy" in l]
    same = subprocess.run(["cmp", "-s", "/dev/shm/ku_out.tsv", "/dev/shm/ku_out_gz.tsv"]).returncode == 0
    print(f".gz input, {label}: wall {time.time() - t:.2f}s rc={r.returncode} output identical to the plain run: {same}", " | ".join(line) if line else r.stderr.decode()[-300:])
os.remove(path)
os.remove(path + ".gz")
for f in ("/dev/shm/ku_out.tsv", "/dev/shm/ku_out2.tsv", "/dev/shm/ku_out_gz.tsv"):
    if os.path.exists(f):
        os.remove(f)
