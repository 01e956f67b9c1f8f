#!/bin/bash
# Parsing rate of the classify executable's input stage (bin/seqio_dump), plain / gz / mate pairs, without and
# with the per-file producer thread (-T).  Run on the GPU box: its host cores are the ones the CLI uses.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
D=/dev/shm/ku_seqio_$$
mkdir -p $D
python3 - "$D" <<'PY'
import sys, numpy as np
d = sys.argv[1]
rng = np.random.default_rng(1)
n = 1_000_000
seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150), dtype=np.uint8)]
q = b"I" * 150
with open(f"{d}/r.fq", "wb") as f:
    for s in range(0, n, 100000):
        f.write(b"".join(b"@read%d some/description\n" % i + seqs[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(s, s + 100000)))
PY
gzip -1 -c $D/r.fq > $D/r1.fq.gz; cp $D/r1.fq.gz $D/r2.fq.gz
for t in "" "-T"; do
  echo "plain $t:  $($REPO/krakenuniq_amd/bin/seqio_dump -n $t $D/r.fq 2>&1)"
  echo "gz $t:     $($REPO/krakenuniq_amd/bin/seqio_dump -n $t $D/r1.fq.gz 2>&1)"
  echo "gz pairs $t: $($REPO/krakenuniq_amd/bin/seqio_dump -n -P $t $D/r1.fq.gz $D/r2.fq.gz 2>&1)"
done
rm -rf $D
