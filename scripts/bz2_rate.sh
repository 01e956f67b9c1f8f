#!/bin/bash
# Rate of the .bz2 side of the input stage on the host it runs on (bin/seqio_dump): the decoder team alone at 1..32 threads,
# the reader and the region parsers behind it, bzip2 -dc beside them.   scripts/bz2_rate.sh [reads]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-400000}
D=/dev/shm/ku_bz2_$$
mkdir -p $D
python3 - "$D" "$N" <<'PY'
import sys, bz2, numpy as np
from concurrent.futures import ProcessPoolExecutor
d, n = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(1)
parts = []
for s in range(0, n, 100000):
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (100000, 150), dtype=np.uint8)]
    quals = (rng.integers(0, 40, (100000, 150)) // 8 * 8 + 33).astype(np.uint8)
    parts.append(b"".join(b"@read%d some/description\n" % (s + i) + seqs[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n" for i in range(100000)))
text = b"".join(parts)
open(f"{d}/r.fq", "wb").write(text)
# ONE bzip2 stream of 900 kB blocks, made quickly: the blocks of a stream are independent, so are streams -- bzip2 -9 itself
# runs below for the file the rates are measured on (a single stream)
PY
( time bzip2 -9 -k $D/r.fq ) 2>&1 | grep real | sed 's/^/bzip2 -9 took /'
DUMP=$REPO/krakenuniq_amd/bin/seqio_dump
echo "host: $(nproc) processors; $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
echo "file: $N reads, $(stat -c %s $D/r.fq) bytes of FASTQ, $(stat -c %s $D/r.fq.bz2) as .bz2 (one stream, 900 kB blocks)"
echo "bzip2 -dc: $( { time bzip2 -dc $D/r.fq.bz2 > /dev/null; } 2>&1 | tr '\n' ' ')"
for t in 1 4 8 16 32; do
  echo "decoder team of $t: $($DUMP -n -Z $t $D/r.fq.bz2 2>&1)"
done
echo "reader (-T, default team):     $($DUMP -n -T $D/r.fq.bz2 2>&1)"
echo "team + 8 region parsers:       $($DUMP -n -j 8 $D/r.fq.bz2 2>&1)"
echo "team of 32 + 8 region parsers: $(KU_PBZIP2_TEAM=32 $DUMP -n -j 8 $D/r.fq.bz2 2>&1)"
rm -rf $D
