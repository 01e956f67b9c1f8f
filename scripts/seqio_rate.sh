#!/bin/bash
# Parsing rate of the classify executable's input stage (bin/seqio_dump) on the host it runs on: plain FASTQ, .gz through
# zlib (one inflate) and through the gzip team (ku_pgzip.h), BGZF, mate pairs; without and with the producer side (-T, as
# the executable reads).  Run on the GPU box: its host cores are the ones the CLI uses.   scripts/seqio_rate.sh [reads]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-2000000}
D=/dev/shm/ku_seqio_$$
mkdir -p $D
python3 - "$D" "$N" <<'PY'
import sys, struct, zlib, numpy as np
d, n = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(1)
q = b"I" * 150
with open(f"{d}/r.fq", "wb") as f:
    for s in range(0, n, 100000):
        seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (100000, 150), dtype=np.uint8)]
        f.write(b"".join(b"@read%d some/description\n" % (s + i) + seqs[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(100000)))
# BGZF (bgzip's layout: independent 64 KiB members with their size in an extra field)
data = open(f"{d}/r.fq", "rb").read()
def one(b):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(b) + c.flush()
    return struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(comp) + 8 - 1) + comp + struct.pack("<II", zlib.crc32(b), len(b))
with open(f"{d}/r.bgzf.gz", "wb") as f:
    for i in range(0, len(data), 65280):
        f.write(one(data[i:i + 65280]))
    f.write(one(b""))
PY
gzip -1 -c $D/r.fq > $D/r1.fq.gz; cp $D/r1.fq.gz $D/r2.fq.gz
gzip -6 -c $D/r.fq > $D/r6.fq.gz
DUMP=$REPO/krakenuniq_amd/bin/seqio_dump
echo "host: $(nproc) processors; $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
echo "files: $N reads, $(stat -c %s $D/r.fq) bytes of FASTQ; gzip -1 $(stat -c %s $D/r1.fq.gz), gzip -6 $(stat -c %s $D/r6.fq.gz), BGZF $(stat -c %s $D/r.bgzf.gz)"
echo "plain:                      $($DUMP -n $D/r.fq 2>&1)"
echo "plain -T:                   $($DUMP -n -T $D/r.fq 2>&1)"
echo "plain, 8 regions:           $($DUMP -n -w -j 8 $D/r.fq 2>&1)"
echo "gz -1, zlib:                $(KU_NO_PGZIP=1 $DUMP -n -T $D/r1.fq.gz 2>&1)"
echo "gz -6, zlib:                $(KU_NO_PGZIP=1 $DUMP -n -T $D/r6.fq.gz 2>&1)"
for t in 4 8 16; do
  echo "gz -1, team of $t:          $(KU_PGZIP_TEAM=$t $DUMP -n -T $D/r1.fq.gz 2>&1)"
  echo "gz -6, team of $t:          $(KU_PGZIP_TEAM=$t $DUMP -n -T $D/r6.fq.gz 2>&1)"
done
echo "gz -1 (default team):       $($DUMP -n -T $D/r1.fq.gz 2>&1)"
echo "BGZF (default team):        $($DUMP -n -T $D/r.bgzf.gz 2>&1)"
echo "gz -1, team + 8 parsers:    $($DUMP -n -j 8 $D/r1.fq.gz 2>&1)"
echo "gz -6, team + 8 parsers:    $($DUMP -n -j 8 $D/r6.fq.gz 2>&1)"
echo "gz -6, team 16 + 8 parsers: $(KU_PGZIP_TEAM=16 $DUMP -n -j 8 $D/r6.fq.gz 2>&1)"
echo "BGZF, team + 8 parsers:     $($DUMP -n -j 8 $D/r.bgzf.gz 2>&1)"
echo "BGZF, team 16 + 8 parsers:  $(KU_BGZF_TEAM=16 $DUMP -n -j 8 $D/r.bgzf.gz 2>&1)"
echo "gz pairs, zlib:             $(KU_NO_PGZIP=1 $DUMP -n -P -T $D/r1.fq.gz $D/r2.fq.gz 2>&1)"
echo "gz pairs, teams:            $($DUMP -n -P -T $D/r1.fq.gz $D/r2.fq.gz 2>&1)"
echo "-- the inflater alone (bytes of text per second)"
for t in 1 2 4 8 16; do
  echo "gz -6, $t threads: $($DUMP -n -z $t $D/r6.fq.gz 2>&1 | tr '\n' ' ')"
done
echo "gz -1, 8 threads: $($DUMP -n -z 8 $D/r1.fq.gz 2>&1 | tr '\n' ' ')"
echo "zcat -6: $( { time zcat $D/r6.fq.gz > /dev/null; } 2>&1 | tr '\n' ' ')"
rm -rf $D
