#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE read shapes on the bench database (run through gpurun):
#   paired 2x150 (configs[3]-style), 10 kbp reads (configs[4]-style), nt = 15 (configs[2] geometry) -> gpurun_out/<tag>_{paired,long,nt15}_stats/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--cpu-sample 0 --no-extras --steps 4 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_paired_stats -- python $REPO/bench.py $A --paired --reads 5000000 > $OUT/${TAG}_paired.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_long_stats -- python $REPO/bench.py $A --read-len 10000 --reads 100000 > $OUT/${TAG}_long.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_nt15_stats -- python $REPO/bench.py $A --nt 15 > $OUT/${TAG}_nt15.log 2>&1
find $OUT -name '*.csv' -size +8M -delete
for c in paired long nt15; do tail -1 $OUT/${TAG}_$c.log | cut -c1-400; done
