#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE read shapes on the bench database (run through gpurun):
#   paired 2x150 (configs[3]-style) and 10 kbp reads (configs[4]-style) -> gpurun_out/<tag>_{paired,long}_stats/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_paired_stats -- python $REPO/bench.py --cpu-sample 0 --steps 2 --warmup 1 --paired --reads 5000000 > $OUT/${TAG}_paired.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_long_stats -- python $REPO/bench.py --cpu-sample 0 --steps 2 --warmup 1 --read-len 10000 --reads 100000 > $OUT/${TAG}_long.log 2>&1
find $OUT -name '*.csv' -size +8M -delete
tail -1 $OUT/${TAG}_paired.log | cut -c1-160; tail -1 $OUT/${TAG}_long.log | cut -c1-160
