#!/bin/bash
# round-3 evidence beyond scripts/profile_bench.sh (run through gpurun from the repo root):
#   sharded lookup with 1/8 ownership on one GPU, the other read shapes, the classify -r run
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--cpu-sample 0 --no-extras --steps 4 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sharded8_stats -- python $REPO/bench.py $A --mode sharded --db-shards 8 > $OUT/${TAG}_sharded8.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_paired_stats -- python $REPO/bench.py $A --paired --reads 5000000 > $OUT/${TAG}_paired.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_long_stats -- python $REPO/bench.py $A --read-len 10000 --reads 100000 > $OUT/${TAG}_long.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_nt15_stats -- python $REPO/bench.py $A --nt 15 > $OUT/${TAG}_nt15.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cli_report -- python $REPO/scripts/e2e_debug.py 2000 10000000 REPORT=1 KU_REPORT_TIMES=1 > $OUT/${TAG}_cli_report.log 2>&1
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete   # (gpurun returns at most 64 MiB)
find $OUT -name '*.csv' -size +8M -delete
for c in sharded8 paired long nt15; do tail -1 $OUT/${TAG}_$c.log | cut -c1-600; done
