#!/bin/bash
# HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) of the fused kernel on the paired and 10 kbp shapes:
#   scripts/pmc_shapes.sh <tag>  -> gpurun_out/<tag>_{paired,long}_pmc{1,2}/
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
A="--cpu-sample 0 --no-extras --steps 3 --warmup 1"
for shape in "paired:--paired --reads 5000000" "long:--read-len 10000 --reads 100000"; do
  name=${shape%%:*}; args=${shape#*:}
  i=0
  for grp in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-include-regex "ku_classify_short_kernel" --output-format csv -d $OUT/${TAG}_${name}_pmc$i -- python $REPO/bench.py $A $args > $OUT/${TAG}_${name}_pmc$i.log 2>&1
    echo "$name $grp rc=$?"
  done
done
find $OUT -name '*.csv' -size +8M -delete
