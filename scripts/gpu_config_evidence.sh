#!/bin/bash
# Evidence for BASELINE.json configs[2..4] on one GPU (one rank's share of the 8-GPU layout): the bench line + kernel stats of
# every configuration in one rocprofv3 run each, and the FETCH_SIZE / WRITE_SIZE passes of configs[2].  Every process
# synthesises the ~44 GB shard again (~90 s); nothing is kept in host memory between them.
#   scripts/gpu_config_evidence.sh <tag> "<configs for stats>" "<configs for pmc>"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r04}
STATS=${2-2 3 4}
PMC=${3-2}
mkdir -p $OUT
B="--gpus 1 --steps 4 --warmup 1 --no-extras --cpu-sample 0"
cd /tmp && export TMPDIR=/tmp
for c in $STATS; do
  ( time timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_config${c}_stats -- python $REPO/bench.py $B --config $c ) > $OUT/${TAG}_config${c}_stats.log 2>&1
  echo "stats $c rc=$?"; grep "^{" $OUT/${TAG}_config${c}_stats.log > $OUT/${TAG}_config${c}_line.json; grep real $OUT/${TAG}_config${c}_stats.log
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
done
for c in $PMC; do
  for grp in FETCH_SIZE WRITE_SIZE; do
    timeout 420 rocprofv3 --pmc $grp --kernel-include-regex "ku_(lookup|resolve|classify_short|route)" --output-format csv \
      -d $OUT/${TAG}_config${c}_$grp -- python $REPO/bench.py $B --config $c > $OUT/${TAG}_config${c}_$grp.log 2>&1
    echo "pmc $c $grp rc=$?"
  done
done
find $OUT -name '*agent_info.csv' -delete
find $OUT -name '*.csv' -size +8M -delete
