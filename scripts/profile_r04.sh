#!/bin/bash
# round-4 evidence beyond scripts/profile_bench.sh (run through gpurun from the repo root), every pass its own process:
#   read shapes of the fused kernel (kernel stats + FETCH_SIZE / WRITE_SIZE), the classify -r run, the owner-routed step
#   (eight ranks on the one device: wall; one rank forced through the routed path: clean kernel split + counters)
set -u
TAG=${1:-r04}
WHAT=${2:-"shapes cli route"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--cpu-sample 0 --no-extras --steps 4 --warmup 1"
REGEX="ku_(lookup|resolve|classify_short|route)"
if [[ $WHAT == *shapes* ]]; then
  for sh in "paired:--paired --reads 5000000" "long:--read-len 10000 --reads 100000" "nt15:--nt 15"; do
    name=${sh%%:*}; args=${sh#*:}
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${name}_stats -- python $REPO/bench.py $A $args > $OUT/${TAG}_${name}.log 2>&1
    for grp in FETCH_SIZE WRITE_SIZE; do
      [ $name = nt15 ] && continue   # (kernel stats only for the nt = 15 shape)
      timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "$REGEX" --output-format csv -d $OUT/${TAG}_${name}_$grp -- python $REPO/bench.py $A $args > $OUT/${TAG}_${name}_$grp.log 2>&1
    done
    echo "shape $name done"; tail -1 $OUT/${TAG}_${name}.log | cut -c1-300
  done
fi
if [[ $WHAT == *cli* ]]; then
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cli_report -- python $REPO/scripts/e2e_debug.py 2000 10000000 REPORT=1 KU_REPORT_TIMES=1 KU_CLI_TIMES=1 > $OUT/${TAG}_cli_report.log 2>&1
  echo "cli rc=$?"; grep -E "processed in|stage busy|Report finished" $OUT/${TAG}_cli_report.log | cut -c1-200
fi
if [[ $WHAT == *route* ]]; then
  ( cd $REPO; for i in 1 2; do timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"; done; timeout 300 python scripts/route_probe.py slots 10000000 8 2>&1 | grep "^slots" ) > $OUT/${TAG}_route_probe.log
  cat $OUT/${TAG}_route_probe.log
  KU_MGPU_FORCE_ROUTE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_route1_stats -- python $REPO/scripts/route_probe.py route 10000000 1 > $OUT/${TAG}_route1_stats.log 2>&1
  grep "^route" $OUT/${TAG}_route1_stats.log
  bash $REPO/scripts/gpu_r04_pmc.sh ${TAG}_route1 10000000 1 | grep "rc="
fi
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
find $OUT -name '*.csv' -size +8M -delete
