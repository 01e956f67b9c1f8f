#!/bin/bash
# round-5 evidence (run through gpurun from the repo root); every pass its own process:
#   main   : bench.py's dominant kernel -- rocprofv3 kernel stats + the PMC groups (scripts/profile_bench.sh)
#   shapes : the windowed instance (mate pairs, 10 kbp reads) and nt = 15 -- kernel stats + FETCH_SIZE / WRITE_SIZE (scripts/profile_r04.sh)
#   cli    : the `classify` executable on 10 M reads with and without -r under rocprofv3 (scripts/cli_probe.py)
#   config2 | config3 | config4: one rank's share of the 8-GPU layout of configs[2..4] on the 195 GB table -- the bench line, and the same step with
#            every round on ONE stream under rocprofv3, so that the kernels' durations do not overlap
# then locally: scripts/summarize_profile.py r05
set -u
WHAT=${1:-"main shapes cli"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
if [[ $WHAT == *main* ]]; then bash $REPO/scripts/profile_bench.sh r05 all; fi
if [[ $WHAT == *shapes* ]]; then bash $REPO/scripts/profile_r04.sh r05 shapes; fi
if [[ $WHAT == *cli* ]]; then
  ( cd $REPO && timeout 400 python scripts/cli_probe.py 2000 10000000 r05 "cli_report:REPORT=1,PROF=1" "cli_plain:PROF=1" "report_x3:REPORT=1,REPEAT=3,KU_RLE_TIMES=1" "plain_x3:REPEAT=3,KU_RLE_TIMES=1" > $OUT/r05_cli_probe.log 2>&1 )
  grep -E "^==|processed in|Report finished|ku_classify_short" $OUT/r05_cli_probe.log | cut -c1-200
fi
for C in 2 3 4; do
if [[ $WHAT == *config$C* ]]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 python $REPO/bench.py --gpus 1 --config $C --steps 6 --warmup 2 --cpu-sample 0 --no-extras > $OUT/r05_config${C}_line.json 2> $OUT/r05_config$C.err
  KU_ROUTE_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_config${C}_stats -- python $REPO/bench.py --gpus 1 --config $C --steps 6 --warmup 2 --cpu-sample 0 --no-extras > $OUT/r05_config${C}_stats.log 2>&1
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
  tail -c 400 $OUT/r05_config${C}_line.json
fi
done
find $OUT -name '*.csv' -size +8M -delete
