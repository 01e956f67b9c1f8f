#!/bin/bash
# round-6 evidence (run through gpurun from the repo root); every pass its own process:
#   main    : bench.py's dominant kernel -- rocprofv3 kernel stats + the PMC groups (scripts/profile_bench.sh)
#   shapes  : the windowed instance (mate pairs, 10 kbp reads) and nt = 15 -- kernel stats + FETCH_SIZE / WRITE_SIZE
#   cli     : the `classify` executable on 10 M reads with and without -r under rocprofv3 (kernel tables), three plain repeats,
#             and the FETCH_SIZE / WRITE_SIZE passes of both instances (lookup_traffic.json: cli)
#   config2 | config3 | config4: one rank's share of the 8-GPU layout of configs[2..4] on the 195 GB table -- the bench line, the same
#             step with every round on ONE stream under rocprofv3 (kernel durations that add up), and FETCH_SIZE / WRITE_SIZE of the
#             routed stages (route_traffic.json -- VERDICT r05 missing #3: the counters on the CURRENT kernel source)
# then locally: scripts/summarize_profile.py r06; scripts/summarize_cli_pmc.py r06; scripts/summarize_configs.py r06;
#               scripts/route_traffic.py r06_config2 <key> 4 ...
set -u
WHAT=${1:-"main shapes cli"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
T=r06
mkdir -p $OUT
if [[ $WHAT == *main* ]]; then bash $REPO/scripts/profile_bench.sh $T all; fi
if [[ $WHAT == *shapes* ]]; then bash $REPO/scripts/profile_r04.sh $T shapes; fi
if [[ $WHAT == *cli* ]]; then
  ( cd $REPO && timeout 700 python scripts/cli_probe.py 2000 10000000 $T "cli_report:REPORT=1,PROF=1" "cli_plain:PROF=1" \
      "report_x3:REPORT=1,REPEAT=3,KU_RLE_TIMES=1" "plain_x3:REPEAT=3,KU_RLE_TIMES=1" \
      "cli_report_fetch:REPORT=1,PMC=FETCH_SIZE" "cli_report_write:REPORT=1,PMC=WRITE_SIZE" "cli_plain_fetch:PMC=FETCH_SIZE" "cli_plain_write:PMC=WRITE_SIZE" \
      > $OUT/${T}_cli_probe.log 2>&1 )
  grep -E "^==|processed in|Report finished|ku_classify_short|FETCH_SIZE|WRITE_SIZE|kernels" $OUT/${T}_cli_probe.log | cut -c1-220
fi
B="--gpus 1 --steps 6 --warmup 2 --cpu-sample 0 --no-extras"
for C in 2 3 4; do
if [[ $WHAT == *config$C* ]]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 python $REPO/bench.py $B --config $C > $OUT/${T}_config${C}_line.json 2> $OUT/${T}_config$C.err
  KU_ROUTE_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${T}_config${C}_stats -- python $REPO/bench.py $B --config $C > $OUT/${T}_config${C}_stats.log 2>&1
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
  if [[ $WHAT == *pmc$C* ]]; then
    for grp in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $grp --kernel-include-regex "ku_(lookup|resolve|classify_short|route)" --output-format csv \
        -d $OUT/${T}_config${C}_$grp -- python $REPO/bench.py --gpus 1 --steps 4 --warmup 1 --cpu-sample 0 --no-extras --config $C > $OUT/${T}_config${C}_$grp.log 2>&1
      echo "pmc config $C $grp rc=$?"
    done
  fi
  tail -c 600 $OUT/${T}_config${C}_line.json
fi
done
find $OUT -name '*agent_info.csv' -delete
find $OUT -name '*.csv' -size +8M -delete
