#!/bin/bash
# routed step: wall at W = 8 on one device, and the clean kernel split of a world of one rank forced through the routed path
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r04b}
mkdir -p $OUT
cd $REPO
timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route" > $OUT/${TAG}_route8.log
cat $OUT/${TAG}_route8.log
cd /tmp && export TMPDIR=/tmp
KU_MGPU_FORCE_ROUTE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_route1_stats -- python $REPO/scripts/route_probe.py route 10000000 1 > $OUT/${TAG}_route1_stats.log 2>&1
echo "stats rc=$?"; grep "^route" $OUT/${TAG}_route1_stats.log
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
f=$(find $OUT/${TAG}_route1_stats -name '*kernel_stats.csv' | head -1)
grep -E "ku_|rocclr" $f | cut -d, -f1-4 | sed 's/(Ku[^"]*//' | head -20
