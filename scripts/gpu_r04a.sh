#!/bin/bash
# round 4, first GPU check of the record-based owner routing: tests, then the W = 8 probe with and without kernel stats
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_mgpu.py -x -q > $OUT/r04a_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/r04a_pytest.log
timeout 300 python scripts/route_probe.py route 10000000 8 > $OUT/r04a_route8.log 2>&1
echo "probe rc=$?"; tail -3 $OUT/r04a_route8.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04a_route_stats -- python $REPO/scripts/route_probe.py route 10000000 8 > $OUT/r04a_route_stats.log 2>&1
echo "stats rc=$?"
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
find $OUT/r04a_route_stats -name '*kernel_stats.csv' | head -2 | xargs -r head -25
