#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r04e}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_mgpu.py -x -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/${TAG}_pytest.log
for bpc in 5 6 7; do
  echo "owner blocks per CU $bpc"; KU_ROUTE_BLOCKS_PER_CU=$bpc timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
done
echo "one stream"; KU_ROUTE_ONE_STREAM=1 timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
for rp in 24000000 96000000 400000000; do
  echo "round $rp"; KU_ROUTE_ROUND=$rp timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
done
