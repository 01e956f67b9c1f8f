#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r04c}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_mgpu.py -x -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/${TAG}_pytest.log
bash scripts/gpu_r04b.sh $TAG
