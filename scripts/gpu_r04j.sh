#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd $REPO
( time timeout 900 python bench.py --gpus 1 --config 2 --steps 4 --warmup 1 --no-extras --cpu-sample 0 ) > $OUT/r04j_config2.log 2>&1
echo "rc=$?"; grep -E "^\{|real|Error|error" $OUT/r04j_config2.log | cut -c1-3000
