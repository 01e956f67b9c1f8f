#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd $REPO
echo "items2"; timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
echo "items3"; KU_LIB=$REPO/scripts/libku_items3.bin timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r04i_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/r04i_pytest.log
