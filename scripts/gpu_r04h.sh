#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout 600 python -m pytest tests/test_gpu_mgpu.py -x -q 2>&1 | tail -2
echo "default"; timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
echo "items2"; KU_LIB=$REPO/scripts/libku_items2.bin timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
export KU_MGPU_FORCE_ROUTE=1
echo "W=1 default"; timeout 300 python scripts/route_probe.py route 10000000 1 2>&1 | grep "^route"
echo "W=1 items2"; KU_LIB=$REPO/scripts/libku_items2.bin timeout 300 python scripts/route_probe.py route 10000000 1 2>&1 | grep "^route"
