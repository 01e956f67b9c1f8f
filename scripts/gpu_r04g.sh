#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
export KU_MGPU_FORCE_ROUTE=1
for rp in 100000000 200000000 400000000; do
  echo "W=1 two streams round $rp"; KU_ROUTE_ROUND=$rp timeout 300 python scripts/route_probe.py route 10000000 1 2>&1 | grep "^route"
  echo "W=1 one stream round $rp"; KU_ROUTE_ONE_STREAM=1 KU_ROUTE_ROUND=$rp timeout 300 python scripts/route_probe.py route 10000000 1 2>&1 | grep "^route"
done
echo "W=1 two streams 200M, owner 5 blocks/CU"; KU_ROUTE_BLOCKS_PER_CU=5 KU_ROUTE_ROUND=200000000 timeout 300 python scripts/route_probe.py route 10000000 1 2>&1 | grep "^route"
echo "W=1 two streams 200M, owner 4 blocks/CU"; KU_ROUTE_BLOCKS_PER_CU=4 KU_ROUTE_ROUND=200000000 timeout 300 python scripts/route_probe.py route 10000000 1 2>&1 | grep "^route"
