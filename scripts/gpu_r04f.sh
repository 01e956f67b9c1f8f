#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rp in 48000000 96000000 400000000; do
  echo "round $rp"; KU_ROUTE_ROUND=$rp timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
done
echo "one stream 48M"; KU_ROUTE_ONE_STREAM=1 timeout 300 python scripts/route_probe.py route 10000000 8 2>&1 | grep "^route"
