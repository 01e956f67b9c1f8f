#!/bin/bash
# end-of-round check (gpurun): the whole GPU suite, smoke(), the default bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r04}
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_final_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/${TAG}_final_pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_final.json 2> $OUT/${TAG}_bench_final.err
echo "bench rc=$?"; cut -c1-400 $OUT/${TAG}_bench_final.json
