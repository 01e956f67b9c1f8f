#!/bin/bash
# extra counter passes for the dominant kernel (instruction issue mix): scripts/pmc_extra.sh <tag>
# NOTE: the TA_* / TCP_* / TD_* groups were tried here and never returned on this pool (each pass ran into its 300 s
# timeout and burned GPU budget) -- do not add them back without a much shorter workload.
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-sample 0 --no-extras --steps 2 --warmup 1 ${BENCH_EXTRA:-}"
i=0
for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "ku_classify_short_kernel" --output-format csv -d $OUT/${TAG}_x$i -- python $REPO/bench.py $ARGS > $OUT/${TAG}_x$i.log 2>&1
  echo "group $i rc=$?"
  f=$(find $OUT/${TAG}_x$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<PY
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c,v in agg.items():
    vv=v[1:] if len(v)>1 else v
    print("  %-44s %.5g"%(c,sum(vv)/len(vv)))
PY
done
find $OUT -name '*.csv' -size +8M -delete
