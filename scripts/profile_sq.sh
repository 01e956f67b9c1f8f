#!/bin/bash
# quick instruction-mix profile of the bench kernels: scripts/profile_sq.sh <tag> [bench args]
set -u
TAG=${1:-sq}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-sample 0 --no-extras --steps 4 --warmup 1 $*"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-include-regex "ku_(lookup|resolve|classify_short)_kernel" --output-format csv -d $OUT/${TAG}_pmcA -- python $REPO/bench.py $ARGS > $OUT/${TAG}_pmcA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-include-regex "ku_(lookup|resolve|classify_short)_kernel" --output-format csv -d $OUT/${TAG}_pmcB -- python $REPO/bench.py $ARGS > $OUT/${TAG}_pmcB.log 2>&1
python - <<PY
import csv, glob, collections
for grp in "AB":
    for f in glob.glob("$OUT/${TAG}_pmc%s/*/*counter_collection.csv" % grp):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'].split('(')[0][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()):
            vv = v[1:] if len(v) > 1 else v
            print(k[0], k[1], '%.4g' % (sum(vv) / len(vv)), len(vv))
PY
find $OUT -name '*.csv' -size +8M -delete
