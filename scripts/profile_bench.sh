#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   scripts/profile_bench.sh <tag> [stats|pmc]   -> gpurun_out/<tag>_*/
# Kernel-trace/stats and every --pmc group are separate runs (gpurun refuses combined modes).
# PMC passes are restricted to our kernels (--kernel-include-regex): counter collection serialises
# every profiled dispatch and the torch DB build launches ~12k kernels.
set -u
TAG=${1:-r01}
WHAT=${2:-all}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-sample 0 --no-extras --steps 4 --warmup 1"
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python $REPO/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "ku_(lookup|resolve|classify_short|rle)_kernel" --output-format csv \
      -d $OUT/${TAG}_pmc$i -- python $REPO/bench.py $ARGS > $OUT/${TAG}_pmc$i.log 2>&1
    echo "pmc group $i ($grp): rc=$?"
  done
fi
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete   # (gpurun returns at most 64 MiB)
find $OUT -name '*.csv' -size +8M -delete
find $OUT -name '*counter_collection.csv' | head
