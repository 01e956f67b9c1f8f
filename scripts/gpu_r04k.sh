#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd $REPO
for nt in 13 15; do
timeout 600 python bench.py --gpus 1 --mode sharded --db-shards 8 --nt $nt --steps 4 --warmup 1 --no-extras --cpu-sample 0 > $OUT/r04k_small_nt$nt.log 2>&1
python - <<PY
import json
for l in open("$OUT/r04k_small_nt$nt.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("nt$nt", d["config"]["hbm_layout"], r["stage_ms_measured"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "rounds", r["rounds"], "kpr", r["kmers_per_record"], "log2", r["mean_ceil_log2_bin"], d["config"]["db_build_split"])
PY
done
