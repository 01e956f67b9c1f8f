#!/bin/bash
# measurement (gpurun): the bench step (configs[1], one launch of 10 M reads) at the probe table's load factors -- bytes of HBM per
# pair against ms per step on the current kernels (DESIGN 2 "Size"; VERDICT r05 #4 asks for <= 40 B per pair)
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
for lf in 0.2 0.3 0.45 0.6 0.8; do
  KU_LOAD_FACTOR=$lf timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-extras > $OUT/r06_load_$lf.json 2> $OUT/r06_load_$lf.err
  python - $lf $OUT/r06_load_$lf.json <<'PY'
import json, sys
lf, p = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(p) if l.startswith("{")][0])
    c = d["config"]
    print(f"load {lf}: {c['hbm_layout']['resident_bytes'] / c['db_pairs_per_gpu']:.1f} B per pair ({c['hbm_layout']['resident_bytes'] / 1e9:.1f} GB), "
          f"{d['ms_per_step']:.2f} ms per step, kernel {d['roofline']['kernel_ms']:.2f} ms, {d['value']:.1f} Mreads/s")
except Exception as e:
    print(f"load {lf}: failed ({e})")
PY
done
