#!/bin/bash
# device work of a sharded step with N = 8 ownership, owner routing vs the position-wise exchange (run through gpurun)
#   scripts/profile_route.sh <round>     -> gpurun_out/prof_<round>_route{,_slots}/ + gpurun_out/route_probe.log
set -u
R=${1:-r03}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for m in route slots; do
  [ -n "${SKIP_PLAIN:-}" ] || timeout 600 python "$REPO/scripts/route_probe.py" $m 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/route_probe.log"
  d=$OUT/prof_${R}_route_$m
  rm -rf "$d"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python "$REPO/scripts/route_probe.py" $m > "$d.log" 2>&1
  find "$d" -type f ! -name "*kernel_stats.csv" -delete   # (the traces of eight ranks are large; gpurun returns at most 64 MiB)
  tail -2 "$d.log"
done
