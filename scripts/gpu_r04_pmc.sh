#!/bin/bash
# PMC passes over the routed step of a world of one rank (forced): owner / scan / resolve kernels
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
TAG=${1:-r04pmc}
N=${2:-10000000}
W=${3:-1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  KU_MGPU_FORCE_ROUTE=1 timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "ku_(lookup|resolve|classify_short|route)" --output-format csv \
    -d $OUT/${TAG}_pmc$i -- python $REPO/scripts/route_probe.py route $N $W > $OUT/${TAG}_pmc$i.log 2>&1
  echo "pmc group $i ($grp): rc=$?"
done
find $OUT -name '*agent_info.csv' -delete
python3 - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$OUT/${TAG}_pmc*/runc/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[(kn, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (kn, c), v in agg.items():
        out.setdefault(kn, {})[c] = {"sum": sum(v), "launches": len(v)}
json.dump(out, open("$OUT/${TAG}_pmc_summary.json", "w"), indent=1, sort_keys=True)
for kn in out:
    print(kn, {c: (round(x["sum"] / 1e6, 1), x["launches"]) for c, x in out[kn].items()})
PY
find $OUT -name '*counter_collection.csv' -size +4M -delete
