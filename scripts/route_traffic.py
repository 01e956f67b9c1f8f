#!/usr/bin/env python3
"""profiles/route_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_r06.sh configN pmcN: HBM bytes per step
of the stages of an owner-routed step (scan, owner, resolve; the prefix-sum kernels count with the owner), per workload,
with the hash of the kernel sources they were measured on (bench.py refuses the file for any other source).
   usage: scripts/route_traffic.py <tag> <workload key> <steps in the profiled run> [<tag> <key> <steps> ...]
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 (MI355X_MICROARCH.md): x 2."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krakenuniq_amd import capi

def stage(kn):
    if "ku_lookup_kernel<3" in kn: return "scan"
    if "ku_route_owner" in kn or "ku_route_prefix" in kn or "ku_route_totals" in kn: return "owner"
    if "ku_classify_short_kernel" in kn or "ku_resolve_kernel" in kn or "ku_route_gather" in kn: return "resolve"
    return None

path = os.path.join(ROOT, "profiles", "route_traffic.json")
out = {"kernel_rev": capi.kernel_rev(), "unit": "bytes", "workloads": {}}
if os.path.exists(path):
    old = json.load(open(path))
    if old.get("kernel_rev") == out["kernel_rev"]:
        out = old
args = sys.argv[1:]
for tag, key, steps in zip(args[0::3], args[1::3], args[2::3]):
    steps = int(steps)
    tot = collections.defaultdict(float)
    for c, mul in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{c}", "*", "*counter_collection.csv"))
        assert fs, (tag, c)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            if r["Counter_Name"] != c:
                continue
            st = stage(r["Kernel_Name"])
            if st:
                tot[st] += float(r["Counter_Value"]) * mul
    out["workloads"][key] = {"hbm_bytes_per_step": {k: v / steps for k, v in tot.items()}, "steps_profiled": steps,
                             "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py ({tag}), FETCH_SIZE x 2 (gfx950), profiles/{tag}_*"}
    print(key, {k: round(v / steps / 1e9, 3) for k, v in tot.items()}, "GB per step")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
