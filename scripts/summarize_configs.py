#!/usr/bin/env python3
"""gpurun_out/<tag>_config{2,3,4}_{line.json,stats/} (scripts/profile_r06.sh configN) -> profiles/<tag>_configN_line.json and
profiles/<tag>_configN_kernel_stats.csv (our kernels + every kernel >= 1 % of GPU time), and the check VERDICT r04 #4 asks
for: the stage kernels' durations in the rocprofv3 table (rounds on ONE stream: they do not overlap), per step, against the
stage times the bench line measured with HIP events.      usage: scripts/summarize_configs.py <tag>"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
for c in (2, 3, 4):
    line = f"gpurun_out/{tag}_config{c}_line.json"
    fs = sorted(glob.glob(f"gpurun_out/{tag}_config{c}_stats/runc/*kernel_stats.csv"), key=os.path.getmtime)
    if not os.path.exists(line) or not fs:
        continue
    shutil.copy(line, f"profiles/{tag}_config{c}_line.json")
    rows = list(csv.reader(open(fs[-1])))
    with open(f"profiles/{tag}_config{c}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "ku_" in r[0] or float(r[4]) >= 1.0:
                w.writerow([r[0][:160]] + r[1:])
    d = json.load(open(line))
    rf = d["roofline"]
    rounds = rf.get("rounds", 1)
    stage = {"scan": ("ku_lookup_kernel<3",), "resolve": ("ku_classify_short_kernel", "ku_resolve_kernel"), "owner": ("ku_route_owner_kernel",),
             "prefix": ("ku_route_prefix", "ku_route_totals")}
    tot, calls = {k: 0.0 for k in stage}, {k: 0 for k in stage}
    for r in rows[1:]:
        for k, pats in stage.items():
            if any(p in r[0] for p in pats) and not (k == "resolve" and "true>" not in r[0].split("(")[0] and "ku_classify_short" in r[0]):
                tot[k] += float(r[2]) / 1e6
                calls[k] = max(calls[k], int(r[1]))
    steps = max(1.0, calls["owner"] / max(rounds, 1))
    per_step = {k: round(v / steps, 3) for k, v in tot.items()}
    print(f"config {c}: {steps:.0f} steps of {rounds:.0f} rounds in the profile; per step (ms): {per_step}, sum {sum(per_step.values()):.2f}; "
          f"the line: stages {rf.get('stage_ms_measured')}, sum {rf.get('measured_stage_sum_ms')}, ms_per_step {d['ms_per_step']}, modelled kernel_ms {rf.get('kernel_ms')}")
