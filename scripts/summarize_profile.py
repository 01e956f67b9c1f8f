#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_{stats,pmc*} (scripts/profile_bench.sh) into profiles/<tag>_kernel_stats.csv,
profiles/<tag>_pmc_summary.json and profiles/lookup_traffic.json.   usage: scripts/summarize_profile.py <tag>
The traffic file carries the hash of the kernel sources it was measured on (capi.kernel_rev()); bench.py refuses it for
any other source."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krakenuniq_amd import capi
tag = sys.argv[1]
rev = capi.kernel_rev()
out = {}
def newest(pattern):
    """one result file per output directory: the latest (gpurun merges into gpurun_out/, older runs may still lie there)"""
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


for f in newest(f'gpurun_out/{tag}_pmc*/runc/*counter_collection.csv'):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[(kn, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (kn, c), v in agg.items():
        vv = v[1:] if len(v) > 1 else v   # first dispatch = untimed warm-up step
        out.setdefault(kn, {})[c] = {"per_launch_mean": sum(vv) / len(vv), "launches": len(vv)}
json.dump(out, open(f'profiles/{tag}_pmc_summary.json', 'w'), indent=1, sort_keys=True)
rows = list(csv.reader(open(newest(f'gpurun_out/{tag}_stats/runc/*kernel_stats.csv')[0])))
with open(f'profiles/{tag}_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f); w.writerow(rows[0])
    for r in rows[1:]:
        if 'ku_' in r[0] or float(r[4]) >= 1.0:
            w.writerow([r[0][:160]] + r[1:])
for r in rows[1:]:
    if 'ku_' in r[0]: print(r[0][:44], 'calls', r[1], 'avg_ns', r[3])
# the other read shapes (scripts/profile_configs.sh)
for c in ("paired", "long", "nt15", "sharded8", "cli_report", "route_route", "route_slots"):
    fs = (newest(f'gpurun_out/{tag}_{c}_stats/runc/*kernel_stats.csv') or newest(f'gpurun_out/{tag}_{c}/runc/*kernel_stats.csv')
          or newest(f'gpurun_out/prof_{tag}_{c}/runc/*kernel_stats.csv'))
    if c == "cli_report":  # the profiler follows the executable the script starts: the file with our kernels
        fs = [f for f in (glob.glob(f'gpurun_out/{tag}_{c}/runc/*kernel_stats.csv')) if 'ku_classify_short' in open(f).read()][-1:]
    if not fs:
        continue
    rr = list(csv.reader(open(fs[0])))
    with open(f'profiles/{tag}_{c}_kernel_stats.csv', 'w', newline='') as f:
        w = csv.writer(f); w.writerow(rr[0])
        for r in rr[1:]:
            if 'ku_' in r[0] or float(r[4]) >= 1.0:
                w.writerow([r[0][:160]] + r[1:])
lk = next(v for k, v in out.items() if k.startswith(('ku_classify_short_kernel', 'ku_lookup_kernel<1')))
kname = next(k for k in out if k.startswith(('ku_classify_short_kernel', 'ku_lookup_kernel<1')))
fetch_kb, write_kb = lk['FETCH_SIZE']['per_launch_mean'], lk['WRITE_SIZE']['per_launch_mean']
j = {"reads": 10000000, "nt": 13, "species": 2000, "read_len": 150, "kernel": kname, "kernel_rev": rev,
     "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)",
     "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
     "correction": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM; scripts/calib_gather.hip: a random "
                   "16-B gather moves one 128-B line): read bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 as is (uncalibrated)",
     "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024)}
# the other read shapes: FETCH_SIZE / WRITE_SIZE passes of scripts/profile_shapes.sh (gpurun_out/<tag>_<shape>_{FETCH,WRITE}_SIZE)
shapes = {}
for name, key in (("paired", "reads5000000_nt13_species2000_len301_paired"), ("long", "reads100000_nt13_species2000_len10000"),
                  ("nt15", "reads10000000_nt15_species2000_len150")):
    tot, kn, launches = {}, None, 0
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = newest(f'gpurun_out/{tag}_{name}_{c}/runc/*counter_collection.csv')
        if not fs:
            break
        v = [(r['Kernel_Name'], float(r['Counter_Value'])) for r in csv.DictReader(open(fs[0]))
             if r['Counter_Name'] == c and 'ku_classify_short_kernel' in r['Kernel_Name']]
        if not v:
            break
        kn = v[0][0].split('(')[0].replace('void ', '')
        vv = [x for _, x in v][1:] if len(v) > 1 else [x for _, x in v]  # first dispatch = untimed warm-up step
        tot[c], launches = sum(vv) / len(vv), len(vv)
    if len(tot) == 2:
        shapes[key] = {"kernel": kn, "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"], "launches": launches,
                       "hbm_bytes_per_launch": int(tot["FETCH_SIZE"] * 2048 + tot["WRITE_SIZE"] * 1024),
                       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py ({tag}_{name}), same correction"}
        print(key, 'traffic GB', shapes[key]["hbm_bytes_per_launch"] / 1e9)
j["shapes"] = shapes
json.dump(j, open('profiles/lookup_traffic.json', 'w'), indent=1)
print({k: '%.4g' % v['per_launch_mean'] for k, v in lk.items()})
print('traffic GB', j['hbm_bytes_per_launch'] / 1e9)
