#!/usr/bin/env python3
"""gpurun_out/<tag>_cli_{report,plain}_{fetch,write}_pmc.json (scripts/cli_probe.py ... PMC=FETCH_SIZE | WRITE_SIZE: the fused
kernel's launches of ONE `classify` run each, 10 M bench reads) -> the `cli` entry of profiles/lookup_traffic.json: HBM bytes
of the instance the executable launches, summed over the run, with the same gfx950 correction as the headline kernel's
(FETCH_SIZE x 2).  Only for the kernel source the file is of.      usage: scripts/summarize_cli_pmc.py <tag>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krakenuniq_amd import capi
tag = sys.argv[1]
path = "profiles/lookup_traffic.json"
tj = json.load(open(path))
if tj.get("kernel_rev") != capi.kernel_rev():
    sys.exit(f"{path} is of kernel source {tj.get('kernel_rev')}, the library is {capi.kernel_rev()}: collect the headline counters first")
cli = {}
for mode in ("report", "plain"):
    got = {}
    for c, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = f"gpurun_out/{tag}_cli_{mode}_{c}_pmc.json"
        if not os.path.exists(f):
            break
        d = json.load(open(f))
        # the instance that counts (DO_COUNTS = true); the warm-up's count-less launches are left out
        ks = {k: v for k, v in d["kernels"].items() if "<2, true" in k}
        if len(ks) != 1:
            break
        k, v = next(iter(ks.items()))
        got[name] = v["sum"]
        got["kernel"], got["launches"], got["reads"] = k, v["launches"], d["reads"]
    if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
        cli[mode] = {"kernel": got["kernel"], "launches": got["launches"], "reads": got["reads"], "FETCH_SIZE_KB": got["FETCH_SIZE"],
                     "WRITE_SIZE_KB": got["WRITE_SIZE"], "hbm_bytes_per_run": int(got["FETCH_SIZE"] * 2048 + got["WRITE_SIZE"] * 1024),
                     "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `classify` ({tag}_cli_{mode}, scripts/cli_probe.py), summed over the run's launches, same correction"}
        print(mode, cli[mode]["kernel"], cli[mode]["launches"], "launches", cli[mode]["hbm_bytes_per_run"] / 1e9, "GB per 10 M reads")
if cli:
    tj["cli"] = cli
    json.dump(tj, open(path, "w"), indent=1)
