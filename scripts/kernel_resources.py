#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every kernel of one .hip file, as hipcc's resource-usage remarks report
them (cross-compiles for gfx950, no GPU needed):   scripts/kernel_resources.py krakenuniq_amd/csrc/ku_short.hip [-D...]"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include",
           f"-I{ROOT}/krakenuniq_amd/csrc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
    err = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stderr
    cur = None
    rows = []
    for line in err.splitlines():
        m = re.search(r"remark: .*?: +Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name)}
            rows.append(cur)
            continue
        m = re.search(r"remark: +([\w][\w \[\]/]*?): +(\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for r in rows:
        print(f"{r['name'][:90]:90s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):3d} SGPR {r.get('TotalSGPRs', -1):4d} "
              f"scratch {r.get('ScratchSize [bytes/lane]', -1):4d} LDS {r.get('LDS Size [bytes/block]', -1):6d} "
              f"occ {r.get('Occupancy [waves/SIMD]', -1)} spillS {r.get('SGPRs Spill', -1)} spillV {r.get('VGPRs Spill', -1)}")


if __name__ == "__main__":
    main()
