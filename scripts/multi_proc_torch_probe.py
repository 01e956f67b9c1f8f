#!/usr/bin/env python3
"""Measurement aid (gpurun): n processes of plain torch (sort / bincount / .item(), nothing of this repo) on ONE device.  On the
round-6 box: 1 and 2 processes 0.3 s each, 4 processes 176 s (they only moved once one of them had been killed) -- which is why
the plain `bench.py --gpus N` flow is rehearsed with two ranks on a 1-GPU box (tests/test_gpu_bench_contract.py,
profiles/r06_gpus2_rehearsal_line.json) and not with four or eight: the database synthesis of four ranks sharing a device does
not return.  python scripts/multi_proc_torch_probe.py <n>"""
import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    r = int(sys.argv[2])
    dev = torch.device("cuda:0")
    t0 = time.time()
    x = torch.randint(0, 1 << 60, (50_000_000,), device=dev)
    for i in range(10):
        y, o = torch.sort(x)
        n = int((y[1:] != y[:-1]).sum().item())
        c = torch.bincount((y & 1023))
    torch.cuda.synchronize()
    print(f"child {r}: 10 sorts of 50 M keys in {time.time() - t0:.1f} s", flush=True)
else:
    n = int(sys.argv[1])
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, __file__, "child", str(i)]) for i in range(n)]
    for p in ps:
        try:
            p.wait(timeout=120)
        except subprocess.TimeoutExpired:
            print("timeout: a child did not finish in 120 s", flush=True)
            p.kill()
    print(f"{n} processes on one device: {time.time() - t0:.1f} s", flush=True)
