#!/usr/bin/env python3
"""Measurement aid (gpurun): what does the fused kernel cost per read as a function of the LAUNCH size?  The bench's one launch of
10 M reads against the launches of ~120 k reads the `classify` executable issues (3.1-3.9 us per thousand
reads against 1.8).  The bench database and one 10 M-read batch stay resident; the same reads are classified in launches of n
reads each, back to back (HIP events around the whole series), in variants -- each in a process of its own:

    python scripts/launch_shape_probe.py [reads] [variant ...]      variant = name:ENV=VAL,ENV=VAL,...  (KU_LIB=... selects a build)

Per variant and launch size: ms per 10 M reads with accounting (runs form, OUT = 1), without accounting, and with the launches
dealt out over two streams (kernels of consecutive launches may overlap: one's tail under the next one's start)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = "/dev/shm/ku_launch_probe"


def child(n_total):
    import numpy as np
    import torch
    from krakenuniq_amd import capi
    dev = torch.device("cuda:0")
    cdb = capi.Db(f"{TMP}/database.kdb", f"{TMP}/database.idx")
    ctax = capi.Tax(f"{TMP}/taxDB")
    ctx = capi.Ctx(0)
    ctx.load_db(cdb)
    ctx.set_taxonomy(ctax)
    reads = torch.from_numpy(np.fromfile(f"{TMP}/reads.bin", dtype=np.uint8)).to(dev)
    stride = 151
    n_all = reads.numel() // stride
    n_total = min(n_total, n_all)
    off = (torch.arange(n_all, dtype=torch.int64, device=dev) * stride)
    lens = torch.full((n_all,), 150, dtype=torch.int32, device=dev)
    n_bytes = reads.numel()
    calls = torch.zeros(n_all, dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    sizes = [int(x) for x in os.environ.get("KU_PROBE_SIZES", "60000,120000,240000,1000000,10000000").split(",")]
    for n in sizes:
        n = min(n, n_total)
        n_launch = n_total // n
        runs_cap = ctx.device_rle_runs_cap(n * stride, n, 150)
        bufs = [(torch.zeros((runs_cap, 2), dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev),
                 torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)) for _ in range(2)]
        res = {}
        for label, flags, n_streams in (("counts", 0, 1), ("no_counts", capi.KU_F_NO_COUNTS, 1), ("counts_2streams", 0, 2)):
            best = None
            for rep in range(3):
                ctx.reset_counts()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                join = [torch.cuda.Event() for _ in range(n_streams)]
                e0.record(streams[0])
                if n_streams == 2:
                    streams[1].wait_event(e0)
                for i in range(n_launch):
                    s = streams[i % n_streams]
                    rb, ro, rc, nr = bufs[i % n_streams]
                    lo = i * n
                    ctx.classify_batch_device_rle(reads.data_ptr(), n_bytes, off.data_ptr() + 8 * lo, lens.data_ptr() + 4 * lo, n, calls.data_ptr() + 4 * lo,
                                                  rb.data_ptr(), runs_cap, ro.data_ptr(), rc.data_ptr(), nr.data_ptr(), max_read_len=150,
                                                  flags=flags, stream=s.cuda_stream)
                if n_streams == 2:
                    join[1].record(streams[1])
                    streams[0].wait_event(join[1])
                e1.record(streams[0])
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                best = ms if best is None else min(best, ms)
            res[label] = best * (10_000_000 / (n * n_launch))
        print(f"   {n:9d} reads per launch x {n_launch:4d}: " + ", ".join(f"{k} {v:7.2f}" for k, v in res.items()) + "   (ms per 10 M reads)", flush=True)
    del ctx


def child_host(n_total):
    """the host-buffer boundary as the executable uses it: batches of n reads through ku_classify_batch_rle_enqueue / _finish, ONE in
    flight (every launch alone on the device, as in a run whose parser cannot keep the device busy) or three; with and without the
    sparse-sketch emulation (OUT = 2 / OUT = 1).  Kernel time: the library's HIP events (KU_RLE_TIMES), printed when the context goes."""
    import numpy as np
    import torch
    from krakenuniq_amd import capi
    stride = 151
    reads = np.fromfile(f"{TMP}/reads.bin", dtype=np.uint8)
    n_all = len(reads) // stride
    n_total = min(n_total, n_all)
    pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()
    host = pin(len(reads), torch.uint8)
    host[:] = reads
    per = int(os.environ.get("KU_PROBE_HOST_READS", "111000"))
    sparse = os.environ.get("KU_PROBE_SPARSE") == "1"
    depth = int(os.environ.get("KU_PROBE_DEPTH", "1"))
    cdb = capi.Db(f"{TMP}/database.kdb", f"{TMP}/database.idx")
    ctax = capi.Tax(f"{TMP}/taxDB")
    ctx = capi.Ctx(0)
    ctx.load_db(cdb)
    ctx.set_taxonomy(ctax)
    if sparse:
        ctx.enable_sparse(500000)
    ctx.rle_reserve(per * stride, per, 150, 4)
    off = pin(per, torch.int64).view(np.uint64)
    lens = pin(per, torch.int32).view(np.uint32)
    off[:] = np.arange(per, dtype=np.uint64) * stride
    lens[:] = 150
    mk = lambda: {"calls": pin(per, torch.int32).view(np.uint32), "hits": pin(per, torch.int32).view(np.uint32),
                  "run_cnt": pin(per, torch.int32).view(np.uint32), "run_off": pin(per, torch.int64).view(np.uint64),
                  "runs": pin((per * 6 + (1 << 16), 2), torch.int32).view(np.uint32)}
    outs = [mk() for _ in range(4)]
    n_b = n_total // per
    t0 = time.perf_counter()
    flying = []
    for i in range(n_b):
        if len(flying) >= depth:
            ctx.rle_finish(flying.pop(0))
        flying.append(ctx.rle_enqueue(host[i * per * stride:(i + 1) * per * stride], off, lens, out=outs[i % 4]))
    while flying:
        ctx.rle_finish(flying.pop(0))
    dt = time.perf_counter() - t0
    print(f"   host batches of {per} reads x {n_b}, depth {depth}, sparse emulation {'on' if sparse else 'off'}: {dt * 1e3:.1f} ms wall for {n_b * per} reads", flush=True)
    del ctx  # (prints the library's timers)


def main():
    n_total = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10_000_000
    if os.environ.get("KU_LAUNCH_PROBE_CHILD"):
        return child_host(n_total) if os.environ.get("KU_PROBE_HOST") == "1" else child(n_total)
    import shutil
    import torch
    from krakenuniq_amd import synth_torch
    shutil.rmtree(TMP, ignore_errors=True)
    os.makedirs(TMP)
    dev = torch.device("cuda:0")
    db = synth_torch.BenchDb(dev, n_species=2000, genome_len=310_000, k=31, nt=13, seed=7)
    db.kmers = db.vals = None
    db.write_files(TMP)
    s, _, _, _ = db.sample_reads(n_total, 150, seed=1)
    s.reshape(-1).cpu().numpy().tofile(f"{TMP}/reads.bin")
    del db, s
    torch.cuda.empty_cache()
    variants = [] if os.environ.get("KU_PROBE_NO_DEFAULT") else [("default", {})]
    for spec in sys.argv[2:]:
        name, _, envs = spec.partition(":")
        variants.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
    for name, env in variants:
        print(f"== {name} {env}", flush=True)
        t0 = time.time()
        r = subprocess.run([sys.executable, __file__, str(n_total)], env=dict(os.environ, KU_LAUNCH_PROBE_CHILD="1", KU_RLE_TIMES="1", **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(r.stdout if r.returncode == 0 else f"   rc {r.returncode}: {r.stderr[-600:]}", flush=True)
        for line in r.stderr.split("\n"):
            if "ku_classify_batch_rle over" in line:
                m = __import__("re").search(r"kernels ([\d.]+) ms for (\d+) reads.*?their sum is ([\d.]+) ms", line)
                if m:
                    print(f"   kernels: {float(m.group(1)) * 1e7 / int(m.group(2)):.2f} ms per 10 M reads covered, {float(m.group(3)) * 1e7 / int(m.group(2)):.2f} summed", flush=True)
        print(f"   ({time.time() - t0:.0f} s)", flush=True)
    shutil.rmtree(TMP, ignore_errors=True)


if __name__ == "__main__":
    main()
