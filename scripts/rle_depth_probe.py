#!/usr/bin/env python3
"""Measurement aid (gpurun): the two-step batch call alone -- no parser, no formatter -- on batches of the size the `classify`
executable sends (default 61 k reads of 150 bp), 164 of them, with 1..4 in flight: what a batch costs the calling thread
(enqueue) and how long it waits (finish), per environment variant (each in a process of its own: HSA / HIP variables are read
at start-up).   python scripts/rle_depth_probe.py [reads_per_batch] [n_batches]
The parent builds the bench database once and writes it to /dev/shm; the children open the files."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = "/dev/shm/ku_depth_probe"


def child(per_batch, n_batches):
    import numpy as np, torch
    from krakenuniq_amd import capi
    cdb = capi.Db(f"{TMP}/database.kdb", f"{TMP}/database.idx")
    ctax = capi.Tax(f"{TMP}/taxDB")
    ctx = capi.Ctx(0)
    ctx.load_db(cdb)
    ctx.set_taxonomy(ctax)
    reads = np.fromfile(f"{TMP}/reads.bin", dtype=np.uint8)
    stride = 151
    n_all = len(reads) // stride
    pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()
    host = pin(len(reads), torch.uint8)
    host[:] = reads
    off = pin(per_batch, torch.int64).view(np.uint64)
    lens = pin(per_batch, torch.int32).view(np.uint32)
    off[:] = np.arange(per_batch, dtype=np.uint64) * stride
    lens[:] = 150
    mk = lambda: {"calls": pin(per_batch, torch.int32).view(np.uint32), "hits": pin(per_batch, torch.int32).view(np.uint32),
                  "run_cnt": pin(per_batch, torch.int32).view(np.uint32), "run_off": pin(per_batch, torch.int64).view(np.uint64),
                  "runs": pin((per_batch * 6 + (1 << 16), 2), torch.int32).view(np.uint32)}
    n_pool = int(os.environ.get("KU_PROBE_POOL", "0"))
    outs = [mk() for _ in range(max(4, n_pool))]
    slices = [host[(i * per_batch % (n_all - per_batch)) * stride:][:per_batch * stride] for i in range(n_batches)]
    if n_pool:  # every batch in a page-locked allocation of its own (sequences, offsets, lengths), recycled like the executable's
        pool = []
        for q in range(n_pool):
            sq = pin(per_batch * stride, torch.uint8)
            sq[:] = slices[q][:per_batch * stride]
            o2 = pin(per_batch, torch.int64).view(np.uint64); o2[:] = off
            l2 = pin(per_batch, torch.int32).view(np.uint32); l2[:] = lens
            pool.append((sq, o2, l2))
    for depth in ((3,) if os.environ.get("KU_PROBE_STAGGER") else (1, 2, 3, 4)):
        best = None
        for rep in range(3):
            ctx.reset_counts()
            flying = []
            t_enq = t_fin = 0.0
            t0 = time.perf_counter()
            for i in range(n_batches):
                if len(flying) >= depth:
                    a = time.perf_counter()
                    ctx.rle_finish(flying.pop(0))
                    t_fin += time.perf_counter() - a
                a = time.perf_counter()
                if n_pool:
                    sq, o2, l2 = pool[i % n_pool]
                    flying.append(ctx.rle_enqueue(sq, o2, l2, out=outs[i % n_pool]))
                else:
                    flying.append(ctx.rle_enqueue(slices[i], off, lens, out=outs[i % 4]))
                t_enq += time.perf_counter() - a
            while flying:
                a = time.perf_counter()
                ctx.rle_finish(flying.pop(0))
                t_fin += time.perf_counter() - a
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, t_enq, t_fin)
        dt, t_enq, t_fin = best
        print(f"   depth {depth}: {dt * 1e3 / n_batches:7.3f} ms per batch ({per_batch * n_batches / dt / 1e6:6.1f} M reads/s); "
              f"enqueue {t_enq * 1e3 / n_batches:.3f} ms, finish {t_fin * 1e3 / n_batches:.3f} ms per batch", flush=True)
    ctx.close() if hasattr(ctx, "close") else None
    del ctx  # (KU_RLE_TIMES: the library's own timers are printed when the context goes)


def main():
    per_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 61_000
    n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 164
    if os.environ.get("KU_DEPTH_PROBE_CHILD"):
        return child(per_batch, n_batches)
    import shutil, torch
    from krakenuniq_amd import synth_torch
    shutil.rmtree(TMP, ignore_errors=True)
    os.makedirs(TMP)
    dev = torch.device("cuda:0")
    db = synth_torch.BenchDb(dev, n_species=2000, genome_len=310_000, k=31, nt=13, seed=7)
    db.kmers = db.vals = None
    db.write_files(TMP)
    s, _, _, _ = db.sample_reads(2_000_000, 150, seed=1)
    s.reshape(-1).cpu().numpy().tofile(f"{TMP}/reads.bin")
    del db, s
    torch.cuda.empty_cache()
    variants = [("default", {})]
    if os.environ.get("KU_PROBE_STAGGER"):  # the -DKS_STAGGER variant of the library: waves that share a SIMD start apart
        lib = os.path.join(ROOT, "krakenuniq_amd", "variants", "libku_stg.so")
        variants = [("default", {}), ("variant build, no stagger", {"KU_LIB": lib, "KU_ABLATE": "0"})]
        for steps in os.environ["KU_PROBE_STAGGER"].split(","):
            variants.append((f"waves of a SIMD {steps} x 64 cycles apart", {"KU_LIB": lib, "KU_ABLATE": str(int(steps) << 16)}))
    for name, env in variants:
        print(f"== {name}", flush=True)
        r = subprocess.run([sys.executable, __file__, str(per_batch), str(n_batches)], env=dict(os.environ, KU_DEPTH_PROBE_CHILD="1", KU_RLE_TIMES="1", **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(r.stdout if r.returncode == 0 else f"   rc {r.returncode}: {r.stderr[-400:]}", flush=True)
        print("   " + " | ".join(l for l in r.stderr.split("\n") if "ku_classify_batch_rle over" in l)[:600], flush=True)
    shutil.rmtree(TMP, ignore_errors=True)


if __name__ == "__main__":
    main()
