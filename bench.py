#!/usr/bin/env python3
"""bench.py -- Mreads/s of the classify hot path on synthetic 150 bp reads (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload at N = 1 (BASELINE.json configs[1]): ~8 GB MiniKraken-style database (k = 31,
minimizer nt = 13, ~0.62 G pairs, ~2000 taxa) built directly in HBM by
krakenuniq_amd/synth_torch.py, 10 M synthetic 150 bp reads resident in HBM.  One "step"
= one pass of the whole hot path over the 10 M-read batch: ku_classify_batch_device, i.e. for reads of up to 222 bp
the fused wave-per-read kernel (scan, canonical k-mer, minimizer, bucket probe, HLL + n_kmers, hit counts,
resolve_tree / LCA, n_reads, per-k-mer taxids); for longer reads, --paired and --mode sharded the two stages
    ku_lookup_device  (scan, canonical k-mer, minimizer, probe / in-bin search, HLL + n_kmers)
    ku_resolve_device (hit counts, resolve_tree / LCA, n_reads, slot -> taxid)
N > 1 (default --mode replicas, weak scaling): every rank holds the database and classifies
its own 10 M reads; the per-taxon state is merged with RCCL inside every step (registers MAX,
counters SUM).  --mode sharded keeps 1/N of the minimizer bins per rank, scans the same batch
on every rank, merges per-k-mer slots with all_reduce(MAX) and resolves 1/N of the reads per
rank (the 300 GB layout of configs[2]).

Prints ONE JSON line (rank 0) with the driver's fields plus `roofline` (the dominant kernel: the fused kernel, or the
lookup kernel of the two-stage path; algorithmic bytes / HIP-event time vs 8 TB/s) and `cpu_baseline` (the compiled reference's
`classify` -- or the C oracle if the binary is absent -- on the host cores, bounded sample).
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s per GPU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--species", type=int, default=2000)
    ap.add_argument("--genome-len", type=int, default=310_000)
    ap.add_argument("--nt", type=int, default=13)
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas")
    ap.add_argument("--paired", action="store_true", help="configs[3]-style reads: mate1 + 'N' + mate2 (2 x read-len + 1)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1 auto, 0 skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def host_cores():
    """usable host cores: min(affinity mask, cgroup cpu quota)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(db, ctx, d_seqs, read_len, n_sample, calls_gpu, taxa_gpu, threads):
    """Time the reference's classify (oracle/_ref, travels as a binary) on the host cores over a
    bounded sample of the same reads against the same database, and check the GPU results
    against its output on that sample.  Falls back to the C oracle ("port") if the binary is absent."""
    from krakenuniq_amd import capi
    cores = threads
    stride = read_len + 1
    host = d_seqs[:n_sample * stride].cpu().numpy()
    ids = [f"r{i}" for i in range(n_sample)]
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 14e9 else None
    tmp = tempfile.mkdtemp(prefix="ku_bench_", dir=tmp_root)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "classify")
    out = {"cores": cores, "unit": "Mreads/s"}
    try:
        off = np.arange(n_sample, dtype=np.uint64) * stride
        lens = np.full(n_sample, read_len, dtype=np.uint32)
        want_text = capi.format_kraken(host, off, lens, ids, db.k, calls_gpu[:n_sample], taxa=taxa_gpu[:n_sample * stride])
        if os.path.exists(ref_bin):
            # the resident pairs hold slot ids after ku_ctx_set_taxonomy: translate back to taxids for the files
            db.write_files(tmp, slot_taxid=torch.from_numpy(ctx.counts()["slot_taxid"].astype(np.int64)).to(db.device))
            with open(os.path.join(tmp, "sample.fa"), "wb") as f:
                rows = host.reshape(n_sample, stride)
                for i in range(n_sample):
                    f.write(b">r%d\n" % i)
                    f.write(rows[i].tobytes())
            cmd = [ref_bin, "-d", f"{tmp}/database.kdb", "-i", f"{tmp}/database.idx", "-a", f"{tmp}/taxDB",
                   "-t", str(cores), "-M", "-o", f"{tmp}/out.tsv", f"{tmp}/sample.fa"]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            err = r.stderr.decode(errors="replace")
            m = re.search(r"(\d+) sequences \(([\d.]+) Mbp\) processed in ([\d.]+)s", err)
            if r.returncode != 0 or not m:
                raise RuntimeError("reference classify failed: " + err[-400:])
            secs = float(m.group(3))  # the reference's own timing window (classify.cpp:248-258)
            out.update(kind="reference", value=n_sample / secs / 1e6,
                       sample=f"{n_sample} of the batch's reads, oracle/_ref/classify -t {cores} -M, "
                              f"its report_stats window {secs:.3f}s")
            got = sorted(open(f"{tmp}/out.tsv").read().split("\n"))
            out["parity_vs_reference_on_sample"] = got == sorted(want_text.split("\n"))
        else:
            from oracle import ku_oracle as ko
            st = torch.from_numpy(ctx.counts()["slot_taxid"].astype(np.int64)).to(db.device)
            raw = db.pairs.clone()
            raw[:, 2] = ((st[raw[:, 2].to(torch.int64)] << 32) >> 32).to(torch.int32)
            pairs = raw.cpu().numpy().view(np.uint8).reshape(-1)
            offs = db.offsets.cpu().numpy().astype(np.uint64)
            ids_t, par_t = db.tax.arrays()
            odb = ko.Db(pairs=pairs, key_ct=db.n_pairs, k=db.k, offsets=offs, nt=db.nt)
            run = ko.Run(odb, ko.Tax(ids=ids_t, parents=par_t), threads=cores)
            t0 = time.time()
            res = run.classify_packed(host, off, lens, want_taxa=False)
            secs = time.time() - t0
            out.update(kind="port", value=n_sample / secs / 1e6,
                       sample=f"{n_sample} of the batch's reads, oracle/libku_oracle.so OpenMP x{cores}, {secs:.3f}s")
            out["parity_vs_reference_on_sample"] = bool((res["calls"] == calls_gpu[:n_sample]).all())
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws != a.gpus and ws > 1:
        a.gpus = ws
    # debugging aid for 1-GPU boxes: KU_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and KU_BENCH_BACKEND=gloo
    # replaces RCCL, so the N > 1 code path can be exercised without N GPUs (numbers are then meaningless)
    if os.environ.get("KU_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("KU_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from krakenuniq_amd import capi, dist as kdist, synth_torch

    k = 31
    sharded = a.mode == "sharded" and ws > 1
    t_build = time.time()
    bin_lo, bin_hi = 0, 4 ** a.nt
    if sharded:
        # every rank derives the same shard plan from the same deterministic sample of the DB's bin keys
        probe = synth_torch.BenchDb(dev, n_species=min(a.species, 32), genome_len=min(a.genome_len, 50_000), k=k,
                                    nt=a.nt, seed=7)
        bins = synth_torch.bin_key(probe.kmers[torch.randperm(probe.n_pairs, device=dev)[:1_000_000]], k, a.nt)
        bounds = kdist.quantile_bin_bounds(bins, 4 ** a.nt, ws)
        bin_lo, bin_hi = int(bounds[rank]), int(bounds[rank + 1])
        del probe, bins
    db = synth_torch.BenchDb(dev, n_species=a.species, genome_len=a.genome_len, k=k, nt=a.nt, seed=7,
                             bin_lo=bin_lo, bin_hi=bin_hi)
    db.kmers = db.vals = None
    torch.cuda.empty_cache()
    # offsets must be *global* pair indices; in a sharded build they start at 0 for this shard, which is fine:
    # the library only uses differences and offsets[0] as the base.
    ctx = capi.Ctx(local_rank)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), k, a.nt, 2, bin_lo, bin_hi, keep=db)
    ids_t, par_t = db.tax.arrays()
    ctax = capi.Tax(ids=ids_t, parents=par_t)
    all_values = kdist.allgather_values(ctx.db_values(), dev) if sharded else None
    ctx.set_taxonomy(ctax, all_values)
    # reads: own batch per rank (replicas) or the same batch on every rank (sharded)
    d_seqs, d_off, d_len, _ = db.sample_reads(a.reads, a.read_len, seed=1 if sharded else 1 + rank)
    if a.paired:  # read_merger.pl semantics: seq1 . "N" . seq2 (scripts/read_merger.pl:187-191)
        m2, _, _, _ = db.sample_reads(a.reads, a.read_len, seed=1001 if sharded else 1001 + rank)
        L = a.read_len
        merged = torch.empty((a.reads, 2 * L + 2), dtype=torch.uint8, device=dev)
        merged[:, :L] = d_seqs.view(a.reads, L + 1)[:, :L]
        merged[:, L] = 78
        merged[:, L + 1:2 * L + 1] = m2.view(a.reads, L + 1)[:, :L]
        merged[:, 2 * L + 1] = 10
        d_seqs = merged.reshape(-1)
        d_off = torch.arange(a.reads, device=dev, dtype=torch.int64) * (2 * L + 2)
        d_len = torch.full((a.reads,), 2 * L + 1, dtype=torch.int32, device=dev)
        a.read_len = 2 * L + 1
        del m2, merged
    n_bytes = d_seqs.numel()
    d_taxa = torch.zeros(n_bytes, dtype=torch.int32, device=dev)
    d_calls = torch.zeros(a.reads, dtype=torch.int32, device=dev)
    ptrs = ctx.counts_device_ptrs()
    build_s = time.time() - t_build

    # a dedicated (non-null) torch stream: kernels, RCCL collectives and the timing events all live on it
    torch.cuda.synchronize()  # inputs were produced on the default stream: finish them before switching streams
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    r_lo, r_hi = kdist.read_slice(a.reads, rank, ws) if sharded else (0, a.reads)

    # torch views (no copy) of the context's live per-taxon state, for the RCCL merge
    def dev_tensor(ptr, nbytes, dtype):
        # build a tensor over existing device memory via __cuda_array_interface__
        class _W:
            pass
        w = _W()
        itemsize = torch.tensor([], dtype=dtype).element_size()
        w.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": {1: "|u1", 8: "<i8"}[itemsize],
                                      "data": (ptr, False), "version": 2}
        return torch.as_tensor(w, device=dev)

    st_regs = dev_tensor(ptrs["registers"], ptrs["register_bytes"], torch.uint8)
    st_kmers = dev_tensor(ptrs["n_kmers"], ptrs["n_slots"] * 8, torch.int64)
    st_reads = dev_tensor(ptrs["n_reads"], ptrs["n_nodes"] * 8, torch.int64)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    fused = not sharded and os.environ.get("KU_NO_FUSED") is None and (a.read_len - k + 1) <= 192 and ctx.db_layout()["hash"]

    def step(i=None):
        if fused:
            # short reads, whole DB resident: ONE fused kernel (wave per read) does lookup + counts + resolve
            if i is not None:
                ev[i][0].record()
            ctx.classify_batch_device(d_seqs.data_ptr(), n_bytes, d_off.data_ptr(), d_len.data_ptr(), a.reads,
                                      d_calls.data_ptr(), d_taxa.data_ptr(), max_read_len=a.read_len, stream=stream)
            if i is not None:
                ev[i][1].record()
        else:
            if i is not None:
                ev[i][0].record()
            ctx.lookup_device(d_seqs.data_ptr(), n_bytes, d_taxa.data_ptr(),
                              flags=capi.KU_F_KEEP_SLOTS if sharded else 0, stream=stream)
            if i is not None:
                ev[i][1].record()
            if sharded:
                kdist.merge_taxa_max(d_taxa)
            ctx.resolve_device(d_seqs.data_ptr(), d_off[r_lo:].data_ptr(), d_len[r_lo:].data_ptr(), r_hi - r_lo,
                               d_calls[r_lo:].data_ptr(), d_taxa.data_ptr(), max_read_len=a.read_len, stream=stream)
        if ws > 1:
            kdist.reduce_state(st_regs, st_kmers, st_reads)

    for _ in range(a.warmup):
        step()
    ctx.reset_counts()
    torch.cuda.synchronize()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if ws > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lookup_ms = float(np.mean([s.elapsed_time(e) for s, e in ev])) if a.steps else 0.0

    total_reads = a.reads * a.steps * (1 if sharded else ws)
    value = total_reads / elapsed / 1e6
    # roofline of the dominant kernel (lookup): algorithmic bytes per launch / HIP-event time
    stats = ctx.lookup_stats_device(d_seqs.data_ptr(), n_bytes)
    bytes_algo = a.reads * (a.read_len + 4) + stats["lookups"] * 20 + 12 * stats["sum_ceil_log2"]
    achieved = bytes_algo / (lookup_ms * 1e-3) / 1e9 if lookup_ms else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "lookup_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if (tj.get("reads") == a.reads and tj.get("nt") == a.nt and tj.get("species") == a.species
                    and tj.get("kernel", "").startswith("ku_classify_short" if fused else "ku_lookup")):
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    default_cfg = (a.reads == 10_000_000 and a.species == 2000 and a.genome_len == 310_000 and a.nt == 13
                   and not a.paired and a.read_len == 150)
    if default_cfg and ws == 1:
        workload = "configs[1]: 8 GB MiniKraken-style DB (k=31), 10M synthetic 150 bp reads, 1xMI355X"
    else:
        workload = (f"{'configs[1] ' if default_cfg else ''}synthetic DB ({a.species} taxa x {a.genome_len} bp, nt={a.nt}), "
                    f"{a.reads} {'paired 2x' + str((a.read_len - 1) // 2) if a.paired else str(a.read_len)} bp reads per GPU "
                    f"per step, {a.mode} x{ws}")
    result = {
        "metric": "Mreads/s (150 bp)", "value": round(value, 3), "unit": "Mreads/s", "n_gpus": ws, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload,
                   "db_pairs_per_gpu": db.n_pairs, "db_bytes_per_gpu": db.n_pairs * 12 + db.offsets.numel() * 8,
                   "hbm_layout": ctx.db_layout(), "k": k, "nt": a.nt, "taxa": a.species, "reads_per_gpu_per_step": a.reads, "read_len": a.read_len,
                   "parallelism": f"{a.mode}{ws}", "db_build_s": round(build_s, 1)},
        "roofline": {"bound": "hbm", "kernel": "ku_classify_short_kernel (fused lookup+resolve)" if fused else "ku_lookup_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": bytes_algo, "lookups_per_launch": stats["lookups"],
                     "mean_ceil_log2_bin": round(stats["sum_ceil_log2"] / max(stats["lookups"], 1), 3),
                     "kernel_ms": round(lookup_ms, 3)},
    }
    if rank == 0 and ws == 1 and a.cpu_sample != 0:
        cores = a.cpu_threads or host_cores()
        n_sample = a.cpu_sample if a.cpu_sample > 0 else min(a.reads, 60_000 * cores, 4_000_000)  # ~20 s of CPU work
        calls = d_calls[:n_sample].cpu().numpy().view(np.uint32)
        taxa = d_taxa[:n_sample * (a.read_len + 1)].cpu().numpy().view(np.uint32)
        try:
            result["cpu_baseline"] = cpu_baseline(db, ctx, d_seqs, a.read_len, n_sample, calls, taxa, cores)
        except Exception as e:  # the baseline must never take the bench line down
            result["cpu_baseline"] = {"value": None, "unit": "Mreads/s", "cores": cores, "kind": "reference",
                                      "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if ws > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
