#!/usr/bin/env python3
"""bench.py -- Mreads/s of the classify hot path on synthetic 150 bp reads (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1: one process per GPU under torch.distributed.run --
                                                            started that way by the driver, or by bench.py itself when it
                                                            is called plainly; fewer than N visible GPUs is an error)
    python bench.py --gpus N --config 2|3|4                (the 8-GPU configurations of BASELINE.json, see PRESETS)

Workload at N = 1 (BASELINE.json configs[1]): ~8 GB MiniKraken-style database (k = 31, minimizer nt = 13, ~0.61 G
pairs, 2000 species) built directly in HBM by krakenuniq_amd/synth_torch.py; FOUR distinct batches of 10 M synthetic
150 bp reads resident in HBM.  One "step" = one pass of the whole hot path over one 10 M-read batch
(ku_classify_batch_device): the fused wave-per-read kernel (scan, canonical k-mer, anchor / minimizer, bucket probe,
HLL + n_kmers, hit counts, resolve_tree / LCA, n_reads, per-k-mer taxids -- `--output runs`: as the Kraken line prints
them, runs of equal codes written by the kernel itself, ku_classify_batch_device_rle; the line times both forms and checks
the expanded runs against the per-position array) -- in one pass for reads of up to 222 bp, in
windows of 128 k-mers for longer ones (--paired, --read-len 10000); with KU_NO_FUSED=1 / KU_NO_WINDOWED=1 or reads beyond
65535 k-mers the two stages ku_lookup_device + ku_resolve_device.  The steps rotate through the batches and the
per-taxon state is zeroed at the start of every rotation (inside the timed region), so each rotation is a fresh 40 M-read
run: HyperLogLog registers start empty and their compare-and-swap updates are part of what is timed.

N > 1 goes through the product's C++ multi-GPU driver (ku_mgpu, RCCL), one process per GPU:
  --mode replicas (default, weak scaling): every rank holds the database and classifies its own batches; the per-taxon
      state is all-reduced once at the end of the run, inside the timed region (registers MAX, counters SUM);
  --mode sharded (strong scaling, the 300 GB layout of configs[2]): rank r holds the minimizer bins of its range; each
      step broadcasts the batch from rank 0, looks up the owned k-mers, reduce-scatters the per-k-mer slots over the read
      dimension and resolves 1/N of the reads per rank (ku_mgpu_step_device).
  A default N > 1 run also times a few sharded steps afterwards and reports them under "sharded".

--config 2 / 3 / 4 (BASELINE.json configs[2..4]): the standard-geometry database (k = 31, minimizer nt = 15) in EIGHT
minimizer-range shards of ~37 GB of pairs each (~300 GB in all; rank r holds shard r -- with fewer than 8 ranks the other
shards' k-mers simply stay unowned, so --gpus 1 --config 2 measures one rank's share of the 8-GPU layout), sharded mode,
10 M x 150 bp reads / 5 M mate pairs 2 x 150 + N / 100 k x 10 kbp reads per step.  Every sharded line carries the rank's
own `roofline` (the sharded lookup kernel, timed alone) and the bytes per read that cross the links.

Prints ONE JSON line (rank 0) with the driver's fields plus `roofline` (dominant kernel; `frac` = `frac_model` =
SURVEY 8(d) algorithmic bytes / HIP-event time / 8 TB/s, `frac_hw` = counter-measured HBM bytes of profiles/ for this
very kernel source / time / 8 TB/s), `cpu_baseline` (the compiled reference's classify on the host cores over a bounded
sample, with a parity check of that sample) and, at N = 1, `device_pipeline` (pinned host buffers -> H2D -> kernels ->
run-length encoding -> D2H through ku_classify_batch_rle) and `e2e` (the classify executable on a FASTQ file in
/dev/shm against the same database, its own timing window).
"""
import argparse
import threading
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

# BASELINE.json configs[2..4]: the ~300 GB standard database as 8 minimizer-range shards (12 000 species x 310 kbp give
# ~3.1 G pairs = 37 GB per shard), sharded mode; reads per step chosen so that the resident batch stays around 1.5 GB
PRESETS = {
    2: {"name": "configs[2]: standard ~300 GB DB (k=31, nt=15) sharded by minimizer bin, 150 bp reads (100 M = 10 steps of 10 M)",
        "nt": 15, "species": 96_000, "db_shards": 8, "reads": 10_000_000, "read_len": 150, "paired": False},
    3: {"name": "configs[3]: standard ~300 GB DB (k=31, nt=15) sharded by minimizer bin, mate pairs 2 x 150 + N (50 M = 10 steps of 5 M)",
        "nt": 15, "species": 96_000, "db_shards": 8, "reads": 5_000_000, "read_len": 150, "paired": True},
    4: {"name": "configs[4]: standard ~300 GB DB (k=31, nt=15) sharded by minimizer bin, 10 kbp reads (1 M = 10 steps of 100 k)",
        "nt": 15, "species": 96_000, "db_shards": 8, "reads": 100_000, "read_len": 10_000, "paired": False},
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--batches", type=int, default=4, help="distinct read batches the steps rotate through")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--species", type=int, default=2000)
    ap.add_argument("--genome-len", type=int, default=310_000)
    ap.add_argument("--nt", type=int, default=13)
    ap.add_argument("--mode", choices=["replicas", "sharded"], default=None,
                    help="N > 1 without --mode / --config: both, the sharded layout (configs[2] scaled to N GPUs) as the headline")
    ap.add_argument("--paired", action="store_true", help="configs[3]-style reads: mate1 + 'N' + mate2 (2 x read-len + 1)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1 auto, 0 skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-extras", action="store_true", help="skip the device-pipeline / end-to-end / sharded legs")
    ap.add_argument("--output", choices=["runs", "taxa"], default="taxa",
                    help="per-k-mer codes of a step: one 32-bit code per base position (ku_classify_batch_device; default) | run-length "
                         "encoded by the fused kernel itself, as the Kraken line prints them and as the classify executable takes them "
                         "(ku_classify_batch_device_rle: 0.2 GB instead of 6 GB of output per 10 M reads; the kernel is bound by "
                         "instruction issue, so the shorter output does not make it faster)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[i]; 2-4 = the 300 GB standard-geometry database in 8 shards, sharded mode")
    ap.add_argument("--db-shards", type=int, default=0,
                    help="sharded mode: minimizer-range shards the database is cut into (rank r holds shard r; default: one per rank)")
    a = ap.parse_args()
    a.mode_given = a.mode is not None
    a.mode = a.mode or "replicas"
    a.preset = None
    if a.config != 1:
        pr = PRESETS[a.config]
        a.preset = pr["name"]
        a.mode = "sharded"
        a.nt, a.species, a.reads, a.read_len, a.paired = pr["nt"], pr["species"], pr["reads"], pr["read_len"], pr["paired"]
        a.db_shards = a.db_shards or pr["db_shards"]
        a.batches = min(a.batches, 2)
    return a


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: N ranks on the first N devices under torch.distributed.run.  Never a
    line that says n_gpus = N from fewer devices."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus and os.environ.get("KU_BENCH_ONE_DEVICE") != "1":
        sys.stderr.write(f"bench.py: --gpus {a.gpus} was asked for but {n_dev} GPU(s) are visible; refusing to run "
                         f"(a {a.gpus}-GPU line cannot be measured on {n_dev})\n")
        sys.exit(2)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


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


def make_batch(db, a, seed, dev):
    """one batch of reads in HBM: (seqs uint8 flat, seq_off int64, seq_len int32, read_len)"""
    if a.paired:  # both mates from one fragment, merged as read_merger.pl does: seq1 . "N" . seq2
        d_seqs, d_off, d_len = db.sample_pairs(a.reads, a.read_len, seed=seed)
        return d_seqs, d_off, d_len, 2 * a.read_len + 1
    d_seqs, d_off, d_len, _ = db.sample_reads(a.reads, a.read_len, seed=seed)
    L = a.read_len
    return d_seqs.reshape(-1), d_off, d_len, L


def write_fastq(path, host_rows, read_len):
    """FASTQ text of the rows of a [n, read_len + 1] uint8 array (ids r000000000 ...), vectorised"""
    n = host_rows.shape[0]
    rec = np.empty((n, 1 + 10 + 1 + read_len + 3 + read_len + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(9):
        rec[:, 2 + d] = 48 + (idx // 10 ** (8 - d)) % 10
    rec[:, 11] = 10
    rec[:, 12:12 + read_len] = host_rows[:, :read_len]
    rec[:, 12 + read_len:15 + read_len] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 15 + read_len:15 + 2 * read_len] = ord("I")
    rec[:, 15 + 2 * read_len] = 10
    with open(path, "wb") as f:
        f.write(rec.tobytes())


def cgroup_quota():
    """the container's CPU quota as text (cgroup v2 cpu.max / v1 cfs quota + period)"""
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            return "unlimited" if q == "max" else f"{int(q) / int(p):g} CPUs"
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return "unlimited" if q < 0 else f"{q / p:g} CPUs"
    except Exception:
        return "unknown"


def host_legs(a, db, ctx, batch, read_len, calls_gpu, taxa_gpu, k, algo_bytes0=None):
    """rank 0, N = 1: CPU baseline (+ parity of the sample), device pipeline, end-to-end executable"""
    from krakenuniq_amd import capi
    out = {}
    cores = a.cpu_threads or host_cores()
    d_seqs, d_off, d_len = batch
    stride = read_len + 1
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 24e9 else None
    tmp = tempfile.mkdtemp(prefix="ku_bench_", dir=tmp_root)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "classify")
    cli_bin = os.path.join(ROOT, "krakenuniq_amd", "bin", "classify")
    try:
        files = False
        if os.path.exists(ref_bin) or os.path.exists(cli_bin):
            # the resident pairs hold slot ids after ku_ctx_set_taxonomy: translate back to taxids for the files
            db.write_files(tmp, slot_taxid=torch.from_numpy(ctx.counts()["slot_taxid"].astype(np.int64)).to(db.device))
            files = True
        # ---- CPU baseline: the reference's classify on a bounded sample of batch 0 + parity of that sample
        if a.cpu_sample != 0:
            n_sample = a.cpu_sample if a.cpu_sample > 0 else min(a.reads, 60_000 * cores, 4_000_000)  # ~20 s of CPU work
            cb = {"cores": cores, "unit": "Mreads/s"}
            try:
                host = d_seqs[:n_sample * stride].cpu().numpy()
                off = np.arange(n_sample, dtype=np.uint64) * stride
                lens = np.full(n_sample, read_len, dtype=np.uint32)
                ids = [f"r{i}" for i in range(n_sample)]
                want_text = capi.format_kraken(host, off, lens, ids, k, calls_gpu[:n_sample], taxa=taxa_gpu[:n_sample * stride])
                if files and os.path.exists(ref_bin):
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
                    cb.update(kind="reference", value=n_sample / secs / 1e6,
                              sample=f"{n_sample} reads of batch 0, oracle/_ref/classify -t {cores} -M, "
                                     f"its report_stats window {secs:.3f}s")
                    got = sorted(open(f"{tmp}/out.tsv").read().split("\n"))
                    cb["parity_vs_reference_on_sample"] = got == sorted(want_text.split("\n"))
                    os.remove(f"{tmp}/out.tsv")
                    # the same binary on ONE thread (-t 1 beside -t nproc), on the first 1 / cores of the sample: tells
                    # the machine's per-core rate on this database from what the thread team makes of it
                    if cores > 1:
                        try:
                            n1 = max(10_000, n_sample // cores)
                            with open(os.path.join(tmp, "sample1.fa"), "wb") as f:
                                rows = host.reshape(n_sample, stride)
                                for i in range(min(n1, n_sample)):
                                    f.write(b">r%d\n" % i)
                                    f.write(rows[i].tobytes())
                            cmd1 = [ref_bin, "-d", f"{tmp}/database.kdb", "-i", f"{tmp}/database.idx", "-a", f"{tmp}/taxDB",
                                    "-t", "1", "-M", "-o", f"{tmp}/out1.tsv", f"{tmp}/sample1.fa"]
                            r1 = subprocess.run(cmd1, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                            m1 = re.search(r"(\d+) sequences \(([\d.]+) Mbp\) processed in ([\d.]+)s", r1.stderr.decode(errors="replace"))
                            if r1.returncode == 0 and m1:
                                s1 = float(m1.group(3))
                                cb["one_thread"] = {"value": round(min(n1, n_sample) / s1 / 1e6, 5), "unit": "Mreads/s", "cores": 1,
                                                    "sample": f"the first {min(n1, n_sample)} reads of that sample, -t 1, its report_stats window {s1:.3f}s",
                                                    "team_speedup": round((n_sample / secs) / (min(n1, n_sample) / s1), 2),
                                                    "team_speedup_note": f"throttled: the box shows {os.cpu_count()} CPUs under a CFS quota of {cgroup_quota()} "
                                                                         f"(so {cores} threads were used) and every thread misses the cache on the same "
                                                                         "8 GB database -- the team's rate is no per-core rate times cores; do not scale it"}
                            for fn in ("sample1.fa", "out1.tsv"):
                                if os.path.exists(f"{tmp}/{fn}"):
                                    os.remove(f"{tmp}/{fn}")
                        except Exception as e:
                            cb["one_thread"] = {"value": None, "error": str(e)[:200]}
                    os.remove(f"{tmp}/sample.fa")
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
                    cb.update(kind="port", value=n_sample / secs / 1e6,
                              sample=f"{n_sample} reads of batch 0, oracle/libku_oracle.so OpenMP x{cores}, {secs:.3f}s")
                    cb["parity_vs_reference_on_sample"] = bool((res["calls"] == calls_gpu[:n_sample]).all())
            except Exception as e:  # the baseline must never take the bench line down
                cb.update(value=None, kind="reference", sample=f"failed: {e}")
            out["cpu_baseline"] = cb
        if a.no_extras:
            return out
        # ---- device pipeline: pinned host buffers -> H2D -> kernels -> RLE -> D2H (ku_classify_batch_rle + ku_fetch_runs)
        try:
            n_dp = min(a.reads, 2_000_000)
            hb = d_seqs[:n_dp * stride].cpu().pin_memory().numpy()
            off = np.arange(n_dp, dtype=np.uint64) * stride
            lens = np.full(n_dp, read_len, dtype=np.uint32)
            r = ctx.classify_batch_rle(hb, off, lens)
            pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()
            n_runs_first = len(r["runs"])
            mk_obuf = lambda: {"calls": pin(n_dp, torch.int32).view(np.uint32), "hits": pin(n_dp, torch.int32).view(np.uint32),
                               "run_cnt": pin(n_dp, torch.int32).view(np.uint32), "run_off": pin(n_dp, torch.int64).view(np.uint64),
                               # (the extent of the run array varies a little from call to call: the waves claim it in chunks)
                               "runs": pin((n_runs_first + n_runs_first // 4 + (1 << 16), 2), torch.int32).view(np.uint32)}
            obuf = mk_obuf()
            off, lens = pin(n_dp, torch.int64).view(np.uint64), pin(n_dp, torch.int32).view(np.uint32)
            off[:] = np.arange(n_dp, dtype=np.uint64) * stride
            lens[:] = read_len
            dts = []
            for _ in range(3):
                t0 = time.perf_counter()
                r = ctx.classify_batch_rle(hb, off, lens, out=obuf)
                dts.append(time.perf_counter() - t0)
            # the same boundary in its two-step form (ku_classify_batch_rle_enqueue / _finish): two batches in flight, the upload
            # of the next and the copies back of the previous under the kernel of the current one; six batches per round
            obuf2 = mk_obuf()
            rounds = []
            n_b = 6
            for _ in range(5):
                flying, bufs = [], [obuf, obuf2]
                t0 = time.perf_counter()
                for i in range(n_b):
                    if len(flying) == 2:
                        r2 = ctx.rle_finish(flying.pop(0))
                    flying.append(ctx.rle_enqueue(hb, off, lens, out=bufs[i & 1]))
                while flying:
                    r2 = ctx.rle_finish(flying.pop(0))
                rounds.append((time.perf_counter() - t0) / n_b)
            rounds.sort()
            dt = rounds[len(rounds) // 2]
            out["device_pipeline"] = {"value": round(n_dp / dt / 1e6, 2), "unit": "Mreads/s", "reads": n_dp,
                                      "what": "median over 5 rounds of 6 batches, two in flight (ku_classify_batch_rle_enqueue / _finish)",
                                      "ms_per_batch_median_min_max": [round(dt * 1e3, 2), round(rounds[0] * 1e3, 2), round(rounds[-1] * 1e3, 2)],
                                      "one_step_ms_of_three_calls": [round(x * 1e3, 2) for x in dts],
                                      "one_step_value_median": round(n_dp / sorted(dts)[1] / 1e6, 2),
                                      "runs_per_read": round(float(r["run_cnt"].sum()) / n_dp, 2),
                                      "calls_match_device_run": bool((r["calls"] == calls_gpu[:n_dp]).all() and (r2["calls"] == calls_gpu[:n_dp]).all()),
                                      "path": "pinned host buffers -> H2D on a copy stream || fused kernel with run-length encoded output "
                                              "-> D2H (calls, runs) on a stream of its own -> pinned host buffers"}
        except Exception as e:
            out["device_pipeline"] = {"value": None, "error": str(e)[:200]}
        # ---- end to end: the classify executable on a FASTQ file (parser team | device | formatter + writer)
        try:
            if not (files and os.path.exists(cli_bin)) or a.paired:
                raise RuntimeError("classify executable or database files missing")
            n_e = min(a.reads, len(calls_gpu))
            write_fastq(f"{tmp}/reads.fq", d_seqs[:n_e * stride].cpu().numpy().reshape(n_e, stride), read_len)
            thr = str(min(cores, 16))
            cmd = [cli_bin, "-d", f"{tmp}/database.kdb", "-i", f"{tmp}/database.idx", "-a", f"{tmp}/taxDB", "-t", thr,
                   "-o", f"{tmp}/e2e.tsv", f"{tmp}/reads.fq"]
            runs_s, errs = [], []
            N_E2E = 5  # five runs, every one of them listed, the median counts -- the first (cold) one included (the host side -- 16 CPUs of
                       # quota on a 256-CPU box -- varies from run to run; round 5 took the median of three and the cold run never counted)

            def cpu_in_window(err_text):
                """the executable's own account of the window (KU_CLI_TIMES): CPU seconds by stage, thread counts"""
                mc = re.search(r"cpu seconds in the window: user ([\d.]+) \+ sys ([\d.]+) in all; parser team ([\d.]+), formatting helpers ([\d.]+), writer ([\d.]+), device thread ([\d.]+)", err_text)
                mt = re.search(r"threads in the window: parser team (\d+), formatting helpers (\d+), others (\d+)", err_text)
                if not mc:
                    return None
                d = {"user": float(mc.group(1)), "sys": float(mc.group(2)), "parser_team": float(mc.group(3)), "formatting_helpers": float(mc.group(4)),
                     "writer": float(mc.group(5)), "device_thread": float(mc.group(6))}
                if mt:
                    d["threads"] = {"parser_team": int(mt.group(1)), "formatting_helpers": int(mt.group(2)), "others": int(mt.group(3))}
                return d
            for rep in range(N_E2E):
                t0 = time.time()
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KU_CLI_TIMES="1"))
                wall = time.time() - t0
                err_i = r.stderr.decode(errors="replace")
                m = re.search(r"(\d+) sequences \(([\d.]+) Mbp\) processed in ([\d.]+)s", err_i)
                if r.returncode != 0 or not m:
                    raise RuntimeError("classify failed: " + err_i[-300:])
                runs_s.append(float(m.group(3)))
                errs.append(err_i)
            secs = sorted(runs_s)[N_E2E // 2]
            err = errs[runs_s.index(secs)]
            mb = re.search(r"stage busy seconds: reader ([\d.]+), device ([\d.]+), writer ([\d.]+)", err)
            busy_plain = float(mb.group(2)) if mb else None
            import pandas as pd
            got = pd.read_csv(f"{tmp}/e2e.tsv", sep="\t", header=None, usecols=[2], dtype=np.uint32)[2].to_numpy()
            out["e2e"] = {"value": round(n_e / secs / 1e6, 2), "unit": "Mreads/s", "reads": n_e, "threads": int(thr),
                          "window": "the executable's report_stats window (classify.cpp:248-258): FASTQ parse -> GPU -> Kraken file",
                          "seconds": secs, "seconds_of_the_runs": runs_s, "value_is": f"the median of {N_E2E} runs, the first (cold) one included",
                          "first_run_seconds": runs_s[0], "cpu_seconds_in_the_window_of_the_median_run": cpu_in_window(err),
                          "cpu_quota": cgroup_quota(), "wall_incl_db_load_s": round(wall, 1),
                          "calls_match_device_run": bool(len(got) == n_e and (got == calls_gpu[:n_e]).all())}
            # the same run as scripts/krakenuniq starts it: with a report (-r), i.e. with the HyperLogLog++ sparse-sketch
            # emulation inside the timing window and the clade roll-up behind it
            try:
                os.remove(f"{tmp}/e2e.tsv")
                env = dict(os.environ, KU_CLI_TIMES="1", KU_REPORT_TIMES="1", KU_RLE_TIMES="1")
                cmd_r = cmd[:-1] + ["-r", f"{tmp}/report.tsv", cmd[-1]]
                runs_r, errs_r = [], []
                for rep in range(N_E2E):
                    for fn in ("e2e.tsv", "report.tsv"):
                        if os.path.exists(f"{tmp}/{fn}"):
                            os.remove(f"{tmp}/{fn}")
                    t0 = time.time()
                    r = subprocess.run(cmd_r, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
                    wall = time.time() - t0
                    err_i = r.stderr.decode(errors="replace")
                    m = re.search(r"(\d+) sequences \(([\d.]+) Mbp\) processed in ([\d.]+)s", err_i)
                    if r.returncode != 0 or not m or not re.search(r"Report finished in ([\d.]+) seconds", err_i):
                        raise RuntimeError("classify -r failed: " + err_i[-300:])
                    runs_r.append(float(m.group(3)))
                    errs_r.append(err_i)
                secs_r = sorted(runs_r)[N_E2E // 2]
                err = errs_r[runs_r.index(secs_r)]
                m2 = re.search(r"Report finished in ([\d.]+) seconds", err)
                m3 = re.search(r"stage busy seconds: reader ([\d.]+), device ([\d.]+), writer ([\d.]+)", err)
                # the fused kernel's emulation instance (OUT = 2: SEEN marks, insert counts, misses into the set) on the stream it
                # runs on, summed over the run's batches by HIP events (KU_RLE_TIMES), against batch 0's algorithmic bytes
                mk = re.search(r"kernels ([\d.]+) ms for (\d+) reads", err)
                mks = re.search(r"their sum is ([\d.]+) ms", err)
                rf_rep = None
                if mk and algo_bytes0 and int(mk.group(2)) == n_e:
                    k_ms = float(mk.group(1))
                    ach = algo_bytes0 / (k_ms * 1e-3) / 1e9
                    rf_rep = {"bound": "hbm", "kernel": "ku_classify_short_kernel<..., OUT = 2> (fused lookup + resolve + runs + sparse-sketch "
                              "bookkeeping), as the executable launches it: one launch per batch of ~120 k reads, the kernels of consecutive "
                              "batches on two streams in turn (one launch's tail under the next one's start)",
                              "kernel_ms": round(k_ms, 3), "launch_ms_source": "HIP events around every batch's kernels on the stream they run on: the time "
                              "the batches' intervals COVER (they overlap; KU_RLE_TIMES; includes the per-batch flag kernel)",
                              "kernel_ms_sum_of_the_intervals": float(mks.group(1)) if mks else None,
                              "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(ach / HBM_PEAK_GBS, 5), "frac_model": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(algo_bytes0),
                              "traffic": None, "frac_hw": None}
                    try:  # counter-measured HBM bytes of this instance over a 10 M-read run of the executable (scripts/summarize_cli_pmc.py)
                        tj = json.load(open(os.path.join(ROOT, "profiles", "lookup_traffic.json")))
                        ent = tj.get("cli", {}).get("report") if tj.get("kernel_rev") == capi.kernel_rev() else None
                        if ent and ent.get("reads") == n_e:
                            rf_rep["traffic"] = ent["hbm_bytes_per_run"]
                            rf_rep["frac_hw"] = round(ent["hbm_bytes_per_run"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                            rf_rep["traffic_source"] = ent["source"]
                    except (OSError, ValueError, KeyError):
                        pass
                got = pd.read_csv(f"{tmp}/e2e.tsv", sep="\t", header=None, usecols=[2], dtype=np.uint32)[2].to_numpy()
                n_rows = sum(1 for _ in open(f"{tmp}/report.tsv"))
                out["e2e"]["with_report"] = {
                    "value": round(n_e / secs_r / 1e6, 2), "unit": "Mreads/s", "seconds": secs_r, "seconds_of_the_runs": runs_r,
                    "value_is": f"the median of {N_E2E} runs, the first one included", "cpu_seconds_in_the_window_of_the_median_run": cpu_in_window(err),
                    "roofline": rf_rep,
                    "report_seconds": float(m2.group(1)), "report_rows": n_rows,
                    "report_stages_ms": {mm.group(1).strip(): float(mm.group(2)) for mm in re.finditer(r"ku_ctx_report: (.+?) +([\d.]+) ms", err)},
                    "device_stage_busy_s": float(m3.group(2)) if m3 else None,
                    "device_stage_busy_s_without_report": busy_plain,
                    "sparse_emulation": "not switched off" if "ran out of device memory" not in err else "gave up",
                    "wall_incl_db_load_s": round(wall, 1),
                    "calls_match_device_run": bool(len(got) == n_e and (got == calls_gpu[:n_e]).all())}
            except Exception as e:
                out["e2e"]["with_report"] = {"value": None, "error": str(e)[:200]}
            # the same reads as ONE gzip stream (how reads mostly arrive): the executable inflates it with a team of threads
            # (ku_pgzip.h) and parses the text in regions while it arrives; once with zlib's one inflate beside it
            try:
                for fn in ("e2e.tsv", "report.tsv"):
                    if os.path.exists(f"{tmp}/{fn}"):
                        os.remove(f"{tmp}/{fn}")
                t0 = time.time()
                subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "write_one_stream_gz.py"), f"{tmp}/reads.fq", f"{tmp}/reads.fq.gz"],
                               check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
                t_gz = time.time() - t0
                os.remove(f"{tmp}/reads.fq")
                cmd_z = cmd[:-1] + [f"{tmp}/reads.fq.gz"]
                runs_z = []
                for rep in range(2):
                    r = subprocess.run(cmd_z, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KU_CLI_TIMES="1"), timeout=300)
                    err_i = r.stderr.decode(errors="replace")
                    m = re.search(r"(\d+) sequences \(([\d.]+) Mbp\) processed in ([\d.]+)s", err_i)
                    if r.returncode != 0 or not m:
                        raise RuntimeError("classify on the .gz failed: " + err_i[-300:])
                    runs_z.append(float(m.group(3)))
                got = pd.read_csv(f"{tmp}/e2e.tsv", sep="\t", header=None, usecols=[2], dtype=np.uint32)[2].to_numpy()
                gz = {"value": round(n_e / min(runs_z) / 1e6, 2), "unit": "Mreads/s", "seconds": min(runs_z), "seconds_of_both_runs": runs_z,
                      "file": "one deflate stream in one gzip member (level 6)", "gz_bytes": os.path.getsize(f"{tmp}/reads.fq.gz"),
                      "written_in_s": round(t_gz, 1), "calls_match_device_run": bool(len(got) == n_e and (got == calls_gpu[:n_e]).all())}
                if n_e <= 20_000_000:  # (zlib's reader takes ~0.45 s per million reads)
                    r = subprocess.run(cmd_z, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KU_NO_PGZIP="1"), timeout=600)
                    m = re.search(r"(\d+) sequences \(([\d.]+) Mbp\) processed in ([\d.]+)s", r.stderr.decode(errors="replace"))
                    if r.returncode == 0 and m:
                        gz["seconds_with_zlib_reader"] = float(m.group(3))
                out["e2e"]["gz"] = gz
            except Exception as e:
                out["e2e"]["gz"] = {"value": None, "error": str(e)[:200]}
        except Exception as e:
            out["e2e"] = {"value": None, "error": str(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def windowed_shapes(a, db, ctx, k, dev, stream, rev):
    """Extras of the default line: the fused kernel's WINDOWED instance (reads beyond 192 k-mers: mate pairs 2 x 150 + N, 10 kbp
    reads) on the same database -- the batches `--paired --reads 5000000` / `--read-len 10000 --reads 100000` time, one warm-up
    and three timed launches each: HIP-event time on the kernel's stream, the model's bytes, and the counter-measured HBM bytes
    of profiles/lookup_traffic.json (`shapes`, accepted for this kernel source only)."""
    import copy
    out = {}
    tj = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "lookup_traffic.json")))
    except Exception:
        pass
    for name, n, L, paired in (("pairs_2x150", 5_000_000, 150, True), ("reads_10kbp", 100_000, 10_000, False)):
        try:
            a2 = copy.copy(a)
            a2.reads, a2.read_len, a2.paired = n, L, paired
            seqs, off, lens, rl = make_batch(db, a2, 1, dev)
            nbytes = seqs.numel()
            taxa = torch.zeros(nbytes, dtype=torch.int32, device=dev)
            calls = torch.zeros(n, dtype=torch.int32, device=dev)
            ms = []
            for i in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ctx.reset_counts()
                torch.cuda.synchronize()
                e0.record()
                ctx.classify_batch_device(seqs.data_ptr(), nbytes, off.data_ptr(), lens.data_ptr(), n, calls.data_ptr(), taxa.data_ptr(),
                                          max_read_len=rl, stream=stream)
                e1.record()
                torch.cuda.synchronize()
                if i:
                    ms.append(e0.elapsed_time(e1))
            st = ctx.lookup_stats_device(seqs.data_ptr(), nbytes)
            k_ms = float(np.mean(ms))
            algo = float(lens.sum().item()) + 4.0 * n + st["lookups"] * 20 + 12 * st["sum_ceil_log2"]
            key = f"reads{n}_nt{a.nt}_species{a.species}_len{rl}" + ("_paired" if paired else "")
            ent = tj.get("shapes", {}).get(key) if tj.get("kernel_rev") == rev else None
            traffic = ent.get("hbm_bytes_per_launch") if ent else None
            out[name] = {"reads_per_launch": n, "read_len": rl, "kernel_ms": round(k_ms, 3), "Mreads_per_s": round(n / k_ms / 1e3, 2),
                         "algorithmic_bytes_per_launch": int(algo), "frac_model": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "frac_hw": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                         "traffic_source": (ent or {}).get("source", "no counter passes of this shape for this kernel source in profiles/lookup_traffic.json")}
            del seqs, off, lens, taxa, calls
            torch.cuda.empty_cache()
        except Exception as e:
            out[name] = {"error": str(e)[:200]}
    return out


def shard_bounds(synth_torch, dev, a, k, n_shards):
    """every rank derives the same shard plan from the same deterministic sample of the DB's bin keys"""
    probe = synth_torch.BenchDb(dev, n_species=min(a.species, 32), genome_len=min(a.genome_len, 50_000), k=k,
                                nt=a.nt, seed=7)
    g = torch.Generator(device=dev)
    g.manual_seed(20260927)  # the SAME sample in every process (an unseeded one gave every rank bounds of its own: gaps between the shards)
    bins = synth_torch.bin_key(probe.kmers[torch.randperm(probe.n_pairs, device=dev, generator=g)[:1_000_000]], k, a.nt)
    return synth_torch.quantile_bin_bounds(bins, 4 ** a.nt, n_shards)


def sharded_run(a, capi, synth_torch, dev, rank, local_rank, ws, uid, k, steps, warmup, stream):
    """the sharded step through ku_mgpu_step_device; returns a dict: elapsed seconds over `steps`, the group, the shard,
    whether every read was resolved exactly once, and the rank's own kernel measurements"""
    n_shards = max(a.db_shards or ws, ws)
    if ws == 1 and n_shards > 1 and os.environ.get("KU_MGPU_EXCHANGE") is None:
        # one rank's share of an n_shards-GPU layout: the rank takes the owner-routed path against itself (scan of the whole
        # batch, the records of the k-mers it owns, owner kernel, resolve) -- the kernels an n_shards-GPU run launches
        os.environ["KU_MGPU_FORCE_ROUTE"] = "1"
    t_b = [time.time()]
    bounds = shard_bounds(synth_torch, dev, a, k, n_shards)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    db = synth_torch.BenchDb(dev, n_species=a.species, genome_len=a.genome_len, k=k, nt=a.nt, seed=7, bin_lo=lo, bin_hi=hi)
    db.kmers = db.vals = None
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_b.append(time.time())
    mg = capi.Mgpu([local_rank], first_rank=rank, world=ws, unique_id=uid if ws > 1 else None)
    mg.ctx(0).adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), k, a.nt, 2, lo, hi)
    ids_t, par_t = db.tax.arrays()
    mg.set_taxonomy(capi.Tax(ids=ids_t, parents=par_t))  # slots cover the values of the shards that are resident
    mg.ctx(0).synchronize()
    t_b.append(time.time())
    db.pairs = None  # the probe table replaced the pairs (hash layout)
    torch.cuda.empty_cache()
    build = {"synthesis_of_the_shard_s": round(t_b[1] - t_b[0], 2),
             "slot_table_and_probe_table_s": round(t_b[2] - t_b[1], 2)}
    nb_batches = max(1, min(a.batches, 2))
    if rank == 0:
        batches = [make_batch(db, a, 1 + 17 * i, dev) for i in range(nb_batches)]
        L = batches[0][3]
    else:
        L = 2 * a.read_len + 1 if a.paired else a.read_len
        stride = L + 1
        one = (torch.zeros(a.reads * stride + 16, dtype=torch.uint8, device=dev),
               torch.zeros(a.reads, dtype=torch.int64, device=dev), torch.zeros(a.reads, dtype=torch.int32, device=dev), L)
        batches = [one]
    stride = L + 1
    n_bytes = a.reads * stride
    d_taxa = torch.zeros(n_bytes + 16, dtype=torch.int32, device=dev)
    d_calls = torch.zeros(a.reads, dtype=torch.int32, device=dev)
    rb = [a.reads * r // ws for r in range(ws + 1)]
    pb = [x * stride for x in rb]

    def step(i):
        b = batches[i % len(batches)]
        mg.step_device([{"d_seqs": b[0].data_ptr(), "d_seq_off": b[1].data_ptr(), "d_seq_len": b[2].data_ptr(),
                         "d_calls": d_calls.data_ptr(), "d_taxa": d_taxa.data_ptr(), "stream": stream}],
                       n_bytes, a.reads, rb, pb, max_read_len=L)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    mg.ctx(0).reset_counts()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    mg.reduce_state([stream])
    torch.cuda.synchronize()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if ws > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # every read was resolved exactly once over the whole world
    total_reads = int(mg.ctx(0).counts()["n_reads"].sum())
    ctx = mg.ctx(0)
    b = batches[0]
    nk = L - k + 1.0
    if os.environ.get("KU_ROUTE_DEBUG"):
        sys.stderr.write(f"[bench] rank {rank}: uses_routing {mg.uses_routing()} uses_rccl {mg.uses_rccl()}\n")
    if mg.uses_routing():
        # ---- the stages of this rank's routed step (HIP events on the streams the kernels run on), mean of three more steps
        mg.set_timing(True)
        acc = []
        for i in range(3):
            step(i)
            torch.cuda.synchronize()
            acc.append(mg.step_times(0))
        mg.set_timing(False)
        tm = {kk: float(np.mean([t[kk] for t in acc])) for kk in acc[0]}
        st = ctx.lookup_stats_device(b[0].data_ptr(), n_bytes)  # the k-mers this rank owns, and their bins
        # with fewer ranks than shards a rank scans and resolves 1 / ws of the reads where the full layout gives it
        # 1 / n_shards: its stage times in that layout
        share = ws / float(n_shards)
        t_equiv = (tm["scan_ms"] + tm["resolve_ms"]) * share + tm["owner_ms"]
        # algorithmic bytes of the rank's step (L + 4 per read of its slice, 16 + 12 * ceil(log2(n_bin + 1)) + 4
        # per lookup it owns): the ranks' figures add up to the single-GPU model
        bytes_algo = a.reads / n_shards * (L + 4.0) + st["lookups"] * 20 + 12 * st["sum_ceil_log2"]
        achieved = bytes_algo / (t_equiv * 1e-3) / 1e9 if t_equiv else 0.0
        rev = capi.kernel_rev()
        traffic, tnote = routed_traffic(os.path.join(ROOT, "profiles", "route_traffic.json"), rev, a.nt, a.species, n_shards, ws, a.reads, L, share)
        rf = {"bound": "hbm", "kernel": "this rank's stages of an owner-routed step: ku_lookup_kernel<3,...> (scan -> records) + "
                                        "ku_route_owner_kernel (probe, HLL, n_kmers) + ku_classify_short_kernel<..., ROUTE> (tickets -> calls), "
                                        "with the prefix sums between them",
              "kernel_rev": rev, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
              "frac_model": round(achieved / HBM_PEAK_GBS, 5),
              "frac_hw": round(traffic / (t_equiv * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and t_equiv else None,
              "traffic": traffic, "traffic_source": tnote,
              "kernel_ms": round(t_equiv, 3), "stage_ms_measured": {kk: round(v, 3) for kk, v in tm.items() if kk.endswith("_ms")},
              "stage_share_of_the_full_layout": {"scan": share, "resolve": share, "owner": 1.0},
              #  with fewer ranks than shards achieved / frac / kernel_ms price the rank's step IN THE FULL LAYOUT from the
              # measured stages (scan and resolve scaled by the share of the reads the rank has there): a model of a layout that
              # did not run, not a measured time.  The measured sum of this run's stages and its own fraction are given beside it.
              "modelled": share != 1.0,
              "measured_stage_sum_ms": round(tm["scan_ms"] + tm["owner_ms"] + tm["resolve_ms"], 3),
              "measured_frac_model_of_this_run": round((a.reads / ws * (L + 4.0) + st["lookups"] * 20 + 12 * st["sum_ceil_log2"]) /
                                                       ((tm["scan_ms"] + tm["owner_ms"] + tm["resolve_ms"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
              "rounds": tm["rounds"], "records_received": int(tm["records_received"]), "kmers_received": int(tm["kmers_received"]),
              "kmers_per_record": round(tm["kmers_received"] / max(tm["records_received"], 1), 2),
              "owned_lookups_per_step": int(st["lookups"]), "owned_fraction_of_kmers": round(st["lookups"] / max(1.0, a.reads * nk), 4),
              "algorithmic_bytes_per_step": int(bytes_algo), "mean_ceil_log2_bin": round(st["sum_ceil_log2"] / max(st["lookups"], 1), 3)}
        rec_b = 16.0 * tm["records_received"] / a.reads  # per read of the batch, what this owner received
        wire = {"exchange": "owner routing (super-k-mer records)", "scatter_in_bytes_per_read": round((stride + 12.0) / ws, 1),
                "records_in_bytes_per_read": round(rec_b * (ws - 1) / max(ws, 1), 1), "slots_out_bytes_per_read": round(4.0 * tm["kmers_received"] / a.reads * (ws - 1) / max(ws, 1), 1),
                "note": "per rank and per read OF THE BATCH: a rank scans 1/N of the reads and sends every run of k-mers that share a minimizer "
                        "occurrence as one 16-byte record to the owner of the bin (round 3: 12 B per k-mer), a 4-byte slot per k-mer comes back; "
                        f"the position-wise exchange (KU_MGPU_EXCHANGE=slots) broadcasts every read to every rank and moves "
                        f"{round(4.0 * stride * (ws - 1) / ws, 1)} B per read per rank"}
        return {"elapsed": elapsed, "mg": mg, "db": db, "ok": total_reads == a.reads * steps, "roofline": rf, "wire": wire,
                "n_shards": n_shards, "read_len": L, "build": build}
    # ---- this rank's kernels alone (outside the timed region): the sharded lookup kernel over the whole batch, the
    # resolve kernel over the rank's slice of the reads
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    reps = 3
    ctx.lookup_device(b[0].data_ptr(), n_bytes, d_taxa.data_ptr(), flags=capi.KU_F_KEEP_SLOTS, stream=stream)
    ev[0].record()
    for _ in range(reps):
        ctx.lookup_device(b[0].data_ptr(), n_bytes, d_taxa.data_ptr(), flags=capi.KU_F_KEEP_SLOTS, stream=stream)
    ev[1].record()
    r0, r1 = rb[rank], rb[rank + 1]
    ev[2].record()
    if r1 > r0:
        ctx.resolve_device(b[0].data_ptr(), b[1][r0:].data_ptr(), b[2][r0:].data_ptr(), r1 - r0, d_calls[r0:].data_ptr(),
                           d_taxa.data_ptr(), flags=capi.KU_F_NO_COUNTS, max_read_len=L, stream=stream)
    ev[3].record()
    torch.cuda.synchronize()
    lookup_ms = ev[0].elapsed_time(ev[1]) / reps
    resolve_ms = ev[2].elapsed_time(ev[3])
    st = ctx.lookup_stats_device(b[0].data_ptr(), n_bytes)
    # algorithmic bytes of the rank's lookup launch: the whole batch is scanned on every rank (L bytes per read) and the
    # per-k-mer slot array written (4 B per k-mer position: what the exchange carries), the owned k-mers are searched
    # (16 + 12 * ceil(log2(n_bin + 1)) + 4 per lookup)
    bytes_algo = a.reads * L + 4.0 * a.reads * (L - k + 1) + st["lookups"] * 20 + 12 * st["sum_ceil_log2"]
    achieved = bytes_algo / (lookup_ms * 1e-3) / 1e9 if lookup_ms else 0.0
    rf = {"bound": "hbm", "kernel": "ku_lookup_kernel<1,1,true,false> (sharded lookup; this rank)", "achieved": round(achieved, 2),
          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
          "kernel_ms": round(lookup_ms, 3), "resolve_slice_ms": round(resolve_ms, 3), "owned_lookups_per_launch": int(st["lookups"]),
          "owned_fraction_of_kmers": round(st["lookups"] / max(1.0, a.reads * (L - k + 1.0)), 4),
          "algorithmic_bytes_per_launch": int(bytes_algo), "mean_ceil_log2_bin": round(st["sum_ceil_log2"] / max(st["lookups"], 1), 3)}
    wire = {"exchange": "position-wise" if ws > 1 else "none", "broadcast_in_bytes_per_read": stride + 12 if ws > 1 else 0,
            "exchange_out_bytes_per_read": round(4.0 * stride * (ws - 1) / ws, 1) if ws > 1 else 0.0,
            "exchange_in_bytes_per_read": round(4.0 * stride * (ws - 1) / ws, 1) if ws > 1 else 0.0,
            "note": "per rank; an all-to-all of 4-byte slots per base position (ku_mgpu.cpp)"}
    return {"elapsed": elapsed, "mg": mg, "db": db, "ok": total_reads == a.reads * steps, "roofline": rf, "wire": wire,
            "n_shards": n_shards, "read_len": L, "build": build}


# ---- N > 1: what the run is about to do, said before it does it, and a bound on how long it may take -----------------------------
# The driver gives `bench.py --gpus N` 1800 s.  KU_BENCH_BUDGET_S (default 1500) is this script's own bound on the whole run: the
# sharded leg gets what the replicas leg left of it (KU_BENCH_SHARDED_LEG_LIMIT fixes it instead), and a rendezvous or a
# collective that never returns ends the run with ONE line that says so instead of the driver's kill.
BUDGET_S = float(os.environ.get("KU_BENCH_BUDGET_S", "1500"))
SYNTH_PAIRS_PER_S = 3.4e7   # shard synthesis on one MI355X (profiles/README.md: 3.1 G pairs in ~90 s)
TABLE_PAIRS_PER_S = 3.8e9   # slot table + probe table behind it (0.61 G pairs in 0.16 s)


def routed_traffic(path, rev, nt, species, n_shards, ws, reads, read_len, share):
    """counter-measured HBM bytes of one rank's routed step (profiles/route_traffic.json, collected by scripts/profile_r06.sh with
    ONE rank doing every rank's scan and resolve and the owner work of shard 0) -> (bytes or None, where they come from).
    The workload as it ran (`..._ws{ws}_...`): scan and resolve at `share` of the profile's; a world that holds the whole layout
    (ws == n_shards > 1) finds its workload under ws1: a rank's scan and resolve are 1 / ws of those passes, its owner work the
    same.  A profile of another kernel source is refused."""
    traffic, tnote = None, "no counter profile of this kernel source in profiles/route_traffic.json"
    if not os.path.exists(path):
        return traffic, tnote
    try:
        tj = json.load(open(path))
        wl = tj.get("workloads", {})
        ent = wl.get(f"nt{nt}_species{species}_shards{n_shards}_ws{ws}_reads{reads}_len{read_len}")
        ent1 = wl.get(f"nt{nt}_species{species}_shards{n_shards}_ws1_reads{reads}_len{read_len}") if ws == n_shards and ws > 1 else None
        if ent and tj.get("kernel_rev") == rev:
            by = ent["hbm_bytes_per_step"]
            traffic = (by.get("scan", 0) + by.get("resolve", 0)) * share + by.get("owner", 0)
            tnote = ent.get("source", "profiles/route_traffic.json")
        elif ent1 and tj.get("kernel_rev") == rev:
            by = ent1["hbm_bytes_per_step"]
            traffic = (by.get("scan", 0) + by.get("resolve", 0)) / ws + by.get("owner", 0)
            tnote = (ent1.get("source", "profiles/route_traffic.json") +
                     f" -- collected with one rank doing all {ws} ranks' scan and resolve: those two taken at 1 / {ws}, the owner stage as measured")
        elif ent or ent1:
            tnote = f"profiles/route_traffic.json is of kernel source {tj.get('kernel_rev')}, this is {rev}: refused"
    except Exception:
        pass
    return traffic, tnote


def planned_table_bytes(n_pairs, free_hbm):
    """the probe table ku_ctx_set_taxonomy will lay out: 128-byte lines of 8 entries at load 0.2 when that takes at
    most 40 % of the free HBM, else 0.3, 0.45, 0.6, 0.8; the sorted layout (12 B per pair) when none fits"""
    for lf in (0.2, 0.3, 0.45, 0.6, 0.8):
        b = int(n_pairs / (8 * lf)) * 128
        if b <= (0.4 if lf == 0.2 else 0.85) * free_hbm:
            return b, lf
    return n_pairs * 12, None


def preflight(a, rank, ws, local_rank, dev, leg, species, genome_len, nt, shards, reads, t_start):
    """one stderr line per rank + (rank 0) the plan of the leg: devices, free HBM, bytes to synthesise and to lay out, ETA, limits"""
    free_b, total_b = torch.cuda.mem_get_info(dev)
    # distinct canonical k-mers of the synthetic genomes: ~ species x genome_len (3 % substitutions between siblings keep most apart)
    pairs_all = species * genome_len
    pairs_rank = pairs_all // max(shards, 1)
    idx_bytes = 8 * (4 ** nt // max(shards, 1) + 1)  # (quantile shard bounds: about 1 / shards of the bins each)
    table_b, lf = planned_table_bytes(pairs_rank, free_b - pairs_rank * 12 - idx_bytes)
    eta = pairs_rank / SYNTH_PAIRS_PER_S + pairs_rank / TABLE_PAIRS_PER_S + 15
    info = {"leg": leg, "rank": rank, "world": ws, "device": local_rank, "device_name": torch.cuda.get_device_name(dev),
            "visible_devices": torch.cuda.device_count(), "hbm_free_gb": round(free_b / 1e9, 1), "hbm_total_gb": round(total_b / 1e9, 1),
            "pairs_this_rank": pairs_rank, "pair_bytes_gb": round(pairs_rank * 12 / 1e9, 1), "index_bytes_gb": round(idx_bytes / 1e9, 2),
            "planned_table_gb": round(table_b / 1e9, 1), "planned_load_factor": lf, "reads_per_step": reads,
            "build_eta_s": round(eta), "elapsed_s": round(time.time() - t_start), "budget_s": BUDGET_S}
    if table_b + pairs_rank * 12 + idx_bytes > free_b:
        info["warning"] = "pairs + index + table exceed the free HBM of this device: the library will fall back to a denser layout or fail"
    sys.stderr.write("[bench preflight] " + json.dumps(info) + "\n")
    sys.stderr.flush()
    return info


def start_budget_watchdog(rank, ws, t_start, state):
    """rank 0 prints one JSON line with what there is when the budget runs out (state["result"], or a line that names the step that
    did not return); every rank leaves.  Cancelled by main() when the run is through."""
    def expire():
        try:  # where every rank stands, for the log
            import faulthandler
            sys.stderr.write(f"[bench] rank {rank}: the budget of {BUDGET_S:.0f} s ran out while: {state.get('doing', '?')}\n")
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            sys.stderr.flush()
        except Exception:
            pass
        if rank == 0:
            r = state.get("result") or {"metric": "Mreads/s (150 bp)", "value": None, "unit": "Mreads/s", "n_gpus": ws,
                                        "higher_is_better": True, "data": "synthetic"}
            r["error"] = (f"bench.py gave up after its own budget of {BUDGET_S:.0f} s (KU_BENCH_BUDGET_S) while: {state.get('doing', '?')}; "
                          f"started {time.time() - t_start:.0f} s ago")
            print(json.dumps(r), flush=True)
        os._exit(3)
    w = threading.Timer(max(30.0, BUDGET_S - (time.time() - t_start)), expire)
    w.daemon = True
    w.start()
    return w


def main():
    t_start = time.time()
    a = parse()
    self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws != a.gpus and ws > 1:
        a.gpus = ws
    # debugging aid for 1-GPU boxes: KU_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 (RCCL refuses that; numbers are
    # meaningless) -- the N > 1 code path is otherwise exercised by tests/test_gpu_mgpu.py inside one process
    if os.environ.get("KU_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from krakenuniq_amd import capi, synth_torch
    # KU_BENCH_ONE_DEVICE=1 (every rank on cuda:0; the C++ driver then needs KU_RCCL_LIB=tests/rccl_shim/libku_rccl_shim.so, RCCL
    # itself refuses two ranks on a device): the few torch.distributed calls of this script go through gloo on host tensors
    one_dev = os.environ.get("KU_BENCH_ONE_DEVICE") == "1"
    cdev = torch.device("cpu") if one_dev else dev

    def fresh_uid():
        """id of one RCCL communicator of the C++ driver: rank 0 makes it, everybody gets it (an id serves one init)"""
        if ws <= 1:
            return None
        t = torch.zeros(128, dtype=torch.uint8, device=cdev)
        if rank == 0:
            t.copy_(torch.from_numpy(capi.mgpu_unique_id()))
        dist.broadcast(t, 0)
        return t.cpu().numpy()

    run_state = {"doing": "torch.distributed rendezvous (init_process_group)"}
    budget_watchdog = start_budget_watchdog(rank, ws, t_start, run_state) if ws > 1 else None
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    run_state["doing"] = "the RCCL id of the C++ driver's communicator (broadcast)"
    uid = fresh_uid()

    k = 31
    sharded = a.mode == "sharded"
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)  # kernels, RCCL collectives and the timing events all live on it
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    result_extra = {}

    if sharded:
        t_build = time.time()
        run_state["doing"] = "the sharded run (shard synthesis, probe table, routed steps)"
        pf = preflight(a, rank, ws, local_rank, dev, "sharded", a.species, a.genome_len, a.nt, max(a.db_shards or ws, ws), a.reads, t_start) if ws > 1 or a.config != 1 else None
        sr = sharded_run(a, capi, synth_torch, dev, rank, local_rank, ws, uid, k, a.steps, a.warmup, stream)
        elapsed, mg, db, L = sr["elapsed"], sr["mg"], sr["db"], sr["read_len"]
        value = a.reads * a.steps / elapsed / 1e6
        ctx = mg.ctx(0)
        what = f"{a.reads} mate pairs 2 x {a.read_len} + N" if a.paired else f"{a.reads} reads of {L} bp"
        workload = (a.preset + f"; rank r of {ws} holds shard r of {sr['n_shards']}") if a.preset else \
            (f"synthetic DB ({a.species} taxa x {a.genome_len} bp, nt={a.nt}) in {sr['n_shards']} minimizer-range shards over "
             f"{ws} GPU(s)")
        result = {
            "metric": "Mreads/s (150 bp)", "value": round(value, 3), "unit": "Mreads/s", "n_gpus": ws, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload + f", {what} per step broadcast from rank 0",
                       "db_pairs_per_gpu": db.n_pairs, "db_bytes_per_gpu": db.n_pairs * 12 + db.offsets.numel() * 8,
                       "hbm_layout": ctx.db_layout(), "k": k, "nt": a.nt, "taxa": a.species,
                       "reads_per_step": a.reads, "read_len": L, "parallelism": f"sharded{ws}", "db_shards": sr["n_shards"],
                       "exchange": "RCCL" if mg.uses_rccl() else "none",
                       "every_read_resolved_once": sr["ok"], "db_build_s": round(time.time() - t_build - elapsed, 1),
                       "db_build_split": sr["build"]},
            "roofline": sr["roofline"], "wire": sr["wire"],
        }
        if pf:
            result["config"]["preflight"] = pf
        if budget_watchdog:
            budget_watchdog.cancel()
        if rank == 0:
            print(json.dumps(result), flush=True)
        mg.close()
        if ws > 1:
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- replicas (N = 1: the plain single-GPU run)
    t_build = time.time()
    run_state["doing"] = "the replicas leg (database synthesis, probe table, steps, end-of-run reduce)"
    pf_rep = preflight(a, rank, ws, local_rank, dev, "replicas", a.species, a.genome_len, a.nt, 1, a.reads, t_start) if ws > 1 else None
    db = synth_torch.BenchDb(dev, n_species=a.species, genome_len=a.genome_len, k=k, nt=a.nt, seed=7)
    db.kmers = db.vals = None
    torch.cuda.empty_cache()
    mg = None
    if ws > 1:
        mg = capi.Mgpu([local_rank], first_rank=rank, world=ws, unique_id=uid, flags=capi.KU_MGPU_REPLICAS)
        ctx = mg.ctx(0)
    else:
        ctx = capi.Ctx(local_rank)
    ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), k, a.nt, 2, keep=db)
    ids_t, par_t = db.tax.arrays()
    ctax = capi.Tax(ids=ids_t, parents=par_t)
    if mg:
        mg.set_taxonomy(ctax)
    else:
        ctx.set_taxonomy(ctax)
    nb_batches = max(1, a.batches)
    batches = [make_batch(db, a, 1 + 17 * i + 1000 * rank, dev) for i in range(nb_batches)]
    read_len = batches[0][3]
    n_bytes = batches[0][0].numel()
    d_taxa = torch.zeros(n_bytes, dtype=torch.int32, device=dev)
    d_calls = torch.zeros(a.reads, dtype=torch.int32, device=dev)
    build_s = time.time() - t_build
    torch.cuda.synchronize()
    n_k0 = read_len - k + 1
    fused0 = (os.environ.get("KU_NO_FUSED") is None and os.environ.get("KU_NO_FUSED_RLE") is None and ctx.db_layout()["hash"]
              and (n_k0 <= int(os.environ.get('KU_SHORT_ONE_PASS_MAX', 192)) or (n_k0 <= 65535 and os.environ.get("KU_NO_WINDOWED") is None)))
    runs_out = a.output == "runs" and fused0
    if fused0:  # calls + {code, first k-mer} runs per read, written by the kernel that classifies (the step's output with --output runs)
        runs_cap = ctx.device_rle_runs_cap(n_bytes, a.reads, read_len)
        d_runs = torch.zeros((runs_cap, 2), dtype=torch.int32, device=dev)
        d_roff = torch.zeros(a.reads, dtype=torch.int64, device=dev)
        d_rcnt = torch.zeros(a.reads, dtype=torch.int32, device=dev)
        d_nruns = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    n_k = read_len - k + 1
    fused = (os.environ.get("KU_NO_FUSED") is None and ctx.db_layout()["hash"]
             and (n_k <= int(os.environ.get('KU_SHORT_ONE_PASS_MAX', 192)) or (n_k <= 65535 and os.environ.get("KU_NO_WINDOWED") is None)))

    def step(i, timed):
        b = batches[i % nb_batches]
        if i % nb_batches == 0:  # a fresh run every rotation: the registers start empty
            tstream.synchronize()
            ctx.reset_counts()
        if fused:  # short reads, whole DB resident: ONE fused kernel (wave per read) does lookup + counts + resolve
            if timed:
                ev[i][0].record()
            if runs_out:
                ctx.classify_batch_device_rle(b[0].data_ptr(), n_bytes, b[1].data_ptr(), b[2].data_ptr(), a.reads, d_calls.data_ptr(),
                                              d_runs.data_ptr(), runs_cap, d_roff.data_ptr(), d_rcnt.data_ptr(), d_nruns.data_ptr(),
                                              max_read_len=read_len, stream=stream)
            else:
                ctx.classify_batch_device(b[0].data_ptr(), n_bytes, b[1].data_ptr(), b[2].data_ptr(), a.reads,
                                          d_calls.data_ptr(), d_taxa.data_ptr(), max_read_len=read_len, stream=stream)
            if timed:
                ev[i][1].record()
        else:
            if timed:
                ev[i][0].record()
            ctx.lookup_device(b[0].data_ptr(), n_bytes, d_taxa.data_ptr(), stream=stream)
            if timed:
                ev[i][1].record()
            ctx.resolve_device(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), a.reads, d_calls.data_ptr(),
                               d_taxa.data_ptr(), max_read_len=read_len, stream=stream)

    for i in range(a.warmup):
        step(i, False)
    torch.cuda.synchronize()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, True)
    if mg:  # end-of-run merge of the per-taxon state over the ranks (C++ driver, RCCL), part of the job
        mg.reduce_state([stream])
    torch.cuda.synchronize()
    if ws > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if ws > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in ev])) if a.steps else 0.0
    state_ok = None
    if mg:  # the reads of the last (possibly partial) rotation of every rank, summed over the world
        last_rot = a.steps - ((a.steps - 1) // nb_batches) * nb_batches
        state_ok = int(ctx.counts()["n_reads"].sum()) == a.reads * last_rot * ws

    total_reads = a.reads * a.steps * ws
    value = total_reads / elapsed / 1e6
    out_info = {"form": "calls + one 32-bit code per base position (ku_classify_batch_device)"}
    if runs_out:
        out_info = {"form": "calls + run-length encoded per-k-mer codes ({code, first k-mer} per run: the hit list of the Kraken line, "
                            "classify.cpp:980-1010), written by the fused kernel itself (ku_classify_batch_device_rle; what the classify "
                            "executable takes); no per-k-mer array exists"}
    if fused0:
        # outside the timed region, both output forms of one batch without accounting (the state of the timed run stays as it
        # is): their times, and the runs expanded again against the per-k-mer array
        b = batches[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms_taxa, ms_runs = [], []
        for rep in range(3):
            e0.record()
            ctx.classify_batch_device_rle(b[0].data_ptr(), n_bytes, b[1].data_ptr(), b[2].data_ptr(), a.reads, d_calls.data_ptr(),
                                          d_runs.data_ptr(), runs_cap, d_roff.data_ptr(), d_rcnt.data_ptr(), d_nruns.data_ptr(),
                                          max_read_len=read_len, flags=capi.KU_F_NO_COUNTS, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            ms_runs.append(e0.elapsed_time(e1))
        extent, n_runs = int(d_nruns.item()), int(d_rcnt.sum().item())
        calls_runs = d_calls.clone()
        for rep in range(3):
            e0.record()
            ctx.classify_batch_device(b[0].data_ptr(), n_bytes, b[1].data_ptr(), b[2].data_ptr(), a.reads, d_calls.data_ptr(),
                                      d_taxa.data_ptr(), max_read_len=read_len, flags=capi.KU_F_NO_COUNTS, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            ms_taxa.append(e0.elapsed_time(e1))
        same = extent <= runs_cap and bool(torch.equal(calls_runs, d_calls))
        if same:
            nk_t = torch.clamp(b[2].to(torch.int64) - (k - 1), min=0)
            flat = synth_torch.expand_runs(d_runs, d_roff, d_rcnt, nk_t)
            stride = read_len + 1
            same = bool(torch.equal(flat.view(a.reads, n_k0), d_taxa[:a.reads * stride].view(a.reads, stride)[:, :n_k0]))
            del flat
        out_info.update({"runs_per_read": round(n_runs / a.reads, 2), "run_bytes_per_step": n_runs * 8, "run_array_extent_used": extent,
                         "expanded_runs_equal_the_per_kmer_array": same,
                         "ms_without_accounting": {"per_position_codes": round(min(ms_taxa), 3), "runs_from_the_kernel": round(min(ms_runs), 3)}})
    # roofline of the dominant kernel: algorithmic bytes per launch (mean over the batches) / HIP-event time
    stats = [ctx.lookup_stats_device(b[0].data_ptr(), n_bytes) for b in batches]
    lookups = float(np.mean([s["lookups"] for s in stats]))
    sum_log = float(np.mean([s["sum_ceil_log2"] for s in stats]))
    bytes_algo = a.reads * (read_len + 4) + lookups * 20 + 12 * sum_log
    achieved = bytes_algo / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
    rev = capi.kernel_rev()
    traffic, traffic_note = None, "no counter profile of this kernel source in profiles/lookup_traffic.json"
    tpath = os.path.join(ROOT, "profiles", "lookup_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # the default workload at the top level, the other read shapes (--paired, --read-len ..., --nt 15) under "shapes"
            skey = f"reads{a.reads}_nt{a.nt}_species{a.species}_len{read_len}" + ("_paired" if a.paired else "")
            ent = tj.get("shapes", {}).get(skey)
            if ent is None and tj.get("reads") == a.reads and tj.get("nt") == a.nt and tj.get("species") == a.species and \
                    tj.get("read_len", 150) == read_len and not a.paired:
                ent = tj
            same_wl = ent is not None and ent.get("kernel", "").startswith("ku_classify_short" if fused else "ku_lookup")
            if same_wl and tj.get("kernel_rev") == rev:
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_note = ent.get("source", "profiles/lookup_traffic.json")
            elif same_wl:
                traffic_note = f"profiles/lookup_traffic.json is of kernel source {tj.get('kernel_rev')}, this is {rev}: refused"
        except Exception:
            pass
    default_cfg = (a.reads == 10_000_000 and a.species == 2000 and a.genome_len == 310_000 and a.nt == 13
                   and not a.paired and a.read_len == 150)
    if default_cfg and ws == 1:
        workload = "configs[1]: 8 GB MiniKraken-style DB (k=31), 10M synthetic 150 bp reads, 1xMI355X"
    else:
        workload = (f"{'configs[1] ' if default_cfg else ''}synthetic DB ({a.species} taxa x {a.genome_len} bp, nt={a.nt}), "
                    f"{a.reads} {'paired 2x' + str((read_len - 1) // 2) if a.paired else str(read_len)} bp reads per GPU "
                    f"per step, {a.mode} x{ws}")
    frac_model = achieved / HBM_PEAK_GBS
    result = {
        "metric": "Mreads/s (150 bp)", "value": round(value, 3), "unit": "Mreads/s", "n_gpus": ws, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(elapsed / max(a.steps, 1) * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload, "distinct_batches": nb_batches,
                   "state": f"per-taxon state zeroed every {nb_batches} steps (a fresh {nb_batches * a.reads // 1_000_000} M-read run per rotation)",
                   "db_pairs_per_gpu": db.n_pairs, "db_bytes_per_gpu": db.n_pairs * 12 + db.offsets.numel() * 8,
                   "hbm_layout": ctx.db_layout(), "k": k, "nt": a.nt, "taxa": a.species, "reads_per_gpu_per_step": a.reads,
                   "read_len": read_len, "parallelism": f"{a.mode}{ws}", "db_build_s": round(build_s, 1), "output": out_info},
        "roofline": {"bound": "hbm", "kernel": "ku_classify_short_kernel (fused lookup+resolve)" if fused else "ku_lookup_kernel",
                     "kernel_rev": rev, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(frac_model, 5), "frac_model": round(frac_model, 5),
                     "frac_hw": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and kernel_ms else None,
                     "traffic": traffic, "traffic_source": traffic_note,
                     "which_fraction_is_which": "frac (= frac_model) prices the REFERENCE algorithm's bytes (SURVEY 8d: a comparison search of 12-byte pairs) "
                                                "against the kernel's time, as the contract defines `achieved`; the bucket probe moves fewer bytes -- "
                                                "frac_hw (counter-measured FETCH_SIZE x 2 + WRITE_SIZE per launch) is the HBM utilisation; the kernel is "
                                                "bound by instruction issue, not by HBM (DESIGN 3.2, 4)",
                     "algorithmic_bytes_per_launch": int(bytes_algo), "lookups_per_launch": int(lookups),
                     "mean_ceil_log2_bin": round(sum_log / max(lookups, 1), 3), "kernel_ms": round(kernel_ms, 3)},
    }
    if mg:
        result["config"]["state_merge"] = "ku_mgpu_reduce_state (RCCL all-reduce) once, inside the timed region"
        result["config"]["merged_read_count_ok"] = state_ok
    if rank == 0 and ws == 1 and default_cfg and fused and not a.no_extras:
        tstream.synchronize()
        result["roofline"]["windowed_instance"] = windowed_shapes(a, db, ctx, k, dev, stream, rev)
    if rank == 0 and ws == 1 and not (a.cpu_sample == 0 and a.no_extras):
        # a run of batch 0 alone for the parity sample / device pipeline legs
        tstream.synchronize()
        ctx.reset_counts()
        b = batches[0]
        ctx.classify_batch_device(b[0].data_ptr(), n_bytes, b[1].data_ptr(), b[2].data_ptr(), a.reads, d_calls.data_ptr(),
                                  d_taxa.data_ptr(), max_read_len=read_len, stream=stream)
        torch.cuda.synchronize()
        calls = d_calls.cpu().numpy().view(np.uint32)
        n_host = min(a.reads, 4_000_000)  # per-k-mer codes only for the CPU sample
        taxa = d_taxa[:n_host * (read_len + 1)].cpu().numpy().view(np.uint32)
        try:
            algo0 = a.reads * (read_len + 4) + stats[0]["lookups"] * 20 + 12 * stats[0]["sum_ceil_log2"]
            result.update(host_legs(a, db, ctx, (b[0], b[1], b[2]), read_len, calls, taxa, k, algo_bytes0=algo0))
        except Exception as e:
            result["cpu_baseline"] = {"value": None, "unit": "Mreads/s", "cores": host_cores(), "kind": "reference",
                                      "sample": f"failed: {e}"}
    if ws > 1 and not a.no_extras and not os.environ.get("KU_BENCH_NO_SHARDED_LEG"):
        # the same world once more with the database sharded by minimizer range (strong scaling, configs[2] layout)
        # an exchange that never returns must not take the line measured above with it
        def give_up():
            if rank == 0:
                result["sharded"] = {"value": None, "error": "the sharded leg did not finish within its time limit"}
                print(json.dumps(result), flush=True)
            os._exit(0)
        # the leg's limit: what the replicas leg left of the run's budget, less a minute for the line (KU_BENCH_SHARDED_LEG_LIMIT fixes it)
        leg_limit = float(os.environ.get("KU_BENCH_SHARDED_LEG_LIMIT", "0")) or max(120.0, BUDGET_S - (time.time() - t_start) - 60.0)
        run_state["doing"] = f"the sharded leg (limit {leg_limit:.0f} s)"
        run_state["result"] = result
        watchdog = threading.Timer(leg_limit, give_up)
        watchdog.daemon = True
        watchdog.start()
        # a rank that fails in this leg (an allocation, say) leaves the others waiting in an exchange: it says so in the rendezvous
        # store, everybody looks there every two seconds, and the run ends with the line measured above instead of at the limit
        leg_store, leg_over, fail_key = dist.distributed_c10d._get_default_store(), threading.Event(), "ku_bench/sharded_leg_failed"

        def watch_the_others():
            while not leg_over.wait(2.0):
                try:
                    if not leg_store.check([fail_key]):
                        continue
                    why = leg_store.get(fail_key).decode(errors="replace")
                except Exception:
                    if rank != 0 and not leg_over.is_set():
                        os._exit(0)  # the store went with rank 0 (or the launcher): nobody is left to exchange with
                    return
                if leg_over.is_set():
                    return
                if rank == 0:
                    result["sharded"] = {"value": None, "error": why[:300]}
                    print(json.dumps(result), flush=True)
                os._exit(0)
        threading.Thread(target=watch_the_others, daemon=True).start()
        try:
            if os.environ.get("KU_BENCH_TEST_FAIL_SHARDED_RANK") == str(rank):
                raise RuntimeError("injected by KU_BENCH_TEST_FAIL_SHARDED_RANK (tests/test_gpu_bench_contract.py)")
            del batches, d_taxa, d_calls
            mg.close()
            mg = None
            del db
            torch.cuda.empty_cache()
            headline = not a.mode_given and a.config == 1  # plain `bench.py --gpus N`: the sharded layout is the headline
            import copy
            a2 = copy.copy(a)
            if headline:
                # configs[2] scaled to this world: the standard geometry, one ~37 GB minimizer-range shard (12 000 species) per GPU
                # -- at N = 8 the ~300 GB database of BASELINE.json --, 10 M x 150 bp reads per step over all ranks
                # (KU_BENCH_SHARD_SPECIES / _GENOME_LEN / _READS / _NT: the same flow at a size a test box holds N times)
                # KU_BENCH_SCALE_DIV=d: the rehearsal of exactly this flow on a test box -- the preset's sizes divided by d (species per
                # shard, reads per step), everything else as the driver runs it (tests/test_gpu_bench_contract.py)
                div = max(1, int(os.environ.get("KU_BENCH_SCALE_DIV", "1")))
                a2.nt, a2.db_shards = int(os.environ.get("KU_BENCH_SHARD_NT", "15")), ws
                a2.species = max(4, int(os.environ.get("KU_BENCH_SHARD_SPECIES", "12000")) // div) * ws
                a2.genome_len = int(os.environ.get("KU_BENCH_SHARD_GENOME_LEN", "310000"))
                a2.reads, a2.read_len, a2.paired, a2.batches = max(1000, int(os.environ.get("KU_BENCH_SHARD_READS", "10000000")) // div), 150, False, 2
            s_steps = a.steps if headline else max(2, min(a.steps, 4))
            s_warm = a.warmup if headline else 1
            pf_sh = preflight(a2, rank, ws, local_rank, dev, "sharded", a2.species, a2.genome_len, a2.nt, max(a2.db_shards or ws, ws), a2.reads, t_start)
            pf_sh["leg_limit_s"] = round(leg_limit)
            sr = sharded_run(a2, capi, synth_torch, dev, rank, local_rank, ws, fresh_uid(), k, s_steps, s_warm, stream)
            el = sr["elapsed"]
            result["sharded"] = {"value": round(a2.reads * s_steps / el / 1e6, 3), "unit": "Mreads/s", "scaling": "strong",
                                 "steps": s_steps, "warmup": s_warm, "ms_per_step": round(el / s_steps * 1e3, 3), "db_pairs_per_gpu": sr["db"].n_pairs,
                                 "headline": headline, "reads_per_step": a2.reads, "nt": a2.nt, "taxa": a2.species, "db_shards": sr["n_shards"],
                                 "db_build_split": sr["build"], "hbm_layout": sr["mg"].ctx(0).db_layout(),
                                 "every_read_resolved_once": sr["ok"], "roofline": sr["roofline"], "wire": sr["wire"], "preflight": pf_sh,
                                 "path": "ku_mgpu_step_device: " + ("scatter of the read slices -> scan of the own slice -> one 16-byte record per run of k-mers to the owner of its bin (all-to-all) -> probe + accounting at the owner -> 4-byte slots back -> per-slice resolve" if sr["wire"].get("exchange", "").startswith("owner routing") else "ncclBroadcast -> owner lookup -> all-to-all (grouped ncclSend/ncclRecv) + max-merge -> per-slice resolve")}
            sr["mg"].close()
        except Exception as e:
            import traceback
            why = f"rank {rank} failed in the sharded leg: {type(e).__name__}: {e}"
            sys.stderr.write(f"[bench] {why}\n{traceback.format_exc()}")
            sys.stderr.flush()
            try:  # the first failure is the cause; a rank that fell over the first one's departure reports that one
                if leg_store.check([fail_key]):
                    why = leg_store.get(fail_key).decode(errors="replace")
                else:
                    leg_store.set(fail_key, why[:300])
            except Exception:
                pass
            result["sharded"] = {"value": None, "error": why[:300]}
        leg_over.set()
        watchdog.cancel()
    sh = result.get("sharded") if ws > 1 else None
    if sh and sh.get("headline") and sh.get("value"):
        # plain `bench.py --gpus N`: the line is the sharded layout's (the 300 GB configuration of BASELINE.json at N = 8);
        # the replicas of the 8 GB database measured above are the second leg
        rep = {kk: result[kk] for kk in ("value", "unit", "steps", "warmup", "ms_per_step", "scaling", "config", "roofline")}
        result = {"metric": "Mreads/s (150 bp)", "value": sh["value"], "unit": "Mreads/s", "n_gpus": ws, "steps": sh["steps"],
                  "warmup": sh["warmup"], "ms_per_step": sh["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                  "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                  "config": {"workload": f"configs[2] scaled to {ws} GPUs: standard-geometry DB (k=31, nt={sh['nt']}, {sh['taxa']} taxa, {sh['db_pairs_per_gpu'] * 12 / 1e9:.1f} GB of pairs per GPU) "
                                         f"sharded by minimizer bin, rank r holds shard r of {sh['db_shards']}; {sh['reads_per_step']} reads of 150 bp per step, "
                                         "scattered from rank 0, owner-routed over RCCL",
                             "db_pairs_per_gpu": sh["db_pairs_per_gpu"], "hbm_layout": sh["hbm_layout"], "k": k, "nt": sh["nt"], "taxa": sh["taxa"],
                             "reads_per_step": sh["reads_per_step"], "read_len": 150, "parallelism": f"sharded{ws}", "db_shards": sh["db_shards"],
                             "every_read_resolved_once": sh["every_read_resolved_once"], "db_build_split": sh["db_build_split"], "path": sh["path"],
                             "preflight": sh.get("preflight"), "budget_s": BUDGET_S, "elapsed_s": round(time.time() - t_start)},
                  "roofline": sh["roofline"], "wire": sh["wire"], "replicas": rep}
    if budget_watchdog:
        budget_watchdog.cancel()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if ws > 1 and (result.get("sharded") or {}).get("error"):
        os._exit(0)  # the other ranks left from inside an exchange: nothing orderly remains to be taken down
    if mg:
        mg.close()
    if ws > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
