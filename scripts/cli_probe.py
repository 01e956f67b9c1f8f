#!/usr/bin/env python3
"""Measurement aid (run through gpurun): the bench database + a FASTQ file written ONCE, then the `classify` executable
several times -- one run per VARIANT -- with its stage times, the report's stage times and (optionally) a rocprofv3 kernel
table per run.

  python scripts/cli_probe.py <species> <reads> <out_tag> VARIANT [VARIANT ...]
  VARIANT = label[:key=value[,key=value...]]      keys: any environment variable, or
            REPORT=1 (add -r), PROF=1 (run under rocprofv3 --kernel-trace --stats), PMC=<counter> (run under rocprofv3 --pmc <counter>, the
            fused kernel's launches only: their sum goes to gpurun_out/<out_tag>_<label>_pmc.json), CPULIST=0-31 (taskset), LIB=<dir> (a directory that holds another
            build of libkrakenuniq_amd.so, e.g. krakenuniq_amd/variants/abl), THREADS=n, REPEAT=n, GZ=1 (.gz input)
Prints one block per variant; kernel tables go to gpurun_out/<out_tag>_<label>_kernel_stats.csv."""
import csv, glob, os, resource, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_stat():
    """CFS throttling counters of this container's cgroup (v2: cpu.stat; v1: cpu/cpu.stat) + its quota"""
    out = {}
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(f):
            for line in open(f):
                k, v = line.split()
                out[k] = int(v)
            break
    return out


def quota():
    for f in ("/sys/fs/cgroup/cpu.max",):
        if os.path.exists(f):
            return open(f).read().strip()
    try:
        return open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read().strip() + " / " + open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
    except OSError:
        return "?"


def main():
    n_species, n_reads, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    variants = sys.argv[4:]
    import numpy as np, torch
    from krakenuniq_amd import synth_torch
    import bench
    tmp = "/dev/shm/ku_probe"
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    dev = torch.device("cuda:0")
    db = synth_torch.BenchDb(dev, n_species=n_species, genome_len=310_000, k=31, nt=13, seed=7)
    db.kmers = db.vals = None
    db.write_files(tmp)
    s, _, _, _ = db.sample_reads(n_reads, 150, seed=1)
    bench.write_fastq(f"{tmp}/reads.fq", s.view(n_reads, 151).cpu().numpy(), 150)
    del db, s
    torch.cuda.empty_cache()
    print(f"database + {n_reads} reads written in {time.time() - t0:.1f}s; cpu quota {quota()}; cpus visible {os.cpu_count()}", flush=True)
    gz_done = False
    for var in variants:
        label, _, kvs = var.partition(":")
        env = dict(os.environ, KU_CLI_TIMES="1", KU_REPORT_TIMES="1")
        report = prof = gz = False
        pmc = None
        threads, repeat, cpulist = "16", 1, None
        for kv in [x for x in kvs.split(",") if x]:
            k, v = kv.split("=", 1)
            if k == "REPORT": report = v != "0"
            elif k == "PROF": prof = v != "0"
            elif k == "PMC": pmc = v
            elif k == "GZ": gz = v != "0"
            elif k == "THREADS": threads = v
            elif k == "REPEAT": repeat = int(v)
            elif k == "CPULIST": cpulist = v.replace("+", ",")  # taskset -c (a+b for a,b)
            elif k == "LIB": env["LD_LIBRARY_PATH"] = os.path.join(ROOT, v) + ":" + env.get("LD_LIBRARY_PATH", "")
            else: env[k] = v
        reads = f"{tmp}/reads.fq"
        if gz:
            if not gz_done:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import write_one_stream_gz
                write_one_stream_gz.write(reads)
                gz_done = True
            reads += ".gz"
        for rep in range(repeat):
            for f in (f"{tmp}/report.tsv", f"{tmp}/out.tsv"):
                if os.path.exists(f): os.remove(f)
            cmd = [f"{ROOT}/krakenuniq_amd/bin/classify", "-d", f"{tmp}/database.kdb", "-i", f"{tmp}/database.idx", "-a", f"{tmp}/taxDB",
                   "-t", threads, "-o", f"{tmp}/out.tsv"] + (["-r", f"{tmp}/report.tsv"] if report else []) + [reads]
            if cpulist:
                cmd = ["taskset", "-c", cpulist] + cmd
            pdir = f"/tmp/prof_{tag}_{label}"
            if prof:
                shutil.rmtree(pdir, ignore_errors=True)
                cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", pdir, "--"] + cmd
            if pmc:
                shutil.rmtree(pdir, ignore_errors=True)
                cmd = ["rocprofv3", "--pmc", pmc, "--kernel-include-regex", "ku_classify_short_kernel", "--output-format", "csv", "-d", pdir, "--"] + cmd
            t = time.time()
            ru0, cs0 = resource.getrusage(resource.RUSAGE_CHILDREN), cpu_stat()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd="/tmp")
            wall = time.time() - t
            ru1, cs1 = resource.getrusage(resource.RUSAGE_CHILDREN), cpu_stat()
            cpu_line = (f"cpu user {ru1.ru_utime - ru0.ru_utime:.2f}s sys {ru1.ru_stime - ru0.ru_stime:.2f}s (whole process, load included); throttled "
                        f"{cs1.get('nr_throttled', 0) - cs0.get('nr_throttled', 0)} periods, {(cs1.get('throttled_usec', cs1.get('throttled_time', 0)) - cs0.get('throttled_usec', cs0.get('throttled_time', 0))) / 1e6:.3f}s")
            err = r.stderr.decode(errors="replace").replace("\r", "\n").split("\n")
            keep = [l.strip()[:520] for l in err if any(w in l for w in ("processed in", "stage busy", "Report finished", "ku_ctx_report:",
                                                                        "ku_classify_batch_rle over", "ku_classify_batch_rle, behind", "rle_first:", "stacks:", "trace:", "cpu seconds", "error", "Error"))]
            print(f"== {label} rep {rep} rc {r.returncode} wall {wall:.2f}s  {cpu_line}")
            print("\n".join("   " + l for l in keep), flush=True)
            if r.returncode != 0:
                print("\n".join(err[-12:]))
            if pmc:
                import json
                tot = {}
                for f in glob.glob(f"{pdir}/**/*counter_collection.csv", recursive=True):
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] != pmc or "ku_classify_short_kernel" not in row["Kernel_Name"]:
                            continue
                        kn = row["Kernel_Name"].split("(")[0].replace("void ", "")
                        e = tot.setdefault(kn, {"sum": 0.0, "launches": 0})
                        e["sum"] += float(row["Counter_Value"])
                        e["launches"] += 1
                json.dump({"counter": pmc, "reads": n_reads, "report": report, "kernels": tot}, open(f"{out_dir}/{tag}_{label}_pmc.json", "w"), indent=1)
                print("   " + pmc + ": " + "; ".join(f"{k}: {v['sum']:.4g} over {v['launches']} launches" for k, v in tot.items()))
                shutil.rmtree(pdir, ignore_errors=True)
            if prof:
                files = [f for f in glob.glob(f"{pdir}/**/*kernel_stats.csv", recursive=True) if "ku_" in open(f).read()]
                if files:
                    rows = list(csv.reader(open(files[-1])))
                    dst = f"{out_dir}/{tag}_{label}_kernel_stats.csv"
                    with open(dst, "w", newline="") as f:
                        w = csv.writer(f)
                        w.writerow(rows[0])
                        for row in rows[1:]:
                            if "ku_" in row[0] or float(row[4]) >= 1.0:
                                w.writerow([row[0][:160]] + row[1:])
                    tot = sum(float(row[2]) for row in rows[1:]) / 1e6
                    print(f"   kernels: {tot:.1f} ms in all")
                    for row in rows[1:12]:
                        print(f"   {row[0][:70]:70s} calls {row[1]:>6s} total_ms {float(row[2]) / 1e6:8.2f} avg_us {float(row[3]) / 1e3:9.1f}")
                shutil.rmtree(pdir, ignore_errors=True)
            sys.stdout.flush()
    if os.path.exists(f"{tmp}/report.tsv"):
        print("".join(open(f"{tmp}/report.tsv").readlines()[:5]))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
