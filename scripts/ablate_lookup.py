"""Within-process ablation of the lookup kernel on the bench workload (measurement aid).
python scripts/ablate_lookup.py [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from krakenuniq_amd import capi, synth_torch
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device('cuda:0')
db = synth_torch.BenchDb(dev, n_species=2000, genome_len=310_000, k=31, nt=13, seed=7)
db.kmers = db.vals = None
ctx = capi.Ctx(0)
ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), 31, 13, 2, keep=db)
ids_t, par_t = db.tax.arrays()
ctx.set_taxonomy(capi.Tax(ids=ids_t, parents=par_t))
d_seqs, d_off, d_len, _ = db.sample_reads(n_reads, 150, seed=1)
d_taxa = torch.zeros(d_seqs.numel(), dtype=torch.int32, device=dev)
d_calls = torch.zeros(n_reads, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
def t_lookup(flags=0, reps=3):
    ts = []
    for i in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.lookup_device(d_seqs.data_ptr(), d_seqs.numel(), d_taxa.data_ptr(), flags=flags, stream=s.cuda_stream); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts[1:])
for name, abl, flags in [("full", 0, 0), ("no counts (MODE 0)", 0, capi.KU_F_NO_COUNTS), ("no probe", 1, 0), ("no HLL", 2, 0),
                         ("no n_kmers", 4, 0), ("no HLL+n_kmers", 6, 0), ("no store", 8, 0), ("no probe, no counts", 1, capi.KU_F_NO_COUNTS),
                         ("no probe/HLL/n_kmers/store", 15, 0)]:
    os.environ["KU_ABLATE"] = str(abl)
    print(f"{name:34s} {t_lookup(flags):8.2f} ms", flush=True)
os.environ["KU_ABLATE"] = "0"
for name, abl in [("fused full", 0), ("fused no probe", 1), ("fused no HLL", 2), ("fused no n_kmers", 4), ("fused no store", 8),
                  ("fused no resolve", 32), ("fused no window/locus", 64), ("fused no probe/counts/store/resolve", 47),
                  ("fused scan only", 111)]:
    os.environ["KU_ABLATE"] = str(abl)
    ts = []
    for i in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.classify_batch_device(d_seqs.data_ptr(), d_seqs.numel(), d_off.data_ptr(), d_len.data_ptr(), n_reads, d_calls.data_ptr(), d_taxa.data_ptr(), max_read_len=150, stream=s.cuda_stream); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name:38s} {min(ts[1:]):8.2f} ms", flush=True)
os.environ["KU_ABLATE"] = "0"
ts = []
for i in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ctx.classify_batch_device(d_seqs.data_ptr(), d_seqs.numel(), d_off.data_ptr(), d_len.data_ptr(), n_reads, d_calls.data_ptr(), d_taxa.data_ptr(), max_read_len=150, stream=s.cuda_stream); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"{'classify_batch_device (fused if eligible)':34s} {min(ts[1:]):8.2f} ms", flush=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ctx.lookup_device(d_seqs.data_ptr(), d_seqs.numel(), d_taxa.data_ptr(), stream=s.cuda_stream)
a.record(); ctx.resolve_device(d_seqs.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n_reads, d_calls.data_ptr(), d_taxa.data_ptr(), max_read_len=150, stream=s.cuda_stream); b.record()
torch.cuda.synchronize(); print(f"{'resolve':34s} {a.elapsed_time(b):8.2f} ms")
