"""PCIe-inclusive rate of the host-buffer entry point ku_classify_batch (H2D + kernels + D2H of the per-k-mer codes),
bench DB, pageable numpy buffers.  python scripts/pcie_rate.py [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from krakenuniq_amd import capi, synth_torch
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device('cuda:0')
db = synth_torch.BenchDb(dev, n_species=2000, genome_len=310_000, k=31, nt=13, seed=7)
db.kmers = db.vals = None
ctx = capi.Ctx(0)
ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), 31, 13, 2, keep=db)
ids_t, par_t = db.tax.arrays()
ctx.set_taxonomy(capi.Tax(ids=ids_t, parents=par_t))
d_seqs, d_off, d_len, _ = db.sample_reads(n_reads, 150, seed=1)
buf = d_seqs.cpu().numpy(); off = d_off.cpu().numpy().astype(np.uint64); lens = d_len.cpu().numpy().astype(np.uint32)
for want in (True, False):
    ctx.classify_batch(buf, off, lens, want_taxa=want)
    t = time.time(); ctx.classify_batch(buf, off, lens, want_taxa=want); dt = time.time() - t
    print(f"ku_classify_batch host buffers, per-k-mer output {'on' if want else 'off'}: {n_reads / dt / 1e6:.1f} Mreads/s ({dt * 1e3:.1f} ms for {n_reads} reads)")
ctx.classify_batch_rle(buf, off, lens)
t = time.time(); r = ctx.classify_batch_rle(buf, off, lens); dt = time.time() - t
print(f"ku_classify_batch_rle + ku_fetch_runs host buffers: {n_reads / dt / 1e6:.1f} Mreads/s ({dt * 1e3:.1f} ms, {len(r['runs']) / n_reads:.2f} runs/read)")
