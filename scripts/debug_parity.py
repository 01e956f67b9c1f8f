import os, sys, subprocess, tempfile, shutil, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from krakenuniq_amd import capi, synth_torch
from oracle import ku_oracle as ko
print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max', os.cpu_count(), len(os.sched_getaffinity(0)))
dev = torch.device('cuda:0')
db = synth_torch.BenchDb(dev, n_species=100, genome_len=100000, k=31, nt=13, seed=7)
n = 200000
d_seqs, d_off, d_len, _ = db.sample_reads(n, 150, seed=1)
ctx = capi.Ctx(0)
ctx.adopt_db(db.pairs.data_ptr(), db.n_pairs, db.offsets.data_ptr(), 31, 13, 2, keep=db)
ids_t, par_t = db.tax.arrays()
# NOTE: pairs get remapped in place by set_taxonomy -> export files BEFORE
tmp = tempfile.mkdtemp(prefix='ku_dbg_', dir='/dev/shm')
db.write_files(tmp)
pairs_host = db.pairs.cpu().numpy().view(np.uint8).reshape(-1).copy()
ctx.set_taxonomy(capi.Tax(ids=ids_t, parents=par_t))
d_taxa = torch.zeros(d_seqs.numel(), dtype=torch.int32, device=dev)
d_calls = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.classify_batch_device(d_seqs.data_ptr(), d_seqs.numel(), d_off.data_ptr(), d_len.data_ptr(), n, d_calls.data_ptr(), d_taxa.data_ptr(), max_read_len=150)
ctx.synchronize()
host = d_seqs.cpu().numpy()
off = np.arange(n, dtype=np.uint64) * 151
lens = np.full(n, 150, dtype=np.uint32)
calls = d_calls.cpu().numpy().view(np.uint32); taxa = d_taxa.cpu().numpy().view(np.uint32)
idsr = [f"r{i}" for i in range(n)]
gpu_text = capi.format_kraken(host, off, lens, idsr, 31, calls, taxa=taxa)
# oracle
offs = db.offsets.cpu().numpy().astype(np.uint64)
odb = ko.Db(pairs=pairs_host, key_ct=db.n_pairs, k=31, offsets=offs, nt=13)
run = ko.Run(odb, ko.Tax(ids=ids_t, parents=par_t), threads=16)
t0 = time.time(); res = run.classify_packed(host, off, lens, want_taxa=True); t_or = time.time() - t0
print('oracle 16 thr: %.2fs  %.1f Kreads/s' % (t_or, n / t_or / 1e3))
print('gpu vs oracle calls equal:', bool((res['calls'] == calls).all()))
with open(f'{tmp}/sample.fa', 'wb') as f:
    rows = host.reshape(n, 151)
    for i in range(n):
        f.write(b'>r%d\n' % i); f.write(rows[i].tobytes())
for t in (1, 8, 32, 64):
    cmd = [f'{ROOT}/oracle/_ref/classify', '-d', f'{tmp}/database.kdb', '-i', f'{tmp}/database.idx', '-a', f'{tmp}/taxDB', '-t', str(t), '-M', '-o', f'{tmp}/out{t}.tsv', f'{tmp}/sample.fa']
    t0 = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE); wall = time.time() - t0
    m = re.search(r'processed in ([\d.]+)s', r.stderr.decode(errors='replace'))
    print('ref -t', t, 'window', m.group(1) if m else r.stderr.decode()[-300:], 'wall %.1f' % wall)
ref = open(f'{tmp}/out8.tsv').read()
a, b = sorted(ref.split('\n')), sorted(gpu_text.split('\n'))
print('ref vs gpu equal:', a == b, len(a), len(b))
if a != b:
    sa, sb = set(a), set(b)
    print('only ref:', list(sa - sb)[:3]); print('only gpu:', list(sb - sa)[:3])
shutil.rmtree(tmp)
