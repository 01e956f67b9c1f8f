"""On-device (torch) twin of synth.py for bench-scale synthetic databases.

Test / bench *data* infrastructure -- not part of the classification path.  Builds a
KrakenDB-format database (12-byte pairs grouped by minimizer bin, sorted inside a
bin, + uint64 bin offsets; reference src/krakendb.cpp:118-148, src/db_sort.cpp:80-128)
directly in HBM, because the 8 GB configuration of BASELINE.json (~620 M pairs)
would take far too long through numpy.  Everything is int64 arithmetic on 62-bit
k-mers; logical right shifts are emulated with masks.  tests/test_synth_torch.py
checks it against synth.py (which in turn is pinned to the reference's db_sort).
"""
from __future__ import annotations

import numpy as np
import torch

from . import synth

INDEX2_XOR_MASK = synth.INDEX2_XOR_MASK


def _s64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x: torch.Tensor, n: int) -> torch.Tensor:
    """logical shift right of int64 bit patterns"""
    return (x >> n) & ((1 << (64 - n)) - 1)


def splitmix64(x: torch.Tensor) -> torch.Tensor:
    z = x + _s64(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def revcomp(x: torch.Tensor, n: int) -> torch.Tensor:
    """reverse complement of n-mers (n <= 31) held in non-negative int64"""
    x = ((x >> 2) & 0x3333333333333333) | ((x & 0x3333333333333333) << 2)
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0F) | ((x & 0x0F0F0F0F0F0F0F0F) << 4)
    x = ((x >> 8) & 0x00FF00FF00FF00FF) | ((x & 0x00FF00FF00FF00FF) << 8)
    x = ((x >> 16) & 0x0000FFFF0000FFFF) | ((x & 0x0000FFFF0000FFFF) << 16)
    x = ((x >> 32) & 0xFFFFFFFF) | (x << 32)
    return ((~x) >> (64 - 2 * n)) & ((1 << (2 * n)) - 1)


def canonical(x: torch.Tensor, n: int) -> torch.Tensor:
    return torch.minimum(x, revcomp(x, n))


def bin_key(canon: torch.Tensor, k: int, nt: int, idx_type: int = 2, chunk: int = 1 << 26) -> torch.Tensor:
    """minimizer bin of canonical k-mers (reference src/krakendb.cpp:182-215)"""
    mask = (1 << (2 * nt)) - 1
    xor = (INDEX2_XOR_MASK if idx_type == 2 else 0) & mask
    out = torch.empty_like(canon)
    for s in range(0, canon.numel(), chunk):
        x = canon[s:s + chunk].clone()
        best = torch.full_like(x, (1 << 62))
        for _ in range(k - nt + 1):
            best = torch.minimum(best, canonical(x & mask, nt) ^ xor)
            x >>= 2
        out[s:s + chunk] = best
    return out


def kmers_of_rows(codes: torch.Tensor, k: int) -> torch.Tensor:
    """codes: uint8 [R, G] -> forward k-mers int64 [R, G-k+1]"""
    n = codes.shape[1] - k + 1
    c = codes.to(torch.int64)
    out = torch.zeros((codes.shape[0], n), dtype=torch.int64, device=codes.device)
    for j in range(k):
        out = (out << 2) | c[:, j:j + n]
    return out


def quantile_bin_bounds(sample_bins: torch.Tensor, n_bins: int, world_size: int) -> np.ndarray:
    """shard bounds from a sample of bin keys (every rank draws the same sample -> same plan)"""
    b = np.zeros(world_size + 1, dtype=np.uint64)
    b[-1] = n_bins
    if world_size > 1:
        q = torch.quantile(sample_bins.to(torch.float64), torch.linspace(0, 1, world_size + 1,
                                                                         dtype=torch.float64)[1:-1].to(sample_bins.device))
        b[1:-1] = np.ceil(q.cpu().numpy()).astype(np.uint64)
    return b


def expand_runs(runs: torch.Tensor, run_off: torch.Tensor, run_cnt: torch.Tensor, n_kmers: torch.Tensor) -> torch.Tensor:
    """run-length encoded per-k-mer codes (ku_classify_batch_device_rle: runs int32 [cap, 2] = {code, first k-mer}, per read
    run_off int64 / run_cnt int32) -> the codes of all k-mers, read after read (n_kmers int64 per read), as one int32
    tensor.  A checker for tests and bench.py: plain tensor arithmetic, no library code."""
    rc = run_cnt.to(torch.int64)
    total = int(rc.sum().item())
    first = torch.cumsum(rc, 0) - rc                      # index of each read's first run in read order
    read_of = torch.repeat_interleave(torch.arange(len(rc), device=rc.device), rc)
    idx = run_off.to(torch.int64)[read_of] + (torch.arange(total, device=rc.device) - first[read_of])
    code, start = runs[idx, 0], runs[idx, 1].to(torch.int64)
    end = torch.empty_like(start)
    end[:-1] = start[1:]
    last = first + rc - 1                                  # the last run of a read ends with the read's k-mers
    has = rc > 0
    end[last[has]] = n_kmers.to(torch.int64)[has]
    assert bool((end > start).all()), "runs of a read must start at increasing k-mers"
    return torch.repeat_interleave(code, end - start)


class BenchDb:
    """Synthetic taxonomy + genomes + (optionally sharded) database resident in HBM."""

    def __init__(self, device, n_species=2000, genome_len=310_000, k=31, nt=13, seed=7, shared_frac=0.1,
                 sub_rate=0.03, bin_lo=0, bin_hi=None, species_chunk=64):
        self.device, self.k, self.nt, self.seed = device, k, nt, seed
        self.n_bins = 4 ** nt
        self.bin_lo, self.bin_hi = bin_lo, self.n_bins if bin_hi is None else bin_hi
        rng = np.random.default_rng(seed)
        lv = tuple(max(1, int(round(n_species * f))) for f in (0.01, 0.03, 0.09, 0.27))
        self.tax = synth.random_taxonomy(n_species, rng, levels=lv)
        species = self.tax.species
        # ancestor table: anc[l][i] = taxid of the l-th ancestor of species i (l = 0 -> itself), padded with root
        depth = max(len(self.tax.path(t)) for t in species)
        anc = np.ones((depth, n_species), dtype=np.int64)
        for i, t in enumerate(species):
            p = self.tax.path(t)
            anc[:len(p), i] = p
        self.anc = torch.from_numpy(anc).to(device)
        genus_ids = {g: i for i, g in enumerate(sorted({self.tax.parent[t] for t in species}))}
        self.genus_of = torch.tensor([genus_ids[self.tax.parent[t]] for t in species], dtype=torch.int64, device=device)
        self.n_species, self.G = n_species, genome_len
        self.shared_len = int(genome_len * shared_frac)
        self.sub_rate = sub_rate
        self.genomes = torch.empty((n_species, genome_len), dtype=torch.uint8, device=device)
        parts_k, parts_v = [], []
        for s0 in range(0, n_species, species_chunk):
            idx = torch.arange(s0, min(s0 + species_chunk, n_species), device=device)
            codes = self._genome_codes(idx)
            self.genomes[idx] = codes
            km = canonical(kmers_of_rows(codes, k).reshape(-1), k)
            sp = idx.repeat_interleave(genome_len - k + 1)
            if self.bin_lo != 0 or self.bin_hi != self.n_bins:  # sharded build: keep only the bins this rank owns
                b = bin_key(km, k, nt)
                keep = (b >= self.bin_lo) & (b < self.bin_hi)
                km, sp = km[keep], sp[keep]
            parts_k.append(km)
            parts_v.append(sp)
        # torch sorts / scans stop at 2^31 elements: large builds (the 36 GB shard of the 300 GB layout is 3 G pairs) go
        # through buckets of the k-mer space (dedup + LCA per bucket) and ranges of the bin space (bin order per range);
        # with one bucket and one range this is the plain  sort by k-mer -> dedup -> stable sort by bin
        total = sum(int(t.numel()) for t in parts_k)
        n_buckets = 1 if total < (1 << 30) else 16
        dedup = []
        for bkt in range(n_buckets):
            if n_buckets == 1:
                kb, vb = torch.cat(parts_k), torch.cat(parts_v)
            else:
                sel = [(t >> (2 * k - 4)) == bkt for t in parts_k]
                kb = torch.cat([t[m] for t, m in zip(parts_k, sel)])
                vb = torch.cat([t[m] for t, m in zip(parts_v, sel)])
                del sel
            dedup.append(self._dedup_lca(kb, vb))
            del kb, vb
        del parts_k, parts_v
        n_unique = sum(int(t[0].numel()) for t in dedup)
        n_ranges = 1 if n_unique < (1 << 30) else -(-n_unique // (1 << 28))
        span = self.bin_hi - self.bin_lo
        bounds = [self.bin_lo + span * r // n_ranges for r in range(n_ranges + 1)]
        by_range = [[] for _ in range(n_ranges)]
        for kb, vb in dedup:
            b = bin_key(kb, k, nt)
            if n_ranges == 1:
                by_range[0].append((kb, vb, b))
            else:
                for r in range(n_ranges):
                    m = (b >= bounds[r]) & (b < bounds[r + 1])
                    by_range[r].append((kb[m], vb[m], b[m]))
            del b
        del dedup
        self.offsets = torch.zeros(span + 1, dtype=torch.int64, device=device)
        out_k, out_v, base = [], [], 0
        for r in range(n_ranges):
            kr = torch.cat([t[0] for t in by_range[r]])
            vr = torch.cat([t[1] for t in by_range[r]])
            br = torch.cat([t[2] for t in by_range[r]])
            by_range[r] = None
            order = torch.sort(br, stable=True).indices  # k-mers are ascending already -> (bin, kmer) order
            out_k.append(kr[order])
            out_v.append(vr[order])
            counts = torch.bincount(br - bounds[r], minlength=bounds[r + 1] - bounds[r])
            seg = self.offsets[bounds[r] - self.bin_lo + 1:bounds[r + 1] - self.bin_lo + 1]
            torch.cumsum(counts, 0, out=seg)
            seg += base
            base += int(kr.numel())
            del kr, vr, br, order, counts
        km = out_k[0] if n_ranges == 1 else torch.cat(out_k)
        vals = out_v[0] if n_ranges == 1 else torch.cat(out_v)
        del out_k, out_v
        self.n_pairs = km.numel()
        pairs = torch.empty((self.n_pairs, 3), dtype=torch.int32, device=device)
        pairs[:, 0] = ((km << 32) >> 32).to(torch.int32)  # low dword, sign-extended so the cast is exact
        pairs[:, 1] = (km >> 32).to(torch.int32)
        pairs[:, 2] = ((vals << 32) >> 32).to(torch.int32)
        self.pairs = pairs
        self.kmers = km  # kept for tests / sampling; callers may delete
        self.vals = vals

    # ---- genomes
    def _genome_codes(self, sp_idx: torch.Tensor) -> torch.Tensor:
        """uint8 codes [len(sp_idx), G]; the first shared_len bases derive from the genus sequence with
        sub_rate substitutions (so sibling species share k-mers -> LCA values above species)."""
        dev = self.device
        j = torch.arange(self.G, device=dev, dtype=torch.int64)[None, :]
        s = sp_idx[:, None]

        def bases(ent, salt):
            key = (self.seed + salt) ^ ((ent + 1) * _s64(0x9E3779B97F4A7C15)) ^ ((j >> 5) * _s64(0xD1342543DE82EF95))
            return (splitmix64(key) >> ((j & 31) * 2)) & 3

        own = bases(s, 0)
        if self.shared_len > 0:
            gj = j[:, :self.shared_len]
            g = self.genus_of[sp_idx][:, None]
            key = (self.seed + 1001) ^ ((g + 1) * _s64(0x9E3779B97F4A7C15)) ^ ((gj >> 5) * _s64(0xD1342543DE82EF95))
            shared = (splitmix64(key) >> ((gj & 31) * 2)) & 3
            r = splitmix64((s * 0x100000000 + gj) ^ (self.seed * 7919))
            mut = (_lsr(r, 11).to(torch.float64) * (1.0 / (1 << 53))) < self.sub_rate
            shared = torch.where(mut, (shared + 1 + (_lsr(r, 2) & 0xFF) % 3) & 3, shared)
            own[:, :self.shared_len] = shared
        return own.to(torch.uint8)

    def _dedup_lca(self, km: torch.Tensor, sp: torch.Tensor):
        """sort by k-mer; duplicate k-mers get the LCA of their species (set_lcas semantics)"""
        km, order = torch.sort(km)
        sp = sp[order]
        del order
        first = torch.ones_like(km, dtype=torch.bool)
        first[1:] = km[1:] != km[:-1]
        run = torch.cumsum(first, 0) - 1
        n_runs = int(run[-1].item()) + 1 if km.numel() else 0
        vals = self.anc[0][sp[first]]  # default: the (single) species' taxid
        dup = ~first
        if bool(dup.any()):
            in_multi = torch.zeros(n_runs, dtype=torch.bool, device=km.device)
            in_multi[run[dup]] = True
            sel = in_multi[run]
            r_sel, s_sel = run[sel], sp[sel]
            done = torch.zeros(n_runs, dtype=torch.bool, device=km.device)
            for lvl in range(self.anc.shape[0]):
                a = self.anc[lvl][s_sel]
                lo = torch.full((n_runs,), (1 << 62), dtype=torch.int64, device=km.device).scatter_reduce(0, r_sel, a, "amin")
                hi = torch.zeros(n_runs, dtype=torch.int64, device=km.device).scatter_reduce(0, r_sel, a, "amax")
                agree = in_multi & ~done & (lo == hi)
                vals = torch.where(agree, lo, vals)
                done |= agree
            vals = torch.where(in_multi & ~done, torch.ones_like(vals), vals)  # nothing in common below the root
        return km[first], vals

    # ---- reads
    def sample_reads(self, n_reads: int, read_len: int = 150, seed: int = 1, frac_random: float = 0.2,
                     sub_rate: float = 0.01, n_rate: float = 0.001, chunk: int = 1 << 20, species=None):
        """ASCII read buffer [n_reads, read_len + 1] (last column '\\n'), seq_off int64, seq_len int32, source taxid.
        species: optional int64 tensor of species indices the reads are drawn from (default: all, uniformly)."""
        dev = self.device
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        buf = torch.empty((n_reads, read_len + 1), dtype=torch.uint8, device=dev)
        src = torch.empty(n_reads, dtype=torch.int64, device=dev)
        ascii_tab = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
        ar = torch.arange(read_len, device=dev)
        for s0 in range(0, n_reads, chunk):
            n = min(chunk, n_reads - s0)
            sp = torch.randint(0, self.n_species, (n,), generator=g, device=dev)
            if species is not None:
                sp = species[sp % species.numel()]
            st = torch.randint(0, self.G - read_len + 1, (n,), generator=g, device=dev)
            codes = self.genomes[sp[:, None], st[:, None] + ar[None, :]].to(torch.int64)
            rc = torch.rand(n, generator=g, device=dev) < 0.5
            codes = torch.where(rc[:, None], 3 - codes.flip(1), codes)
            sub = torch.rand((n, read_len), generator=g, device=dev) < sub_rate
            codes = torch.where(sub, (codes + torch.randint(1, 4, (n, read_len), generator=g, device=dev)) & 3, codes)
            rnd = torch.rand(n, generator=g, device=dev) < frac_random
            codes = torch.where(rnd[:, None], torch.randint(0, 4, (n, read_len), generator=g, device=dev), codes)
            a = ascii_tab[codes]
            isn = torch.rand((n, read_len), generator=g, device=dev) < n_rate
            a = torch.where(isn, torch.full_like(a, 78), a)
            buf[s0:s0 + n, :read_len] = a
            src[s0:s0 + n] = torch.where(rnd, torch.zeros_like(sp), self.anc[0][sp])
        buf[:, read_len] = 10
        seq_off = torch.arange(n_reads, device=dev, dtype=torch.int64) * (read_len + 1)
        seq_len = torch.full((n_reads,), read_len, dtype=torch.int32, device=dev)
        return buf.reshape(-1), seq_off, seq_len, src

    def sample_pairs(self, n_pairs: int, read_len: int = 150, seed: int = 1, frag_lo: int = 200, frag_hi: int = 600,
                     frac_random: float = 0.2, sub_rate: float = 0.01, n_rate: float = 0.001, chunk: int = 1 << 20):
        """mate pairs as a sequencer makes them: both mates read inwards from the two ends of ONE fragment of a
        genome (mate 2 is the reverse complement of the fragment's far end).  Returns the merged records
        mate1 + 'N' + mate2 (scripts/read_merger.pl:187-191) as an ASCII buffer [n_pairs, 2 * read_len + 2]
        (last column '\\n'), seq_off int64, seq_len int32."""
        dev = self.device
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        L = read_len
        W = 2 * L + 2
        buf = torch.empty((n_pairs, W), dtype=torch.uint8, device=dev)
        ascii_tab = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
        ar = torch.arange(L, device=dev)
        frag_lo = max(frag_lo, L)
        for s0 in range(0, n_pairs, chunk):
            n = min(chunk, n_pairs - s0)
            sp = torch.randint(0, self.n_species, (n,), generator=g, device=dev)
            fl = torch.randint(frag_lo, frag_hi + 1, (n,), generator=g, device=dev)
            st = (torch.rand(n, generator=g, device=dev) * (self.G - fl).to(torch.float32)).to(torch.int64)
            m1 = self.genomes[sp[:, None], st[:, None] + ar[None, :]].to(torch.int64)
            m2 = 3 - self.genomes[sp[:, None], (st + fl)[:, None] - 1 - ar[None, :]].to(torch.int64)  # far end, reverse complemented
            flip = torch.rand(n, generator=g, device=dev) < 0.5  # the fragment came from the other strand
            a = torch.where(flip[:, None], m2, m1)
            b = torch.where(flip[:, None], m1, m2)
            both = torch.cat([a, b], 1)
            sub = torch.rand((n, 2 * L), generator=g, device=dev) < sub_rate
            both = torch.where(sub, (both + torch.randint(1, 4, (n, 2 * L), generator=g, device=dev)) & 3, both)
            rnd = torch.rand(n, generator=g, device=dev) < frac_random
            both = torch.where(rnd[:, None], torch.randint(0, 4, (n, 2 * L), generator=g, device=dev), both)
            t = ascii_tab[both]
            isn = torch.rand((n, 2 * L), generator=g, device=dev) < n_rate
            t = torch.where(isn, torch.full_like(t, 78), t)
            buf[s0:s0 + n, :L] = t[:, :L]
            buf[s0:s0 + n, L + 1:2 * L + 1] = t[:, L:]
        buf[:, L] = 78
        buf[:, 2 * L + 1] = 10
        seq_off = torch.arange(n_pairs, device=dev, dtype=torch.int64) * W
        seq_len = torch.full((n_pairs,), 2 * L + 1, dtype=torch.int32, device=dev)
        return buf.reshape(-1), seq_off, seq_len

    # ---- export (CPU baseline / parity sample)
    def write_files(self, dirname: str, slot_taxid: torch.Tensor = None):
        """database.kdb / database.idx / taxDB in the reference's on-disk format (full range only).
        slot_taxid: pass the context's slot table when the resident values were already remapped to slot ids."""
        import os
        assert self.bin_lo == 0 and self.bin_hi == self.n_bins
        os.makedirs(dirname, exist_ok=True)
        with open(os.path.join(dirname, "database.kdb"), "wb") as f:
            f.write(synth.kdb_header(self.k, self.n_pairs))
            step = 1 << 26
            for s in range(0, self.n_pairs, step):
                p = self.pairs[s:s + step]
                if slot_taxid is not None:
                    p = p.clone()
                    p[:, 2] = ((slot_taxid[p[:, 2].to(torch.int64)] << 32) >> 32).to(torch.int32)
                f.write(p.cpu().numpy().tobytes())
        with open(os.path.join(dirname, "database.idx"), "wb") as f:
            f.write(b"KRAKIX2" + bytes([self.nt]))
            f.write(self.offsets.cpu().numpy().astype("<u8").tobytes())
        self.tax.write(os.path.join(dirname, "taxDB"))
