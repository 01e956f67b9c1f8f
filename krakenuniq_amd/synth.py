"""Deterministic synthetic KrakenDB / taxonomy / read generator (numpy).

Test and bench *data* infrastructure -- not part of the classification path.
The reference ships no fixtures and needs jellyfish-1 to build a database
(SURVEY.md 4, 8c), so this module writes the reference's on-disk formats
directly:

* ``database.kdb``: Jellyfish-1 list header + ``key_ct`` packed (8-byte LE k-mer,
  4-byte LE taxid) pairs, grouped by minimizer bin, sorted inside a bin
  (reference src/krakendb.cpp:60-78,177; src/db_sort.cpp:80-128).
* ``database.idx``: ``KRAKIX2`` + nt byte + ``uint64[4^nt + 1]`` bin offsets
  (reference src/krakendb.cpp:118-148).
* ``taxDB``: ``id \\t parent \\t name \\t rank`` (reference src/taxdb.hpp:563-605).

Everything is vectorised numpy on uint64; the (much larger) bench database is
built with the torch twin in ``krakenuniq_amd/synth_torch.py``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

INDEX2_XOR_MASK = 0xE37E28C4271B5A2D  # reference src/krakendb.cpp:45
U64 = np.uint64
_MASK64 = (1 << 64) - 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        z = (x.astype(U64) + U64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
        return z ^ (z >> U64(31))


def procedural_genome(seed: int, g: int, length: int) -> np.ndarray:
    """2-bit codes (A=0,C=1,G=2,T=3) of genome ``g``: base i = 2 bits of
    splitmix64(seed ^ g*phi ^ (i >> 5))."""
    i = np.arange(length, dtype=U64)
    with np.errstate(over="ignore"):
        key = U64(seed) ^ (U64(g) * U64(0x9E3779B97F4A7C15)) ^ (i >> U64(5)) * U64(0xD1342543DE82EF95)
    h = splitmix64(key)
    return ((h >> ((i & U64(31)) * U64(2))) & U64(3)).astype(np.uint8)


def mutate(codes: np.ndarray, rate: float, rng: np.random.Generator) -> np.ndarray:
    out = codes.copy()
    hit = rng.random(len(codes)) < rate
    out[hit] = (out[hit] + rng.integers(1, 4, hit.sum(), dtype=np.uint8)) & 3
    return out


_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def codes_to_ascii(codes: np.ndarray) -> bytes:
    return _ASCII[codes].tobytes()


def ascii_to_codes(seq: bytes) -> np.ndarray:
    """inverse of codes_to_ascii for upper-case ACGT text"""
    lut = np.full(256, 255, dtype=np.uint8)
    lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
    codes = lut[np.frombuffer(seq, dtype=np.uint8)]
    assert not (codes == 255).any(), "ACGT only"
    return codes


def revcomp_codes(codes: np.ndarray) -> np.ndarray:
    return (3 - codes[::-1]).astype(np.uint8)


# --------------------------------------------------------------------------- k-mers
def kmers_forward(codes: np.ndarray, k: int) -> np.ndarray:
    """All forward k-mers (oldest base in the high bits), uint64[len-k+1]."""
    n = len(codes) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=U64)
    out = np.zeros(n, dtype=U64)
    c = codes.astype(U64)
    for j in range(k):
        out = (out << U64(2)) | c[j:j + n]
    return out


def revcomp_kmer(x: np.ndarray, n: int) -> np.ndarray:
    """Reverse complement of n-mers packed in uint64 (reference src/krakendb.cpp:218-225)."""
    x = x.astype(U64)
    x = ((x >> U64(2)) & U64(0x3333333333333333)) | ((x & U64(0x3333333333333333)) << U64(2))
    x = ((x >> U64(4)) & U64(0x0F0F0F0F0F0F0F0F)) | ((x & U64(0x0F0F0F0F0F0F0F0F)) << U64(4))
    x = ((x >> U64(8)) & U64(0x00FF00FF00FF00FF)) | ((x & U64(0x00FF00FF00FF00FF)) << U64(8))
    x = ((x >> U64(16)) & U64(0x0000FFFF0000FFFF)) | ((x & U64(0x0000FFFF0000FFFF)) << U64(16))
    x = (x >> U64(32)) | (x << U64(32))
    return (~x) >> U64(64 - 2 * n)


def canonical(x: np.ndarray, n: int) -> np.ndarray:
    return np.minimum(x, revcomp_kmer(x, n))


def bin_key(canon: np.ndarray, k: int, nt: int, idx_type: int = 2) -> np.ndarray:
    """Minimizer bin key of canonical k-mers (reference src/krakendb.cpp:182-215)."""
    mask = U64((1 << (2 * nt)) - 1)
    xor = U64((INDEX2_XOR_MASK if idx_type == 2 else 0) & ((1 << (2 * nt)) - 1))
    best = np.full(canon.shape, np.iinfo(np.uint64).max, dtype=U64)
    x = canon.astype(U64)
    for _ in range(k - nt + 1):
        best = np.minimum(best, xor ^ canonical(x & mask, nt))
        x = x >> U64(2)
    return best


# --------------------------------------------------------------------------- taxonomy
@dataclass
class Taxonomy:
    ids: list = field(default_factory=list)
    parent: dict = field(default_factory=dict)
    name: dict = field(default_factory=dict)
    rank: dict = field(default_factory=dict)

    def add(self, tid: int, parent: int, name: str, rank: str) -> None:
        self.ids.append(tid)
        self.parent[tid] = parent
        self.name[tid] = name
        self.rank[tid] = rank

    def path(self, t: int) -> list:
        p = []
        while t and t not in p:
            p.append(t)
            nxt = self.parent.get(t, 0)
            if nxt == t:
                break
            t = nxt
        return p

    def lca(self, a: int, b: int) -> int:
        if a == 0 or b == 0:
            return a or b
        pa = self.path(a)
        for x in self.path(b):
            if x in pa:
                return x
        return 1

    def write(self, path: str) -> None:
        with open(path, "w") as f:
            for t in self.ids:
                f.write(f"{t}\t{self.parent[t]}\t{self.name[t]}\t{self.rank[t]}\n")

    def arrays(self):
        ids = np.array(self.ids, dtype=np.uint32)
        par = np.array([self.parent[t] for t in self.ids], dtype=np.uint32)
        return ids, par


def small_taxonomy() -> Taxonomy:
    """root(1) -> genus 2 -> species 4,5 ; genus 3 -> species 6 -> sequence 1000000001."""
    t = Taxonomy()
    t.add(1, 1, "root", "root")
    t.add(2, 1, "G2", "genus")
    t.add(3, 1, "G3", "genus")
    t.add(4, 2, "S4", "species")
    t.add(5, 2, "S5", "species")
    t.add(6, 3, "S6", "species")
    t.add(1000000001, 6, "S6 plasmid pX", "sequence")
    return t


def random_taxonomy(n_species: int, rng: np.random.Generator, levels=(20, 60, 180, 540)) -> Taxonomy:
    """Random 5-level tree root -> phyla -> ... -> species with sparse taxids."""
    t = Taxonomy()
    t.add(1, 1, "root", "root")
    ranks = ["phylum", "class", "order", "genus"]
    prev = [1]
    next_id = 2
    for lvl, width in enumerate(levels):
        width = max(1, min(width, n_species))
        cur = []
        rank = ranks[lvl] if lvl < len(ranks) else f"clade{lvl}"
        for i in range(width):
            tid = next_id
            next_id += int(rng.integers(1, 40))
            t.add(tid, prev[i % len(prev)] if i < len(prev) else prev[int(rng.integers(0, len(prev)))],
                  f"{rank}_{tid}", rank)
            cur.append(tid)
        prev = cur
    species = []
    for i in range(n_species):
        tid = next_id if i % 7 else 1000000001 + i  # some pseudo-taxids >= 1e9 (set_lcas.cpp:51)
        next_id += int(rng.integers(1, 40))
        t.add(tid, prev[i % len(prev)], f"species_{tid}", "species")
        species.append(tid)
    t.species = species  # type: ignore[attr-defined]
    return t


# --------------------------------------------------------------------------- DB writers
def kdb_header(k: int, key_ct: int) -> bytes:
    """Jellyfish-1 list header as KrakenDB reads it (reference src/krakendb.cpp:70-72,177)."""
    key_bits = 2 * k
    size = 72 + 2 * (4 + 8 * key_bits)
    h = bytearray(size)
    h[0:8] = b"JFLISTDN"
    h[8:16] = int(key_bits).to_bytes(8, "little")
    h[16:24] = (4).to_bytes(8, "little")
    h[48:56] = int(key_ct).to_bytes(8, "little")
    return bytes(h)


PAIR_DT = np.dtype([("key", "<u8"), ("val", "<u4")])  # 12-byte packed pair (k = 31)


def pack_pairs(kmers: np.ndarray, vals: np.ndarray) -> np.ndarray:
    p = np.empty(len(kmers), dtype=PAIR_DT)
    p["key"] = kmers
    p["val"] = vals
    return p


def write_jdb(path: str, kmers: np.ndarray, vals: np.ndarray, k: int) -> None:
    """Unsorted Jellyfish-style file: input for the reference's db_sort."""
    assert (2 * k + 7) // 8 == 8, "writer handles key_len == 8 (k in 29..32)"
    with open(path, "wb") as f:
        f.write(kdb_header(k, len(kmers)))
        f.write(pack_pairs(kmers, vals).tobytes())


def sort_db(kmers: np.ndarray, vals: np.ndarray, k: int, nt: int):
    """Group by minimizer bin, sort by k-mer inside the bin; returns (kmers, vals, offsets)."""
    bk = bin_key(kmers, k, nt)
    order = np.lexsort((kmers, bk))
    counts = np.bincount(bk.astype(np.int64), minlength=4 ** nt)
    offsets = np.zeros(4 ** nt + 1, dtype=np.uint64)
    np.cumsum(counts, out=offsets[1:])
    return kmers[order], vals[order], offsets


def write_db(dirname: str, kmers: np.ndarray, vals: np.ndarray, offsets: np.ndarray, k: int, nt: int) -> None:
    os.makedirs(dirname, exist_ok=True)
    with open(os.path.join(dirname, "database.kdb"), "wb") as f:
        f.write(kdb_header(k, len(kmers)))
        f.write(pack_pairs(kmers, vals).tobytes())
    with open(os.path.join(dirname, "database.idx"), "wb") as f:
        f.write(b"KRAKIX2" + bytes([nt]))
        f.write(offsets.astype("<u8").tobytes())


def read_db(dirname: str, kdb: str = "database.kdb", idx: str = "database.idx"):
    """Parse database.kdb/.idx back into (kmers, vals, offsets, k, nt, idx_type)."""
    raw = np.fromfile(os.path.join(dirname, kdb), dtype=np.uint8)
    assert raw[:8].tobytes() == b"JFLISTDN"
    key_bits = int.from_bytes(raw[8:16].tobytes(), "little")
    key_ct = int.from_bytes(raw[48:56].tobytes(), "little")
    hdr = 72 + 2 * (4 + 8 * key_bits)
    pairs = raw[hdr:hdr + 12 * key_ct].view(PAIR_DT)
    idx = np.fromfile(os.path.join(dirname, idx), dtype=np.uint8)
    magic = idx[:7].tobytes()
    nt = int(idx[7])
    offsets = idx[8:].view("<u8")
    return (pairs["key"].copy(), pairs["val"].copy(), offsets.copy(), key_bits // 2, nt,
            1 if magic == b"KRAKIDX" else 2)


def lca_database(genomes: dict, tax: Taxonomy, k: int):
    """Canonical k-mer -> LCA of all taxa whose genome contains it (set_lcas semantics)."""
    db: dict = {}
    for tid, codes in genomes.items():
        for km in np.unique(canonical(kmers_forward(codes, k), k)).tolist():
            db[km] = tax.lca(db[km], tid) if km in db else tid
    kmers = np.fromiter(db.keys(), dtype=U64, count=len(db))
    vals = np.fromiter(db.values(), dtype=np.uint32, count=len(db))
    return kmers, vals


# --------------------------------------------------------------------------- reads
def sample_reads(genomes: dict, n_reads: int, read_len: int, rng: np.random.Generator,
                 frac_random: float = 0.2, sub_rate: float = 0.01, n_rate: float = 0.001):
    """80 % sampled from genomes (50 % rev-comp, 1 % subs, 0.1 % N), 20 % uniform random.

    Returns (list of ascii bytes, list of source taxid (0 = random))."""
    tids = list(genomes.keys())
    reads, src = [], []
    for _ in range(n_reads):
        if rng.random() < frac_random:
            codes = rng.integers(0, 4, read_len, dtype=np.uint8)
            tid = 0
        else:
            tid = tids[int(rng.integers(0, len(tids)))]
            g = genomes[tid]
            s = int(rng.integers(0, max(1, len(g) - read_len + 1)))
            codes = g[s:s + read_len]
            if rng.random() < 0.5:
                codes = revcomp_codes(codes)
            codes = mutate(codes, sub_rate, rng)
        a = bytearray(codes_to_ascii(codes))
        for p in np.nonzero(rng.random(len(a)) < n_rate)[0]:
            a[int(p)] = ord("N")
        reads.append(bytes(a))
        src.append(tid)
    return reads, src


def write_fastq(path: str, reads, ids=None) -> None:
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            rid = ids[i] if ids else f"r{i}"
            f.write(b"@" + rid.encode() + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")


def write_fasta(path: str, reads, ids=None, width: int = 0) -> None:
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            rid = ids[i] if ids else f"r{i}"
            f.write(b">" + rid.encode() + b"\n")
            if width:
                for s in range(0, len(r), width):
                    f.write(r[s:s + width] + b"\n")
                if len(r) == 0:
                    f.write(b"\n")
            else:
                f.write(r + b"\n")


def read_seqfile(path: str):
    """Minimal FASTA/FASTQ parser with the reference's record semantics
    (src/seqreader.cpp:26-133): returns (ids, seqs)."""
    data = open(path, "rb").read()
    ids, seqs = [], []
    lines = data.split(b"\n")
    if data[:1] == b"@":
        i = 0
        while i + 3 < len(lines) + 1 and i < len(lines):
            h = lines[i]
            if not h or h[:1] != b"@":
                break
            ids.append(h[1:].split()[0].decode() if h[1:].split() else "")
            seqs.append(lines[i + 1] if i + 1 < len(lines) else b"")
            i += 4
    else:
        cur = None
        for ln in lines:
            if ln[:1] == b">":
                if cur is not None:
                    seqs.append(b"".join(cur))
                ids.append(ln[1:].split()[0].decode() if ln[1:].split() else "")
                cur = []
            elif cur is not None:
                cur.append(ln)
        if cur is not None:
            seqs.append(b"".join(cur))
    return ids, seqs
