"""ctypes binding of oracle/libku_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package (krakenuniq_amd/)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libku_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("ku_oracle.c", "ku_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    sig = {
        "ko_scan": (C.c_size_t, [C.c_char_p, C.c_size_t, C.c_int, u64p, u8p]),
        "ko_revcomp": (C.c_uint64, [C.c_uint64, C.c_int]),
        "ko_canonical": (C.c_uint64, [C.c_uint64, C.c_int]),
        "ko_bin_key": (C.c_uint64, [C.c_uint64, C.c_int, C.c_int, C.c_int]),
        "ko_hash": (C.c_uint64, [C.c_uint64]),
        "ko_db_open": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
        "ko_db_wrap": (C.c_void_p, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int]),
        "ko_db_close": (None, [C.c_void_p]),
        "ko_db_key_ct": (C.c_uint64, [C.c_void_p]),
        "ko_db_k": (C.c_int, [C.c_void_p]),
        "ko_db_nt": (C.c_int, [C.c_void_p]),
        "ko_db_idx_type": (C.c_int, [C.c_void_p]),
        "ko_db_query": (C.c_int64, [C.c_void_p, C.c_uint64]),
        "ko_db_count_taxons": (C.c_size_t, [C.c_void_p, u32p, u64p]),
        "ko_tax_load": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_size_t]),
        "ko_tax_from_arrays": (C.c_void_p, [u32p, u32p, C.c_size_t]),
        "ko_tax_free": (None, [C.c_void_p]),
        "ko_tax_size": (C.c_size_t, [C.c_void_p]),
        "ko_tax_parent": (C.c_uint32, [C.c_void_p, C.c_uint32]),
        "ko_lca": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32]),
        "ko_resolve_tree": (C.c_uint32, [C.c_void_p, u32p, u32p, C.c_size_t]),
        "ko_hll_new": (C.c_void_p, [C.c_int, C.c_int]),
        "ko_hll_free": (None, [C.c_void_p]),
        "ko_hll_insert": (None, [C.c_void_p, C.c_uint64]),
        "ko_hll_merge": (None, [C.c_void_p, C.c_void_p]),
        "ko_hll_cardinality": (C.c_uint64, [C.c_void_p, C.c_int]),
        "ko_hll_n_observed": (C.c_uint64, [C.c_void_p]),
        "ko_hll_is_sparse": (C.c_int, [C.c_void_p]),
        "ko_hll_sparse_size": (C.c_size_t, [C.c_void_p]),
        "ko_hll_sparse_dump": (C.c_size_t, [C.c_void_p, u32p, C.c_size_t]),
        "ko_hll_registers": (None, [C.c_void_p, u8p]),
        "ko_ertl_from_registers": (C.c_uint64, [u8p, C.c_int, C.c_uint64, C.c_int]),
        "ko_classify_read": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32,
                                          u32p, u8p, C.POINTER(C.c_size_t), u32p]),
        "ko_hitlist_string": (C.c_size_t, [u32p, u8p, C.c_size_t, C.c_char_p]),
        "ko_run_new": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_int]),
        "ko_run_add_db": (C.c_int, [C.c_void_p, C.c_void_p]),
        "ko_run_free": (None, [C.c_void_p]),
        "ko_run_classify": (None, [C.c_void_p, C.c_void_p, u64p, u32p, C.c_size_t, u32p, u32p, u8p, u64p, u32p,
                                   u32p]),
        "ko_run_total_sequences": (C.c_uint64, [C.c_void_p]),
        "ko_run_total_classified": (C.c_uint64, [C.c_void_p]),
        "ko_run_n_taxa": (C.c_size_t, [C.c_void_p]),
        "ko_run_get": (None, [C.c_void_p, C.c_size_t, u32p, u64p, u64p, u64p, C.POINTER(C.c_int)]),
        "ko_run_sketch": (C.c_void_p, [C.c_void_p, C.c_size_t]),
        "ko_run_report": (C.c_void_p, [C.c_void_p, C.c_char_p, C.c_char_p]),
        "ko_free": (None, [C.c_void_p]),
        "ko_set_hll_sparse": (None, [C.c_int]),
        "ko_resolve_uids3": (C.c_uint32, [C.c_void_p, u32p, C.c_size_t, u32p, C.c_size_t]),
        "ko_umap_order": (C.c_size_t, [u32p, C.c_size_t, u32p]),
        "ko_run_set_uid_map": (None, [C.c_void_p, u32p, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def scan(seq: bytes, k: int):
    n = max(len(seq) - k + 1, 0)
    fwd = np.zeros(n, dtype=np.uint64)
    amb = np.zeros(n, dtype=np.uint8)
    got = lib().ko_scan(seq, len(seq), k, _p(fwd, u64p), _p(amb, u8p))
    assert got == n
    return fwd, amb


class Hll:
    def __init__(self, p=12, sparse=True, handle=None):
        self._own = handle is None
        self.h = handle if handle is not None else lib().ko_hll_new(p, int(sparse))
        self.p = p

    def __del__(self):
        if self._own and self.h and _lib is not None:
            _lib.ko_hll_free(self.h)
            self.h = None

    def insert(self, x: int):
        lib().ko_hll_insert(self.h, x & 0xFFFFFFFFFFFFFFFF)

    def insert_seq(self, n, mult, start=0):
        f = lib().ko_hll_insert
        for i in range(n):
            f(self.h, ((start + i) * mult) & 0xFFFFFFFFFFFFFFFF)

    def merge(self, other: "Hll"):
        lib().ko_hll_merge(self.h, other.h)

    def cardinality(self, use_n=True):
        return lib().ko_hll_cardinality(self.h, int(use_n))

    @property
    def n_observed(self):
        return lib().ko_hll_n_observed(self.h)

    @property
    def is_sparse(self):
        return bool(lib().ko_hll_is_sparse(self.h))

    def sparse_list(self):
        n = lib().ko_hll_sparse_size(self.h)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        lib().ko_hll_sparse_dump(self.h, _p(out, u32p), n)
        return out[:n]

    def registers(self):
        out = np.zeros(1 << self.p, dtype=np.uint8)
        lib().ko_hll_registers(self.h, _p(out, u8p))
        return out


class Tax:
    def __init__(self, path=None, ids=None, parents=None):
        if path is not None:
            err = C.create_string_buffer(256)
            self.h = lib().ko_tax_load(path.encode(), err, 256)
            if not self.h:
                raise RuntimeError(err.value.decode())
        else:
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            parents = np.ascontiguousarray(parents, dtype=np.uint32)
            self.h = lib().ko_tax_from_arrays(_p(ids, u32p), _p(parents, u32p), len(ids))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ko_tax_free(self.h)
            self.h = None

    def parent(self, t):
        return lib().ko_tax_parent(self.h, t)

    def lca(self, a, b):
        return lib().ko_lca(self.h, a, b)

    def resolve_uids(self, uids, uid_map):
        """resolve_uids3 (uid_mapping.cpp:212-274): uids = a read's non-zero DB values in k-mer order, uid_map = uint32
        [n, 2] {taxid, parent uid} blocks"""
        u = np.ascontiguousarray(uids, dtype=np.uint32)
        m = np.ascontiguousarray(uid_map, dtype=np.uint32).reshape(-1)
        return lib().ko_resolve_uids3(self.h, _p(u, u32p), len(u), _p(m, u32p), len(m) // 2)

    def resolve(self, hits: dict):
        t = np.array(list(hits.keys()), dtype=np.uint32)
        c = np.array(list(hits.values()), dtype=np.uint32)
        return lib().ko_resolve_tree(self.h, _p(t, u32p), _p(c, u32p), len(t))


class Db:
    def __init__(self, kdb=None, idx=None, pairs=None, key_ct=None, k=None, offsets=None, nt=None, idx_type=2):
        if kdb is not None:
            err = C.create_string_buffer(256)
            self.h = lib().ko_db_open(kdb.encode(), idx.encode(), err, 256)
            if not self.h:
                raise RuntimeError(err.value.decode())
        else:
            self._keep = (pairs, offsets)
            self.h = lib().ko_db_wrap(pairs.ctypes.data, key_ct, k, offsets.ctypes.data, nt, idx_type)
        self.k = lib().ko_db_k(self.h)
        self.nt = lib().ko_db_nt(self.h)
        self.idx_type = lib().ko_db_idx_type(self.h)
        self.key_ct = lib().ko_db_key_ct(self.h)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ko_db_close(self.h)
            self.h = None

    def query(self, canon):
        return lib().ko_db_query(self.h, canon)

    def count_taxons(self):
        n = lib().ko_db_count_taxons(self.h, None, None)
        t = np.zeros(n, dtype=np.uint32)
        c = np.zeros(n, dtype=np.uint64)
        lib().ko_db_count_taxons(self.h, _p(t, u32p), _p(c, u64p))
        return t, c


def pack_reads(seqs):
    """list of bytes -> (buffer with '\\n' after each read, off uint64[n], len uint32[n])."""
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    off = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
    buf = b"".join(s + b"\n" for s in seqs)
    return buf, off, lens


class Run:
    """One classify run (per-read results + per-taxon counts + report)."""

    def __init__(self, db: Db, tax: Tax, work_unit_nt=500000, quick=False, min_hits=1, threads=1, extra_dbs=()):
        self.db, self.tax = db, tax
        self.h = lib().ko_run_new(db.h, tax.h, work_unit_nt, int(quick), min_hits, threads)
        self.quick = quick
        self.extra_dbs = list(extra_dbs)  # further -d/-i pairs, searched in order after `db`
        for e in self.extra_dbs:
            if lib().ko_run_add_db(self.h, e.h) != 0:
                raise ValueError("databases must share k (at most 8)")

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ko_run_free(self.h)
            self.h = None

    def set_uid_map(self, uid_map):
        """classify -I: the database's values are UIDs; uid_map = uint32 [n, 2] {taxid, parent uid} (the map file)"""
        self._uid_map = np.ascontiguousarray(uid_map, dtype=np.uint32).reshape(-1)
        lib().ko_run_set_uid_map(self.h, _p(self._uid_map, u32p), len(self._uid_map) // 2)

    def classify(self, seqs, want_taxa=True):
        buf, off, lens = pack_reads(seqs)
        return self.classify_packed(buf, off, lens, want_taxa)

    def classify_packed(self, buf, off, lens, want_taxa=True):
        n = len(lens)
        k = self.db.k
        nk = np.maximum(lens.astype(np.int64) - k + 1, 0).astype(np.uint64)
        toff = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(nk, out=toff[1:])
        calls = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
        nsl = np.zeros(n, dtype=np.uint32)
        hits = np.zeros(n, dtype=np.uint32)
        taxa = np.zeros(max(int(toff[-1]), 1), dtype=np.uint32) if want_taxa else None
        amb = np.zeros(max(int(toff[-1]), 1), dtype=np.uint8) if want_taxa else None
        bufp = buf if isinstance(buf, (bytes, bytearray)) else buf.ctypes.data
        lib().ko_run_classify(self.h, bufp, _p(off, u64p), _p(lens, u32p), n, _p(calls, u32p), _p(taxa, u32p),
                              _p(amb, u8p), _p(toff, u64p), _p(nsl, u32p), _p(hits, u32p))
        return {"calls": calls, "taxa": taxa, "ambig": amb, "taxa_off": toff, "n_slots": nsl, "hits": hits}

    def counts(self):
        out = {}
        L = lib()
        for i in range(L.ko_run_n_taxa(self.h)):
            t, nr, nk, card, sp = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
            L.ko_run_get(self.h, i, C.byref(t), C.byref(nr), C.byref(nk), C.byref(card), C.byref(sp))
            out[t.value] = {"n_reads": nr.value, "n_kmers": nk.value, "cardinality": card.value,
                            "sparse": bool(sp.value), "sketch": Hll(handle=L.ko_run_sketch(self.h, i))}
        return out

    def report(self, taxdb_path, counts_path=None):
        p = lib().ko_run_report(self.h, taxdb_path.encode(), counts_path.encode() if counts_path else None)
        s = C.string_at(p).decode()
        lib().ko_free(p)
        return s


def umap_order(keys):
    """iteration order of a libstdc++ std::unordered_map<uint32_t, T> after operator[] on `keys` in that order"""
    k = np.ascontiguousarray(keys, dtype=np.uint32)
    out = np.zeros(max(len(k), 1), dtype=np.uint32)
    n = lib().ko_umap_order(_p(k, u32p), len(k), _p(out, u32p))
    return out[:n]


def set_hll_sparse(sparse: bool):
    """False = sketches dense from the start (the GPU path's HLL model); True = reference behaviour."""
    lib().ko_set_hll_sparse(int(sparse))


def hitlist(taxa, ambig):
    n = len(taxa)
    buf = C.create_string_buffer(24 * n + 16)
    taxa = np.ascontiguousarray(taxa, dtype=np.uint32)
    ambig = np.ascontiguousarray(ambig, dtype=np.uint8)
    m = lib().ko_hitlist_string(_p(taxa, u32p), _p(ambig, u8p), n, buf)
    return buf.raw[:m].decode()


def kraken_lines(ids, seqs, res, quick=False, only_classified=False, print_seq=False):
    """Kraken output lines (classify.cpp:980-1010) from a Run.classify() result."""
    out = []
    for i, (rid, s) in enumerate(zip(ids, seqs)):
        call = int(res["calls"][i])
        if call == 0xFFFFFFFF:
            continue  # read dropped by the reference's empty-work-unit rule
        if call == 0 and only_classified:
            continue
        if quick:
            hl = f"Q:{int(res['hits'][i])}"
        else:
            a = int(res["taxa_off"][i])
            n = int(res["n_slots"][i])
            hl = hitlist(res["taxa"][a:a + n], res["ambig"][a:a + n])
        line = f"{'C' if call else 'U'}\t{rid}\t{call}\t{len(s)}\t{hl}"
        if print_seq:
            line += "\t" + s.decode()
        out.append(line)
    return "\n".join(out) + ("\n" if out else "")
