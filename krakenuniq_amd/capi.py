"""ctypes binding of libkrakenuniq_amd.so (the C ABI in include/krakenuniq_amd.h).

Thin plumbing only: every method maps 1:1 onto a ku_* entry point and raises
KuError on a non-zero status.  There is no Python/CPU classification fallback:
if the shared library (HIP kernels included) is missing, importing fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KU_LIB") or os.path.join(_HERE, "libkrakenuniq_amd.so")  # KU_LIB: tuning experiments, and the tests' -DKU_TEST_HOOKS build

KU_AMBIG = 0xFFFFFFFF
KU_HLL_M = 4096
KU_F_QUICK, KU_F_NO_COUNTS, KU_F_KEEP_SLOTS, KU_F_MERGE_CHUNK = 1, 2, 4, 8
KU_P_ONLY_CLASSIFIED, KU_P_SEQUENCE, KU_P_QUICK = 1, 2, 4

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


class KuError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        detail = lib().ku_last_error().decode()
        super().__init__(f"{where}: {lib().ku_strerror(status).decode()} ({status}): {detail}")


class DbInfo(C.Structure):
    _fields_ = [("k", C.c_uint32), ("nt", C.c_uint32), ("idx_type", C.c_uint32), ("key_len", C.c_uint32),
                ("key_ct", C.c_uint64), ("n_bins", C.c_uint64)]


class Opts(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("min_hits", C.c_uint32), ("max_read_len", C.c_uint32),
                ("reserved", C.c_uint32)]


class Run(C.Structure):
    _fields_ = [("code", C.c_uint32), ("start", C.c_uint32)]


class CountsDims(C.Structure):
    _fields_ = [("n_slots", C.c_uint64), ("n_nodes", C.c_uint64)]


class DevBatch(C.Structure):  # ku_mgpu_dev_batch
    _fields_ = [("d_seqs", C.c_void_p), ("d_seq_off", C.c_void_p), ("d_seq_len", C.c_void_p), ("d_calls", C.c_void_p),
                ("d_taxa", C.c_void_p), ("stream", C.c_void_p)]


# name -> (restype, argtypes); mirrors include/krakenuniq_amd.h one to one
SIGNATURES = {
    "ku_strerror": (C.c_char_p, [C.c_int]),
    "ku_last_error": (C.c_char_p, []),
    "ku_abi_version": (C.c_int, []),
    "ku_device_count": (C.c_int, []),
    "ku_db_open": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "ku_db_wrap": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                             C.POINTER(C.c_void_p)]),
    "ku_db_close": (None, [C.c_void_p]),
    "ku_db_get_info": (C.c_int, [C.c_void_p, C.POINTER(DbInfo)]),
    "ku_db_shard_plan": (C.c_int, [C.c_void_p, C.c_uint32, u64p]),
    "ku_db_chunk_plan": (C.c_int, [C.c_void_p, C.c_uint64, u64p, C.c_uint32, u32p]),
    "ku_db_values": (C.c_int, [C.c_void_p, u32p, u64p]),
    "ku_setlcas_open": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ku_setlcas_add": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32]),
    "ku_setlcas_finish": (C.c_int, [C.c_void_p, u32p, u64p]),
    "ku_setlcas_uid_map": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]),
    "ku_setlcas_close": (None, [C.c_void_p]),
    "ku_db_sort_files": (C.c_int, [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_int]),
    "ku_ctx_swap_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "ku_ctx_prefetch_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "ku_ctx_mem_info": (C.c_int, [C.c_void_p, u64p, u64p]),
    "ku_batch_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u32p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "ku_batch_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Opts)]),
    "ku_batch_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Opts), u32p, u32p, u64p, u32p, u64p]),
    "ku_batch_destroy": (None, [C.c_void_p]),
    "ku_batch_absorb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ku_ctx_merge_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ku_tax_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ku_tax_from_arrays": (C.c_int, [u32p, u32p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "ku_tax_close": (None, [C.c_void_p]),
    "ku_tax_size": (C.c_uint64, [C.c_void_p]),
    "ku_tax_parent": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "ku_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ku_ctx_destroy": (None, [C.c_void_p]),
    "ku_ctx_load_db": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "ku_ctx_adopt_db": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_uint64, C.c_uint64]),
    "ku_ctx_add_db": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ku_ctx_db_layout": (C.c_int, [C.c_void_p, u32p, u64p]),
    "ku_ctx_db_values": (C.c_int, [C.c_void_p, u32p, u64p]),
    "ku_ctx_set_taxonomy": (C.c_int, [C.c_void_p, C.c_void_p, u32p, C.c_uint64]),
    "ku_ctx_count_taxons": (C.c_int, [C.c_void_p, u32p, u64p, u64p]),
    "ku_ctx_count_taxons_db": (C.c_int, [C.c_void_p, C.c_uint32, u32p, u64p, u64p]),
    "ku_ctx_reset_counts": (C.c_int, [C.c_void_p]),
    "ku_ctx_enable_sparse": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "ku_sparse_export": (C.c_int, [C.c_void_p, u8p, u64p, u64p]),
    "ku_sparse_close_unit": (C.c_int, [C.c_void_p]),
    "ku_report_sparse": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, u32p, u64p, u8p, u8p, u64p, C.c_uint64,
                                   C.c_uint64, u32p, u64p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_hll_cardinality_sparse": (C.c_uint64, [u32p, C.c_uint64, C.c_uint64]),
    "ku_tax_ids": (C.c_int, [C.c_void_p, u32p]),
    "ku_report_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, u8p, u64p, u64p, u64p, u64p, C.c_uint64,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_ctx_sparse_state": (C.c_int, [C.c_void_p]),
    "ku_ctx_disable_sparse": (C.c_int, [C.c_void_p]),
    "ku_ctx_report": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_size_t)]),
    "ku_ctx_report_cols": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_size_t)]),
    "ku_report_rows_cols": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, u8p, u64p, u64p, u64p, u64p, C.c_uint64, C.c_uint32,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_ctx_enable_exact": (C.c_int, [C.c_void_p, C.c_uint32]),
    "ku_counts_export_exact": (C.c_int, [C.c_void_p, u64p]),
    "ku_classify_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u32p, C.c_uint64, C.POINTER(Opts),
                                    u32p, u32p, u32p]),
    "ku_classify_batch_rle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u32p, C.c_uint64, C.POINTER(Opts), u32p, u32p,
                                        u64p, u32p, u64p]),
    "ku_fetch_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "ku_classify_batch_rle_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u32p, C.c_uint64, C.POINTER(Opts), u32p, u32p,
                                                u64p, u32p, C.c_void_p, C.c_uint64]),
    "ku_classify_batch_rle_copied": (C.c_uint64, [C.c_void_p]),
    "ku_classify_batch_rle_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
    "ku_classify_batch_rle_finish": (C.c_int, [C.c_void_p, u64p]),
    "ku_classify_batch_rle_in_flight": (C.c_int, [C.c_void_p]),
    "ku_classify_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ku_classify_batch_device_rle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                               C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "ku_device_rle_runs_cap": (C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]),
    "ku_lookup_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Opts), C.c_void_p, C.c_void_p]),
    "ku_resolve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Opts),
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ku_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "ku_lookup_stats_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.c_void_p]),
    "ku_counts_dims_get": (C.c_int, [C.c_void_p, C.POINTER(CountsDims)]),
    "ku_counts_export": (C.c_int, [C.c_void_p, u32p, u64p, u8p, u32p, u64p]),
    "ku_counts_device_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), u64p, C.POINTER(C.c_void_p), u64p,
                                        C.POINTER(C.c_void_p), u64p]),
    "ku_hll_cardinality": (C.c_uint64, [u8p, C.c_uint32, C.c_uint64]),
    "ku_hitlist_string": (C.c_size_t, [u32p, C.c_size_t, C.c_char_p]),
    "ku_format_kraken": (C.c_int, [C.c_void_p, u64p, u32p, C.c_uint64, C.c_char_p, C.c_uint32, u32p, u32p, u32p,
                                   C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_format_kraken_rle": (C.c_int, [C.c_void_p, u64p, u32p, C.c_uint64, C.c_char_p, C.c_uint32, u32p, C.c_void_p, u64p, u32p,
                                       u32p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_report": (C.c_int, [C.c_void_p, C.c_char_p, u32p, u64p, u8p, C.c_uint64, u32p, u64p, C.c_uint64,
                            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_report_multi": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, u32p, u64p, u8p, C.c_uint64, u32p, u64p,
                                  C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_report_exact": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, u32p, u64p, u64p, C.c_uint64, u32p, u64p,
                                  C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "ku_mgpu_unique_id": (C.c_int, [u8p]),
    "ku_mgpu_create": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint32,
                                 C.POINTER(C.c_void_p)]),
    "ku_mgpu_destroy": (None, [C.c_void_p]),
    "ku_mgpu_ctx": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "ku_mgpu_uses_rccl": (C.c_int, [C.c_void_p]),
    "ku_mgpu_uses_routing": (C.c_int, [C.c_void_p]),
    "ku_mgpu_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ku_mgpu_step_times": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]),
    "ku_mgpu_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ku_mgpu_load_dbs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p]),
    "ku_mgpu_enable_sparse": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "ku_mgpu_sparse_close_unit": (C.c_int, [C.c_void_p]),
    "ku_mgpu_sparse_state": (C.c_int, [C.c_void_p]),
    "ku_mgpu_enable_exact": (C.c_int, [C.c_void_p, C.c_uint32]),
    "ku_mgpu_set_taxonomy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ku_mgpu_classify_batch_rle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u32p, C.c_uint64, C.POINTER(Opts), u32p,
                                             u32p, u64p, u32p, u64p]),
    "ku_mgpu_fetch_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "ku_mgpu_step_device": (C.c_int, [C.c_void_p, C.POINTER(DevBatch), C.c_uint64, C.c_uint64, u64p, u64p, C.POINTER(Opts)]),
    "ku_mgpu_reduce_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ku_mgpu_count_taxons": (C.c_int, [C.c_void_p, u32p, u64p, u64p]),
    "ku_uid_map_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ku_uid_map_from_blocks": (C.c_int, [u32p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "ku_uid_map_close": (None, [C.c_void_p]),
    "ku_uid_map_size": (C.c_uint64, [C.c_void_p]),
    "ku_resolve_uids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64p, u32p, u32p, C.c_uint64, C.c_uint32, C.c_uint32, u32p]),
    "ku_ctx_replace_calls": (C.c_int, [C.c_void_p, u32p, C.c_uint64, u64p]),
    "ku_free": (None, [C.c_void_p]),
    "ku_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "ku_host_free": (None, [C.c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C krakenuniq_amd/csrc` "
                              "(or __graft_entry__.build()); there is no fallback path")
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's).
        # Importing torch first makes this library bind to the runtime torch uses, so torch tensors, RCCL
        # and our kernels share one device context; loading ours first leaves torch with "No HIP GPUs".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the ABI symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def kernel_rev():
    """identity of the kernel sources the library was built from (profiles are only valid for the source they measured)"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for fn in ("ku_device.h", "ku_short.hip", "ku_kernels.hip", "ku_route.hip"):  # device code of the classify kernels (not the host-side declarations)
        h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:12]


def _chk(status, where):
    if status != 0:
        raise KuError(status, where)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Db:
    """Host view of database.kdb/.idx (ku_db)."""

    def __init__(self, kdb=None, idx=None, pairs=None, key_ct=None, k=None, offsets=None, nt=None, idx_type=2):
        self.h = C.c_void_p()
        if kdb is not None:
            _chk(lib().ku_db_open(kdb.encode(), idx.encode(), C.byref(self.h)), "ku_db_open")
        else:
            self._keep = (pairs, offsets)
            _chk(lib().ku_db_wrap(pairs.ctypes.data, key_ct, k, offsets.ctypes.data, nt, idx_type, C.byref(self.h)),
                 "ku_db_wrap")
        self.info = DbInfo()
        _chk(lib().ku_db_get_info(self.h, C.byref(self.info)), "ku_db_get_info")

    def close(self):
        if self.h and _lib is not None:
            _lib.ku_db_close(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def shard_plan(self, n):
        b = np.zeros(n + 1, dtype=np.uint64)
        _chk(lib().ku_db_shard_plan(self.h, n, _p(b, u64p)), "ku_db_shard_plan")
        return b

    def chunk_plan(self, max_bytes, cap=4096):
        b = np.zeros(cap + 1, dtype=np.uint64)
        n = C.c_uint32()
        _chk(lib().ku_db_chunk_plan(self.h, max_bytes, _p(b, u64p), cap, C.byref(n)), "ku_db_chunk_plan")
        return b[:n.value + 1]

    def values(self):
        """ascending distinct non-zero taxids of the whole database (host scan)"""
        n = C.c_uint64()
        _chk(lib().ku_db_values(self.h, None, C.byref(n)), "ku_db_values")
        out = np.zeros(max(n.value, 1), dtype=np.uint32)
        n2 = C.c_uint64(len(out))
        _chk(lib().ku_db_values(self.h, _p(out, u32p), C.byref(n2)), "ku_db_values")
        return out[:n2.value]


class Tax:
    """Host taxonomy (ku_tax)."""

    def __init__(self, path=None, ids=None, parents=None):
        self.h = C.c_void_p()
        if path is not None:
            _chk(lib().ku_tax_open(path.encode(), C.byref(self.h)), "ku_tax_open")
        else:
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            parents = np.ascontiguousarray(parents, dtype=np.uint32)
            _chk(lib().ku_tax_from_arrays(_p(ids, u32p), _p(parents, u32p), len(ids), C.byref(self.h)),
                 "ku_tax_from_arrays")

    def ids(self):
        """taxids of the entries in row order (the layout report_rows expects)"""
        out = np.zeros(max(int(lib().ku_tax_size(self.h)), 1), dtype=np.uint32)
        _chk(lib().ku_tax_ids(self.h, _p(out, u32p)), "ku_tax_ids")
        return out[:int(lib().ku_tax_size(self.h))]

    def close(self):
        if self.h and _lib is not None:
            _lib.ku_tax_close(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def parent(self, taxid):
        return lib().ku_tax_parent(self.h, taxid)

    def __len__(self):
        return lib().ku_tax_size(self.h)


class UidMap:
    """UID-to-taxid map of a UID database (set_lcas -I): {taxid, parent uid} blocks"""

    def __init__(self, path=None, blocks=None):
        self.h = C.c_void_p()
        if path is not None:
            _chk(lib().ku_uid_map_open(path.encode(), C.byref(self.h)), "ku_uid_map_open")
        else:
            b = np.ascontiguousarray(blocks, dtype=np.uint32).reshape(-1)
            _chk(lib().ku_uid_map_from_blocks(_p(b, u32p), len(b) // 2, C.byref(self.h)), "ku_uid_map_from_blocks")

    def __len__(self):
        return int(lib().ku_uid_map_size(self.h))

    def close(self):
        if self.h:
            lib().ku_uid_map_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resolve_uids(tax, uid_map, rle, lens, k, n_threads=1):
    """resolve_uids3 for a batch from its run-length encoded codes (ku_resolve_uids); rle as classify_batch_rle returns"""
    runs = np.ascontiguousarray(rle["runs"], dtype=np.uint32)
    roff = np.ascontiguousarray(rle["run_off"], dtype=np.uint64)
    rcnt = np.ascontiguousarray(rle["run_cnt"], dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    calls = np.zeros(max(len(lens), 1), dtype=np.uint32)
    _chk(lib().ku_resolve_uids(tax.h, uid_map.h, runs.ctypes.data, _p(roff, u64p), _p(rcnt, u32p), _p(lens, u32p), len(lens), k,
                               n_threads, _p(calls, u32p)), "ku_resolve_uids")
    return calls[:len(lens)]


class Ctx:
    """Per-GPU context (ku_ctx)."""

    def __init__(self, device=0, borrowed=None):
        self._keep = []
        self._borrowed = borrowed is not None
        if borrowed is not None:  # a context owned by a multi-GPU group (ku_mgpu_ctx)
            self.h = C.c_void_p(borrowed)
            return
        self.h = C.c_void_p()
        _chk(lib().ku_ctx_create(device, C.byref(self.h)), "ku_ctx_create")

    def close(self):
        if getattr(self, "h", None) and _lib is not None and not getattr(self, "_borrowed", False):
            _lib.ku_ctx_destroy(self.h)
        self.h = C.c_void_p()

    __del__ = close

    def load_db(self, db: Db, bin_lo=0, bin_hi=None):
        if bin_hi is None:
            bin_hi = db.info.n_bins
        _chk(lib().ku_ctx_load_db(self.h, db.h, bin_lo, bin_hi), "ku_ctx_load_db")

    def swap_shard(self, db: Db, bin_lo, bin_hi):
        """out-of-core run: make bins [bin_lo, bin_hi) the resident shard, keeping slots and per-taxon state"""
        _chk(lib().ku_ctx_swap_shard(self.h, db.h, int(bin_lo), int(bin_hi)), "ku_ctx_swap_shard")

    def prefetch_shard(self, db: Db, bin_lo, bin_hi):
        """upload + lay out the next chunk next to the resident one (ku_ctx_prefetch_shard); swap_shard then exchanges them"""
        _chk(lib().ku_ctx_prefetch_shard(self.h, db.h, int(bin_lo), int(bin_hi)), "ku_ctx_prefetch_shard")

    def batch(self, buf, off, lens):
        return Batch(self, buf, off, lens)

    def add_db(self, db: Db):
        """a further database of a hierarchical run, searched after the ones already resident"""
        _chk(lib().ku_ctx_add_db(self.h, db.h), "ku_ctx_add_db")

    def adopt_db(self, d_pairs_ptr, n_pairs, d_offsets_ptr, k, nt, idx_type=2, bin_lo=0, bin_hi=None, keep=None):
        if bin_hi is None:
            bin_hi = 4 ** nt
        self._keep.append(keep)
        _chk(lib().ku_ctx_adopt_db(self.h, d_pairs_ptr, n_pairs, d_offsets_ptr, k, nt, idx_type, bin_lo, bin_hi),
             "ku_ctx_adopt_db")

    def db_layout(self):
        h, b = C.c_uint32(), C.c_uint64()
        _chk(lib().ku_ctx_db_layout(self.h, C.byref(h), C.byref(b)), "ku_ctx_db_layout")
        return {"hash": bool(h.value), "resident_bytes": b.value}

    def db_values(self):
        n = C.c_uint64()
        _chk(lib().ku_ctx_db_values(self.h, None, C.byref(n)), "ku_ctx_db_values")
        out = np.zeros(max(n.value, 1), dtype=np.uint32)
        n2 = C.c_uint64(len(out))
        _chk(lib().ku_ctx_db_values(self.h, _p(out, u32p), C.byref(n2)), "ku_ctx_db_values")
        return out[:n2.value]

    def set_taxonomy(self, tax: Tax, all_values=None):
        if all_values is None:
            _chk(lib().ku_ctx_set_taxonomy(self.h, tax.h, None, 0), "ku_ctx_set_taxonomy")
        else:
            v = np.ascontiguousarray(all_values, dtype=np.uint32)
            _chk(lib().ku_ctx_set_taxonomy(self.h, tax.h, _p(v, u32p), len(v)), "ku_ctx_set_taxonomy")

    def count_taxons(self, db_index=0):
        n = C.c_uint64()
        _chk(lib().ku_ctx_count_taxons_db(self.h, db_index, None, None, C.byref(n)), "ku_ctx_count_taxons_db")
        t = np.zeros(max(n.value, 1), dtype=np.uint32)
        c = np.zeros(max(n.value, 1), dtype=np.uint64)
        n2 = C.c_uint64(len(t))
        _chk(lib().ku_ctx_count_taxons_db(self.h, db_index, _p(t, u32p), _p(c, u64p), C.byref(n2)), "ku_ctx_count_taxons_db")
        return t[:n2.value], c[:n2.value]

    def enable_exact(self, capacity_log2=20):
        _chk(lib().ku_ctx_enable_exact(self.h, capacity_log2), "ku_ctx_enable_exact")

    def exact_counts(self):
        d = CountsDims()
        _chk(lib().ku_counts_dims_get(self.h, C.byref(d)), "ku_counts_dims_get")
        out = np.zeros(d.n_slots, dtype=np.uint64)
        _chk(lib().ku_counts_export_exact(self.h, _p(out, u64p)), "ku_counts_export_exact")
        return out

    def reset_counts(self):
        _chk(lib().ku_ctx_reset_counts(self.h), "ku_ctx_reset_counts")

    def enable_sparse(self, work_unit_nt=500000, global_log2=0):
        """HyperLogLog++ sparse-mode emulation (ku_ctx_enable_sparse): exact reproduction of the reference's report"""
        _chk(lib().ku_ctx_enable_sparse(self.h, work_unit_nt, global_log2), "ku_ctx_enable_sparse")

    def disable_sparse(self):
        _chk(lib().ku_ctx_disable_sparse(self.h), "ku_ctx_disable_sparse")

    def sparse_close_unit(self):
        _chk(lib().ku_sparse_close_unit(self.h), "ku_sparse_close_unit")

    def sparse_state(self):
        """0 = emulation off, 1 = on, 2 = gave up for lack of device memory (ku_ctx_sparse_state)"""
        return lib().ku_ctx_sparse_state(self.h)

    def report(self, tax: "Tax", counts_paths=(), flags=0):
        """the report from the device-resident state, clade roll-up on the GPU (ku_ctx_report; flags = 1: the six columns of
        `classify -p 0`, ku_ctx_report_cols)"""
        out, n = C.c_void_p(), C.c_size_t()
        paths = (C.c_char_p * len(counts_paths))(*[p.encode() for p in counts_paths])
        if flags:
            _chk(lib().ku_ctx_report_cols(self.h, tax.h, paths, len(counts_paths), flags, C.byref(out), C.byref(n)), "ku_ctx_report_cols")
        else:
            _chk(lib().ku_ctx_report(self.h, tax.h, paths, len(counts_paths), C.byref(out), C.byref(n)), "ku_ctx_report")
        s = C.string_at(out, n.value).decode()
        lib().ku_free(out)
        return s

    def sparse_export(self):
        """(slot_is_sparse uint8[n_slots], pairs uint64[n] = slot << 32 | encoded hash) -- closes the last work unit"""
        d = CountsDims()
        _chk(lib().ku_counts_dims_get(self.h, C.byref(d)), "ku_counts_dims_get")
        flags = np.zeros(d.n_slots, dtype=np.uint8)
        n = C.c_uint64()
        _chk(lib().ku_sparse_export(self.h, _p(flags, u8p), None, C.byref(n)), "ku_sparse_export")
        pairs = np.zeros(max(n.value, 1), dtype=np.uint64)
        n2 = C.c_uint64(len(pairs))
        _chk(lib().ku_sparse_export(self.h, _p(flags, u8p), _p(pairs, u64p), C.byref(n2)), "ku_sparse_export")
        return flags, pairs[:n2.value]

    def classify_batch(self, buf, off, lens, flags=0, min_hits=1, want_taxa=True):
        """Host-buffer entry point.  buf: bytes/np.uint8 with a non-ACGT byte after every read."""
        arr = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(lens)
        calls = np.zeros(max(n, 1), dtype=np.uint32)
        hits = np.zeros(max(n, 1), dtype=np.uint32)
        taxa = np.zeros(max(len(arr), 1), dtype=np.uint32) if want_taxa else None
        o = Opts(flags, min_hits, 0, 0)
        _chk(lib().ku_classify_batch(self.h, arr.ctypes.data, len(arr), _p(off, u64p), _p(lens, u32p), n, C.byref(o),
                                     _p(calls, u32p), _p(taxa, u32p), _p(hits, u32p)), "ku_classify_batch")
        return {"calls": calls[:n], "taxa": taxa, "hits": hits[:n]}

    def classify_batch_rle(self, buf, off, lens, flags=0, min_hits=1, out=None):
        """Host-buffer entry point with run-length encoded per-k-mer codes (ku_classify_batch_rle + ku_fetch_runs).
        out: optional dict of caller-owned arrays (page-locked ones make the copies back DMA transfers): calls, hits,
        run_cnt (uint32[n]), run_off (uint64[n]), runs (uint32[cap, 2], cap >= the batch's run total)."""
        arr = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(lens)
        out = out or {}
        calls = out["calls"] if "calls" in out else np.zeros(max(n, 1), dtype=np.uint32)
        hits = out["hits"] if "hits" in out else np.zeros(max(n, 1), dtype=np.uint32)
        roff = out["run_off"] if "run_off" in out else np.zeros(max(n, 1), dtype=np.uint64)
        rcnt = out["run_cnt"] if "run_cnt" in out else np.zeros(max(n, 1), dtype=np.uint32)
        o = Opts(flags, min_hits, 0, 0)
        total = C.c_uint64()
        _chk(lib().ku_classify_batch_rle(self.h, arr.ctypes.data, len(arr), _p(off, u64p), _p(lens, u32p), n, C.byref(o),
                                         _p(calls, u32p), _p(hits, u32p), _p(roff, u64p), _p(rcnt, u32p),
                                         C.byref(total)), "ku_classify_batch_rle")
        runs = out.get("runs")
        if runs is None or len(runs) < total.value:
            runs = np.zeros((max(total.value, 1), 2), dtype=np.uint32)
        _chk(lib().ku_fetch_runs(self.h, runs.ctypes.data, total.value), "ku_fetch_runs")
        return {"calls": calls[:n], "hits": hits[:n], "runs": runs[:total.value], "run_off": roff[:n], "run_cnt": rcnt[:n]}

    def replace_calls(self, new_calls):
        """the last batch's reads are counted under new_calls (ku_ctx_replace_calls); returns the calls that could not be"""
        nc = np.ascontiguousarray(new_calls, dtype=np.uint32)
        dropped = C.c_uint64()
        _chk(lib().ku_ctx_replace_calls(self.h, _p(nc, u32p), len(nc), C.byref(dropped)), "ku_ctx_replace_calls")
        return int(dropped.value)

    def lookup_device(self, d_seqs, n_bytes, d_taxa, flags=0, stream=None):
        o = Opts(flags, 1, 0, 0)
        _chk(lib().ku_lookup_device(self.h, d_seqs, n_bytes, C.byref(o), d_taxa, stream), "ku_lookup_device")

    def resolve_device(self, d_seqs, d_off, d_len, n_reads, d_calls, d_taxa, d_hits=None, flags=0, min_hits=1,
                       max_read_len=0, stream=None):
        o = Opts(flags, min_hits, max_read_len, 0)
        _chk(lib().ku_resolve_device(self.h, d_seqs, d_off, d_len, n_reads, C.byref(o), d_calls, d_taxa, d_hits,
                                     stream), "ku_resolve_device")

    def rle_enqueue(self, buf, off, lens, flags=0, min_hits=1, runs_cap=0, out=None):
        """First step of a batch through the two-step form of ku_classify_batch_rle (up to KU_RLE_MAX_IN_FLIGHT = four batches in flight).  Returns the
        handle rle_finish() takes; the arrays stay alive with it.  out: caller-owned result arrays as for classify_batch_rle
        (page-locked ones make the copies asynchronous); out["runs"] / runs_cap: where the runs go with the other results."""
        arr = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(off)
        out = out or {}
        job = {"arr": arr, "off": off, "lens": lens, "n": n,
               "calls": out["calls"] if "calls" in out else np.zeros(max(n, 1), np.uint32),
               "hits": out["hits"] if "hits" in out else np.zeros(max(n, 1), np.uint32),
               "run_off": out["run_off"] if "run_off" in out else np.zeros(max(n, 1), np.uint64),
               "run_cnt": out["run_cnt"] if "run_cnt" in out else np.zeros(max(n, 1), np.uint32)}
        o = Opts(flags, min_hits, 0, 0)
        job["runs"] = out["runs"] if "runs" in out else (np.zeros((runs_cap, 2), np.uint32) if runs_cap else None)
        rc = len(job["runs"]) if job["runs"] is not None else 0
        _chk(lib().ku_classify_batch_rle_enqueue(self.h, arr.ctypes.data, len(arr), _p(off, u64p), _p(lens, u32p), n, C.byref(o),
                                                 _p(job["calls"], u32p), _p(job["hits"], u32p), _p(job["run_off"], u64p),
                                                 _p(job["run_cnt"], u32p), job["runs"].ctypes.data if rc else None, rc),
             "ku_classify_batch_rle_enqueue")
        return job

    def rle_finish(self, job):
        """Second step: waits for the OLDEST batch in flight (which must be `job`), fetches its runs.  Same dictionary as
        classify_batch_rle."""
        total = C.c_uint64(0)
        _chk(lib().ku_classify_batch_rle_finish(self.h, C.byref(total)), "ku_classify_batch_rle_finish")
        if job.get("runs") is not None and total.value <= lib().ku_classify_batch_rle_copied(self.h):
            runs = job["runs"][:total.value]  # they came with the other results
        else:
            runs = np.zeros((total.value, 2), np.uint32)
            _chk(lib().ku_fetch_runs(self.h, runs.ctypes.data, total.value), "ku_fetch_runs")
        n = job.get("n", len(job.get("calls", ())))
        return {"calls": job["calls"][:n], "hits": job["hits"][:n], "run_off": job["run_off"][:n], "run_cnt": job["run_cnt"][:n], "runs": runs}

    def rle_in_flight(self):
        return lib().ku_classify_batch_rle_in_flight(self.h)

    def rle_reserve(self, n_bytes, n_reads, max_read_len, n_jobs=4):
        """ku_classify_batch_rle_reserve: buffers for n_jobs batches + the warm-up (count-less synthetic batches)"""
        _chk(lib().ku_classify_batch_rle_reserve(self.h, n_bytes, n_reads, max_read_len, n_jobs), "ku_classify_batch_rle_reserve")

    def classify_batch_device(self, d_seqs, n_bytes, d_off, d_len, n_reads, d_calls, d_taxa, d_hits=None, flags=0,
                              min_hits=1, max_read_len=0, stream=None):
        o = Opts(flags, min_hits, max_read_len, 0)
        _chk(lib().ku_classify_batch_device(self.h, d_seqs, n_bytes, d_off, d_len, n_reads, C.byref(o), d_calls,
                                            d_taxa, d_hits, stream), "ku_classify_batch_device")

    def classify_batch_device_rle(self, d_seqs, n_bytes, d_off, d_len, n_reads, d_calls, d_runs, runs_cap, d_run_off, d_run_cnt,
                                  d_n_runs, max_read_len, flags=0, stream=None):
        """device buffers in, run-length encoded per-k-mer codes out (the fused kernel's own runs; no per-k-mer array)"""
        o = Opts(flags, 1, max_read_len, 0)
        _chk(lib().ku_classify_batch_device_rle(self.h, d_seqs, n_bytes, d_off, d_len, n_reads, C.byref(o), d_calls, d_runs,
                                                runs_cap, d_run_off, d_run_cnt, d_n_runs, stream), "ku_classify_batch_device_rle")

    def device_rle_runs_cap(self, n_bytes, n_reads, max_read_len):
        return int(lib().ku_device_rle_runs_cap(self.h, n_bytes, n_reads, max_read_len))

    def lookup_stats_device(self, d_seqs, n_bytes, stream=None):
        out = np.zeros(4, dtype=np.uint64)
        _chk(lib().ku_lookup_stats_device(self.h, d_seqs, n_bytes, _p(out, u64p), stream), "ku_lookup_stats_device")
        return {"lookups": int(out[0]), "sum_ceil_log2": int(out[1]), "nonempty": int(out[2]), "sum_nb": int(out[3])}

    def synchronize(self):
        _chk(lib().ku_ctx_synchronize(self.h), "ku_ctx_synchronize")

    def counts(self):
        d = CountsDims()
        _chk(lib().ku_counts_dims_get(self.h, C.byref(d)), "ku_counts_dims_get")
        ns, nn = d.n_slots, d.n_nodes
        out = {"slot_taxid": np.zeros(ns, dtype=np.uint32), "n_kmers": np.zeros(ns, dtype=np.uint64),
               "registers": np.zeros((ns, KU_HLL_M), dtype=np.uint8), "node_taxid": np.zeros(nn, dtype=np.uint32),
               "n_reads": np.zeros(nn, dtype=np.uint64)}
        _chk(lib().ku_counts_export(self.h, _p(out["slot_taxid"], u32p), _p(out["n_kmers"], u64p),
                                    _p(out["registers"], u8p), _p(out["node_taxid"], u32p), _p(out["n_reads"], u64p)),
             "ku_counts_export")
        return out

    def counts_device_ptrs(self):
        r, k, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nb, ns, nn = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().ku_counts_device_ptrs(self.h, C.byref(r), C.byref(nb), C.byref(k), C.byref(ns), C.byref(n),
                                         C.byref(nn)), "ku_counts_device_ptrs")
        return {"registers": r.value, "register_bytes": nb.value, "n_kmers": k.value, "n_slots": ns.value,
                "n_reads": n.value, "n_nodes": nn.value}


KU_MGPU_REPLICAS, KU_MGPU_NO_RCCL = 1, 2


def mgpu_unique_id():
    """the RCCL id rank 0 makes and the launcher hands to every process (ku_mgpu_unique_id)"""
    buf = np.zeros(128, dtype=np.uint8)
    _chk(lib().ku_mgpu_unique_id(_p(buf, u8p)), "ku_mgpu_unique_id")
    return buf


class Mgpu:
    """Several GPUs driven through the C++ multi-GPU driver (ku_mgpu): n_local ranks of `world` from this process."""

    def __init__(self, devices, first_rank=0, world=None, unique_id=None, flags=0):
        devices = list(devices)
        self.n_local = len(devices)
        self.world = self.n_local if world is None else world
        self.first_rank = first_rank
        arr = (C.c_int * self.n_local)(*devices)
        idb = np.ascontiguousarray(unique_id, dtype=np.uint8) if unique_id is not None else None
        self.h = C.c_void_p()
        _chk(lib().ku_mgpu_create(arr, self.n_local, first_rank, self.world, _p(idb, u8p), flags, C.byref(self.h)),
             "ku_mgpu_create")
        self._keep = []

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ku_mgpu_destroy(self.h)
        self.h = None  # (at interpreter exit the ctypes module may be gone already: no new objects here)

    __del__ = close

    def ctx(self, i=0):
        return Ctx(borrowed=lib().ku_mgpu_ctx(self.h, i))

    def uses_rccl(self):
        return bool(lib().ku_mgpu_uses_rccl(self.h))

    def uses_routing(self):
        return bool(lib().ku_mgpu_uses_routing(self.h))

    def set_timing(self, on=True):
        _chk(lib().ku_mgpu_set_timing(self.h, 1 if on else 0), "ku_mgpu_set_timing")

    def step_times(self, local=0):
        """stages of the last owner-routed step of one local rank (HIP events): ms of scan / owner / resolve, rounds,
        records and k-mers received"""
        out = (C.c_double * 6)()
        _chk(lib().ku_mgpu_step_times(self.h, local, out), "ku_mgpu_step_times")
        return {"scan_ms": out[0], "owner_ms": out[1], "resolve_ms": out[2], "rounds": int(out[3]), "records_received": int(out[4]),
                "kmers_received": int(out[5])}

    def load(self, db: Db, tax: Tax):
        self._keep += [db, tax]
        _chk(lib().ku_mgpu_load(self.h, db.h, tax.h), "ku_mgpu_load")

    def load_dbs(self, dbs, tax: Tax):
        """several databases searched in order per k-mer (replicas only)"""
        arr = (C.c_void_p * len(dbs))(*[d.h for d in dbs])
        self._keep += list(dbs) + [tax]
        _chk(lib().ku_mgpu_load_dbs(self.h, arr, len(dbs), tax.h), "ku_mgpu_load_dbs")

    def enable_sparse(self, work_unit_nt=500000, global_log2=0):
        _chk(lib().ku_mgpu_enable_sparse(self.h, work_unit_nt, global_log2), "ku_mgpu_enable_sparse")

    def sparse_close_unit(self):
        _chk(lib().ku_mgpu_sparse_close_unit(self.h), "ku_mgpu_sparse_close_unit")

    def sparse_state(self):
        return lib().ku_mgpu_sparse_state(self.h)

    def enable_exact(self, capacity_log2=20):
        _chk(lib().ku_mgpu_enable_exact(self.h, capacity_log2), "ku_mgpu_enable_exact")

    def set_taxonomy(self, tax: Tax):
        self._keep.append(tax)
        _chk(lib().ku_mgpu_set_taxonomy(self.h, tax.h), "ku_mgpu_set_taxonomy")

    def classify_batch_rle(self, buf, off, lens, flags=0, min_hits=1):
        arr = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(lens)
        calls = np.zeros(max(n, 1), dtype=np.uint32)
        hits = np.zeros(max(n, 1), dtype=np.uint32)
        roff = np.zeros(max(n, 1), dtype=np.uint64)
        rcnt = np.zeros(max(n, 1), dtype=np.uint32)
        o = Opts(flags, min_hits, 0, 0)
        total = C.c_uint64()
        _chk(lib().ku_mgpu_classify_batch_rle(self.h, arr.ctypes.data, len(arr), _p(off, u64p), _p(lens, u32p), n,
                                              C.byref(o), _p(calls, u32p), _p(hits, u32p), _p(roff, u64p), _p(rcnt, u32p),
                                              C.byref(total)), "ku_mgpu_classify_batch_rle")
        runs = np.zeros((max(total.value, 1), 2), dtype=np.uint32)
        _chk(lib().ku_mgpu_fetch_runs(self.h, runs.ctypes.data, total.value), "ku_mgpu_fetch_runs")
        return {"calls": calls[:n], "hits": hits[:n], "runs": runs[:total.value], "run_off": roff[:n], "run_cnt": rcnt[:n]}

    def step_device(self, batches, n_bytes, n_reads, read_bounds, pos_bounds, flags=0, min_hits=1, max_read_len=0):
        """batches: one dict per local rank with the device pointers d_seqs, d_seq_off, d_seq_len, d_calls, d_taxa, stream"""
        arr = (DevBatch * self.n_local)()
        for i, b in enumerate(batches):
            arr[i] = DevBatch(b["d_seqs"], b["d_seq_off"], b["d_seq_len"], b["d_calls"], b["d_taxa"], b.get("stream"))
        rb = np.ascontiguousarray(read_bounds, dtype=np.uint64)
        pb = np.ascontiguousarray(pos_bounds, dtype=np.uint64)
        o = Opts(flags, min_hits, max_read_len, 0)
        _chk(lib().ku_mgpu_step_device(self.h, arr, n_bytes, n_reads, _p(rb, u64p), _p(pb, u64p), C.byref(o)),
             "ku_mgpu_step_device")

    def reduce_state(self, streams=None):
        arr = None
        if streams is not None:
            arr = (C.c_void_p * self.n_local)(*streams)
        _chk(lib().ku_mgpu_reduce_state(self.h, arr), "ku_mgpu_reduce_state")

    def count_taxons(self):
        n = C.c_uint64()
        _chk(lib().ku_mgpu_count_taxons(self.h, None, None, C.byref(n)), "ku_mgpu_count_taxons")
        t = np.zeros(max(n.value, 1), dtype=np.uint32)
        c = np.zeros(max(n.value, 1), dtype=np.uint64)
        n2 = C.c_uint64(len(t))
        _chk(lib().ku_mgpu_count_taxons(self.h, _p(t, u32p), _p(c, u64p), C.byref(n2)), "ku_mgpu_count_taxons")
        return t[:n2.value], c[:n2.value]


class Batch:
    """Reads of one batch resident on the device across the chunk passes of an out-of-core run (ku_batch)."""

    def __init__(self, ctx, buf, off, lens):
        arr = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf
        self.ctx, self.n = ctx, len(lens)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        self.h = C.c_void_p()
        _chk(lib().ku_batch_create(ctx.h, arr.ctypes.data, len(arr), _p(off, u64p), _p(lens, u32p), self.n,
                                   C.byref(self.h)), "ku_batch_create")

    def lookup(self, flags=0, min_hits=1):
        o = Opts(flags, min_hits, 0, 0)
        _chk(lib().ku_batch_lookup(self.ctx.h, self.h, C.byref(o)), "ku_batch_lookup")

    def finish(self, flags=0, min_hits=1):
        n = self.n
        calls = np.zeros(max(n, 1), dtype=np.uint32)
        hits = np.zeros(max(n, 1), dtype=np.uint32)
        roff = np.zeros(max(n, 1), dtype=np.uint64)
        rcnt = np.zeros(max(n, 1), dtype=np.uint32)
        total = C.c_uint64()
        o = Opts(flags, min_hits, 0, 0)
        _chk(lib().ku_batch_finish(self.ctx.h, self.h, C.byref(o), _p(calls, u32p), _p(hits, u32p), _p(roff, u64p),
                                   _p(rcnt, u32p), C.byref(total)), "ku_batch_finish")
        runs = np.zeros((max(total.value, 1), 2), dtype=np.uint32)
        _chk(lib().ku_fetch_runs(self.ctx.h, runs.ctypes.data, total.value), "ku_fetch_runs")
        return {"calls": calls[:n], "hits": hits[:n], "runs": runs[:total.value], "run_off": roff[:n], "run_cnt": rcnt[:n]}

    def close(self):
        if self.h:
            lib().ku_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


KU_SL_RESET, KU_SL_FORCE_CONTAMINANT = 1, 2


class SetLcas:
    """set_lcas on the GPU (ku_setlcas_*): fold library sequences into the database values."""

    def __init__(self, db: Db, tax: Tax, flags=0, device=0):
        self.h = C.c_void_p()
        self.n = db.info.key_ct
        _chk(lib().ku_setlcas_open(device, db.h, tax.h, flags, C.byref(self.h)), "ku_setlcas_open")
        self._keep = (db, tax)

    def add(self, seq: bytes, taxid: int):
        _chk(lib().ku_setlcas_add(self.h, seq, len(seq), taxid), "ku_setlcas_add")

    def finish(self):
        vals = np.zeros(max(self.n, 1), dtype=np.uint32)
        miss = C.c_uint64()
        _chk(lib().ku_setlcas_finish(self.h, _p(vals, u32p), C.byref(miss)), "ku_setlcas_finish")
        return vals[:self.n], miss.value

    def close(self):
        if self.h:
            lib().ku_setlcas_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def db_sort_files(in_path, out_kdb, out_idx, nt, zero_vals=False, device=0):
    """db_sort on the GPU: Jellyfish-format list -> database.kdb + database.idx"""
    _chk(lib().ku_db_sort_files(device, in_path.encode(), out_kdb.encode(), out_idx.encode(), nt, int(zero_vals)),
         "ku_db_sort_files")


def hll_cardinality(registers, n_observed, p=12):
    r = np.ascontiguousarray(registers, dtype=np.uint8)
    return lib().ku_hll_cardinality(_p(r, u8p), p, int(n_observed))


def hitlist_string(taxa):
    t = np.ascontiguousarray(taxa, dtype=np.uint32)
    buf = C.create_string_buffer(24 * len(t) + 16)
    n = lib().ku_hitlist_string(_p(t, u32p), len(t), buf)
    return buf.raw[:n].decode()


def format_kraken(buf, off, lens, ids, k, calls, taxa=None, hits=None, flags=0):
    arr = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf
    off = np.ascontiguousarray(off, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    calls = np.ascontiguousarray(calls, dtype=np.uint32)
    idbuf = b"".join(i.encode() + b"\0" for i in ids)
    out, n = C.c_void_p(), C.c_size_t()
    _chk(lib().ku_format_kraken(arr.ctypes.data, _p(off, u64p), _p(lens, u32p), len(lens), idbuf, k,
                                _p(calls, u32p), _p(taxa, u32p), _p(hits, u32p), flags, C.byref(out), C.byref(n)),
         "ku_format_kraken")
    s = C.string_at(out, n.value).decode()
    lib().ku_free(out)
    return s


def format_kraken_rle(buf, off, lens, ids, k, res, flags=0):
    arr = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else buf
    off = np.ascontiguousarray(off, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    idbuf = b"".join(i.encode() + b"\0" for i in ids)
    runs = np.ascontiguousarray(res["runs"], dtype=np.uint32)
    out, n = C.c_void_p(), C.c_size_t()
    _chk(lib().ku_format_kraken_rle(arr.ctypes.data, _p(off, u64p), _p(lens, u32p), len(lens), idbuf, k,
                                    _p(np.ascontiguousarray(res["calls"], dtype=np.uint32), u32p), runs.ctypes.data,
                                    _p(np.ascontiguousarray(res["run_off"], dtype=np.uint64), u64p),
                                    _p(np.ascontiguousarray(res["run_cnt"], dtype=np.uint32), u32p),
                                    _p(np.ascontiguousarray(res["hits"], dtype=np.uint32), u32p), flags,
                                    C.byref(out), C.byref(n)), "ku_format_kraken_rle")
    s = C.string_at(out, n.value).decode()
    lib().ku_free(out)
    return s


def report_exact(tax: Tax, counts: dict, unique, counts_paths):
    """classifyExact's report: `unique` = Ctx.exact_counts(), counts_paths = list of database.kdb.counts files"""
    out, n = C.c_void_p(), C.c_size_t()
    paths = (C.c_char_p * len(counts_paths))(*[p.encode() for p in counts_paths])
    unique = np.ascontiguousarray(unique, dtype=np.uint64)
    _chk(lib().ku_report_exact(tax.h, paths, len(counts_paths), _p(counts["slot_taxid"], u32p), _p(counts["n_kmers"], u64p),
                               _p(unique, u64p), len(counts["slot_taxid"]), _p(counts["node_taxid"], u32p),
                               _p(counts["n_reads"], u64p), len(counts["node_taxid"]), C.byref(out), C.byref(n)),
         "ku_report_exact")
    s = C.string_at(out, n.value).decode()
    lib().ku_free(out)
    return s


def hll_cardinality_sparse(encoded, n_observed):
    e = np.ascontiguousarray(encoded, dtype=np.uint32)
    return int(lib().ku_hll_cardinality_sparse(_p(e, u32p), len(e), n_observed))


def report_rows(tax: Tax, present, clade_reads, tax_reads, clade_kmers, clade_uniq, counts_paths=()):
    """the report text from per-taxDB-entry clade summaries (ku_report_rows)"""
    out, n = C.c_void_p(), C.c_size_t()
    paths = (C.c_char_p * len(counts_paths))(*[p.encode() for p in counts_paths])
    arrs = [np.ascontiguousarray(present, dtype=np.uint8)] + [np.ascontiguousarray(a, dtype=np.uint64)
                                                              for a in (clade_reads, tax_reads, clade_kmers, clade_uniq)]
    _chk(lib().ku_report_rows(tax.h, paths, len(counts_paths), _p(arrs[0], u8p), *[_p(a, u64p) for a in arrs[1:]], len(arrs[0]),
                              C.byref(out), C.byref(n)), "ku_report_rows")
    s = C.string_at(out, n.value).decode()
    lib().ku_free(out)
    return s


def report_sparse(tax: Tax, counts: dict, slot_is_sparse, pairs, counts_paths):
    """the report with the reference's sparse sketches (ku_report_sparse); counts_paths: list of database.kdb.counts"""
    out, n = C.c_void_p(), C.c_size_t()
    regs = np.ascontiguousarray(counts["registers"], dtype=np.uint8)
    flags = np.ascontiguousarray(slot_is_sparse, dtype=np.uint8)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint64)
    paths = (C.c_char_p * len(counts_paths))(*[p.encode() for p in counts_paths])
    _chk(lib().ku_report_sparse(tax.h, paths, len(counts_paths), _p(counts["slot_taxid"], u32p), _p(counts["n_kmers"], u64p),
                                _p(regs, u8p), _p(flags, u8p), _p(pairs, u64p), len(pairs), len(counts["slot_taxid"]),
                                _p(counts["node_taxid"], u32p), _p(counts["n_reads"], u64p), len(counts["node_taxid"]),
                                C.byref(out), C.byref(n)), "ku_report_sparse")
    s = C.string_at(out, n.value).decode()
    lib().ku_free(out)
    return s


def report(tax: Tax, counts: dict, counts_path=None):
    """counts_path: database.kdb.counts, or a list of them (one per database of a hierarchical run)"""
    out, n = C.c_void_p(), C.c_size_t()
    regs = np.ascontiguousarray(counts["registers"], dtype=np.uint8)
    if isinstance(counts_path, (list, tuple)):
        paths = (C.c_char_p * len(counts_path))(*[p.encode() for p in counts_path])
        _chk(lib().ku_report_multi(tax.h, paths, len(counts_path), _p(counts["slot_taxid"], u32p),
                                   _p(counts["n_kmers"], u64p), _p(regs, u8p), len(counts["slot_taxid"]),
                                   _p(counts["node_taxid"], u32p), _p(counts["n_reads"], u64p),
                                   len(counts["node_taxid"]), C.byref(out), C.byref(n)), "ku_report_multi")
        s = C.string_at(out, n.value).decode()
        lib().ku_free(out)
        return s
    _chk(lib().ku_report(tax.h, counts_path.encode() if counts_path else None, _p(counts["slot_taxid"], u32p),
                         _p(counts["n_kmers"], u64p), _p(regs, u8p), len(counts["slot_taxid"]),
                         _p(counts["node_taxid"], u32p), _p(counts["n_reads"], u64p), len(counts["node_taxid"]),
                         C.byref(out), C.byref(n)), "ku_report")
    s = C.string_at(out, n.value).decode()
    lib().ku_free(out)
    return s
