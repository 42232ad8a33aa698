"""ctypes mirror of include/mcq_host.h (reference shard reader, taxonomy keys, classify)."""
import ctypes as C
import os

import numpy as np

from .build import host_lib_path

NO_TAXON = 0xFFFFFFFF


class Info(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sketch_size", C.c_uint32), ("winlen", C.c_uint32), ("winstride", C.c_uint32),
                ("q_sketch_size", C.c_uint32), ("q_winlen", C.c_uint32), ("q_winstride", C.c_uint32),
                ("max_locs_per_feature", C.c_uint32), ("n_ranks", C.c_uint32), ("n_targets", C.c_uint32),
                ("n_taxa", C.c_uint32), ("n_keys", C.c_uint64), ("n_locs", C.c_uint64)]


class TaxonRec(C.Structure):
    _fields_ = [("id", C.c_int64), ("parent", C.c_int64), ("rank", C.c_uint8), ("name", C.c_char_p), ("file", C.c_char_p),
                ("index", C.c_uint64), ("windows", C.c_uint64)]


class ShardParams(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("k", "sketch_size", "winlen", "winstride", "q_k", "q_sketch_size", "q_winlen",
                                         "q_winstride", "max_locs_per_feature")]


_lib = None


def write_shard(path, params, taxa, n_targets, keys, list_off, locs):
    """mcq_refdb_write_shard.  params: dict with the ShardParams field names; taxa: list of dicts
    (id, parent, rank, name, file, index, windows); keys u32, list_off u64 [n+1], locs u64 (tgt<<32|win)."""
    sp = ShardParams(**{k: int(v) for k, v in params.items()})
    arr = (TaxonRec * len(taxa))()
    keep = []
    for i, t in enumerate(taxa):
        nm, fl = t["name"].encode("latin-1"), t["file"].encode("latin-1")
        keep += [nm, fl]
        arr[i] = TaxonRec(t["id"], t["parent"], t["rank"], nm, fl, t["index"], t["windows"])
    keys = np.ascontiguousarray(keys, np.uint32); list_off = np.ascontiguousarray(list_off, np.uint64)
    locs = np.ascontiguousarray(locs, np.uint64)
    rc = lib().mcq_refdb_write_shard(path.encode(), C.byref(sp), arr, len(taxa), n_targets, keys.ctypes.data_as(C.c_void_p),
                                     list_off.ctypes.data_as(C.c_void_p), locs.ctypes.data_as(C.c_void_p), len(keys))
    if rc != 0:
        raise RuntimeError(lib().mcq_host_last_error().decode())


def lib():
    global _lib
    if _lib is None:
        p = host_lib_path()
        if not os.path.exists(p):
            raise ImportError("host library %s missing: run __graft_entry__.build()" % p)
        L = C.CDLL(p)
        L.mcq_refdb_open.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mcq_refdb_close.argtypes = [C.c_void_p]
        L.mcq_refdb_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        for f, t in (("mcq_refdb_keys", C.POINTER(C.c_uint32)), ("mcq_refdb_list_off", C.POINTER(C.c_uint64)),
                     ("mcq_refdb_locs", C.POINTER(C.c_uint64))):
            getattr(L, f).restype = t; getattr(L, f).argtypes = [C.c_void_p]
        L.mcq_refdb_tgt2tax.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.mcq_refdb_taxon_id.restype = C.c_int64; L.mcq_refdb_taxon_id.argtypes = [C.c_void_p, C.c_uint32]
        L.mcq_refdb_taxon_rank.restype = C.c_uint32; L.mcq_refdb_taxon_rank.argtypes = [C.c_void_p, C.c_uint32]
        L.mcq_refdb_taxon_name.restype = C.c_char_p; L.mcq_refdb_taxon_name.argtypes = [C.c_void_p, C.c_uint32]
        L.mcq_refdb_ancestor.restype = C.c_uint32; L.mcq_refdb_ancestor.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.mcq_refdb_classify.restype = C.c_uint32
        L.mcq_refdb_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32]
        L.mcq_default_hits_min.restype = C.c_uint32; L.mcq_default_hits_min.argtypes = [C.c_uint32]
        L.mcq_rank_from_name.restype = C.c_uint32; L.mcq_rank_from_name.argtypes = [C.c_char_p]
        L.mcq_rank_name.restype = C.c_char_p; L.mcq_rank_name.argtypes = [C.c_uint32]
        L.mcq_host_last_error.restype = C.c_char_p
        L.mcq_refdb_write_shard.argtypes = [C.c_char_p, C.POINTER(ShardParams), C.POINTER(TaxonRec), C.c_uint64, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.mcq_refdb_open_meta.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mcq_refdb_tgt_windows.argtypes = [C.c_void_p, C.c_void_p]
        L.mcq_refdb_file_stats.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mcq_shard_stream_open.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mcq_shard_stream_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.mcq_shard_stream_close.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class RefDb:
    """The reference's <prefix>.db_<r> shard files, parsed and unioned on the host."""

    def __init__(self, prefix, n_ranks, meta_only=False):
        """meta_only: mcq_refdb_open_meta -- parameters and taxa only, the tables are streamed (stream())"""
        h = C.c_void_p()
        opener = lib().mcq_refdb_open_meta if meta_only else lib().mcq_refdb_open
        if opener(prefix.encode(), n_ranks, C.byref(h)) != 0:
            raise RuntimeError(lib().mcq_host_last_error().decode())
        self.h = h
        self.info = Info()
        lib().mcq_refdb_get_info(self.h, C.byref(self.info))

    def close(self):
        if getattr(self, "h", None):
            lib().mcq_refdb_close(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tgt_windows(self):
        out = np.zeros(self.info.n_targets, np.uint32)
        if lib().mcq_refdb_tgt_windows(self.h, out.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError(lib().mcq_host_last_error().decode())
        return out

    def file_stats(self, rank):
        b, k, l = C.c_uint64(), C.c_uint64(), C.c_uint64()
        if lib().mcq_refdb_file_stats(self.h, rank, C.byref(b), C.byref(k), C.byref(l)) != 0:
            raise RuntimeError(lib().mcq_host_last_error().decode())
        return int(b.value), int(k.value), int(l.value)

    def stream(self, rank, chunk=1 << 20):
        """yields (feature, target, window) uint32 arrays, chunk by chunk in file order (meta_only handles)"""
        sh = C.c_void_p()
        if lib().mcq_shard_stream_open(self.h, rank, C.byref(sh)) != 0:
            raise RuntimeError(lib().mcq_host_last_error().decode())
        try:
            f, t, w = (np.zeros(chunk, np.uint32) for _ in range(3))
            n = C.c_uint64()
            while True:
                if lib().mcq_shard_stream_next(sh, f.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                               chunk, C.byref(n)) != 0:
                    raise RuntimeError(lib().mcq_host_last_error().decode())
                if n.value == 0:
                    return
                yield f[:n.value].copy(), t[:n.value].copy(), w[:n.value].copy()
        finally:
            lib().mcq_shard_stream_close(sh)

    def table(self):
        n, m = self.info.n_keys, self.info.n_locs
        keys = np.ctypeslib.as_array(lib().mcq_refdb_keys(self.h), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        off = np.ctypeslib.as_array(lib().mcq_refdb_list_off(self.h), shape=(n + 1,)).copy() if n else np.zeros(1, np.uint64)
        locs = np.ctypeslib.as_array(lib().mcq_refdb_locs(self.h), shape=(m,)).copy() if m else np.zeros(0, np.uint64)
        return keys, off, locs

    def tgt2tax(self, merge_below_rank):
        out = np.zeros(self.info.n_targets, np.uint32)
        if lib().mcq_refdb_tgt2tax(self.h, merge_below_rank, out.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError(lib().mcq_host_last_error().decode())
        return out

    def taxon_id(self, key):
        return int(lib().mcq_refdb_taxon_id(self.h, int(key)))

    def classify(self, cands, hits_min, hits_diff_fraction, highest_rank):
        """cands: array [n, 4] of (tax key, hits, beg, end) -> taxon index or NO_TAXON"""
        c = np.ascontiguousarray(cands, np.uint32).reshape(-1, 4)
        return int(lib().mcq_refdb_classify(self.h, c.ctypes.data_as(C.c_void_p), len(c), hits_min,
                                            C.c_float(hits_diff_fraction), highest_rank))


def rank_from_name(name):
    return int(lib().mcq_rank_from_name(name.encode()))
