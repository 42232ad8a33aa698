"""oracle/mc_oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/mc_oracle.cpp (the CPU restatement of the reference's
query path).  Importable only from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); the product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmc_oracle.so")
_lib = None

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "mc_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-std=c++14", "-O3", "-shared", "-fPIC", "-pthread", src, "-o", _SO])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_tmh.restype = C.c_uint32; L.orc_tmh.argtypes = [C.c_uint32]
        L.orc_revcomp.restype = C.c_uint32; L.orc_revcomp.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_canonical.restype = C.c_uint32; L.orc_canonical.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_windows.restype = C.c_int
        L.orc_windows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p, C.c_int]
        L.orc_sketch.restype = C.c_int
        L.orc_sketch.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, u32p]
        L.orc_db_create.restype = C.c_void_p
        L.orc_db_create.argtypes = [C.c_uint64, u32p, u64p, u64p, C.c_uint32, u32p,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_db_destroy.argtypes = [C.c_void_p]
        L.orc_query.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, u64p, C.c_int, C.c_uint32, C.c_uint32,
                                C.c_uint64, C.c_uint32, u32p, u32p, u64p, C.c_int]
        L.orc_query_matches.restype = C.c_uint64
        L.orc_query_matches.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, u64p, C.c_uint64]
        L.orc_query_target_cands.restype = C.c_uint64
        L.orc_query_target_cands.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64,
                                             C.c_uint64, u32p, C.c_uint64]
        L.orc_reduce_query.restype = C.c_uint32
        L.orc_reduce_query.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, u32p]
        L.orc_tree_fold.restype = C.c_uint32
        L.orc_tree_fold.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p, C.c_uint32, u32p]
        L.orc_insert_sequence.restype = C.c_uint32
        L.orc_insert_sequence.argtypes = [C.c_uint32, u32p, u32p, C.c_uint32, u32p]
        L.orc_classify.restype = C.c_uint32
        L.orc_classify.argtypes = [u32p, C.c_uint32, u32p, C.POINTER(C.c_uint8), C.c_uint32, C.c_float, C.c_uint32]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def tmh(x): return lib().orc_tmh(x)
def revcomp(x, k): return lib().orc_revcomp(x, k)
def canonical(x, k): return lib().orc_canonical(x, k)


def windows(n, winlen=128, stride=113):
    cap = 4 + n // max(1, stride)
    b = np.zeros(cap, np.uint64); e = np.zeros(cap, np.uint64)
    c = lib().orc_windows(n, winlen, stride, _p(b, u64p), _p(e, u64p), cap)
    return [(int(b[i]), int(e[i])) for i in range(c)]


def sketch(seq, k=16, s=16):
    if isinstance(seq, str):
        seq = seq.encode()
    out = np.zeros(max(1, s), np.uint32)
    n = lib().orc_sketch(seq, len(seq), k, s, _p(out, u32p))
    return out[:n].copy()


def pack_reads(seqs):
    """list of str/bytes -> (bases bytes, off uint64[n+1])."""
    bs = [s.encode() if isinstance(s, str) else s for s in seqs]
    off = np.zeros(len(bs) + 1, np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), off


class OracleDb:
    """keys u32[n], off u64[n+1], locs u64 ((tgt<<32)|win), tgt2tax u32[n_targets]."""

    def __init__(self, keys, off, locs, tgt2tax, k=16, s=16, winlen=128, winstride=113, tgt_winstride=0):
        self.keys = np.ascontiguousarray(keys, np.uint32)
        self.off = np.ascontiguousarray(off, np.uint64)
        self.locs = np.ascontiguousarray(locs, np.uint64)
        self.tgt2tax = np.ascontiguousarray(tgt2tax, np.uint32)
        self.h = lib().orc_db_create(len(self.keys), _p(self.keys, u32p), _p(self.off, u64p), _p(self.locs, u64p),
                                     len(self.tgt2tax), _p(self.tgt2tax, u32p), k, s, winlen, winstride, tgt_winstride)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_db_destroy(self.h); self.h = None
        except Exception:       # interpreter shutdown
            pass

    def query(self, bases, off, paired, max_cand=2, emulate_ranks=1, insert_size_max=0, quirk_seq_drop=0,
              threads=1, want_stats=False):
        off = np.ascontiguousarray(off, np.uint64)
        n_seq = len(off) - 1
        nq = n_seq // 2 if paired else n_seq
        cand = np.zeros((nq, max_cand, 4), np.uint32)
        ncand = np.zeros(nq, np.uint32)
        stats = np.zeros(4, np.uint64)
        lib().orc_query(self.h, n_seq, bases, _p(off, u64p), 1 if paired else 0, max_cand, emulate_ranks,
                        insert_size_max, quirk_seq_drop, _p(cand, u32p), _p(ncand, u32p), _p(stats, u64p), threads)
        return (cand, ncand, stats) if want_stats else (cand, ncand)

    def reduce_query(self, locs, query_len, max_cand=2, emulate_ranks=1, insert_size_max=0, quirk_seq_drop=0):
        locs = np.ascontiguousarray(locs, np.uint64)
        out = np.zeros((max_cand, 4), np.uint32)
        n = lib().orc_reduce_query(self.h, _p(locs, u64p), len(locs), query_len, max_cand, emulate_ranks, insert_size_max,
                                   quirk_seq_drop, _p(out, u32p))
        return out, n

    def matches(self, s1, s2=b""):
        s1 = s1.encode() if isinstance(s1, str) else s1
        s2 = s2.encode() if isinstance(s2, str) else s2
        cap = 1 << 12
        while True:
            out = np.zeros(cap, np.uint64)
            n = lib().orc_query_matches(self.h, s1, len(s1), s2, len(s2), _p(out, u64p), cap)
            if n <= cap:
                return out[:n].copy()
            cap = int(n)

    def target_cands(self, s1, s2=b"", insert_size_max=0):
        s1 = s1.encode() if isinstance(s1, str) else s1
        s2 = s2.encode() if isinstance(s2, str) else s2
        cap = 1 << 10
        while True:
            out = np.zeros((cap, 4), np.uint32)
            n = lib().orc_query_target_cands(self.h, s1, len(s1), s2, len(s2), insert_size_max, _p(out, u32p), cap)
            if n <= cap:
                return out[:n].copy()
            cap = int(n)


def tree_fold(lists, max_cand, quirk_seq_drop=0):
    """lists: P lists of (tax, hits)."""
    P = len(lists)
    arr = np.zeros((P, max_cand, 2), np.uint32)
    n = np.zeros(P, np.uint32)
    for r, l in enumerate(lists):
        n[r] = len(l)
        for i, (t, h) in enumerate(l):
            arr[r, i] = (t, h)
    out = np.zeros((max_cand, 2), np.uint32)
    m = lib().orc_tree_fold(P, max_cand, _p(arr, u32p), _p(n, u32p), quirk_seq_drop, _p(out, u32p))
    return [tuple(int(x) for x in out[i]) for i in range(m)]


def insert_sequence(tax, hits, max_cand):
    """reference insert() applied to a candidate sequence; returns [(tax, hits, source index)]."""
    tax = np.ascontiguousarray(tax, np.uint32); hits = np.ascontiguousarray(hits, np.uint32)
    out = np.zeros((max_cand, 3), np.uint32)
    n = lib().orc_insert_sequence(len(tax), _p(tax, u32p), _p(hits, u32p), max_cand, _p(out, u32p))
    return [tuple(int(x) for x in out[i]) for i in range(n)]


def classify(cands, lineage, rank_of, hits_min, hits_diff_fraction, highest_rank):
    """cands: sequence of (tax_key, hits); returns taxon index or 0xFFFFFFFF."""
    c = np.ascontiguousarray(np.asarray(cands, np.uint32).reshape(-1, 2))
    lineage = np.ascontiguousarray(lineage, np.uint32)
    rank_of = np.ascontiguousarray(rank_of, np.uint8)
    return lib().orc_classify(_p(c, u32p), len(c), _p(lineage, u32p), _p(rank_of, C.POINTER(C.c_uint8)),
                              hits_min, C.c_float(hits_diff_fraction), highest_rank)
