"""ctypes mirror of include/mcq.h.  Names and argument meaning follow the header.

Device buffers are passed as raw pointers (e.g. torch.Tensor.data_ptr()); host
buffers as numpy arrays.  The HIP library is mandatory: there is no fallback.
"""
import ctypes as C
import os

import numpy as np

from .build import lib_path

MCQ_DEVICE_PTRS = 1
MCQ_QUIRK_SEQ_DROP = 2
MCQ_FOLD_BY_LISTS = 0x8000
MCQ_BATCH_RANGES = 8             # seq_off = (begin,end) pairs into `bases` (device pointers only)
MCQ_BATCH_PACKED = 0x10          # bases in the packed form of mcq_pack_bases (3 bits per base)
MCQ_FORCE_BLOCK_PATH = 0x100     # debug: send every query down the block-per-query path
MCQ_DB_LOCS_64 = 0x200           # Database(flags=...): keep 64-bit locations
MCQ_DB_LOCS_GW = 0x2000          # Database(flags=...): 32-bit locations in the global-window form
MCQ_DB_SLOTS_16 = 0x4000         # Database(flags=...): 16-B slots, every list behind the slot array
MCQ_DB_BUCKETS_64 = 0x8000       # Database(flags=...): 64-B buckets with inline lists
MCQ_LOC_FIELDS64, MCQ_LOC_FIELDS32, MCQ_LOC_GLOBAL_WINDOW = 0, 1, 2
MCQ_BUILD_REMOVE_OVERPOPULATED = 0x1000   # Table / Database.build: -remove-overpopulated-features
MCQ_FORCE_RAW_SORT = 0x400       # debug: wave path without the de-duplicating pass
MCQ_NO_WAVE16 = 0x800            # debug: 513..1024 locations take the workgroup path, not the second wave stage
MCQ_NO_TWO_CLASS = 0x4000        # debug: long match lists are sorted whole (no light / heavy split)

MCQ_OK, MCQ_E_ARG, MCQ_E_HIP, MCQ_E_CAPACITY, MCQ_E_UNSUPPORTED = 0, -1, -2, -3, -4


class McqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mcq error %d: %s" % (code, msg))
        self.code = code


class DbDesc(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sketch_size", C.c_uint32), ("winlen", C.c_uint32), ("winstride", C.c_uint32),
                ("tgt_winstride", C.c_uint32), ("n_targets", C.c_uint32), ("n_keys", C.c_uint64), ("n_locs", C.c_uint64),
                ("keys", C.c_void_p), ("list_off", C.c_void_p), ("locs", C.c_void_p), ("tgt2tax", C.c_void_p),
                ("n_shards", C.c_uint32), ("shard_id", C.c_uint32), ("flags", C.c_uint32), ("device", C.c_int32),
                ("loc_win_bits", C.c_uint32), ("tgt_windows", C.c_void_p)]


class DbLayout(C.Structure):
    _fields_ = [("loc_bytes", C.c_uint32), ("loc_format", C.c_uint32), ("win_bits", C.c_uint32), ("bucket_bytes", C.c_uint32),
                ("slots_per_key", C.c_uint32), ("n_slots", C.c_uint64), ("n_keys", C.c_uint64), ("n_locs", C.c_uint64),
                ("n_ext_locs", C.c_uint64), ("n_windows", C.c_uint64), ("bytes", C.c_uint64), ("gw_offsets", C.c_void_p)]

    def as_dict(self):
        return {k: (getattr(self, k) or 0) for k, _ in self._fields_}


class BuildDesc(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sketch_size", C.c_uint32), ("winlen", C.c_uint32), ("winstride", C.c_uint32),
                ("n_targets", C.c_uint32), ("bases", C.c_void_p), ("seq_off", C.c_void_p), ("tgt2tax", C.c_void_p),
                ("emulate_ranks", C.c_uint32), ("max_locs", C.c_uint32), ("n_shards", C.c_uint32),
                ("shard_id", C.c_uint32), ("flags", C.c_uint32), ("device", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("bases", C.c_void_p), ("seq_off", C.c_void_p),
                ("paired", C.c_uint32), ("flags", C.c_uint32), ("n_bases", C.c_uint64)]


class QueryOpts(C.Structure):
    _fields_ = [("max_cand", C.c_uint32), ("emulate_ranks", C.c_uint32), ("insert_size_max", C.c_uint64),
                ("flags", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("cands", C.c_void_p), ("n_cand", C.c_void_p), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n_features", C.c_uint64), ("n_hit_features", C.c_uint64),
                ("n_locations", C.c_uint64), ("n_cands", C.c_uint64), ("n_overflow", C.c_uint64), ("n_two_class", C.c_uint64), ("n_two_class_retry", C.c_uint64), ("n_narrow_queued", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PartsBuilderDesc(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sketch_size", C.c_uint32), ("winlen", C.c_uint32), ("winstride", C.c_uint32),
                ("tgt_winstride", C.c_uint32), ("n_targets", C.c_uint32), ("tgt_windows", C.c_void_p), ("expected_locations", C.c_uint64),
                ("n_ranges", C.c_uint32), ("n_shards", C.c_uint32), ("shard_id", C.c_uint32), ("device", C.c_int32)]


class ShardCfg(C.Structure):
    _fields_ = [("n_ranks", C.c_uint32), ("rank", C.c_uint32), ("max_queries", C.c_uint64), ("max_seqs", C.c_uint64),
                ("max_bases", C.c_uint64), ("max_locs_per_query", C.c_uint64), ("max_features_per_peer", C.c_uint64),
                ("max_locations_per_peer", C.c_uint64)]


# mcq_exchange_fn
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p,
                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32)
MCQ_SHARD_EXACT = 1
MCQ_SHARD_UNIQUE_ID_BYTES = 128

_lib = None


def lib():
    """Loads libmcq_hip.so; raises if it has not been built (no CPU fallback)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError("HIP library %s missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % p)
        L = C.CDLL(p)
        L.mcq_last_error.restype = C.c_char_p
        L.mcq_version.restype = C.c_char_p
        L.mcq_db_create.argtypes = [C.POINTER(DbDesc), C.POINTER(C.c_void_p)]
        L.mcq_db_destroy.argtypes = [C.c_void_p]
        L.mcq_db_bytes.restype = C.c_uint64; L.mcq_db_bytes.argtypes = [C.c_void_p]
        L.mcq_ws_create.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
        L.mcq_ws_destroy.argtypes = [C.c_void_p]
        L.mcq_query.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.POINTER(QueryOpts), C.POINTER(Result), C.c_void_p]
        L.mcq_ws_sync.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.mcq_query_pipelined.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.POINTER(QueryOpts), C.POINTER(Result), C.POINTER(C.c_uint64)]
        L.mcq_ws_wait.argtypes = [C.c_void_p, C.c_uint64]
        L.mcq_owner.restype = C.c_uint32; L.mcq_owner.argtypes = [C.c_uint32, C.c_uint32]
        L.mcq_debug_matches.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64]
        L.mcq_ws_timing.argtypes = [C.c_void_p, C.c_int]
        L.mcq_ws_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mcq_ws_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mcq_count_windows.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p]
        L.mcq_sketch.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcq_lookup_count.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcq_lookup_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcq_db_loc_bytes.restype = C.c_uint32; L.mcq_db_loc_bytes.argtypes = [C.c_void_p]
        L.mcq_db_win_bits.restype = C.c_uint32; L.mcq_db_win_bits.argtypes = [C.c_void_p]
        L.mcq_db_layout_get.argtypes = [C.c_void_p, C.POINTER(DbLayout)]
        L.mcq_bucket_features.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcq_assemble.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(Batch),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcq_fastq_index.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.mcq_fasta_index.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.mcq_build_table.argtypes = [C.POINTER(BuildDesc), C.POINTER(C.c_void_p)]
        L.mcq_db_build.argtypes = [C.POINTER(BuildDesc), C.POINTER(C.c_void_p)]
        L.mcq_table_info.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.mcq_table_free.argtypes = [C.c_void_p]
        L.mcq_build_last_error.restype = C.c_char_p
        L.mcq_build_parts.argtypes = [C.POINTER(BuildDesc), C.POINTER(C.c_void_p)]
        L.mcq_parts_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.mcq_db_from_parts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.mcq_parts_free.argtypes = [C.c_void_p]
        if hasattr(L, "mcq_parts_builder_create"):          # (an older build of the library under MCQ_HIP_LIB, for same-box A/B runs, lacks the r04 entry points)
            L.mcq_parts_builder_create.argtypes = [C.POINTER(PartsBuilderDesc), C.POINTER(C.c_void_p)]
            L.mcq_parts_builder_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
            L.mcq_parts_builder_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
            L.mcq_parts_builder_free.argtypes = [C.c_void_p]
        L.mcq_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(QueryOpts), C.POINTER(Result), C.c_void_p]
        L.mcq_packed_bytes.restype = C.c_uint64; L.mcq_packed_bytes.argtypes = [C.c_uint64]
        L.mcq_pack_bases.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
        L.mcq_shard_create.argtypes = [C.c_void_p, C.POINTER(ShardCfg), C.POINTER(C.c_void_p)]
        L.mcq_shard_destroy.argtypes = [C.c_void_p]
        L.mcq_shard_unique_id.argtypes = [C.c_void_p]
        L.mcq_shard_comm_rccl.argtypes = [C.c_void_p, C.c_void_p]
        L.mcq_shard_set_exchange.argtypes = [C.c_void_p, EXCHANGE_FN, C.c_void_p]
        L.mcq_shard_query.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(QueryOpts), C.POINTER(Result), C.c_void_p,
                                      C.c_uint32, C.POINTER(Batch)]
        L.mcq_shard_sync.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.mcq_shard_set_caps.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.mcq_shard_get_caps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mcq_shard_timing.argtypes = [C.c_void_p, C.c_int]
        L.mcq_shard_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mcq_shard_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.mcq_shard_exchange_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise McqError(rc, lib().mcq_last_error().decode())


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Database:
    """GPU-resident feature -> locations multimap (one shard).  Mirrors the query half
    of the reference's sketch_database: built from the union of its shard tables."""

    def __init__(self, keys, list_off, locs, tgt2tax, k=16, sketch_size=16, winlen=128, winstride=113,
                 tgt_winstride=0, n_shards=1, shard_id=0, device=0, device_ptrs=None, flags=0, loc_win_bits=0, tgt_windows=None):
        """keys/list_off/locs/tgt2tax: numpy arrays (host) -- or, with device_ptrs=dict(
        keys=ptr, list_off=ptr, locs=ptr, tgt2tax=ptr, n_keys=, n_locs=, n_targets=[, tgt_windows=ptr]), raw device pointers.
        tgt_windows (host: numpy u32 [n_targets]): windows per target for the global-window location form."""
        d = DbDesc()
        d.k, d.sketch_size, d.winlen, d.winstride, d.tgt_winstride = k, sketch_size, winlen, winstride, tgt_winstride
        d.n_shards, d.shard_id, d.device = n_shards, shard_id, device
        d.loc_win_bits = loc_win_bits
        if device_ptrs is None:
            self._keep = (np.ascontiguousarray(keys, np.uint32), np.ascontiguousarray(list_off, np.uint64),
                          np.ascontiguousarray(locs, np.uint64), np.ascontiguousarray(tgt2tax, np.uint32))
            kk, oo, ll, tt = self._keep
            assert len(oo) == len(kk) + 1
            d.n_keys, d.n_locs, d.n_targets = len(kk), len(ll), len(tt)
            d.keys, d.list_off, d.locs, d.tgt2tax = _np_ptr(kk), _np_ptr(oo), _np_ptr(ll), _np_ptr(tt)
            if tgt_windows is not None:
                tw = np.ascontiguousarray(tgt_windows, np.uint32)
                assert len(tw) == len(tt)
                self._keep = self._keep + (tw,)
                d.tgt_windows = _np_ptr(tw)
            d.flags = flags
        else:
            p = device_ptrs
            d.n_keys, d.n_locs, d.n_targets = p["n_keys"], p["n_locs"], p["n_targets"]
            d.keys, d.list_off, d.locs, d.tgt2tax = p["keys"], p["list_off"], p["locs"], p["tgt2tax"]
            d.tgt_windows = p.get("tgt_windows")
            d.flags = MCQ_DEVICE_PTRS | flags
        self.k, self.sketch_size, self.winlen, self.winstride = k, sketch_size, winlen, winstride
        self.device = device
        h = C.c_void_p()
        _chk(lib().mcq_db_create(C.byref(d), C.byref(h)))
        self.h = h

    @classmethod
    def build(cls, bases_ptr, seq_off_ptr, tgt2tax_ptr, n_targets, emulate_ranks=1, k=16, sketch_size=16, winlen=128,
              winstride=113, max_locs=0, n_shards=1, shard_id=0, device=0, flags=0, device_ptrs=True):
        """mcq_db_build: reference sequences -> queryable handle, all on the GPU"""
        d = _build_desc(bases_ptr, seq_off_ptr, tgt2tax_ptr, n_targets, emulate_ranks, k, sketch_size, winlen, winstride,
                        max_locs, n_shards, shard_id, device, flags, device_ptrs)
        self = cls.__new__(cls)
        self.k, self.sketch_size, self.winlen, self.winstride, self.device = k, sketch_size, winlen, winstride, device
        h = C.c_void_p()
        rc = lib().mcq_db_build(C.byref(d), C.byref(h))
        if rc != 0:
            raise McqError(rc, (lib().mcq_build_last_error() or b"").decode())
        self.h = h
        return self

    def bytes(self):
        return int(lib().mcq_db_bytes(self.h))

    # ---- staged entry points (device pointers only) -------------------------------
    def count_windows(self, bases_ptr, seq_off_ptr, n_seqs, win_off_ptr, stream=None):
        b = Batch(n_seqs, bases_ptr, seq_off_ptr, 0, MCQ_DEVICE_PTRS)
        _chk(lib().mcq_count_windows(self.h, C.byref(b), win_off_ptr, stream))

    def sketch(self, bases_ptr, seq_off_ptr, n_seqs, win_off_ptr, features_ptr, n_feat_ptr, stream=None):
        b = Batch(n_seqs, bases_ptr, seq_off_ptr, 0, MCQ_DEVICE_PTRS)
        _chk(lib().mcq_sketch(self.h, C.byref(b), win_off_ptr, features_ptr, n_feat_ptr, stream))

    def lookup_count(self, features_ptr, n, list_len_ptr, list_src_ptr=None, stream=None):
        _chk(lib().mcq_lookup_count(self.h, features_ptr, n, list_len_ptr, list_src_ptr, stream))

    def lookup_gather(self, features_ptr, n, out_off_ptr, out_locs_ptr, list_len_ptr=None, list_src_ptr=None, stream=None):
        """out_locs in the handle's native width (loc_bytes())"""
        _chk(lib().mcq_lookup_gather(self.h, features_ptr, n, list_len_ptr, list_src_ptr, out_off_ptr, out_locs_ptr, stream))

    def loc_bytes(self):
        return int(lib().mcq_db_loc_bytes(self.h))

    def win_bits(self):
        return int(lib().mcq_db_win_bits(self.h))

    def layout(self):
        """what the handle was built as: location format / width, bucket size, load factor, sizes (mcq_db_layout)"""
        lo = DbLayout()
        _chk(lib().mcq_db_layout_get(self.h, C.byref(lo)))
        return lo.as_dict()

    def assemble(self, n_lists, list_len_ptr, src_slot_ptr, n_slots, src_locs_ptr, bases_ptr, seq_off_ptr, n_seqs, paired,
                 win_off_ptr, loc_off_ptr, query_len_ptr, dst_locs_ptr, stream=None):
        b = Batch(n_seqs, bases_ptr, seq_off_ptr, 1 if paired else 0, MCQ_DEVICE_PTRS)
        _chk(lib().mcq_assemble(self.h, n_lists, list_len_ptr, src_slot_ptr, n_slots, src_locs_ptr, C.byref(b), win_off_ptr,
                                loc_off_ptr, query_len_ptr, dst_locs_ptr, stream))

    def close(self):
        if getattr(self, "h", None):
            lib().mcq_db_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass


class Workspace:
    def __init__(self, db, max_queries, max_bases, max_locs_per_query=0):
        self.db = db
        self.max_queries = max_queries
        h = C.c_void_p()
        _chk(lib().mcq_ws_create(db.h, max_queries, max_bases, max_locs_per_query, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().mcq_ws_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def sync(self, stream=None):
        st = Stats()
        _chk(lib().mcq_ws_sync(self.h, stream, C.byref(st)))
        return st.as_dict()

    # ---- host-buffer call: numpy in, numpy out (copies + sync inside) -------------
    def query_host(self, bases, seq_off, paired, max_cand=2, emulate_ranks=1, insert_size_max=0, flags=0, packed=False):
        """packed: `bases` holds the packed form (pack_bases) of the batch instead of ASCII"""
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        n_seqs = len(seq_off) - 1
        nq = n_seqs // 2 if paired else n_seqs
        bases_arr = np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases
        bases_arr = np.ascontiguousarray(bases_arr)
        b = Batch(n_seqs, _np_ptr(bases_arr) if len(bases_arr) else None, _np_ptr(seq_off), 1 if paired else 0,
                  MCQ_BATCH_PACKED if packed else 0, int(seq_off[-1]) if packed else 0)
        o = QueryOpts(max_cand, emulate_ranks, insert_size_max, flags)
        cands = np.zeros((max(nq, 1), max_cand, 4), np.uint32)
        ncand = np.zeros(max(nq, 1), np.uint32)
        r = Result(_np_ptr(cands), _np_ptr(ncand), 0)
        _chk(lib().mcq_query(self.db.h, self.h, C.byref(b), C.byref(o), C.byref(r), None))
        return cands[:nq], ncand[:nq]

    # ---- device-buffer call: raw pointers, enqueue only --------------------------
    def query_device(self, bases_ptr, seq_off_ptr, n_seqs, paired, cands_ptr, ncand_ptr, max_cand=2,
                     emulate_ranks=1, insert_size_max=0, flags=0, stream=None, ranges=False, packed_bases=0):
        """packed_bases > 0: bases_ptr is the packed form of a batch of that many bases"""
        b = Batch(n_seqs, bases_ptr, seq_off_ptr, 1 if paired else 0,
                  MCQ_DEVICE_PTRS | (MCQ_BATCH_RANGES if ranges else 0) | (MCQ_BATCH_PACKED if packed_bases else 0), packed_bases)
        o = QueryOpts(max_cand, emulate_ranks, insert_size_max, flags)
        r = Result(cands_ptr, ncand_ptr, MCQ_DEVICE_PTRS)
        _chk(lib().mcq_query(self.db.h, self.h, C.byref(b), C.byref(o), C.byref(r), stream))

    # ---- host buffers, pipelined: raw HOST pointers (pinned), returns a ticket for wait()
    def query_pipelined(self, bases_ptr, seq_off_ptr, n_seqs, paired, cands_ptr, ncand_ptr, max_cand=2, emulate_ranks=1,
                        insert_size_max=0, flags=0, packed_bases=0):
        b = Batch(n_seqs, bases_ptr, seq_off_ptr, 1 if paired else 0, MCQ_BATCH_PACKED if packed_bases else 0, packed_bases)
        o = QueryOpts(max_cand, emulate_ranks, insert_size_max, flags)
        r = Result(cands_ptr, ncand_ptr, 0)
        t = C.c_uint64(0)
        _chk(lib().mcq_query_pipelined(self.db.h, self.h, C.byref(b), C.byref(o), C.byref(r), C.byref(t)))
        return int(t.value)

    def wait(self, ticket):
        _chk(lib().mcq_ws_wait(self.h, ticket))

    def timing(self, enable):
        _chk(lib().mcq_ws_timing(self.h, 1 if enable else 0))

    def kernel_time(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        _chk(lib().mcq_ws_kernel_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    def phase_clocks(self):
        """diagnostic builds (-DMCQ_PHASE_CLOCK): shader clocks per phase of the workgroup kernel of the last synchronised call"""
        out = (C.c_uint64 * 22)()
        if hasattr(lib(), "mcq_debug_phase_clocks"):
            _chk(lib().mcq_debug_phase_clocks(self.h, out))
        return [int(x) for x in out]

    def kernel_times(self):
        """(ms of first wave stage, second wave stage, workgroup kernel summed over the timed batches; n batches)"""
        ms, n = (C.c_double * 3)(), C.c_uint64(0)
        _chk(lib().mcq_ws_kernel_times(self.h, ms, C.byref(n)))
        return [float(x) for x in ms], int(n.value)

    def reduce_device(self, n_queries, loc_off_ptr, locs_ptr, query_len_ptr, cands_ptr, ncand_ptr, max_cand=2,
                      emulate_ranks=1, insert_size_max=0, flags=0, stream=None):
        o = QueryOpts(max_cand, emulate_ranks, insert_size_max, flags)
        r = Result(cands_ptr, ncand_ptr, MCQ_DEVICE_PTRS)
        _chk(lib().mcq_reduce(self.db.h, self.h, n_queries, loc_off_ptr, locs_ptr, query_len_ptr, C.byref(o), C.byref(r), stream))

    def debug_matches(self, bases, seq_off, paired, path_flags=0):
        """sorted match list of every query, tapped on the path the query takes (path_flags = 0) or the one a
        test hook (MCQ_FORCE_BLOCK_PATH, MCQ_FORCE_RAW_SORT, MCQ_NO_WAVE16) selects"""
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        n_seqs = len(seq_off) - 1
        nq = n_seqs // 2 if paired else n_seqs
        bases_arr = np.ascontiguousarray(np.frombuffer(bases, dtype=np.uint8))
        b = Batch(n_seqs, _np_ptr(bases_arr) if len(bases_arr) else None, _np_ptr(seq_off), 1 if paired else 0, 0)
        moff = np.zeros(nq + 1, np.uint64)
        _chk(lib().mcq_debug_matches(self.db.h, self.h, C.byref(b), path_flags, _np_ptr(moff), None, 0))
        m = np.zeros(max(1, int(moff[nq])), np.uint64)
        _chk(lib().mcq_debug_matches(self.db.h, self.h, C.byref(b), path_flags, _np_ptr(moff), _np_ptr(m), len(m)))
        return moff, m[:int(moff[nq])]


class Shard:
    """mcq_shard_*: the feature-sharded multi-GPU path of one rank (one process per GPU).  `db` is this rank's shard
    (Database(..., n_shards=n_ranks, shard_id=rank)).  All calls are collective."""

    def __init__(self, db, n_ranks, rank, max_queries, max_bases, max_seqs=0, max_locs_per_query=0,
                 max_features_per_peer=0, max_locations_per_peer=0):
        self.db, self.n_ranks, self.rank = db, n_ranks, rank
        cfg = ShardCfg(n_ranks, rank, max_queries, max_seqs, max_bases, max_locs_per_query, max_features_per_peer,
                       max_locations_per_peer)
        h = C.c_void_p()
        _chk(lib().mcq_shard_create(db.h, C.byref(cfg), C.byref(h)))
        self.h = h
        self._xfn = None

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(MCQ_SHARD_UNIQUE_ID_BYTES)
        _chk(lib().mcq_shard_unique_id(buf))
        return buf.raw

    def comm_rccl(self, unique_id):
        assert len(unique_id) == MCQ_SHARD_UNIQUE_ID_BYTES
        _chk(lib().mcq_shard_comm_rccl(self.h, C.create_string_buffer(unique_id, MCQ_SHARD_UNIQUE_ID_BYTES)))

    def set_exchange(self, fn):
        """fn: an EXCHANGE_FN (kept alive here)"""
        self._xfn = fn
        _chk(lib().mcq_shard_set_exchange(self.h, fn, None))

    def query(self, bases_ptr, seq_off_ptr, n_seqs, paired, cands_ptr, ncand_ptr, max_cand=2, emulate_ranks=1,
              insert_size_max=0, flags=0, stream=None, exact=False, next_batch=None):
        """next_batch = (bases_ptr, seq_off_ptr, n_seqs) of the following call (resident): sketched under this batch's exchange"""
        b = Batch(n_seqs, bases_ptr, seq_off_ptr, 1 if paired else 0, MCQ_DEVICE_PTRS)
        o = QueryOpts(max_cand, emulate_ranks, insert_size_max, flags)
        r = Result(cands_ptr, ncand_ptr, MCQ_DEVICE_PTRS)
        nb = None
        if next_batch is not None:
            nb = C.byref(Batch(next_batch[2], next_batch[0], next_batch[1], 1 if paired else 0, MCQ_DEVICE_PTRS))
        _chk(lib().mcq_shard_query(self.h, C.byref(b), C.byref(o), C.byref(r), stream, MCQ_SHARD_EXACT if exact else 0, nb))

    def sync(self, stream=None):
        st = Stats()
        _chk(lib().mcq_shard_sync(self.h, stream, C.byref(st)))
        return st.as_dict()

    def caps(self):
        f, l = C.c_uint64(0), C.c_uint64(0)
        _chk(lib().mcq_shard_get_caps(self.h, C.byref(f), C.byref(l)))
        return int(f.value), int(l.value)

    def set_caps(self, features_per_peer, locations_per_peer):
        _chk(lib().mcq_shard_set_caps(self.h, features_per_peer, locations_per_peer))

    def timing(self, enable):
        _chk(lib().mcq_shard_timing(self.h, 1 if enable else 0))

    def kernel_times(self):
        ms, n = (C.c_double * 3)(), C.c_uint64(0)
        _chk(lib().mcq_shard_kernel_times(self.h, ms, C.byref(n)))
        return [float(x) for x in ms], int(n.value)

    def stage_times(self):
        """({'S1': ms, 'X1': ms, 'S2': ms, 'X2': ms} summed over the batches, n batches)"""
        ms, n = (C.c_double * 4)(), C.c_uint64(0)
        _chk(lib().mcq_shard_stage_times(self.h, ms, C.byref(n)))
        return dict(zip(("S1", "X1", "S2", "X2"), [float(x) for x in ms])), int(n.value)

    def exchange_bytes(self):
        out = (C.c_uint64 * 8)()
        _chk(lib().mcq_shard_exchange_bytes(self.h, out))
        return dict(zip(("batches", "x1", "x2_ends", "x2_locations", "own_blocks", "rccl_ranks", "block_features", "block_locations"), [int(x) for x in out]))

    def close(self):
        if getattr(self, "h", None):
            lib().mcq_shard_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def packed_bytes(n_bases):
    return int(lib().mcq_packed_bytes(n_bases))


def pack_bases_host(bases):
    """ASCII bytes -> the packed form (numpy uint8 array of packed_bytes(len) bytes)"""
    src = np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else np.ascontiguousarray(bases, np.uint8)
    out = np.zeros(packed_bytes(len(src)), np.uint8)
    _chk(lib().mcq_pack_bases(_np_ptr(src) if len(src) else None, len(src), _np_ptr(out), 0, None))
    return out


def pack_bases_device(bases_ptr, n_bases, out_ptr, stream=None):
    _chk(lib().mcq_pack_bases(bases_ptr, n_bases, out_ptr, MCQ_DEVICE_PTRS, stream))


def bucket_features(features_ptr, n, n_shards, counts_ptr, bucketed_ptr, src_index_ptr, stream=None):
    _chk(lib().mcq_bucket_features(features_ptr, n, n_shards, counts_ptr, bucketed_ptr, src_index_ptr, stream))


def _build_desc(bases_ptr, seq_off_ptr, tgt2tax_ptr, n_targets, emulate_ranks, k, sketch_size, winlen, winstride,
                max_locs, n_shards, shard_id, device, flags, device_ptrs):
    d = BuildDesc()
    d.k, d.sketch_size, d.winlen, d.winstride, d.n_targets = k, sketch_size, winlen, winstride, n_targets
    d.bases, d.seq_off, d.tgt2tax = bases_ptr, seq_off_ptr, tgt2tax_ptr
    d.emulate_ranks, d.max_locs, d.n_shards, d.shard_id, d.device = emulate_ranks, max_locs, n_shards, shard_id, device
    d.flags = flags | (MCQ_DEVICE_PTRS if device_ptrs else 0)
    return d


class Table:
    """mcq_build_table: keys / list_off / locs of the union table as device arrays"""

    def __init__(self, bases_ptr, seq_off_ptr, n_targets, emulate_ranks=1, k=16, sketch_size=16, winlen=128, winstride=113,
                 max_locs=0, device=0, device_ptrs=True, flags=0):
        d = _build_desc(bases_ptr, seq_off_ptr, None, n_targets, emulate_ranks, k, sketch_size, winlen, winstride,
                        max_locs, 1, 0, device, flags, device_ptrs)
        h = C.c_void_p()
        rc = lib().mcq_build_table(C.byref(d), C.byref(h))
        if rc != 0:
            raise McqError(rc, (lib().mcq_build_last_error() or b"").decode())
        self.h, self.n_targets, self.device = h, n_targets, device
        nk, nl = C.c_uint64(), C.c_uint64()
        pk, po, pl, pw = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(lib().mcq_table_info(h, C.cast(C.byref(nk), C.c_void_p), C.cast(C.byref(nl), C.c_void_p),
                                  C.cast(C.byref(pk), C.c_void_p), C.cast(C.byref(po), C.c_void_p),
                                  C.cast(C.byref(pl), C.c_void_p), C.cast(C.byref(pw), C.c_void_p)))
        self.n_keys, self.n_locs = int(nk.value), int(nl.value)
        self.keys_ptr, self.list_off_ptr, self.locs_ptr, self.win_off_ptr = pk.value, po.value, pl.value, pw.value

    def to_host(self):
        """(keys u32, list_off u64, locs u64, win_off u64) as numpy arrays (hipMemcpy device -> host)"""
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def pull(ptr, n, dt):
            out = np.empty(n, dt)
            if n and hip.hipMemcpy(_np_ptr(out), ptr, out.nbytes, 2) != 0:      # hipMemcpyDeviceToHost
                raise McqError(MCQ_E_HIP, "hipMemcpy of a table array failed")
            return out
        return (pull(self.keys_ptr, self.n_keys, np.uint32), pull(self.list_off_ptr, self.n_keys + 1, np.uint64),
                pull(self.locs_ptr, self.n_locs, np.uint64), pull(self.win_off_ptr, self.n_targets + 1, np.uint64))

    def close(self):
        if getattr(self, "h", None):
            lib().mcq_table_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Parts:
    """mcq_build_parts: the table built in feature-hash ranges and kept as keys / list lengths / 32-bit global-window words
    (device memory) -- for tables whose one-piece build does not fit (RefSeq scale).  database() makes the queryable
    handle (any shard of it); the sequences may be released before that."""

    def __init__(self, bases_ptr, seq_off_ptr, n_targets, emulate_ranks=1, k=16, sketch_size=16, winlen=128, winstride=113,
                 max_locs=0, n_shards=1, shard_id=0, device=0, flags=0):
        d = _build_desc(bases_ptr, seq_off_ptr, None, n_targets, emulate_ranks, k, sketch_size, winlen, winstride,
                        max_locs, n_shards, shard_id, device, flags, True)
        h = C.c_void_p()
        rc = lib().mcq_build_parts(C.byref(d), C.byref(h))
        if rc != 0:
            raise McqError(rc, (lib().mcq_build_last_error() or b"").decode())
        self.h, self.n_targets, self.device = h, n_targets, device
        self.k, self.sketch_size, self.winlen, self.winstride = k, sketch_size, winlen, winstride
        nk, nl, nw, nb = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        npar = C.c_uint32()
        _chk(lib().mcq_parts_info(h, C.byref(nk), C.byref(nl), C.byref(nw), C.byref(npar), C.byref(nb)))
        self.n_keys, self.n_locs, self.n_windows, self.n_parts, self.bytes = int(nk.value), int(nl.value), int(nw.value), int(npar.value), int(nb.value)

    def database(self, tgt2tax_ptr, n_shards=1, shard_id=0, flags=0, device_ptrs=True):
        db = Database.__new__(Database)
        db.k, db.sketch_size, db.winlen, db.winstride, db.device = self.k, self.sketch_size, self.winlen, self.winstride, self.device
        h = C.c_void_p()
        rc = lib().mcq_db_from_parts(self.h, tgt2tax_ptr, n_shards, shard_id, flags | (MCQ_DEVICE_PTRS if device_ptrs else 0), C.byref(h))
        if rc != 0:
            raise McqError(rc, (lib().mcq_build_last_error() or b"").decode())
        db.h = h
        return db

    def close(self):
        if getattr(self, "h", None):
            lib().mcq_parts_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PartsBuilder:
    """mcq_parts_builder_*: table parts from streamed (feature, target, window) triples -- the reference's shard files without a
    host-side union (host.RefDb(meta_only=True).stream() feeds it).  finish() returns a Parts object."""

    def __init__(self, tgt_windows, k=16, sketch_size=16, winlen=128, winstride=113, tgt_winstride=0, expected_locations=0, n_ranges=0,
                 n_shards=1, shard_id=0, device=0):
        tw = np.ascontiguousarray(tgt_windows, np.uint32)
        d = PartsBuilderDesc(k, sketch_size, winlen, winstride, tgt_winstride, len(tw), tw.ctypes.data_as(C.c_void_p), expected_locations,
                             n_ranges, n_shards, shard_id, device)
        h = C.c_void_p()
        rc = lib().mcq_parts_builder_create(C.byref(d), C.byref(h))
        if rc != 0:
            raise McqError(rc, (lib().mcq_build_last_error() or b"").decode())
        self.h, self.n_targets, self.device = h, len(tw), device
        self.k, self.sketch_size, self.winlen, self.winstride = k, sketch_size, winlen, winstride

    def add(self, feat, tgt, win):
        f, t, w = (np.ascontiguousarray(x, np.uint32) for x in (feat, tgt, win))
        rc = lib().mcq_parts_builder_add(self.h, f.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), len(f), 0)
        if rc != 0:
            raise McqError(rc, (lib().mcq_build_last_error() or b"").decode())

    def finish(self):
        ph = C.c_void_p()
        rc = lib().mcq_parts_builder_finish(self.h, C.byref(ph))
        self.h = None if rc == 0 else self.h
        if rc != 0:
            raise McqError(rc, (lib().mcq_build_last_error() or b"").decode())
        p = Parts.__new__(Parts)
        p.h, p.n_targets, p.device = ph, self.n_targets, self.device
        p.k, p.sketch_size, p.winlen, p.winstride = self.k, self.sketch_size, self.winlen, self.winstride
        nk, nl, nw, nb = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        npar = C.c_uint32()
        _chk(lib().mcq_parts_info(ph, C.byref(nk), C.byref(nl), C.byref(nw), C.byref(npar), C.byref(nb)))
        p.n_keys, p.n_locs, p.n_windows, p.n_parts, p.bytes = int(nk.value), int(nl.value), int(nw.value), int(npar.value), int(nb.value)
        return p

    def close(self):
        if getattr(self, "h", None):
            lib().mcq_parts_builder_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fastq_index(text_ptr, n_bytes, ranges_ptr, max_seqs, n_seqs_ptr, stream=None):
    """raw FASTQ text in HBM -> (begin,end) ranges of the sequence lines (device buffers)"""
    _chk(lib().mcq_fastq_index(text_ptr, n_bytes, ranges_ptr, max_seqs, n_seqs_ptr, stream))


def fasta_index(text_ptr, n_bytes, ranges_ptr, max_seqs, n_seqs_ptr, stream=None):
    """the same for FASTA text with one sequence line per record"""
    _chk(lib().mcq_fasta_index(text_ptr, n_bytes, ranges_ptr, max_seqs, n_seqs_ptr, stream))


def owner(feature, n_shards):
    return int(lib().mcq_owner(feature, n_shards))
