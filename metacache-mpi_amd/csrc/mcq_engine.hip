// mcq_engine.hip -- kernels + C ABI of the MI355X query-path engine (see include/mcq.h).
//
// Three kernels carry the whole per-query path (rows 1-11 of SURVEY.md 8a):
//
//   k_query_wave    one wavefront per query.  Sketch (<= 4 windows), 64 parallel table
//                   probes, list gather into registers, distinct-key counting in the
//                   wave's LDS segment, register sort, per-target window sweep, top
//                   lists + tree fold.  Per read it touches the bases, one 16-B slot per probe, the
//                   location lists once and the candidates out (the algorithmic bytes); the HBM traffic
//                   is 2.3 x that, because slots and short lists come in 64-B sectors.
//   k_query_wave16  second wave stage: queries of 513..1024 locations (32-bit keys),
//                   16 keys per lane.
//   k_query_block   one 1024-thread workgroup per query that fits neither (long reads
//                   with many windows, longer lists): same steps with workgroup
//                   barriers, LDS up to 8192 locations, global scratch beyond.
//
// Queries that overflow the first stage are queued through device counters (two queues in
// one array); the other kernels drain them, so no host round trip sits inside a batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <type_traits>

#include "../../include/mcq.h"
#include "mcq_device.hpp"

using namespace mcq;

// ------------------------------------------------------------------ error handling
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(MCQ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// like HIPCHK inside a constructor-like function: releases what the half-built object already holds before returning
#define HIPCHK_OR(expr, cleanup) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup; \
    return fail(MCQ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)

extern "C" const char* mcq_last_error(void) { return g_err.c_str(); }
extern "C" const char* mcq_version(void) { return "mcq 0.1 (gfx950)"; }

// ------------------------------------------------------------------ persistent grids
// The fused query kernel loops over its queries (grid stride).  Its grid is exactly what the device holds at
// once -- occupancy x CUs -- and no more: 2.26 ms at 24 workgroups per CU, 2.07 ms at the resident 8 (long items,
// all waves alive from start to end).  The staged kernels with short items keep several rounds of workgroups.
template <class Kernel>
static u32 resident_blocks(Kernel kernel, int block_size, int device) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_size, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount < 1) prop.multiProcessorCount = 256;
    if (const char* e = getenv("MCQ_WAVE_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(e));       // tuning knob
    return std::min<u32>((u32)per_cu * (u32)prop.multiProcessorCount, 8192u);     // 4 x 8192 waves: what MCQ_OVF_TAIL covers
}
// (computed once per workspace for its device and kept in the handle: nothing static, so two devices or two
// threads in one process do not disturb each other)
static u32 grid_for(u32 cap, u64 want) { return (u32)std::min<u64>(want ? want : 1, cap ? cap : 1); }

// ------------------------------------------------------------------ handles
struct mcq_db {
    DbDev d;
    int device;
    u64 nslots;
    u64 n_keys_local, n_locs_local;
    uint4* slots;             // one allocation: the buckets, then the lists too long for a bucket
    u32* tgt2tax;
    u32* gw_off; u32* gw_blk; // global-window form: first window of every target, block -> target (see LocGW)
    GwDev g;                  // ... as the kernels take them
    u32 n_shards, shard_id;
    u32 bucket_bytes, slots_per_key;
    u64 n_ext, n_windows;
    u64 bytes;
    bool seq_taxa;            // tgt2tax holds sequence-level taxa (bit 31; see make_opt)
    u64 fmt_sig;              // what the location words of this handle mean (format, field widths, window offsets of the targets, sketch
                              // parameters), hashed: the ranks of a sharded run compare it before the first words travel (mcq_shard.hpp)
};

struct ScratchDev {
    u32* feat; u32* fpos; u64* foff; u64* gbuf; u64* ghits;
    u32 fmax; u32 lmax;
};

struct DebugDev {
    int mode;                 // 0 off, 1 = write match counts, 2 = write matches
    u64* match_cnt;           // [nq]
    const u64* match_off;     // [nq+1]
    u64* matches;
};

#define MCQ_N_TIMED 3           // kernels of one batch that are timed separately: first wave stage, second wave stage, workgroup kernel
struct TimedLaunch { hipEvent_t ev[MCQ_N_TIMED + 1]; };
struct mcq_ws {
    int device;
    u64 max_queries, max_bases;
    CountersDev* ctr;         // device
    CountersDev* ctr_host;    // pinned
    u32* ovf_list;            // [ovf_capacity(max_queries)]
    unsigned long long* probe_buf;   // [(2 x max_queries + 3 x MCQ_OVF_TAIL) x 64]: rows of the back queue, then of the front queue; see CountersDev
    ScratchDev sc;
    int n_block_wgs;
    u32 cap_wave, cap_wave16, cap_reduce16, cap_wave32, cap_wave_many;   // resident workgroups of the wave-per-query kernels on this device
    // staging for host-pointer calls
    char* d_bases; u64* d_seq_off; u32* d_cands; u32* d_ncand;
    u64 last_nq;
    // host-buffer pipeline (mcq_query_pipelined): two staging sets, copy streams on both sides of the compute stream
    struct Pipe {
        char* d_bases[2]; u64* d_seq_off[2]; u32* d_cands[2]; u32* d_ncand[2];
        hipStream_t s_in, s_k, s_out;
        hipEvent_t ev_in[2], ev_k[2], ev_out[2];
        u64 issued;             // calls so far; call i uses set i & 1
        bool ready;
    } pipe;
    // optional per-launch timing of the path's kernels (events between them on the call's stream)
    int timing;
    std::vector<TimedLaunch>* ev_used;
    std::vector<TimedLaunch>* ev_free;
    double timed_ms[MCQ_N_TIMED]; u64 timed_launches;
};

// ------------------------------------------------------------------ kernels: table build
__global__ void k_fill_slots(uint4* slots, u64 n_uint4) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_uint4; i += stride) slots[i] = make_uint4(MCQ_EMPTY, 0, 0, 0);
}

// A table is handed over in one piece (mcq_db_create: offsets + public 64-bit locations) or in parts (mcq_db_create_parts:
// list lengths + locations that are global-window words already).  PartView is what the build kernels see of either.
struct PartView {
    u64 n_keys, n_locs;
    const u32* keys;
    const u64* off;           // [n_keys + 1] exclusive offsets of the lists inside `locs`
    const void* locs;         // u64 (tgt << 32) | win, or (gw_words) u32 global window indices
    u32 gw_words;
};
// source location i of a part -> the handle's native word: bit fields (tgt << wb) | win, or, with gw_off, the global window index
// gw_off[tgt] + win; a source that holds global-window words already is copied (the handle then keeps that form)
template <class KeyT>
__device__ __forceinline__ KeyT loc_native(const PartView& pv, u64 i, u32 wb, const u32* __restrict__ gw_off) {
    if (pv.gw_words) return (KeyT)static_cast<const u32*>(pv.locs)[i];
    const u64 l = static_cast<const u64*>(pv.locs)[i];
    if (sizeof(KeyT) == 4 && gw_off) return (KeyT)(gw_off[(u32)(l >> 32)] + (u32)l);
    return (KeyT)(((l >> 32) << wb) | (l & 0xFFFFFFFFull));
}
// one thread per key: claim a bucket with CAS on the key word, then fill it: length, and either the list itself
// (64-B buckets: up to 14 compact / 7 wide locations) or the offset of the list among the long ones (ext_off + ext_base).
// bq = uint4 per bucket (4 or 1); inl = longest inline list (0 with 16-B slots)
template <class KeyT>
__global__ void k_insert_keys(uint4* slots, u32 mask, u32 bq, u32 inl, PartView pv, const u64* own_len, const u64* ext_off, u64 ext_base,
                              u32 wb, const u32* gw_off) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pv.n_keys) return;
    const u32 len = (u32)own_len[i];
    if (len == 0) return;                                 // foreign or empty
    const u32 key = pv.keys[i];
    u32 idx = tmh(key) & mask;
    while (true) {
        u32* w = reinterpret_cast<u32*>(&slots[(u64)idx * bq]);
        const u32 prev = atomicCAS(w, MCQ_EMPTY, key);
        if (prev == MCQ_EMPTY) {
            w[1] = len;
            if (len <= inl) {
                KeyT* dst = reinterpret_cast<KeyT*>(w + 2);
                const u64 src = pv.off[i];
                for (u32 t = 0; t < len; ++t) dst[t] = loc_native<KeyT>(pv, src + t, wb, gw_off);
            } else { const u64 b = ext_base + ext_off[i]; w[2] = (u32)b; w[3] = (u32)(b >> 32); }
            return;
        }
        idx = (idx + 1) & mask;
    }
}

// list length per key if owned by this shard, else 0
__global__ void k_owned_len(PartView pv, u32 n_shards, u32 shard_id, u64* out_len) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pv.n_keys) return;
    u32 own = (u32)(((u64)tmh(pv.keys[i]) * n_shards) >> 32);
    out_len[i] = (own == shard_id) ? (pv.off[i + 1] - pv.off[i]) : 0;
}
// owned non-empty keys, owned locations, and of those the ones in lists longer than inl64 (what a 64-B bucket cannot hold): totals[3]
__global__ void k_owned_totals(const u64* own_len, u64 n_keys, u32 inl64, unsigned long long* totals) {
    unsigned long long k = 0, l = 0, x = 0;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += stride) { const u64 v = own_len[i]; k += v > 0; l += v; x += v > inl64 ? v : 0; }
    for (int d = 32; d > 0; d >>= 1) { k += __shfl_xor(k, d, 64); l += __shfl_xor(l, d, 64); x += __shfl_xor(x, d, 64); }
    if ((threadIdx.x & 63) == 0 && l) { atomicAdd(&totals[0], k); atomicAdd(&totals[1], l); if (x) atomicAdd(&totals[2], x); }
}
__global__ void k_ext_len(const u64* own_len, u64 n_keys, u32 inline_max, u64* ext_len) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_keys) { const u64 len = own_len[i]; ext_len[i] = len > inline_max ? len : 0; }
}

// copy the owned long lists behind the buckets
template <class KeyT>
__global__ void k_copy_lists(PartView pv, const u64* ext_off, KeyT* out, u32 wb, const u32* gw_off) {
    // one wave per key, grid-stride (the grid is bounded: total threads must stay < 2^32)
    const u32 lane = threadIdx.x & 63;
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 key = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; key < pv.n_keys; key += nwaves) {
        u64 b = ext_off[key], n = ext_off[key + 1] - b, src = pv.off[key];
        for (u64 t = lane; t < n; t += 64) out[b + t] = loc_native<KeyT>(pv, src + t, wb, gw_off);
    }
}

// ---- global-window form: extents of the targets, offsets, block table
// ext[t] = 1 + largest window id of target t among the locations (a racy read first: the maximum only grows, and most
// locations lose against it without an atomic)
__global__ void k_tgt_extent(const u64* locs, u64 n, u32 n_targets, u32* ext) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 l = locs[i];
        const u32 t = (u32)(l >> 32), w = (u32)l;
        if (t < n_targets && w != 0xFFFFFFFFu && *reinterpret_cast<volatile const u32*>(&ext[t]) <= w) atomicMax(&ext[t], w + 1);
    }
}
// gw_blk[b] = (last target t with gw_off[t] <= b << shift, gw_off[t]) (targets without windows are skipped)
__global__ void k_gw_blocks(const u32* gw_off, u32 n_targets, u32 shift, u64 n_blk, uint2* blk) {
    const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    const u64 w = b << shift;
    u32 lo = 0, hi = n_targets ? n_targets - 1 : 0;
    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if ((u64)gw_off[mid] <= w) lo = mid; else hi = mid - 1; }
    blk[b] = make_uint2(lo, gw_off[lo]);
}
__global__ void k_u64_to_u32(const u64* in, u32* out, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (u32)in[i];
}

// largest window id over all locations (decides whether locations fit 32 bits)
__global__ void k_max_win(const u64* locs, u64 n, u32* out) {
    u32 m = 0;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { u32 w = (u32)locs[i]; m = w > m ? w : m; }
    for (int d = 32; d > 0; d >>= 1) { u32 o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// exclusive scan of u64 array (single workgroup of 256 or 1024 threads)
template <class InT>
__global__ __launch_bounds__(1024) void k_scan_u64(const InT* in, u64* out, u64 n) {
    __shared__ u64 s_w[16];
    __shared__ u64 s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < n; base += NT) {
        u64 i = base + tid;
        u64 v = (i < n) ? in[i] : 0, x = v;
        for (int d = 1; d < 64; d <<= 1) {
            u64 t = __shfl_up(x, d, 64);
            if (lane >= (u32)d) x += t;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        u64 woff = 0;
        for (u32 w = 0; w < wave; ++w) woff += s_w[w];
        u64 carry = s_carry;
        if (i < n) out[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == NT - 1) s_carry = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) out[n] = s_carry;
}

// exclusive scan of n u64 values in three launches: per-tile scan + tile sums, scan of the
// tile sums (one workgroup), add.  out has n + 1 entries (out[n] = total).
#define MCQ_SCAN_TILE 8192
template <class InT>
__global__ __launch_bounds__(1024) void k_scan_tiles(const InT* in, u64* out, u64 n, u64* tile_sums) {
    __shared__ u64 s_w[16];
    __shared__ u64 s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    const u64 t0 = (u64)blockIdx.x * MCQ_SCAN_TILE;
    for (u64 base = t0; base < t0 + MCQ_SCAN_TILE; base += NT) {
        const u64 i = base + tid;
        u64 v = (i < n) ? in[i] : 0, x = v;
        for (int d = 1; d < 64; d <<= 1) { u64 t = __shfl_up(x, d, 64); if (lane >= (u32)d) x += t; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        u64 woff = 0;
        for (u32 w = 0; w < wave; ++w) woff += s_w[w];
        const u64 carry = s_carry;
        if (i < n) out[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == NT - 1) s_carry = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) tile_sums[blockIdx.x] = s_carry;
}
__global__ void k_scan_add(u64* out, u64 n, const u64* tile_off) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += tile_off[i / MCQ_SCAN_TILE];
    if (i == 0) out[n] = tile_off[(n + MCQ_SCAN_TILE - 1) / MCQ_SCAN_TILE];
}

// ------------------------------------------------------------------ kernel: wave per query
#define MCQ_LCAP_WAVE16 1024    // longest match list of the second wave stage (16 keys per lane)
__device__ __forceinline__ u32 pow2ceil(u32 x) { return x <= 1 ? 1u : 1u << (32 - __builtin_clz(x - 1)); }

template <class KeyT> __device__ __forceinline__ KeyT key_pad() { return ~(KeyT)0; }
// location word -> the public (tgt << 32) | win form
template <class KeyT, class LF> __device__ __forceinline__ u64 key_expand(KeyT k, const LF& lf) {
    u32 t; KeyT tb;
    lf.locate(k, t, tb);
    return ((u64)t << 32) | (u64)(k - tb);
}

// Gather E*64 list elements into registers: r[e] = element e*64 + lane of the concatenated lists.
// Which list an element belongs to: every non-empty list marks its first element's slot with (its lane + 1)
// in `mark` (64 * E words of the wave's LDS), an inclusive prefix maximum over the slots spreads the marks
// to the right (6 DPP steps per register instead of a 6-step shuffle search per element).
template <class KeyT, int E>
__device__ __forceinline__ void gather_regs_marks(const DbDev& db, KeyT (&r)[E], u32 T, u32 pos, u32 len, u64 off, u32 lane, u32* mark) {
    const KeyT* __restrict__ locs = static_cast<const KeyT*>(db.locs);
#pragma unroll
    for (int e = 0; e < E; ++e) mark[e * 64 + lane] = 0;
    wave_sync();
    if (len > 0) mark[pos] = lane + 1;
    wave_sync();
    u32 carry = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u32 t = e * 64 + lane;
        u32 v = wave_incl_max_dpp(mark[t]);
        v = v > carry ? v : carry;
        carry = bcast(v, 63);
        const u32 j = v - 1;                             // v >= 1: the first list starts at slot 0
        const u32 pj = __shfl(pos, (int)j, 64);
        const u32 olo = __shfl((u32)off, (int)j, 64), ohi = __shfl((u32)(off >> 32), (int)j, 64);
        r[e] = key_pad<KeyT>();
        if (t < T) r[e] = locs[(((u64)ohi << 32) | olo) + (t - pj)];
    }
    wave_sync();                                         // mark[] is the caller's again
}
// The same for up to 128 lists, two per lane (list l of the lane's first feature, list 64 + l of its second):
// the marks run to 128, and the list's start comes out of the first or the second register set.
// E > 32 (the 64-register form of the third wave stage): the marks are bytes, so that 4096 of them fit the 2048 words
template <int E>
__device__ __forceinline__ void gather_regs2_marks(const DbDev& db, u32 (&r)[E], u32 T, u32 pos0, u32 len0, u64 off0,
                                             u32 pos1, u32 len1, u64 off1, bool two, u32 lane, u32* mark) {
    const u32* __restrict__ locs = static_cast<const u32*>(db.locs);
    unsigned char* mark8 = reinterpret_cast<unsigned char*>(mark);
    if constexpr (E > 32) {
#pragma unroll
        for (int e = 0; e < E / 4; ++e) mark[e * 64 + lane] = 0;
        wave_sync();
        if (len0 > 0) mark8[pos0] = (unsigned char)(lane + 1);
        if (len1 > 0) mark8[pos1] = (unsigned char)(lane + 65);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) mark[e * 64 + lane] = 0;
        wave_sync();
        if (len0 > 0) mark[pos0] = lane + 1;
        if (len1 > 0) mark[pos1] = lane + 65;
    }
    wave_sync();
    u32 carry = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u32 t = e * 64 + lane;
        if (e > 0 && (u32)(e * 64) >= T) { r[e] = MCQ_EMPTY; continue; }      // wave-uniform: nothing up here
        u32 v = wave_incl_max_dpp(E > 32 ? (u32)mark8[t] : mark[t]);
        v = v > carry ? v : carry;
        carry = bcast(v, 63);
        const u32 j = v - 1;                             // v >= 1: the first list starts at slot 0
        u32 pj = __shfl(pos0, (int)(j & 63), 64);
        u32 olo = __shfl((u32)off0, (int)(j & 63), 64), ohi = __shfl((u32)(off0 >> 32), (int)(j & 63), 64);
        if (two) {                                       // wave-uniform
            const u32 pj1 = __shfl(pos1, (int)(j & 63), 64);
            const u32 olo1 = __shfl((u32)off1, (int)(j & 63), 64), ohi1 = __shfl((u32)(off1 >> 32), (int)(j & 63), 64);
            if (j >= 64) { pj = pj1; olo = olo1; ohi = ohi1; }
        }
        r[e] = MCQ_EMPTY;
        if (t < T) r[e] = locs[(((u64)ohi << 32) | olo) + (t - pj)];
    }
    wave_sync();                                         // mark[] is the caller's again
}
// r03: the list starts as a BITMAP instead (bit t = a list starts at element t; one ds_or per non-empty list), and the
// non-empty lists' (offset - start) compacted into a small table: row e takes its 64 bits from a lane of the register that
// holds the bitmap (v_readlane with a constant lane), the list of element t is the number of starts at or before t -- two
// v_mbcnt on the row's bits plus the starts of the rows before (a scalar popcount) -- and one ds_read_b64 fetches that list's
// offset.  ~16 instructions per register row instead of ~40 (12 of them a DPP chain with wait states, three shuffles),
// and the rows no longer depend on one another.  mark: 64 + 128 words (2E <= 64 bitmap words, then the table, 8-B aligned).
#ifdef MCQ_GATHER_MARKS          // tuning knob (A/B): the prefix-maximum form
#define gather_regs gather_regs_marks
#define gather_regs2 gather_regs2_marks
#else
template <class KeyT, int E>
__device__ __forceinline__ void gather_regs(const DbDev& db, KeyT (&r)[E], u32 T, u32 pos, u32 len, u64 off, u32 lane, u32* mark) {
    static_assert(E <= 32, "the bitmap of one register");
    const KeyT* __restrict__ locs = static_cast<const KeyT*>(db.locs);
    u32* bm = mark;
    unsigned long long* fb = reinterpret_cast<unsigned long long*>(mark + 64);
    bm[lane] = 0;
    wave_sync();
    const bool has = len > 0;
    const u64 hm = __ballot(has);
    if (has) { atomicOr(&bm[pos >> 5], 1u << (pos & 31)); fb[lane_rank(hm)] = off - pos; }
    wave_sync();
    const u32 bw = bm[lane];                             // word l of the bitmap in lane l
    u32 before = 0;                                      // starts in the rows before (wave-uniform)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u32 t = e * 64 + lane;
        r[e] = key_pad<KeyT>();
        if (e > 0 && (u32)(e * 64) >= T) continue;       // wave-uniform: nothing up here
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)bw, 2 * e), hi = (u32)__builtin_amdgcn_readlane((int)bw, 2 * e + 1);
        const u32 cnt = before + __builtin_amdgcn_mbcnt_hi(hi >> 1, __builtin_amdgcn_mbcnt_lo((lo >> 1) | (hi << 31), 0u)) + (lo & 1u);
        before += (u32)__builtin_popcount(lo) + (u32)__builtin_popcount(hi);
        const unsigned long long base = fb[(cnt - 1u) & 63u];          // cnt >= 1: the first list starts at element 0
        if (t < T) r[e] = locs[base + t];
    }
    wave_sync();                                         // mark[] is the caller's again
}
// The same for up to 128 lists, two per lane (list l of the lane's first feature, list 64 + l of its second); E <= 64:
// 128 bitmap words in two registers, the table behind them (mark: 128 + 256 words)
template <int E>
__device__ __forceinline__ void gather_regs2(const DbDev& db, u32 (&r)[E], u32 T, u32 pos0, u32 len0, u64 off0,
                                             u32 pos1, u32 len1, u64 off1, bool two, u32 lane, u32* mark) {
#ifndef MCQ_GATHER_SB            // tuning knob (A/B): the bitmap form for 32 and 64 registers too, with scheduling barriers every 8 rows
    // (32 / 64 registers per lane: with independent rows the compiler hoists their table reads and loads and spills 100
    // VGPRs in k_query_wave32 -- RefSeq-scale pairs 13.3 -> 13.9 ms; the prefix-maximum form's carry keeps them in order)
    if constexpr (E >= 32) { gather_regs2_marks<E>(db, r, T, pos0, len0, off0, pos1, len1, off1, two, lane, mark); return; }
#endif
    const u32* __restrict__ locs = static_cast<const u32*>(db.locs);
    u32* bm = mark;
    unsigned long long* fb = reinterpret_cast<unsigned long long*>(mark + 128);
    bm[lane] = 0; bm[64 + lane] = 0;
    wave_sync();
    const u64 m0 = __ballot(len0 > 0), m1 = __ballot(len1 > 0);
    const u32 n0 = (u32)__builtin_popcountll(m0);
    if (len0 > 0) { atomicOr(&bm[pos0 >> 5], 1u << (pos0 & 31)); fb[lane_rank(m0)] = off0 - pos0; }
    if (len1 > 0) { atomicOr(&bm[pos1 >> 5], 1u << (pos1 & 31)); fb[n0 + lane_rank(m1)] = off1 - pos1; }
    wave_sync();
    const u32 bw0 = bm[lane], bw1 = E > 32 ? bm[64 + lane] : 0u;
    u32 before = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u32 t = e * 64 + lane;
        r[e] = MCQ_EMPTY;
        if (e > 0 && (u32)(e * 64) >= T) continue;       // wave-uniform: nothing up here
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)(e < 32 ? bw0 : bw1), (2 * e) & 63);
        const u32 hi = (u32)__builtin_amdgcn_readlane((int)(e < 32 ? bw0 : bw1), (2 * e + 1) & 63);
        const u32 cnt = before + __builtin_amdgcn_mbcnt_hi(hi >> 1, __builtin_amdgcn_mbcnt_lo((lo >> 1) | (hi << 31), 0u)) + (lo & 1u);
        before += (u32)__builtin_popcount(lo) + (u32)__builtin_popcount(hi);
        const unsigned long long base = fb[(cnt - 1u) & 127u];
        if (t < T) r[e] = locs[base + t];
#ifdef MCQ_GATHER_SB
        if constexpr (E >= 32) { if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
#endif
    }
    wave_sync();                                         // mark[] is the caller's again
}
#endif
// ... sort them there and leave the sorted keys in the wave's LDS segment for the sweep.
template <class KeyT, int E>
__device__ __forceinline__ void gather_sort_store(const DbDev& db, KeyT* buf, u32* hits, u32 T, u32 pos, u32 len, u64 off, u32 lane, int stop) {
    KeyT r[E];
    gather_regs<KeyT, E>(db, r, T, pos, len, off, lane, hits);
    if (stop != 3) wave_regsort<KeyT, E>(r, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) buf[e * 64 + lane] = r[e];
}

// ---- rows 8-9 on DISTINCT (tgt,win) keys (32-bit keys, T <= 512) ------------------------------
// A read's match list repeats the same (target, window) many times (C2: 107 locations, 33 distinct;
// 2x150 bp pairs: 214 / 65; on an 8 x larger table 218 / 140), and everything after the gather only needs
// the distinct keys and how often each occurs.  The wave counts them in a 512-slot open-addressing table
// in its LDS segment (ds_cmpst claims a slot, ds_add counts in a 16-bit lane), the lanes that claimed a
// slot compact their keys, <= 256 distinct keys are sorted in one, two or four registers per lane instead
// of 64*E raw locations, and the multiplicities come back by probing the table with the sorted keys.
// Output: SK[0..D) sorted distinct keys, WP[0..D) inclusive prefix sums of the multiplicities (for
// sweep_targets_weighted).  Returns D, or ~0u when there are more than 256 distinct keys (the caller then
// sorts the raw list); the insertion stops at the first register that takes D past 256, so the table never
// holds more than 320 keys.
//   LDS (u32 words): buf[0..512) table keys, later the sweep's H in buf[0..256) and the top lists' scratch
//   behind it;  hits[0..256) table counts (16 bits per slot: any multiset of up to 512 locations is counted
//   exactly, whatever the caller of mcq_reduce hands over), later WP;  hits[256..512) compaction list, later SK.
#ifndef MCQ_DEDUP_MAX_T
#define MCQ_DEDUP_MAX_T 512u    // tuning knobs: -DMCQ_DEDUP_MAX_T=384u -DMCQ_DEDUP_MAX_D=128u is the two-register form
#endif
#ifndef MCQ_DEDUP_MAX_D
#define MCQ_DEDUP_MAX_D 256u
#endif
#ifdef MCQ_TOPK_DPP        // tuning knob: DPP reductions per rank instead of LDS maxima for all ranks at once
#define MCQ_TOPK_DEDUP(db, opt, out, sk, h, D, nw, lf, q, lane, t1) topk_fold_write<u32, u32, 9>(db, opt, out, sk, h, D, nw, lf, q, lane)
#else
#define MCQ_TOPK_DEDUP(db, opt, out, sk, h, D, nw, lf, q, lane, t1) topk_dedup(db, opt, out, sk, h, D, nw, lf, q, lane, t1)
#endif
// top lists of the dedup path: more than 64 distinct keys (two to four rounds of 64 run heads) take all heads at once
// t1 (D <= 64): the target of sorted key j in lane j, as dedup_finish looked it up
template <class LF>
__device__ __forceinline__ u32 topk_dedup(const DbDev& db, const OptDev& opt, const OutDev& out, const u32* sk, u32* H, u32 D,
                                          u32 numWindows, const LF& lf, u64 q, u32 lane, u32 t1) {
    // one selection for all ranks (zero words among the heads do no harm).  (Tried: the taxon keys of the <= 64 sorted words
    // loaded before the sweep and shuffled in here -- one more live register in the 64-VGPR kernel, +1.5 % on configs[1].)
    if (MCQ_OPT_LIN(opt)) {
        if (D <= 64) return topk_lin_write<u32, 9, LF::lookup>(db, opt, out, sk, H, D, lf, q, lane, t1);
        return topk_lin_write<u32, 9>(db, opt, out, sk, H, D, lf, q, lane);
    }
#ifndef MCQ_TOPK_DEDUP_CHUNKED                                      // tuning knob (A/B)
    if (D > 64) {
        u32 nheads = 0;
        for (u32 base = 0; base < D; base += 64) {                  // in place: writes trail reads
            const u32 j = base + lane;
            const u32 hv = (j < D) ? H[j] : 0;
            const u64 hm = __ballot(hv != 0);
            if (hv != 0) H[nheads + lane_rank(hm)] = hv;
            nheads += (u32)__builtin_popcountll(hm);
        }
        wave_sync();
        return topk_all_lds<9, 4>(db, opt, out, sk, H, nheads, numWindows, lf, q, lane, H + 256);
    }
#endif
    return topk_fold_write_lds<9, LF::lookup>(db, opt, out, sk, H, D, numWindows, lf, q, lane, H + 256, t1);
}
// -DMCQ_PHASE_CLOCK (diagnostic builds only): lane 0 of every wave of the second wave stage adds the shader clocks between the marks
// below into LDS (loads drained first), folded into CountersDev::pad_[3..9], [11] when the kernel ends
#ifdef MCQ_PHASE_CLOCK
__shared__ u64 g_wph[4][12];
#define WCLK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) { const u64 t_ = __builtin_amdgcn_s_memtime(); \
    u64* p_ = g_wph[threadIdx.x >> 6]; p_[i] += t_ - p_[11]; p_[11] = t_; } } while (0)
#else
#define WCLK(i) do { } while (0)
#endif
__device__ __forceinline__ u32 dedup_slot(u32 key) { return (key * 0x9E3779B1u) >> 23; }
__device__ __forceinline__ u32* dedup_sk(u32* hits) { return hits + 256; }
__device__ __forceinline__ u32* dedup_wp(u32* hits) { return hits; }
__device__ __forceinline__ u32 dedup_count(const u32* tabkey, const u32* tabcnt, u32 k) {
    u32 slot = dedup_slot(k);
    while (tabkey[slot] != k) slot = (slot + 1) & 511u;
    return (tabcnt[slot >> 1] >> (16 * (slot & 1))) & 0xFFFFu;
}

// Insertion half (depends on the number of registers E): counts the keys in the table and leaves the distinct
// ones in the compaction list; returns their number D.
template <int E>
__device__ __forceinline__ u32 dedup_insert(const u32 (&r)[E], u32* buf, u32* hits, u32 T, u32 lane) {
    u32* tabkey = buf; u32* tabcnt = hits; u32* list = hits + 256;
    reinterpret_cast<uint4*>(tabkey)[lane] = make_uint4(MCQ_EMPTY, MCQ_EMPTY, MCQ_EMPTY, MCQ_EMPTY);
    reinterpret_cast<uint4*>(tabkey)[64 + lane] = make_uint4(MCQ_EMPTY, MCQ_EMPTY, MCQ_EMPTY, MCQ_EMPTY);
    u32 zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));     // not hoistable: a loop-invariant zero quad gets spilled to scratch otherwise
    reinterpret_cast<uint4*>(tabcnt)[lane] = make_uint4(zero, zero, zero, zero);
    wave_sync();
    u32 D = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e > 0 && (u32)(e * 64) >= T) break;  // wave-uniform: nothing left in the higher registers
        const u32 key = r[e];
        bool created = false;
        if ((u32)(e * 64) + lane < T) {
            u32 slot = dedup_slot(key);
            while (true) {                       // <= 320 keys in 512 slots: an empty slot always exists
                const u32 old = atomicCAS(&tabkey[slot], MCQ_EMPTY, key);
                created = old == MCQ_EMPTY;
                if (created || old == key) { atomicAdd(&tabcnt[slot >> 1], 1u << (16 * (slot & 1))); break; }
                slot = (slot + 1) & 511u;
            }
        }
        const u64 cm = __ballot(created);
        const u32 rank = D + lane_rank(cm);
        if (created && rank < MCQ_DEDUP_MAX_D) list[rank] = key;
        D += (u32)__builtin_popcountll(cm);
        if (D > MCQ_DEDUP_MAX_D) break;          // wave-uniform: the raw list gets sorted instead
    }
    wave_sync();
    return D;
}
// Second half, the same code for every E: sort the D distinct keys, fetch their multiplicities, prefix sums.
// k1/incl1: the sorted keys and inclusive sums one per lane when D <= 64 (for sweep_targets_regs); t1/tb1: target and
// first word of the target of k1 -- looked up here, right behind the sort, so that a format that has to go to memory
// for them (LocGW) does so once per read and under the LDS work that follows.
// Returns D, or ~0u when D > 256.
template <class LF>
__device__ __forceinline__ u32 dedup_finish(u32 D, u32* buf, u32* hits, u32 lane, const LF& lf, u32& k1, u32& incl1, u32& t1, u32& tb1) {
    u32* tabkey = buf; u32* tabcnt = hits; u32* list = hits + 256; u32* SK = dedup_sk(hits); u32* WP = dedup_wp(hits);
    k1 = MCQ_EMPTY; incl1 = 0; t1 = 0; tb1 = 0;
    if (D > MCQ_DEDUP_MAX_D) return ~0u;
    if (D <= 64) {
        u32 k = lane < D ? list[lane] : MCQ_EMPTY;
        k = (D <= 32) ? wave_sort_blocks32_1(k) : wave_sort64_1(k);
        if constexpr (LF::lookup) lf.locate(lane < D ? k : list[0], t1, tb1);      // (padding lanes look up a real word)
        const u32 c = lane < D ? dedup_count(tabkey, tabcnt, k) : 0u;
        const u32 incl = wave_incl_scan_dpp(c);
        wave_sync();                             // counts consumed: WP overwrites them
        SK[lane] = k; WP[lane] = incl;
        k1 = k; incl1 = incl;
    } else if (D <= 128) {
        u32 k[2];
        k[0] = list[lane]; k[1] = (64 + lane < D) ? list[64 + lane] : MCQ_EMPTY;
        wave_regsort<u32, 2>(k, lane);
        const u32 c0 = dedup_count(tabkey, tabcnt, k[0]);
        const u32 c1 = (64 + lane < D) ? dedup_count(tabkey, tabcnt, k[1]) : 0u;
        const u32 i0 = wave_incl_scan_dpp(c0);
        const u32 i1 = wave_incl_scan_dpp(c1) + bcast(i0, 63);
        wave_sync();
        SK[lane] = k[0]; SK[64 + lane] = k[1]; WP[lane] = i0; WP[64 + lane] = i1;
    } else {
        u32 k[4], c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) k[e] = ((u32)(64 * e) + lane < D) ? list[64 * e + lane] : MCQ_EMPTY;
        wave_regsort<u32, 4>(k, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = ((u32)(64 * e) + lane < D) ? dedup_count(tabkey, tabcnt, k[e]) : 0u;
        u32 carry = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { c[e] = wave_incl_scan_dpp(c[e]) + carry; carry = bcast(c[e], 63); }
        wave_sync();                             // list and counts consumed: SK and WP overwrite them
#pragma unroll
        for (int e = 0; e < 4; ++e) { SK[64 * e + lane] = k[e]; WP[64 * e + lane] = c[e]; }
    }
    wave_sync();                                 // table dead from here: buf[] becomes the sweep's H
    return D;
}
// gather (or load) + insertion; the caller finishes with dedup_finish.  ~1u: stage-ablation stop after the gather.
template <int E>
__device__ __forceinline__ u32 gather_dedup_insert(const DbDev& db, u32* buf, u32* hits, u32 T, u32 pos, u32 len, u64 off, u32 lane, int stop) {
    u32 r[E];
    gather_regs<u32, E>(db, r, T, pos, len, off, lane, hits);
    if (stop == 3) {                             // stage-ablation hook: keep the loads alive, skip the rest
        u32 x = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) x ^= r[e];
        buf[lane] = x;
        return ~1u;
    }
    return dedup_insert<E>(r, buf, hits, T, lane);
}
// the same with two lists per lane (second wave stage, reads of 65..128 features)
template <int E>
__device__ __forceinline__ u32 gather2_dedup_insert(const DbDev& db, u32* buf, u32* hits, u32 T, u32 pos0, u32 len0, u64 off0,
                                                    u32 pos1, u32 len1, u64 off1, bool two, u32 lane) {
    u32 r[E];
    gather_regs2<E>(db, r, T, pos0, len0, off0, pos1, len1, off1, two, lane, hits);
    return dedup_insert<E>(r, buf, hits, T, lane);
}
// the same for a match list that already sits in global memory (staged / sharded path)
template <int E>
__device__ __forceinline__ u32 load_dedup_insert(const u32* __restrict__ src, u32* buf, u32* hits, u32 T, u32 lane) {
    u32 r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const u32 t = e * 64 + lane; r[e] = t < T ? src[t] : MCQ_EMPTY; }
    return dedup_insert<E>(r, buf, hits, T, lane);
}

// the same for a list that sits in the wave's own LDS (src may lie inside `hits`: everything is in registers before the
// table is cleared)
template <int E>
__device__ __forceinline__ u32 lds_dedup_insert(const u32* src, u32* buf, u32* hits, u32 T, u32 lane) {
    u32 r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const u32 t = e * 64 + lane; r[e] = t < T ? src[t] : MCQ_EMPTY; }
    wave_sync();
    return dedup_insert<E>(r, buf, hits, T, lane);
}

// Heavy words of the two-class tail: nH of them staged in hits[0..nH), E = registers per lane they need.  They repeat
// (a true target is hit by up to s features per window), so the distinct-key table is tried first -- any number of words,
// up to 256 distinct ones -- and only a list with more distinct words is sorted raw (E >= 8 then).  Either way the sweep
// leaves the sorted words SK[0..D) and the run heads' packed words H[0..D) (nine: packed with 9 index bits -- the distinct-key
// sweeps -- or 10), which go, with the light prefix, into the lists (heavy_lists).  false: not sorted here.
template <int E, class LF>
__device__ __forceinline__ bool heavy_sweep(u32 nH, u32 numWindows, const LF& lf, u32 lane, u32* buf, u32* hits, u32& D, u32*& SK, u32*& H, bool& nine) {
    constexpr int JB = 10;                                   // entry index of the lists: up to 1024 sorted words
    D = 0; SK = dedup_sk(hits); H = buf; nine = true;
    if (nH) {
        u32 rh[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { const u32 t = e * 64 + lane; rh[e] = t < nH ? hits[t] : MCQ_EMPTY; }
        wave_sync();
        u32 k1 = MCQ_EMPTY, incl1 = 0, t1 = 0, tb1 = 0;
        const u32 Dd = dedup_insert<E>(rh, buf, hits, nH, lane);
        WCLK(4);
        if (E > 16 && Dd > MCQ_DEDUP_MAX_D) return false;     // (32 registers of heavy words with more than 256 distinct ones: not sorted here)
        if (E <= 4 || Dd <= MCQ_DEDUP_MAX_D) {
            D = dedup_finish(Dd, buf, hits, lane, lf, k1, incl1, t1, tb1);
            if (D <= 64 && numWindows <= 8) sweep_targets_regs(k1, incl1, tb1, H, D, numWindows, lf, lane);
            else sweep_targets_weighted(SK, dedup_wp(hits), H, D, numWindows, lf, lane);
        } else if constexpr (E > 4 && E <= 16) {
            wave_sync();
            wave_regsort<u32, E>(rh, lane);
            SK = buf; H = hits; D = nH; nine = false;
#pragma unroll
            for (int e = 0; e < E; ++e) SK[e * 64 + lane] = rh[e];
            wave_sync();
            sweep_targets_wave<u32, JB>(SK, H, D, numWindows, lf, lane);
        }
    } else wave_sync();
    return true;
}
// The lists from what heavy_sweep left (one copy of this code per kernel, whatever E was)
template <class LF>
__device__ __forceinline__ u32 heavy_lists(const DbDev& db, const OptDev& opt, const OutDev& out, u32 D, u32* SK, u32* H, bool nine, u32 lkey, bool omitted,
                                           u32 theta, u32 safe, u32 numWindows, const LF& lf, u64 q, u32 lane) {
    constexpr int JB = 10;
    WCLK(5);
    // run heads to the front of H (in place: writes trail reads), re-packed with JB index bits.  A head with ONE hit is no
    // better than a light word: it stays only below theta -- the others join the omitted light words (same proof: one hit,
    // word >= theta)
    u32 nheads = 0;
    bool dropped = false;
    for (u32 base = 0; base < D; base += 64) {
        const u32 j = base + lane;
        const u32 hv = (j < D) ? H[j] : 0;
        const u32 h = nine ? hv >> 9 : hv >> JB, jb = nine ? 511u - (hv & 511u) : ((1u << JB) - 1) - (hv & ((1u << JB) - 1));
        const bool one = hv != 0 && h == 1u && SK[jb] >= theta;
        const u64 hb = __ballot(hv != 0 && !one);
        if (hv != 0 && !one) H[nheads + lane_rank(hb)] = (h << JB) | (((1u << JB) - 1) - jb);
        nheads += (u32)__builtin_popcountll(hb);
        dropped = dropped || __ballot(one) != 0;
    }
    wave_sync();
    if (D == 0 && lane == 0) SK[0] = safe;                   // (idle lanes look up SK[0])
    wave_sync();
    if (nheads > 256) return ~1u;                            // (heads with two or more hits, or below theta: a dozen or two)
    // scratch of the lists: 192 words of H's segment that neither the heads (<= 256 words) nor, in the other segment, SK touch
    u32 n;
    const u32 nP = (u32)__builtin_popcountll(__ballot(lkey != MCQ_EMPTY));      // (the light prefix sits in lanes [0, nP))
    if (nheads + nP <= 64u) {
        // the usual case -- a dozen or two heads, ~40 light words: ONE chunk of entries, heads in the lanes behind the light words
        // (one look-up of targets and taxa instead of two in a row, a fifth of the list rounds' compares)
        const u32 i = lane - nP;
        const u32 v = (lane >= nP && i < nheads) ? H[i] : 0u;
        const u32 mkey = v ? SK[((1u << JB) - 1) - (v & ((1u << JB) - 1))] : lkey;
        n = topk_two_class<JB, 0>(db, opt, out, SK, D, H, 0u, mkey, omitted || dropped, theta, numWindows, lf, q, lane, H + 512 + 64, v ? v >> JB : 1u);
    } else
        n = topk_two_class<JB, 4>(db, opt, out, SK, D, H, nheads, lkey, omitted || dropped, theta, numWindows, lf, q, lane, H + 512 + 64);
    WCLK(6);
    return n == ~0u ? ~1u : n;
}

// ---- the two-class tail of a wave (mcq_device.hpp, "rows 8-11 in two classes") for a raw match list held in registers,
// r[e] = word e * 64 + lane, T words.  buf / hits: the wave's two LDS segments (>= 1024 words each).
// Returns the number of candidates written; ~0u = not attempted or not taken, r[] untouched, the caller sorts the raw
// list as before; ~1u = given up after the registers were spent: the caller queues the query for the exact path.
#define MCQ_TWO_CLASS_MAX_PM 16u        // beyond P x M = 16 the light prefix rarely fills the lists: not attempted
#ifndef MCQ_TWO_CLASS_EXPECT
#define MCQ_TWO_CLASS_EXPECT 40.0f      // light words expected below theta (tuning knob)
#endif
template <int E, class LF>
__device__ __forceinline__ u32 two_class_tail(const DbDev& db, const OptDev& opt, const OutDev& out, u32 (&r)[E], u32 T, u32 numWindows,
                                              float word_space, const LF& lf, u64 q, u32 lane, u32* buf, u32* hits) {
    if ((opt.lin ? opt.P > 8u : opt.P * opt.max_cand > MCQ_TWO_CLASS_MAX_PM) || (opt.hooks & 8u)) return ~0u;     // (as launch_query's tc_lists)
    const u32 cs = cell_shift(numWindows);
    constexpr u32 LOG = E > 16 ? 16u : MCQ_CELL_LOG;         // each map fills one LDS segment (64 x E words)
    u32* occ = buf; u32* multi = hits;
    cells_clear<LOG>(occ, multi, lane, 64);
    wave_sync();
    // (tried: every register's atomics issued back to back without a branch, zeros ORed where nothing is to be set -- twice
    // the LDS atomics and 5 spilled VGPRs: second wave stage 5.1 -> 7.5 ms on the RefSeq-scale table)
#pragma unroll
    for (int e = 0; e < E; ++e) if ((u32)(e * 64) < T && (u32)(e * 64) + lane < T) cells_insert<LOG>(r[e], cs, occ, multi);
    wave_sync();
    typename std::conditional<(E > 32), unsigned long long, u32>::type hm = 0;       // bit e: r[e] is heavy
#pragma unroll
    for (int e = 0; e < E; ++e) if ((u32)(e * 64) < T && (u32)(e * 64) + lane < T && cells_heavy<LOG>(r[e], cs, occ, multi)) hm |= (decltype(hm))1 << e;
    wave_sync();                                             // the maps are dead: heavy words -> hits[0..nH), light prefix -> buf[0..nP)
    WCLK(2);
    const float th = word_space * MCQ_TWO_CLASS_EXPECT * __builtin_amdgcn_rcpf((float)T);
    const u32 theta = th >= 4294967040.0f ? 0xFFFFFFFEu : (u32)th;
    u32 nH = 0, nL = 0, nP = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if ((u32)(e * 64) >= T) break;                       // wave-uniform
        const bool valid = (u32)(e * 64) + lane < T, heavy = (hm >> e) & 1u;
        const u64 bh = __ballot(valid && heavy);
        if constexpr (E > 32) { if (valid && heavy) { const u32 i = nH + lane_rank(bh); if (i < 2048u) hits[i] = r[e]; } }    // (a segment holds 2048)
        else if (valid && heavy) hits[nH + lane_rank(bh)] = r[e];
        nH += (u32)__builtin_popcountll(bh);
        const bool pre = valid && !heavy && r[e] < theta;
        const u64 bp = __ballot(pre);
        if (pre) { const u32 i = nP + lane_rank(bp); if (i < 64) buf[i] = r[e]; }
        nP += (u32)__builtin_popcountll(bp);
    }
    nL = T - nH;                                             // (every word is one or the other)
    wave_sync();
    WCLK(3);
    if (nP > 64) return ~0u;                                 // theta too generous for this read (words far from uniform)
    const u32 safe = bcast(r[0], 0);                         // any real word (T >= 1)
    const u32 lkey = lane < nP ? buf[lane] : MCQ_EMPTY;
    const bool omitted = nL > nP;
    // from here on r[] is spent: the heavy words come back from LDS into as many registers as they need
    u32 D, *SK, *H; bool nine, ok;
    if (nH <= 64)       ok = heavy_sweep<1>(nH, numWindows, lf, lane, buf, hits, D, SK, H, nine);
    else if (nH <= 128) ok = heavy_sweep<2>(nH, numWindows, lf, lane, buf, hits, D, SK, H, nine);
    else if (nH <= 256) ok = heavy_sweep<4>(nH, numWindows, lf, lane, buf, hits, D, SK, H, nine);
    else if (nH <= 512) ok = heavy_sweep<8>(nH, numWindows, lf, lane, buf, hits, D, SK, H, nine);
    else if (E > 8 && nH <= 1024) { if constexpr (E > 8) ok = heavy_sweep<16>(nH, numWindows, lf, lane, buf, hits, D, SK, H, nine); }
    else if (E > 16 && nH <= 2048) { if constexpr (E > 16) ok = heavy_sweep<32>(nH, numWindows, lf, lane, buf, hits, D, SK, H, nine); }
    else return ~1u;
    if (!ok) return ~1u;
    return heavy_lists(db, opt, out, D, SK, H, nine, lkey, omitted, theta, safe, numWindows, lf, q, lane);
}

// geometry of one read (or pair) on the wave path
struct ReadGeom {
    u64 o0, o1;          // byte offsets of the mates
    u32 n1, n2;          // their lengths
    u32 nw1, nw2;        // their window counts
    u64 qlen;            // l1 + l2
    bool ovf;            // not for the first wave stage
    bool wide;           // 65..128 features (5..8 windows, e.g. 2 x 250 bp): second wave stage, two features per lane
};
__device__ __forceinline__ ReadGeom read_geom(const DbDev& db, const BatchDev& b, u64 q, int force_block) {
    ReadGeom g;
    const u64 a = b.paired ? 2 * q : q;
    u64 e0, e1;
    seq_bounds(b.seq_off, b.ranges, a, g.o0, e0);
    if (b.paired) seq_bounds(b.seq_off, b.ranges, a + 1, g.o1, e1); else { g.o1 = e0; e1 = e0; }
    const u64 l1 = e0 - g.o0, l2 = e1 - g.o1;
    g.qlen = l1 + l2;
    g.ovf = (force_block & 1) || ((l1 | l2) >> 20) != 0;
    g.n1 = (u32)l1; g.n2 = (u32)l2; g.nw1 = 0; g.nw2 = 0; g.wide = false;
    if (!g.ovf) {
        g.nw1 = num_windows32(g.n1, db.winlen, db.winstride, db.magic_stride);
        g.nw2 = b.paired ? num_windows32(g.n2, db.winlen, db.winstride, db.magic_stride) : 0;
        g.ovf = (g.nw1 + g.nw2) * db.s > 64;
        g.wide = g.ovf && (g.nw1 + g.nw2) * db.s <= 128;
    }
    return g;
}
// window w of the read (mate 1's windows, then mate 2's): offset into bases and length
__device__ __forceinline__ void window_span(const DbDev& db, const ReadGeom& g, u32 w, u64& at, u32& wl) {
    const bool m2 = w >= g.nw1;
    u32 beg;
    window_of32(m2 ? g.n2 : g.n1, db.winlen, db.winstride, db.magic_stride, m2 ? w - g.nw1 : w, beg, wl);
    at = (m2 ? g.o1 : g.o0) + beg;
}
// Top lists after a raw sort (32-bit keys).  The run heads are compacted to the front of the hit words; when they
// leave the last 128 of the CAP hit words free, those serve the LDS maxima of topk_all_lds (up to CAP / 2 heads at
// once, all virtual ranks in the same round) or topk_fold_write_lds (64 heads at a time), else the DPP reductions per
// rank and 64 heads.
template <int JB, int CAP, class LF>
__device__ __forceinline__ u32 topk_heads(const DbDev& db, const OptDev& opt, const OutDev& out, const u32* buf, u32* hits,
                                          u32 T, u32 numWindows, const LF& lf, u64 q, u32 lane) {
    u32 nheads = 0;
    for (u32 base = 0; base < T; base += 64) {                      // in place: writes trail reads
        const u32 j = base + lane;
        const u32 hv = (j < T) ? hits[j] : 0;
        const u64 hm = __ballot(hv != 0);
        if (hv != 0) hits[nheads + lane_rank(hm)] = hv;
        nheads += (u32)__builtin_popcountll(hm);
    }
    wave_sync();
    if (MCQ_OPT_LIN(opt)) return topk_lin_write<u32, JB>(db, opt, out, buf, hits, nheads, lf, q, lane);
#ifndef MCQ_TOPK_CHUNKED                                            // tuning knob (A/B): M rounds per 64 heads only
    constexpr int NC = CAP / 128;                                   // register budget: 2 words per 64 heads
    if (nheads <= 64u * NC)
        return topk_all_lds<JB, NC>(db, opt, out, buf, hits, nheads, numWindows, lf, q, lane, hits + (CAP - 128));
#endif
    if (nheads <= (u32)CAP - 128u)
        return topk_fold_write_lds<JB>(db, opt, out, buf, hits, nheads, numWindows, lf, q, lane, hits + (CAP - 128));
    return topk_fold_write<u32, u32, JB>(db, opt, out, buf, hits, nheads, numWindows, lf, q, lane);
}

// 64-bit keys: the lists by DPP reductions per rank (topk_fold_write compacts the heads itself), or the one selection
template <class KeyT, class LF>
__device__ __forceinline__ u32 topk_heads64(const DbDev& db, const OptDev& opt, const OutDev& out, const KeyT* buf, u32* hits,
                                            u32 T, u32 numWindows, const LF& lf, u64 q, u32 lane) {
    if (MCQ_OPT_LIN(opt)) return topk_lin_write<KeyT, 9>(db, opt, out, buf, hits, T, lf, q, lane);
    return topk_fold_write<KeyT, u32, 9>(db, opt, out, buf, hits, T, numWindows, lf, q, lane);
}

// Top lists when the P lists do not fit the 64 lanes of a wave (NL list registers per lane: the reference's -n 32 / -n 64
// with -maxcand 4): the run heads to the front of H, all of them at once through the LDS maxima.  ~0u: more than 256 heads
// -- the caller hands the query to the workgroup kernel.  scr: 128 words of LDS behind the heads.
template <int NL, class LF>
__device__ __forceinline__ u32 topk_many_lists(const DbDev& db, const OptDev& opt, const OutDev& out, const u32* sk, u32* H, u32 D,
                                               u32 numWindows, const LF& lf, u64 q, u32 lane, u32* scr) {
    u32 nheads = 0;
    for (u32 base = 0; base < D; base += 64) {                  // in place: writes trail reads
        const u32 j = base + lane;
        const u32 hv = (j < D) ? H[j] : 0;
        const u64 hm = __ballot(hv != 0);
        if (hv != 0) H[nheads + lane_rank(hm)] = hv;
        nheads += (u32)__builtin_popcountll(hm);
    }
    wave_sync();
    if (nheads > 256) return ~0u;
    return topk_all_lds_n<9, 4, NL>(db, opt, out, sk, H, nheads, numWindows, lf, q, lane, scr);
}

#ifndef MCQ_WAVE_OCC
#define MCQ_WAVE_OCC 8          // waves per SIMD the 32-bit-key kernel is compiled for (tuning knob)
#endif
// match-list tap of the wave kernels (TAP instantiations only: mcq_debug_matches): the sorted match list of a query
// as rows 7-8 define it, written out from whichever form the path holds it in
template <class KeyT, class LF>
__device__ __forceinline__ void tap_sorted(const DebugDev& dbg, const KeyT* buf, u32 T, const LF& lf, u64 q, u32 lane) {
    for (u32 t = lane; t < T; t += 64) dbg.matches[dbg.match_off[q] + t] = key_expand<KeyT>(buf[t], lf);
}
// ... from the distinct sorted keys SK[0..D) and the inclusive sums WP of their multiplicities
template <class LF>
__device__ __forceinline__ void tap_distinct(const DebugDev& dbg, const u32* SK, const u32* WP, u32 D, const LF& lf, u64 q, u32 lane) {
    for (u32 j = lane; j < D; j += 64) {
        const u64 v = key_expand<u32>(SK[j], lf);
        for (u32 c = j ? WP[j - 1] : 0u; c < WP[j]; ++c) dbg.matches[dbg.match_off[q] + c] = v;
    }
}

// SH (feature-sharded path, home rank): the same kernel, but the probe results of a query's feature slots come from the
// exchange (shard_fetch) instead of sketch + probe, and db.locs is the received location buffer.
// GW: 32-bit locations in the global-window form (LocGW), else bit fields (LocShift).  BSH: table layout at compile time (2 = 64-B
// buckets, 0 = 16-B slots; -1 = run-time: the TAP instantiations), see probe()
// NL > 1: the P virtual-rank lists take NL registers per lane (P x M up to 256; see topk_many_lists): an instantiation of its
// own, so that the usual one carries none of it
template <class KeyT, int LCAP, bool TAP = false, bool SH = false, bool GW = false, int BSH = -1, int NL = 1>
__global__ __launch_bounds__(256, sizeof(KeyT) == 4 ? (NL > 1 ? 5 : MCQ_WAVE_OCC) : 5) void k_query_wave(DbDev db, BatchDev b, OptDev opt, OutDev out,
                                                    CountersDev* ctr, u32* ovf_list, int force_block, DebugDev dbg, ShardDev sh, GwDev gwd) {
    static_assert(LCAP == 512, "wave path: 8 keys per lane at most, entry index packed into 9 bits");
    static_assert(!GW || sizeof(KeyT) == 4, "the global-window form is a 32-bit word");
    const typename LocOf<KeyT, GW>::type lf = loc_format<KeyT, GW>(db, gwd);
    __shared__ KeyT s_buf[4][LCAP];
    __shared__ u32 s_hits[4][LCAP];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    KeyT* buf = s_buf[wave];
    u32* hits = s_hits[wave];
    u32* sk_tmp = hits;                // sketch scratch aliases the (not yet used) hit words
    u32* feat = hits + 64;
    const u64 nwaves = (u64)gridDim.x * 4;
    unsigned long long st_feat = 0, st_hit = 0, st_loc = 0, st_cand = 0;
    u32 st_geom = 0, st_long = 0;
#ifdef MCQ_PROFILE_HOOKS
    const int stop = force_block >> 4;              // profiling builds: stop after stage 1..5 (results invalid); 0 = run everything
#else
    constexpr int stop = 0;
#endif
    __shared__ u32 s_ovf[4][5];                     // this wave's queue reservations (mcq_device.hpp, "overflow queues")
    if (lane == 0) ovf_init(s_ovf[wave]);
    // The distinct-key count of a read is only known after the attempt; a failed attempt costs a gather and ~300 table
    // inserts before the raw list is gathered again.  Reads of one batch are alike, so the wave remembers the shortest
    // list that recently failed and sends lists at least that long straight to the raw sort (results are the same
    // either way; only the path differs).
    // (One word of LDS per wave, touched only by lists longer than 256: a loop-carried register costs the short reads.)
#ifndef MCQ_NO_TFAIL                                   // tuning knob (A/B)
    __shared__ u32 s_tfail[4];
    if (lane == 0) s_tfail[wave] = MCQ_DEDUP_MAX_T + 1;
    wave_sync();
#endif

#ifndef MCQ_NO_DIRECT_MODE                             // tuning knob (A/B)
    // Direct mode (CountersDev::direct_mode, set by the batch before): most queries of such a batch leave this stage anyway -- it
    // would sketch, probe, scan, write 512 B of probe results and push one queue entry per wave and query (2.3 ms per 1 M reads at
    // RefSeq scale).  Instead one LANE per query looks at its geometry and every query is queued: wide and narrow ones to the back
    // queue, flagged MCQ_Q_UNPROBED (the second stage sketches and probes two features per lane as it does for wide reads), more
    // than 128 features to the front queue.  Whole chunks of queue slots per 64 queries, one atomic per queue.
    if constexpr (sizeof(KeyT) == 4 && !TAP && NL == 1) {
        if ((ctr->direct_mode & (SH ? 1u : 2u)) && !(force_block & 7) && stop == 0) {
            for (u64 q0 = ((u64)blockIdx.x * 4 + wave) * 64; q0 < b.nq; q0 += nwaves * 64) {
                const u64 q = q0 + lane;
                const bool valid = q < b.nq;
                bool front = false, lng = false;
                if (valid) { const ReadGeom g = read_geom(db, b, q, 0); front = g.ovf && !g.wide; lng = MCQ_BLOCK_LONG_FIRST && g.qlen >= MCQ_BLOCK_LONG_FIRST; }
#pragma unroll
                for (int back = 0; back < 2; ++back) {
                    const u64 m = __ballot(valid && (back ? !front : front));
                    const u32 cnt = (u32)__builtin_popcountll(m);
                    if (!cnt) continue;                                 // (wave-uniform)
                    const u32 padded = (cnt + MCQ_OVF_CHUNK - 1) / MCQ_OVF_CHUNK * MCQ_OVF_CHUNK;
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(back ? &ctr->ovf_mid_count : &ctr->ovf_count, padded);
                    base = bcast(base, 0);
                    if (valid && (back ? !front : front)) ovf_list[ovf_slot(b.nq, back, base + lane_rank(m))] = (u32)q | (back ? MCQ_Q_UNPROBED : 0u);
                    if (lane < padded - cnt) ovf_list[ovf_slot(b.nq, back, base + cnt + lane)] = MCQ_EMPTY;
                }
                const u32 nv = (u32)__builtin_popcountll(__ballot(valid)), nf = (u32)__builtin_popcountll(__ballot(valid && front));      // (by every lane)
                const u32 nl = (u32)__builtin_popcountll(__ballot(lng));
                if (lane == 0) { atomicAdd(&ctr->n_ovf, nv); if (nf) atomicAdd(&ctr->n_geom, (unsigned long long)nf); if (nl) atomicAdd(&ctr->n_long, (unsigned long long)nl); }
            }
            return;
        }
    }
#endif
    for (u64 q = (u64)blockIdx.x * 4 + wave; q < b.nq; q += nwaves) {
        const ReadGeom g = read_geom(db, b, q, force_block);
        bool ovf = g.ovf;
        u32 myf = MCQ_EMPTY, nfeat = 0, T = 0, len = 0, pos = 0;
        u64 off = 0;
        if (!ovf) {
            if constexpr (SH) {
                nfeat = (g.nw1 + g.nw2) * db.s;          // feature slots (unused ones have no list)
                if (lane < nfeat) shard_fetch(sh, sh.win_off[b.paired ? 2 * q : q] * db.s + lane, off, len);
            } else {
            for (u32 w = 0; w < g.nw1 + g.nw2; ++w) {
                u64 at; u32 wl;
                window_span(db, g, w, at, wl);
                nfeat += wave_sketch_b(b, at, wl, db.k, db.s, lane, sk_tmp, feat + nfeat);
            }
            if (lane < nfeat) myf = feat[lane];
            if (stop == 1) { if (myf == 12345u) out.ncand[q] = nfeat; continue; }
            probe<BSH>(db, myf, off, len);
            if (stop == 2) { if (len == 0x7FFFFFFFu) out.ncand[q] = (u32)off; continue; }
            }
            u32 incl = wave_incl_scan_dpp(len);
            pos = incl - len;
            T = bcast(incl, 63);
            if (T > (u32)LCAP) ovf = true;
        }
        if (ovf) {
            if (g.ovf && !g.wide) ++st_geom;
            if (MCQ_BLOCK_LONG_FIRST && g.qlen >= MCQ_BLOCK_LONG_FIRST) ++st_long;
            // two queues in one array: 32-bit keys and either 513..1024 locations or 65..128 features from the back
            // (k_query_wave16: still one wave per query), everything else from the front (k_query_block)
            if (lane == 0) {
#ifdef MCQ_NO_WAVE16_ROUTE                             // tuning knob (A/B): one queue, as before the second wave stage
                if (false) {
#else
                if (sizeof(KeyT) == 4 && ((!g.ovf && T <= (u32)MCQ_LCAP_WAVE16) || g.wide) && !(force_block & 4)) {
#endif
                    ovf_push(s_ovf[wave], 1, ctr, ovf_list, b.nq, (u32)q);
                } else ovf_push(s_ovf[wave], 0, ctr, ovf_list, b.nq, (u32)q);
            }
#if !defined(MCQ_NO_PROBE_HANDOVER) && !defined(MCQ_NO_WAVE16_ROUTE)     // tuning knob (A/B)
            if (sizeof(KeyT) == 4 && !g.ovf && T <= (u32)MCQ_LCAP_WAVE16 && !(force_block & 4)) {
                wave_sync();                           // the slot lane 0 just took: next - 1 of the back queue
                const u32 slot = s_ovf[wave][1] - 1;
                ctr->probe_buf[(u64)slot * 64 + lane] = (off << 16) | len;
                if constexpr (!SH) st_feat += nfeat;
                st_hit += (u32)__builtin_popcountll(__ballot(len > 0));     // counted here, not in the second stage
            }
#ifndef MCQ_NO_FRONT_HANDOVER                          // tuning knob (A/B)
            else if constexpr (sizeof(KeyT) == 4 && !SH) {
                if (!g.ovf && opt.tc_limit != 0) {     // a front-queue entry the third wave stage will look at: the same, by front slot
                    wave_sync();
                    const u32 slot = s_ovf[wave][0] - 1;
                    ctr->probe_front[(u64)slot * 64 + lane] = lane < nfeat ? ((off << 16) | len) : 0xFFFFull;
                }
            }
#endif
#endif
            continue;
        }
        if constexpr (!SH) st_feat += nfeat;             // (sharded: the sketch kernel counted the features)
        st_hit += (u32)__builtin_popcountll(__ballot(len > 0)); st_loc += T;
        if constexpr (TAP) { if (dbg.mode == 1 && lane == 0) dbg.match_cnt[q] = T; }
        if (T == 0) { if (lane == 0) out.ncand[q] = 0; continue; }

        wave_sync();                                   // feat[] (aliasing hits) has been consumed
        const u32 numWindows = range_width(g.qlen, opt.insert_size_max, db.tgt_winstride, db.magic_tgt_stride);
        if constexpr (sizeof(KeyT) == 4) {
            if (T <= MCQ_DEDUP_MAX_T && !(force_block & 2)) {
                u32 D, k1 = MCQ_EMPTY, incl1 = 0, t1 = 0, tb1 = 0;
                bool skipped = false;
                if (T <= 64)       D = gather_dedup_insert<1>(db, buf, hits, T, pos, len, off, lane, stop);
                else if (T <= 128) D = gather_dedup_insert<2>(db, buf, hits, T, pos, len, off, lane, stop);
                else if (T <= 192) D = gather_dedup_insert<3>(db, buf, hits, T, pos, len, off, lane, stop);
                else if (T <= 256) D = gather_dedup_insert<4>(db, buf, hits, T, pos, len, off, lane, stop);
                else {
#ifndef MCQ_NO_TFAIL
                    const u32 t_fail = s_tfail[wave];
                    skipped = T >= t_fail;
#endif
                    if (skipped) {                 // creep up, so that such lists are tried again now and then
#ifndef MCQ_NO_TFAIL
                        if (lane == 0) s_tfail[wave] = t_fail + 2;
#endif
                        D = MCQ_DEDUP_MAX_D + 1;
                    }
                    else if (T <= 384) D = gather_dedup_insert<6>(db, buf, hits, T, pos, len, off, lane, stop);
                    else               D = gather_dedup_insert<8>(db, buf, hits, T, pos, len, off, lane, stop);
                }
                if (D != ~1u) D = dedup_finish(D, buf, hits, lane, lf, k1, incl1, t1, tb1);
                if (stop == 3 || stop == 4) { if (buf[lane] == 0x1234u && D == 77u) out.ncand[q] = 1; wave_sync(); continue; }
                if (D != ~0u) {
                    if constexpr (TAP) { if (dbg.mode == 2) tap_distinct(dbg, dedup_sk(hits), dedup_wp(hits), D, lf, q, lane); }
                    if (D <= 64 && numWindows <= 8) sweep_targets_regs(k1, incl1, tb1, reinterpret_cast<u32*>(buf), D, numWindows, lf, lane);
                    else sweep_targets_weighted(dedup_sk(hits), dedup_wp(hits), reinterpret_cast<u32*>(buf), D, numWindows, lf, lane);
                    if (stop == 5) { if (buf[lane] == 0x12345u) out.ncand[q] = 1; wave_sync(); continue; }
                    if constexpr (NL > 1) {
                        const u32 nc = topk_many_lists<NL>(db, opt, out, dedup_sk(hits), reinterpret_cast<u32*>(buf), D, numWindows, lf, q, lane, reinterpret_cast<u32*>(buf) + 256);
                        if (nc == ~0u) {            // (cannot happen here: at most 256 distinct keys) -- the workgroup kernel
                            if (lane == 0) ovf_push(s_ovf[wave], 0, ctr, ovf_list, b.nq, (u32)q);
                            if constexpr (!SH) st_feat -= nfeat;
                            st_hit -= (u32)__builtin_popcountll(__ballot(len > 0)); st_loc -= T;
                        } else st_cand += nc;
                    } else
                    st_cand += MCQ_TOPK_DEDUP(db, opt, out, dedup_sk(hits), reinterpret_cast<u32*>(buf), D, numWindows, lf, q, lane, t1);
                    wave_sync();
                    continue;
                }
                wave_sync();                       // more than 256 distinct keys: the raw list is sorted below
#ifndef MCQ_NO_TFAIL
                if (lane == 0 && !skipped) s_tfail[wave] = T;  // (T > 256) lists this long and longer skip the attempt for a while
#endif
            }
        }
        if (T <= 64)       gather_sort_store<KeyT, 1>(db, buf, hits, T, pos, len, off, lane, stop);
        else if (T <= 128) gather_sort_store<KeyT, 2>(db, buf, hits, T, pos, len, off, lane, stop);
        else if (T <= 256) gather_sort_store<KeyT, 4>(db, buf, hits, T, pos, len, off, lane, stop);
        else               gather_sort_store<KeyT, 8>(db, buf, hits, T, pos, len, off, lane, stop);
        wave_sync();
        if (stop == 3 || stop == 4) { if (buf[lane] == (KeyT)0x1234) out.ncand[q] = 1; continue; }
        if constexpr (TAP) { if (dbg.mode == 2) tap_sorted<KeyT>(dbg, buf, T, lf, q, lane); }
        sweep_targets_wave<KeyT>(buf, hits, T, numWindows, lf, lane);
        if (stop == 5) { if (hits[lane] == 0x12345u) out.ncand[q] = 1; continue; }
        if constexpr (NL > 1) {
            const u32 nc = topk_many_lists<NL>(db, opt, out, reinterpret_cast<const u32*>(buf), hits, T, numWindows, lf, q, lane, hits + (LCAP - 128));
            if (nc == ~0u) {                        // more than 256 run heads: the workgroup kernel (lists in its LDS)
                if (lane == 0) ovf_push(s_ovf[wave], 0, ctr, ovf_list, b.nq, (u32)q);
                if constexpr (!SH) st_feat -= nfeat;
                st_hit -= (u32)__builtin_popcountll(__ballot(len > 0)); st_loc -= T;
            } else st_cand += nc;
        } else
        if constexpr (sizeof(KeyT) == 4) st_cand += topk_heads<9, LCAP>(db, opt, out, reinterpret_cast<const u32*>(buf), hits, T, numWindows, lf, q, lane);
        else st_cand += topk_heads64<KeyT>(db, opt, out, buf, hits, T, numWindows, lf, q, lane);
        wave_sync();
    }
    if (lane == 0) ovf_flush(s_ovf[wave], ctr, ovf_list, b.nq);
    if (lane == 0 && st_geom) atomicAdd(&ctr->n_geom, (unsigned long long)st_geom);
    if (lane == 0 && st_long) atomicAdd(&ctr->n_long, (unsigned long long)st_long);
    if (lane == 0 && (st_feat | st_loc | st_hit)) {
        if (st_feat) atomicAdd(&ctr->n_features, st_feat);
        atomicAdd(&ctr->n_hit_features, st_hit);
        atomicAdd(&ctr->n_locations, st_loc);
        atomicAdd(&ctr->n_cands, st_cand);
    }
}

// ------------------------------------------------------------------ kernel: wave per query, 16 keys per lane
// Second stage of the wave path for the queries k_query_wave queued from the back of ovf_list: at most 64 features
// and 513..1024 locations (paired reads on a large table).  Same steps, one wave per query: sketch and probe again
// (cheap beside the rest), gather into 16 registers per lane, register sort of the raw list, sweep and top lists
// with a 10-bit entry index.  8 KB of LDS per wave; four waves per SIMD (128 VGPRs).  Per query this issues a fraction of
// the instructions of a 1024-thread workgroup, whose barrier phases leave most of its waves idle at this size.
#ifndef MCQ_WAVE16_OCC
#define MCQ_WAVE16_OCC 4        // waves per SIMD it is compiled for: 5 fit the LDS, but then 10 VGPRs spill (+30 % time)
#endif
template <bool TAP = false, bool SH = false, bool GW = false, int BSH = -1>
__global__ __launch_bounds__(256, MCQ_WAVE16_OCC) void k_query_wave16(DbDev db, BatchDev b, OptDev opt, OutDev out,
                                                                      CountersDev* ctr, u32* ovf_list, DebugDev dbg, ShardDev sh, GwDev gwd) {
    constexpr int LCAP = MCQ_LCAP_WAVE16, JB = 10;
    const typename LocOf<u32, GW>::type lf = loc_format<u32, GW>(db, gwd);
    __shared__ u32 s_buf[4][LCAP];
    __shared__ u32 s_hits[4][LCAP];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32* buf = s_buf[wave];
    u32* hits = s_hits[wave];
    u32* sk_tmp = hits;
    u32* feat = hits + 64;
    const u32 nwaves = gridDim.x * 4;
    const u32 n_mid = ctr->ovf_mid_count;
    unsigned long long st_feat = 0, st_hit = 0, st_loc = 0, st_cand = 0, st_two = 0, st_retry = 0, st_short = 0;
#ifdef MCQ_PHASE_CLOCK
    if (lane < 12) g_wph[wave][lane] = lane == 11 ? (u64)__builtin_amdgcn_s_memtime() : 0;
    wave_sync();
#endif
    u32 fq_next = 0, fq_left = 0;                      // this wave's reservation in the front queue (wide reads with > 1024 locations)
    // size of the space the location words live in (for the light-word threshold of the two-class tail)
    float word_space;
    if constexpr (GW) word_space = (float)gwd.off[db.n_targets];
    else word_space = (db.wb < 32 && ((u64)db.n_targets << db.wb) < 0xFFFFFFFFull) ? (float)((u64)db.n_targets << db.wb) : 4294967040.0f;
    // (-DMCQ_WAVE_DYNQ, tuning knob: the waves take their entries from a shared cursor, MCQ_WAVE_DYNQ_CH at a time, instead of striding)
#ifdef MCQ_WAVE_DYNQ
    for (u32 cb = 0;;) {
        if (lane == 0) cb = atomicAdd(&ctr->w_cursor[0], (u32)MCQ_WAVE_DYNQ_CH);
        cb = bcast(cb, 0);
        if (cb >= n_mid) break;
        const u32 ce = cb + (u32)MCQ_WAVE_DYNQ_CH < n_mid ? cb + (u32)MCQ_WAVE_DYNQ_CH : n_mid;
    for (u32 it = cb; it < ce; ++it) {
#else
    {
    for (u32 it = blockIdx.x * 4 + wave; it < n_mid; it += nwaves) {
#endif
        WCLK(7);
        const u32 qe = ovf_list[ovf_slot(b.nq, 1, ovf_visit(it, n_mid))];
        if (qe == MCQ_EMPTY) continue;                 // unused tail of a wave's reservation
        const u32 q32 = qe & ~MCQ_Q_UNPROBED;
        const u64 q = q32;
        const ReadGeom g = read_geom(db, b, q, 0);
        // fresh: nothing is known of the query's features yet -- a wide read, or any read the first stage queued unseen (direct mode)
        const bool fresh = g.wide || (qe & MCQ_Q_UNPROBED) != 0;
        u32 nfeat = 0;                                 // <= 128: the first stage queued nothing wider
        bool two = false;
        u64 off0 = 0, off1 = 0; u32 len0 = 0, len1 = 0;
#ifndef MCQ_NO_PROBE_HANDOVER
        if (!fresh) {                                  // queued by its length: the first stage left its probe results
            const u64 pw = ctr->probe_buf[(u64)ovf_visit(it, n_mid) * 64 + lane];
            off0 = pw >> 16; len0 = (u32)(pw & 0xFFFFu);
        } else
#endif
        {
            if constexpr (SH) {                        // feature slots of the exchange, two per lane
                nfeat = (g.nw1 + g.nw2) * db.s;
                two = nfeat > 64;
                const u64 sb = sh.win_off[b.paired ? 2 * q : q] * db.s;
                if (lane < nfeat) shard_fetch(sh, sb + lane, off0, len0);
                if (two && 64 + lane < nfeat) shard_fetch(sh, sb + 64 + lane, off1, len1);
            } else {
            for (u32 w = 0; w < g.nw1 + g.nw2; ++w) {
                u64 at; u32 wl;
                window_span(db, g, w, at, wl);
                nfeat += wave_sketch_b(b, at, wl, db.k, db.s, lane, sk_tmp, feat + nfeat);
            }
            two = nfeat > 64;                          // wave-uniform
            const u32 myf0 = lane < nfeat ? feat[lane] : MCQ_EMPTY;
            const u32 myf1 = (two && 64 + lane < nfeat) ? feat[64 + lane] : MCQ_EMPTY;
            probe<BSH>(db, myf0, off0, len0);
            if (two) probe<BSH>(db, myf1, off1, len1);
            }
        }
        const u32 incl0 = wave_incl_scan_dpp(len0);
        const u32 T0 = bcast(incl0, 63);
        const u32 incl1 = wave_incl_scan_dpp(len1) + T0;
        const u32 pos0 = incl0 - len0, pos1 = incl1 - len1;
        const u32 T = bcast(incl1, 63);                // <= 1024 for the queries queued by their length
        WCLK(0);
        if (T > (u32)LCAP) {                           // a longer list than this stage takes: on to the front queue
            if (fq_left == 0) {
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&ctr->ovf_count, MCQ_OVF_CHUNK);
                fq_next = bcast(base, 0); fq_left = MCQ_OVF_CHUNK;
            }
            if (lane == 0) ovf_list[fq_next] = q32;
#ifndef MCQ_NO_FRONT_HANDOVER
            // (direct mode) <= 64 features: the third wave stage reads such an entry's probe results from the slot's row, as it does
            // for the entries the first stage queues there itself; it counts the features (no bit 63)
            if constexpr (!SH) { if (!g.wide) ctr->probe_front[(u64)fq_next * 64 + lane] = lane < nfeat ? ((off0 << 16) | len0) : 0xFFFFull; }
#endif
            ++fq_next; --fq_left;
            continue;
        }
        st_loc += T;
        if constexpr (TAP) { if (dbg.mode == 1 && lane == 0) dbg.match_cnt[q] = T; }
#ifdef MCQ_NO_PROBE_HANDOVER
        {
#else
        if (fresh) {                                   // (the first stage counted the features of the others)
#endif
            if constexpr (!SH) st_feat += nfeat;
            st_hit += (u32)__builtin_popcountll(__ballot(len0 > 0)) + (u32)__builtin_popcountll(__ballot(len1 > 0));
        }
        if (T == 0) { if (lane == 0) out.ncand[q] = 0; continue; }
        wave_sync();                                   // feat[] (aliasing hits) has been consumed
        const u32 numWindows = range_width(g.qlen, opt.insert_size_max, db.tgt_winstride, db.magic_tgt_stride);
        if (T <= MCQ_DEDUP_MAX_T && (qe & MCQ_Q_UNPROBED) && !g.wide) st_short += 1;       // (direct mode: the first stage would have kept this one)
        if (T <= MCQ_DEDUP_MAX_T) {                    // a short list (a wide read, or direct mode): the distinct-key tail of the first stage
            u32 D, k1 = MCQ_EMPTY, incl1 = 0, t1 = 0, tb1 = 0;
            if (T <= 128)      D = gather2_dedup_insert<2>(db, buf, hits, T, pos0, len0, off0, pos1, len1, off1, two, lane);
            else if (T <= 256) D = gather2_dedup_insert<4>(db, buf, hits, T, pos0, len0, off0, pos1, len1, off1, two, lane);
            else               D = gather2_dedup_insert<8>(db, buf, hits, T, pos0, len0, off0, pos1, len1, off1, two, lane);
            D = dedup_finish(D, buf, hits, lane, lf, k1, incl1, t1, tb1);
            if (D != ~0u) {
                if constexpr (TAP) { if (dbg.mode == 2) tap_distinct(dbg, dedup_sk(hits), dedup_wp(hits), D, lf, q, lane); }
                if (D <= 64 && numWindows <= 8) sweep_targets_regs(k1, incl1, tb1, buf, D, numWindows, lf, lane);
                else sweep_targets_weighted(dedup_sk(hits), dedup_wp(hits), buf, D, numWindows, lf, lane);
                st_cand += MCQ_TOPK_DEDUP(db, opt, out, dedup_sk(hits), buf, D, numWindows, lf, q, lane, t1);
                wave_sync();
                continue;
            }
            wave_sync();                               // more than 256 distinct keys: the raw list below
        }
        {
            u32 r[16];
            gather_regs2<16>(db, r, T, pos0, len0, off0, pos1, len1, off1, two, lane, hits);
            WCLK(1);
            if constexpr (!TAP) {                       // (the taps want the whole sorted list)
                const u32 n2 = two_class_tail<16>(db, opt, out, r, T, numWindows, word_space, lf, q, lane, buf, hits);
                if (n2 == ~1u) {                        // given up after the registers were spent: the workgroup kernel takes it
                    st_loc -= T; st_retry += 1;
                    if (fresh) {                        // (the next stage counts them again)
                        if constexpr (!SH) st_feat -= nfeat;
                        st_hit -= (u32)__builtin_popcountll(__ballot(len0 > 0)) + (u32)__builtin_popcountll(__ballot(len1 > 0));
                    }
                    if (fq_left == 0) {
                        u32 base = 0;
                        if (lane == 0) base = atomicAdd(&ctr->ovf_count, MCQ_OVF_CHUNK);
                        fq_next = bcast(base, 0); fq_left = MCQ_OVF_CHUNK;
                    }
                    if (lane == 0) ovf_list[fq_next] = q32;
#ifndef MCQ_NO_FRONT_HANDOVER
                    // <= 64 features: the third wave stage reads every such front entry's probe results from the slot's row
                    // (bit 63: features and hit features are counted already -- the first stage did that at its hand-over)
                    if constexpr (!SH) { if (!g.wide) ctr->probe_front[(u64)fq_next * 64 + lane] = fresh ? (lane < nfeat ? ((off0 << 16) | len0) : 0xFFFFull)
                                                                                                           : ((1ull << 63) | (off0 << 16) | len0); }
#endif
                    ++fq_next; --fq_left;
                    wave_sync();
                    continue;
                }
                if (n2 != ~0u) { st_cand += n2; st_two += 1; wave_sync(); continue; }
            }
            wave_regsort<u32, 16>(r, lane);
#pragma unroll
            for (int e = 0; e < 16; ++e) buf[e * 64 + lane] = r[e];
        }
        wave_sync();
        if constexpr (TAP) { if (dbg.mode == 2) tap_sorted<u32>(dbg, buf, T, lf, q, lane); }
        sweep_targets_wave<u32, JB>(buf, hits, T, numWindows, lf, lane);
        st_cand += topk_heads<JB, LCAP>(db, opt, out, buf, hits, T, numWindows, lf, q, lane);
        wave_sync();
    }
    }
#ifdef MCQ_PHASE_CLOCK
    WCLK(7);
    wave_sync();
    if (lane < 8 && g_wph[wave][lane]) atomicAdd(&ctr->pad_[lane < 7 ? 3 + lane : 11], (unsigned long long)g_wph[wave][lane]);
#endif
    if (lane == 0) for (; fq_left; --fq_left, ++fq_next) ovf_list[fq_next] = MCQ_EMPTY;
    if (lane == 0 && st_two) atomicAdd(&ctr->n_two_class, st_two);
    if (lane == 0 && st_retry) atomicAdd(&ctr->n_two_class_retry, st_retry);
    if (lane == 0 && st_short) atomicAdd(&ctr->n_short, st_short);
    if (lane == 0 && (st_feat | st_loc | st_hit)) {
        if (st_feat) atomicAdd(&ctr->n_features, st_feat);
        atomicAdd(&ctr->n_hit_features, st_hit);
        atomicAdd(&ctr->n_locations, st_loc);
        atomicAdd(&ctr->n_cands, st_cand);
    }
}

// ------------------------------------------------------------------ kernel: wave per query, 32 keys per lane (two-class tail only)
// Third wave stage, for what a RefSeq-scale table does to short reads: 1025..2048 locations for at most 128 features -- one
// read in five there -- which until round 3 took a 1024-thread workgroup each (67 us per read; the second wave stage takes
// 6 us); and 2049..4096 locations (one 2 x 150 bp pair in eight there) in 64 registers per lane, same LDS: the gather's marks
// are bytes, the cell maps keep their 2^16 cells, at most 2048 heavy words.  It walks the FRONT queue before the workgroup kernels do: an entry it can answer -- few features, a list that fits
// 32 registers per lane, lists provable by the two-class tail -- is answered and overwritten with the empty marker; every
// other entry stays for the workgroup kernels, and the narrow ones among those are counted for them (see k_query_block).
// Sketch and probe again (no hand-over: ~8 us of the ~50), two features per lane as in the second stage.  16 KB of LDS per
// wave, two waves per SIMD.  Launched only when the two-class tail is (32-bit words, P x M <= 16).
template <bool SH = false, bool GW = false, int BSH = -1>
__global__ __launch_bounds__(256, 2) void k_query_wave32(DbDev db, BatchDev b, OptDev opt, OutDev out, CountersDev* ctr, u32* ovf_list,
                                                         int force_block, ShardDev sh, GwDev gwd) {
    constexpr int LSEG = 2048;                         // words per LDS segment
#ifdef MCQ_WAVE32_ONLY                                 // tuning knob (A/B): without the 64-register form
    constexpr int LCAP = 2048;
#else
    constexpr int LCAP = 4096;
#endif
    const typename LocOf<u32, GW>::type lf = loc_format<u32, GW>(db, gwd);
    __shared__ u32 s_buf[4][LSEG];
    __shared__ u32 s_hits[4][LSEG];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32* buf = s_buf[wave];
    u32* hits = s_hits[wave];
    u32* sk_tmp = hits;
    u32* feat = hits + 64;
    const u32 nwaves = gridDim.x * 4;
    const u32 n_front = ctr->ovf_count;                // (nothing appends to the front queue while this kernel runs)
    unsigned long long st_feat = 0, st_hit = 0, st_loc = 0, st_cand = 0, st_two = 0, st_narrow = 0;
    float word_space;
    if constexpr (GW) word_space = (float)gwd.off[db.n_targets];
    else word_space = (db.wb < 32 && ((u64)db.n_targets << db.wb) < 0xFFFFFFFFull) ? (float)((u64)db.n_targets << db.wb) : 4294967040.0f;
#ifdef MCQ_WAVE_DYNQ
    for (u32 cb = 0;;) {
        if (lane == 0) cb = atomicAdd(&ctr->w_cursor[1], (u32)MCQ_WAVE_DYNQ_CH);
        cb = bcast(cb, 0);
        if (cb >= n_front) break;
        const u32 ce = cb + (u32)MCQ_WAVE_DYNQ_CH < n_front ? cb + (u32)MCQ_WAVE_DYNQ_CH : n_front;
    for (u32 it = cb; it < ce; ++it) {
#else
    {
    for (u32 it = blockIdx.x * 4 + wave; it < n_front; it += nwaves) {
#endif
        const u32 slot = ovf_visit(it, n_front);
        const u32 q32 = ovf_list[slot];
        if (q32 == MCQ_EMPTY) continue;                // unused tail of a wave's reservation
        const u64 q = q32;
        const ReadGeom g = read_geom(db, b, q, force_block);          // (as the first stage saw it)
        const u32 narrow = g.qlen < opt.tc_limit ? 1u : 0u;
        if (g.ovf && !g.wide) { st_narrow += narrow; continue; }      // more than 128 features: a workgroup's
        u32 nfeat = 0;
        bool two = false, counted = false;
        u64 off0 = 0, off1 = 0; u32 len0 = 0, len1 = 0;
        if constexpr (SH) {
            nfeat = (g.nw1 + g.nw2) * db.s;
            two = nfeat > 64;
            const u64 sb = sh.win_off[b.paired ? 2 * q : q] * db.s;
            if (lane < nfeat) shard_fetch(sh, sb + lane, off0, len0);
            if (two && 64 + lane < nfeat) shard_fetch(sh, sb + 64 + lane, off1, len1);
        }
#ifndef MCQ_NO_FRONT_HANDOVER
        else if (!g.ovf) {                             // <= 64 features: the first stage left its probe results in the slot's row
            const u64 pw = ctr->probe_front[(u64)slot * 64 + lane];
            const u32 l = (u32)(pw & 0xFFFFu);
            nfeat = (u32)__builtin_popcountll(__ballot(l != 0xFFFFu));
            len0 = l == 0xFFFFu ? 0u : l; off0 = (pw << 1) >> 17;
            counted = __ballot((pw >> 63) != 0) != 0;             // handed on by the second stage: counted by the first one
        }
#endif
        else {
            for (u32 w = 0; w < g.nw1 + g.nw2; ++w) {
                u64 at; u32 wl;
                window_span(db, g, w, at, wl);
                nfeat += wave_sketch_b(b, at, wl, db.k, db.s, lane, sk_tmp, feat + nfeat);
            }
            two = nfeat > 64;                          // wave-uniform
            const u32 myf0 = lane < nfeat ? feat[lane] : MCQ_EMPTY;
            const u32 myf1 = (two && 64 + lane < nfeat) ? feat[64 + lane] : MCQ_EMPTY;
            probe<BSH>(db, myf0, off0, len0);
            if (two) probe<BSH>(db, myf1, off1, len1);
        }
        const u32 incl0 = wave_incl_scan_dpp(len0);
        const u32 T0 = bcast(incl0, 63);
        const u32 incl1 = wave_incl_scan_dpp(len1) + T0;
        const u32 pos0 = incl0 - len0, pos1 = incl1 - len1;
        const u32 T = bcast(incl1, 63);
        if (T > (u32)LCAP || T == 0) { st_narrow += narrow; continue; }          // (an empty list: the workgroup kernel writes the zero)
        wave_sync();                                   // feat[] (aliasing hits) has been consumed
        const u32 numWindows = range_width(g.qlen, opt.insert_size_max, db.tgt_winstride, db.magic_tgt_stride);
        u32 n2;
        if (T <= 2048u) {                               // wave-uniform
            u32 r[32];
            gather_regs2<32>(db, r, T, pos0, len0, off0, pos1, len1, off1, two, lane, hits);
            n2 = two_class_tail<32>(db, opt, out, r, T, numWindows, word_space, lf, q, lane, buf, hits);
        } else {
#ifndef MCQ_WAVE32_ONLY
            u32 r[64];
            gather_regs2<64>(db, r, T, pos0, len0, off0, pos1, len1, off1, two, lane, hits);
            n2 = two_class_tail<64>(db, opt, out, r, T, numWindows, word_space, lf, q, lane, buf, hits);
#else
            n2 = ~0u;
#endif
        }
        wave_sync();
        if (n2 >= ~1u) { st_narrow += narrow; continue; }             // not taken / not provable: the entry stays
        if (lane == 0) ovf_list[slot] = MCQ_EMPTY;                    // answered
        if (!counted) {
            if constexpr (!SH) st_feat += nfeat;
            st_hit += (u32)__builtin_popcountll(__ballot(len0 > 0)) + (u32)__builtin_popcountll(__ballot(len1 > 0));
        }
        st_loc += T; st_cand += n2; st_two += 1;
    }
    }
    if (lane == 0 && st_two) atomicAdd(&ctr->n_two_class, st_two);
    if (lane == 0 && st_narrow) atomicAdd(&ctr->n_narrow, st_narrow);
    if (lane == 0 && (st_feat | st_loc | st_hit)) {
        if (st_feat) atomicAdd(&ctr->n_features, st_feat);
        atomicAdd(&ctr->n_hit_features, st_hit);
        atomicAdd(&ctr->n_locations, st_loc);
        atomicAdd(&ctr->n_cands, st_cand);
    }
}

// ------------------------------------------------------------------ kernel: workgroup per query
// -DMCQ_PHASE_CLOCK (diagnostic builds only): thread 0 of every workgroup adds the shader clocks between the phase marks of the
// workgroup kernel into 24 words of LDS, folded into CountersDev::pad_ when the kernel ends (mcq_debug_phase_clocks).
#ifdef MCQ_PHASE_CLOCK
#define PHCLK(ph, i) do { if (threadIdx.x == 0) { const u64 t_ = __builtin_amdgcn_s_memtime(); (ph)[i] += t_ - (ph)[23]; (ph)[23] = t_; } } while (0)
#else
#define PHCLK(ph, i) do { } while (0)
#endif
// exclusive scan of a[0..n) in place by the whole workgroup; returns the total
__device__ __forceinline__ u32 block_excl_scan(u32* a, u32 n, u32 tid, u32* s_w /* >= 18 words */) {
    const u32 NTB = blockDim.x;
    const u32 lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_w[17] = 0;
    __syncthreads();
    for (u32 base = 0; base < n; base += NTB) {
        u32 i = base + tid;
        u32 v = (i < n) ? a[i] : 0;
        u32 x = wave_incl_scan_dpp(v);
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        u32 woff = 0;
        for (u32 w = 0; w < wave; ++w) woff += s_w[w];
        u32 carry = s_w[17];
        if (i < n) a[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == NTB - 1) s_w[17] = carry + woff + x;
        __syncthreads();
    }
    return s_w[17];
}

// The same for n <= 4 x blockDim.x values of which thread tid holds a[tid + c x blockDim.x] (it wrote them itself: no barrier needed
// in front) in TWO barriers whatever n is: wave scans of every chunk, the (chunk, wave) totals through 64 words of LDS, their
// prefix by one more wave scan in every wave.  (block_excl_scan takes four barriers per 1024 values: 9 for the 1136 features of
// an 8 kb read, with the barrier in front -- measured with the phase clocks of r04: probe + scan were 19 % of the kernel.)
__device__ __forceinline__ u32 block_excl_scan4(u32* a, u32 n, u32 tid) {
    __shared__ u32 s_tot[64];
    const u32 NTB = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = NTB >> 6;
    u32 v[4], x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u32 i = (u32)c * NTB + tid;
        v[c] = i < n ? a[i] : 0u;
        x[c] = wave_incl_scan_dpp(v[c]);
        if (lane == 63) s_tot[c * nwv + wave] = x[c];
    }
    __syncthreads();
    const u32 t = lane < 4 * nwv ? s_tot[lane] : 0u;              // (nwv <= 16)
    const u32 incl = wave_incl_scan_dpp(t);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u32 i = (u32)c * NTB + tid;
        const u32 j = c * nwv + wave;                               // (wave-uniform)
        const u32 before = bcast(incl - t, j);
        if (i < n) a[i] = before + x[c] - v[c];
    }
    const u32 total = bcast(incl, 63);
    __syncthreads();
    return total;
}

// Tail of the workgroup path: fill B[0..n2p) through `load(t)`, sort, sweep, top lists.
// HT/JB: packed (hits << JB | index) word of the sweep: u32 with JB = 13 when the list fits the workgroup's LDS
// (<= 8192 entries, hits <= 8192), u64 with JB = 32 in global scratch
// filled: B[0..T) holds the unsorted list already
template <class KeyT, class HT, int JB, int BIG, bool RTLIN = false, class LF, class Fill, bool HGLOBAL = false>
__device__ __forceinline__ void block_tail(const DbDev& db, const OptDev& opt, const OutDev& out, CountersDev* ctr,
                                           KeyT* B, HT* H, u32 T, u32 numWindows, const LF& lf, u64 q, u32 tid,
                                           const DebugDev& dbg, u32* biglist, Fill fill, bool filled = false) {
    const u32 n2p = pow2ceil(T), NTB = blockDim.x;
#ifdef MCQ_SORT_PAD_FULL                                       // tuning knob (A/B): the whole power-of-two network
    const u32 npad = n2p;
#else
    const u32 npad = n2p < 256 ? n2p : ((T + 127u) & ~127u);  // padded to whole 128-key chunks (see bitonic_sort_block)
#endif
    if (!filled) fill(B);                                      // B[0..T) = the unsorted match list
    for (u32 t = T + tid; t < npad; t += NTB) B[t] = key_pad<KeyT>();
    __syncthreads();
    bitonic_sort_block(B, n2p, npad, tid, NTB, [] { __syncthreads(); });
    if (dbg.mode == 2) {
        for (u32 t = tid; t < T; t += NTB) dbg.matches[dbg.match_off[q] + t] = key_expand<KeyT>(B[t], lf);
    }
    __shared__ TopkBlockScratch<HT> s_topk;
    sweep_targets<KeyT, HT, JB, LF, HGLOBAL>(B, H, T, numWindows, lf, tid, NTB, s_topk.fmx, [] { __syncthreads(); });
    const u32 n = topk_block<KeyT, HT, JB, BIG, RTLIN>(db, opt, out, B, H, T, numWindows, lf, q, tid, NTB, &s_topk, biglist, [] { __syncthreads(); });
    if (tid == 0) atomicAdd(&ctr->n_cands, (unsigned long long)n);
    __syncthreads();
}

// ---- the two-class tail of the workgroup kernel (32-bit words, the list in LDS) ---------------------------------------
// Long reads and the short reads of a RefSeq-scale table that outgrow the wave stages: a few thousand locations, most of
// them chance hits alone on their target.  Same split as in the wave stage (mcq_device.hpp, "rows 8-11 in two classes"), by
// the whole workgroup: the list B[0..T) is filled as before; the hit words' LDS holds the two cell maps (2^17 cells each);
// every thread classifies its words; the heavy ones are packed into the hit words' segment, sorted there (a 2430-word list
// of an 8 kb read keeps ~1050 of them), swept with the packed heads in B's segment; and ONE wave builds the lists from
// the heads that matter and the light prefix (topk_two_class_lds: entries in B's segment).  Returns TC_DONE when the query is
// answered; else the caller runs the exact tail (filling B again unless TC_HEAVY left it intact).
#define MCQ_TC_MAX_WINDOWS 16u          // widest window range the workgroup kernel's two-class tail takes (see k_query_block)
#define MCQ_TC_MIN_QUEUED 4096ull       // fewest narrow queries in the front queue that get the two-class workgroup kernel
enum { TC_NOT_TRIED = 0, TC_DONE = 1, TC_HEAVY = 2, TC_FAILED = 3 };      // TC_HEAVY: given up early, B is filled and intact
template <int LCAPB, class LF, class Fill>
__device__ __forceinline__ int block_two_class(const DbDev& db, const OptDev& opt, const OutDev& out, CountersDev* ctr, u32* B, u32* HW, u32 T,
                                               u32 numWindows, float word_space, const LF& lf, u64 q, u32 tid, u32* s_x /* 8 words */, Fill fill) {
    constexpr u32 LOG = 31u - __builtin_clz((u32)LCAPB) + 4;               // both maps fill the LCAPB hit words: 2 x 2^(LOG - 5) words
    constexpr u32 MAPW = 1u << (LOG - 5), NPRE = 64, JB = 13;
    static_assert(2 * MAPW <= (u32)LCAPB && LCAPB <= 8192, "cell maps live in the hit words; 13 index bits");
    const u32 NTB = blockDim.x, lane = tid & 63;
    // (numWindows > 16 = long reads: their locations are clusters around the strains' loci, hundreds of words per cell -- the
    // LDS atomics of the maps collide on a few words and the tail, measured on 8 kb reads, costs 5 % instead of saving)
    if (T > (u32)LCAPB - 256u || numWindows > MCQ_TC_MAX_WINDOWS) return TC_NOT_TRIED;          // (P x M and the hook: decided by the host, which launches this kernel or not)
    u32* occ = HW; u32* multi = HW + MAPW;
    fill(B);
    __syncthreads();                                         // every wave has read the list starts (they live where the maps go)
    cells_clear<LOG>(occ, multi, tid, NTB);
    if (tid < 8) s_x[tid] = 0;
    __syncthreads();
    const u32 cs = cell_shift(numWindows);
    for (u32 t = tid; t < T; t += NTB) cells_insert<LOG>(B[t], cs, occ, multi);
    __syncthreads();
    u32 hm = 0;                                              // bit i: my i-th word is heavy (T <= 8192: at most 8 words per thread)
    for (u32 t = tid, i = 0; t < T; t += NTB, ++i) hm |= cells_heavy<LOG>(B[t], cs, occ, multi) ? 1u << i : 0u;
    {   // mostly heavy words (a long read on a small table: every location belongs to a strain's cluster): nothing to gain
        const u32 mine = (u32)__builtin_popcount(hm);
        u32 wsum = mine;
        for (int d = 32; d > 0; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
        if (lane == 0 && wsum) atomicAdd(&s_x[3], wsum);
    }
    __syncthreads();                                         // the maps are dead: heavy words -> HW[0..nH), light prefix -> HW[LCAPB - 64 ..)
    if (4 * s_x[3] > 3 * T) return TC_HEAVY;
    const float th = word_space * MCQ_TWO_CLASS_EXPECT * __builtin_amdgcn_rcpf((float)T);
    const u32 theta = th >= 4294967040.0f ? 0xFFFFFFFEu : (u32)th;
    u32* pre = HW + (LCAPB - NPRE);
    for (u32 t0 = 0, i = 0; t0 < T; t0 += NTB, ++i) {        // (uniform trip count: the ballots need every lane)
        const u32 t = t0 + tid;
        const bool valid = t < T, heavy = (hm >> i) & 1u;
        const u32 k = valid ? B[t] : 0u;
        const u64 bh = __ballot(valid && heavy);
        u32 base = 0;
        if (lane == 0 && bh) base = atomicAdd(&s_x[0], (u32)__builtin_popcountll(bh));
        base = bcast(base, 0);
        if (valid && heavy) HW[base + lane_rank(bh)] = k;
        const bool lp = valid && !heavy && k < theta;
        const u64 bl = __ballot(valid && !heavy), bp = __ballot(lp);
        u32 pb = 0;
        if (lane == 0) { if (bl) atomicAdd(&s_x[1], (u32)__builtin_popcountll(bl)); if (bp) pb = atomicAdd(&s_x[2], (u32)__builtin_popcountll(bp)); }
        pb = bcast(pb, 0);
        if (lp) { const u32 i2 = pb + lane_rank(bp); if (i2 < NPRE) pre[i2] = k; }
    }
    __syncthreads();
    const u32 nH = s_x[0], nL = s_x[1], nP = s_x[2];
    if (nP > NPRE) return TC_FAILED;                         // theta too generous for this read
    // sort the heavy words in place (padded to whole 128-word chunks), sweep them with the packed heads in B's segment
    const u32 n2p = pow2ceil(nH), npad = n2p < 256 ? n2p : ((nH + 127u) & ~127u);
    for (u32 t = nH + tid; t < npad; t += NTB) HW[t] = MCQ_EMPTY;
    const u32 safe = B[0];
    __syncthreads();
    bitonic_sort_block(HW, n2p, npad, tid, NTB, [] { __syncthreads(); });
    u32* H = B;
    if (nH) sweep_targets<u32, u32, JB>(HW, H, nH, numWindows, lf, tid, NTB, HW + (LCAPB - 128) /* 17 words behind the sorted words */, [] { __syncthreads(); });
    __syncthreads();
    // ONE wave: the run heads that matter (two or more hits, or a word below theta) to the front, then the lists
    if (tid < 64) {
        u32 nheads = 0;
        bool dropped = false;
        for (u32 base = 0; base < nH; base += 64) {
            const u32 j = base + lane;
            const u32 hv = (j < nH) ? H[j] : 0;
            const u32 h = hv >> JB, jb = ((1u << JB) - 1) - (hv & ((1u << JB) - 1));
            const bool one = hv != 0 && h == 1u && HW[jb] >= theta;
            const u64 hb = __ballot(hv != 0 && !one);
            if (hv != 0 && !one) H[nheads + lane_rank(hb)] = hv;
            nheads += (u32)__builtin_popcountll(hb);
            dropped = dropped || __ballot(one) != 0;
        }
        wave_sync();
        if (nH == 0 && lane == 0) HW[0] = safe;              // (idle lanes look up SK[0])
        wave_sync();
        u32 n = ~0u;
        const u32 lkey = lane < nP ? pre[lane] : MCQ_EMPTY;
        if (nheads <= 1024) n = topk_two_class_lds<JB, 1088>(db, opt, out, HW, nH, H, nheads, lkey, nP, nL > nP || dropped, theta, numWindows, lf, q, lane, H + 1024);
        if (lane == 0) { s_x[7] = n; if (n != ~0u) { atomicAdd(&ctr->n_cands, (unsigned long long)n); atomicAdd(&ctr->n_two_class, 1ull); } else atomicAdd(&ctr->n_two_class_retry, 1ull); }
    }
    __syncthreads();
    const bool done = s_x[7] != ~0u;
    __syncthreads();
    return done ? TC_DONE : TC_FAILED;
}

// NT threads per workgroup; LCAPB entries of the match list fit its LDS (longer lists are sorted in global scratch).
// 32-bit keys: 8192 x (4 + 4) B = 64 KB of LDS and 64 VGPRs, so two 1024-thread workgroups share a CU -- the phases
// of a query are serialised by workgroup barriers, and the second workgroup fills the gaps (+45 % on 8 kb reads;
// <4096, 512> with four per CU is slower: 15 % of those reads then sort in global scratch).
// TC: the instantiation with the two-class tail.  Merely carrying that code costs the kernel 10 % on long reads (register
// allocation: 20 spilled VGPRs around the query loop), which never take it -- so it is a kernel of its own, launched behind the
// plain one over the same queue: queries with narrow window ranges (numWindows <= 16: short reads and pairs whose lists
// outgrew the wave stages) are left to it (OptDev::hooks bit 16 tells the plain kernel), everything else to the plain one.
template <class KeyT, int LCAPB, int NT, int BIG = 0, bool SH = false, bool GW = false, bool TC = false>
__global__ __launch_bounds__(NT, sizeof(KeyT) == 4 ? 8 : 4) void k_query_block(DbDev db, BatchDev b, OptDev opt, OutDev out,
                                                      CountersDev* ctr, const u32* ovf_list, ScratchDev sc, DebugDev dbg, ShardDev sh, GwDev gwd) {
    static_assert(LCAPB <= 8192, "packed sweep word: 13 index bits");
    const typename LocOf<KeyT, GW>::type lf = loc_format<KeyT, GW>(db, gwd);
    constexpr u32 NW16 = NT / 64;
    // the key segment and the hit words behind it, in ONE allocation: lists of 8193 .. 16384 32-bit words sort across both (r04)
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[(size_t)LCAPB * (sizeof(KeyT) + 4)];
    KeyT* const s_buf = reinterpret_cast<KeyT*>(s_mem);
    u32* const s_hits = reinterpret_cast<u32*>(s_mem + (size_t)LCAPB * sizeof(KeyT));
    __shared__ u32 s_w[20];
    __shared__ u32 s_biglist[BIG == 1 ? 2 * MCQ_BIGLIST_MAX : 1];  // P lists of M entries when they do not fit a wave (OptDev::big)
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 W = db.winlen, S = db.winstride;
    u32* g_feat = sc.feat + (u64)blockIdx.x * sc.fmax;
    u32* g_fpos = sc.fpos + (u64)blockIdx.x * ((u64)sc.fmax + 1);
    u64* g_foff = sc.foff + (u64)blockIdx.x * sc.fmax;
    KeyT* gbuf = reinterpret_cast<KeyT*>(sc.gbuf + (u64)blockIdx.x * sc.lmax);
    u64* ghits = sc.ghits + (u64)blockIdx.x * sc.lmax;
    const KeyT* __restrict__ locs = static_cast<const KeyT*>(db.locs);
    const u32 n_ovf = ctr->ovf_count;
    u32* sk = s_hits + wave * 128;                             // per-wave sketch scratch (hit words unused yet)
    u32 tc_skip = 0;                                           // queries for which the two-class tail is not attempted (uniform)
    // a kernel of its own for the narrow queries only pays when there are enough of them (a batch of long reads has a few
    // hundred short ones: the plain kernel takes them along, the other one returns at once)
    const bool tc_worth = ctr->n_narrow >= MCQ_TC_MIN_QUEUED;
    if (TC && !tc_worth) return;

    // The queue entries of this workgroup's next 64 visits are fetched at once: most entries are empty when the wave stages
    // have answered them (a RefSeq-scale batch of pairs: 99 %), and one dependent load per visit was 0.5 ms per kernel there.
    __shared__ u32 s_q[64];
#ifdef MCQ_PHASE_CLOCK
    __shared__ u64 s_ph[24];
    if (tid < 24) s_ph[tid] = 0;
    __syncthreads();
    if (tid == 0) s_ph[23] = __builtin_amdgcn_s_memtime();
    u64* const ph = s_ph;
#else
    u64* const ph = nullptr;
#endif
    // The workgroups take their queue entries from a shared cursor, a chunk at a time (r04; until then entry k belonged to workgroup
    // k mod 512): queries cost by their length -- ONT-like reads: log-normal, sigma 0.6 -- and of 512 workgroups with 32 reads each
    // the unluckiest took a third longer than the average while the others had gone idle (2.55 -> 2.03 ms per 16 384 reads).  The
    // chunk: 1/32 of a workgroup's even share, at most 64 (a RefSeq-scale batch of pairs leaves a million entries of which 99 %
    // are empty markers: fetched 64 at a time), at least 1.  Two passes over the queue, each with its cursor: first the queries of
    // MCQ_BLOCK_LONG_FIRST bases and more -- the longest ones take a workgroup many times the average, and one that is started last
    // is what the whole grid then waits for -- then the others.  Every workgroup leaves a pass when its cursor has passed the end.
    __shared__ u32 s_chunk;
    const bool ctr_long = ctr->n_long != 0;            // (written by the first wave stage, which is done)
#ifndef MCQ_BLOCK_QUEUE_DIV
#define MCQ_BLOCK_QUEUE_DIV 32u                         // tuning knob: chunks per workgroup's even share (measured: 2 -> 2.45, 8 -> 2.11, 32 -> 2.03 ms)
#endif
    const u32 share = n_ovf / (gridDim.x * MCQ_BLOCK_QUEUE_DIV);
    const u32 C1 = share < 1u ? 1u : (share > 64u ? 64u : share);
    for (u32 pass = (MCQ_BLOCK_LONG_FIRST && ctr_long) ? 0u : 1u; pass < 2u; ++pass) {       // (no long queries in the batch: no first pass)
    const u32 C = pass ? C1 : (C1 * 16u > 64u ? 64u : C1 * 16u);       // (the long ones are few: the first pass mostly skips)
    for (;;) {
    __syncthreads();                                   // (the entries of the chunk before are consumed)
    if (tid == 0) s_chunk = atomicAdd(&ctr->blk_cursor[(TC ? 2 : 0) + pass], C);
    __syncthreads();
    const u32 it0 = s_chunk;
    if (it0 >= n_ovf) break;                           // (uniform)
    if (tid < C) { const u32 it = it0 + tid; s_q[tid] = it < n_ovf ? ovf_list[ovf_visit(it, n_ovf)] : MCQ_EMPTY; }
    __syncthreads();
    for (u32 k = 0; k < C; ++k) {
        const u32 q32 = s_q[k];
        if (q32 == MCQ_EMPTY) continue;                // answered, or the unused tail of a wave's reservation (uniform over the workgroup)
        const u64 q = q32;
        const u64 a = b.paired ? 2 * q : q;
        u64 o0, e0, o1, e1;
        seq_bounds(b.seq_off, b.ranges, a, o0, e0);
        if (b.paired) seq_bounds(b.seq_off, b.ranges, a + 1, o1, e1); else { o1 = e0; e1 = e0; }
        const u64 n1 = e0 - o0, n2 = e1 - o1;
        if (MCQ_BLOCK_LONG_FIRST && ctr_long && (n1 + n2 >= MCQ_BLOCK_LONG_FIRST) != (pass == 0u)) continue;      // (the other pass's)
        const u32 nw1 = num_windows(n1, W, S), nw2 = b.paired ? num_windows(n2, W, S) : 0;
        const u64 NW = (u64)nw1 + nw2;
        // which of the two workgroup kernels takes this query (uniform): narrow window ranges = short queries (opt.tc_limit: the
        // first length with numWindows > MCQ_TC_MAX_WINDOWS, 0 = no two-class kernel behind this one)
        if (((n1 + n2 < opt.tc_limit) && tc_worth) != TC) continue;
        if (NW * db.s > sc.fmax) {                        // beyond the workspace: flag, no result
            if (tid == 0) { out.ncand[q] = 0; atomicAdd(&ctr->err_count, 1u); if (dbg.mode == 1) dbg.match_cnt[q] = 0; }
            continue;
        }
        // features, list lengths and list starts of the query live in LDS when it has at most 2048 features
        // (behind the waves' sketch scratch in the hit words, which the sweep only needs later), else in global scratch
        const bool f_lds = NW * db.s <= 2048 && NT == 1024;
        u32* feat = f_lds ? s_hits + 2048 : g_feat;
        u32* fpos = f_lds ? s_hits + 2048 : g_fpos;             // a feature's word becomes its list length, then its start
        u64* foff = f_lds ? reinterpret_cast<u64*>(s_hits + 4096) : g_foff;
        if (tid == 0) { s_w[18] = 0; s_w[19] = 0; }
        __syncthreads();
        PHCLK(ph, 0);
        if constexpr (!SH) {
        // (Loading the next window's bases before this one is sketched -- a wave sketches four or five windows of an 8 kb read one after
        // the other -- was measured in r04: +4 % kernel time, same box; removed.)
        for (u32 w = wave; w < (u32)NW; w += NW16) {
            const bool m2 = w >= nw1;
            const u64 n = m2 ? n2 : n1;
            const u64 sb = m2 ? o1 : o0;
            u64 beg; u32 wl;
            window_of(n, W, S, m2 ? w - nw1 : w, beg, wl);
            u32 m = wave_sketch_b(b, sb + beg, wl, db.k, db.s, lane, sk, sk + 64);
            u32 base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_w[18], m);
            base = bcast(base, 0);
            if (lane < m) feat[base + lane] = sk[64 + lane];
            wave_sync();
        }
        __syncthreads();
        }
        PHCLK(ph, 1);
        const u32 F = SH ? (u32)(NW * db.s) : s_w[18];     // sharded: every feature slot of the query (unused ones have no list)
        u32 nhit = 0;
        for (u32 i = tid; i < F; i += NT) {
            u64 off; u32 len;
            if constexpr (SH) shard_fetch(sh, sh.win_off[a] * db.s + i, off, len);
            else probe(db, feat[i], off, len);
            foff[i] = off; fpos[i] = len; nhit += (len > 0);
        }
        nhit = wave_incl_scan_dpp(nhit);
        if (lane == 63 && nhit) atomicAdd(&s_w[19], nhit);
        u32 T;
#ifndef MCQ_NO_SCAN4                                    // tuning knob (A/B)
        if (F <= 4 * NT && NT >= 256) T = block_excl_scan4(fpos, F, tid);       // (uniform; the thread scans the lengths it wrote itself)
#else
        if (false) { }
#endif
        else { __syncthreads(); T = block_excl_scan(fpos, F, tid, s_w); }
        PHCLK(ph, 2);
        if (tid == 0) {
            if (!SH) atomicAdd(&ctr->n_features, (unsigned long long)F);
            atomicAdd(&ctr->n_locations, (unsigned long long)T);
            atomicAdd(&ctr->n_hit_features, (unsigned long long)s_w[19]);
        }
        if (dbg.mode == 1) { if (tid == 0) dbg.match_cnt[q] = T; }
        if (T == 0) { if (tid == 0) out.ncand[q] = 0; continue; }
        if (T > sc.lmax || pow2ceil(T) > sc.lmax) {
            if (tid == 0) { out.ncand[q] = 0; atomicAdd(&ctr->err_count, 1u); }
            continue;
        }
        const u32 numWindows = range_width(n1 + n2, opt.insert_size_max, db.tgt_winstride, db.magic_tgt_stride);
        // every wave copies the lists of 64 features at a time (fpos = exclusive scan of the list lengths): list of
        // an element by a shuffle search inside the group, four 64-element chunks of loads in flight
        auto fill = [&](KeyT* B) {
            for (u32 g = wave; g * 64 < F; g += NW16) {
                const u32 i = g * 64 + lane;
                const bool valid = i < F;
                const u32 start = valid ? fpos[i] : T;
                const u32 next = (i + 1 < F) ? fpos[i + 1] : T;
                const u32 len = valid ? next - start : 0;
                const u64 off = valid ? foff[i] : 0;
                const u32 incl = wave_incl_scan_dpp(len);
                const u32 pos = incl - len;
                const u32 Tg = bcast(incl, 63);
                const u32 gbase = bcast(start, 0);
                for (u32 base = 0; base < Tg; base += 256) {
                    KeyT v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v[u] = 0;
                        if (base + (u32)(u * 64) < Tg) {
                            const u32 t = base + u * 64 + lane;
                            const u32 tt = t < Tg ? t : Tg - 1;
                            u32 lo = 0;
#pragma unroll
                            for (u32 step = 32; step > 0; step >>= 1) {
                                u32 c = lo + step;
                                u32 pc = __shfl(pos, (int)(c & 63), 64);
                                if (c < 64 && pc <= tt) lo = c;
                            }
                            const u32 pj = __shfl(pos, (int)lo, 64);
                            const u32 olo = __shfl((u32)off, (int)lo, 64), ohi = __shfl((u32)(off >> 32), (int)lo, 64);
                            if (t < Tg) v[u] = locs[(((u64)ohi << 32) | olo) + (tt - pj)];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 t = base + u * 64 + lane;
                        if (t < Tg) B[gbase + t] = v[u];
                    }
                }
            }
        };
#ifdef MCQ_NO_BLOCK_MID                                // tuning knob (A/B)
        constexpr bool mid_ok = false;
#else
        constexpr bool mid_ok = sizeof(KeyT) == 4 && BIG == 2;
#endif
        // (the sort is padded to whole 128-key chunks only, so a list fits the LDS whenever that many keys do)
        if (((T + 127u) & ~127u) <= (u32)LCAPB && pow2ceil(T) <= 8192u) {
            int tc = TC_NOT_TRIED;
            if constexpr (TC) {
                // (reads of one batch are alike: after a query that was mostly heavy words the workgroup skips the attempt for
                // the next 15 -- results are the same either way)
                if (dbg.mode == 0 && tc_skip == 0) {
                    // the two-class tail keeps its cell maps where the query's list starts live (LDS case): a copy in global
                    // scratch lets the exact tail fill the list again, should the lists not be provable
                    if (f_lds) {
                        for (u32 i = tid; i < F; i += NT) { g_fpos[i] = fpos[i]; g_foff[i] = foff[i]; }
                        __syncthreads();
                    }
                    // size of the space the location words live in (light-word threshold)
                    float word_space;
                    if constexpr (GW) word_space = (float)gwd.off[db.n_targets];
                    else word_space = (db.wb < 32 && ((u64)db.n_targets << db.wb) < 0xFFFFFFFFull) ? (float)((u64)db.n_targets << db.wb) : 4294967040.0f;
                    tc = block_two_class<LCAPB>(db, opt, out, ctr, reinterpret_cast<u32*>(s_buf), s_hits, T, numWindows, word_space, lf, q, tid, s_w, fill);
                    if (tc != TC_DONE && f_lds) { fpos = g_fpos; foff = g_foff; __threadfence_block(); __syncthreads(); }
                    if (tc == TC_HEAVY) tc_skip = 15;
                } else if (tc_skip) --tc_skip;
            }
            const bool done = tc == TC_DONE, filled = tc == TC_HEAVY;
            if (!done) { block_tail<KeyT, u32, 13, BIG>(db, opt, out, ctr, s_buf, s_hits, T, numWindows, lf, q, tid, dbg, s_biglist, fill, filled); PHCLK(ph, 10); }
        }
        // 8193 .. 16384 words (ONT-like reads of 27 .. 55 kb: one in a hundred, and until r04 a sixth of the kernel's time -- a workgroup
        // took 23-27 us per kb of such a read against 6.5 below, everything in global scratch): the list sorts in the LDS of BOTH
        // segments (32-bit words; the query's feature arrays are in global scratch at this size), only the sweep's packed words and
        // the lists' scans go through global memory, the atomics reduced per run inside a wave first (sweep_targets<..., true>)
        else if (mid_ok && !f_lds && dbg.mode == 0 && ((T + 127u) & ~127u) <= 2u * (u32)LCAPB) {
            if constexpr (sizeof(KeyT) == 4 && BIG == 2)
                block_tail<KeyT, u32, 14, BIG, false, decltype(lf), decltype(fill), true>(db, opt, out, ctr, s_buf, reinterpret_cast<u32*>(ghits), T, numWindows, lf, q, tid, dbg, s_biglist, fill);
        }
        else                           block_tail<KeyT, u64, 32, BIG>(db, opt, out, ctr, gbuf, ghits, T, numWindows, lf, q, tid, dbg, s_biglist, fill);
    }
    }
    }
#ifdef MCQ_PHASE_CLOCK
    __syncthreads();
    if (tid < 17 && s_ph[tid]) atomicAdd(&ctr->pad_[tid], (unsigned long long)s_ph[tid]);
#endif
}

// ------------------------------------------------------------------ staged kernels (sharded path, DB build)
__global__ void k_count_windows(const u64* seq_off, u32 ranges, u64 n_seqs, u32 W, u32 S, u64* cnt) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_seqs) { u64 bg, en; seq_bounds(seq_off, ranges, i, bg, en); cnt[i] = num_windows(en - bg, W, S); }
}

// one wave per window: window w belongs to the last sequence i with win_off[i] <= w
__global__ __launch_bounds__(256) void k_sketch_windows(DbDev db, const char* bases, const u64* seq_off, u32 ranges, u64 n_seqs,
                                                        const u64* win_off, u32* features, u32* n_feat) {
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 n_win = win_off[n_seqs];
    const u64 nwaves = (u64)gridDim.x * 4;
    __shared__ u32 s_sk[4][128];
    u32* sk = s_sk[wave];
    for (u64 w = (u64)blockIdx.x * 4 + wave; w < n_win; w += nwaves) {
        u64 lo = 0, hi = n_seqs;
        while (hi - lo > 1) { u64 mid = (lo + hi) >> 1; if (win_off[mid] <= w) lo = mid; else hi = mid; }
        u64 o0, oe; seq_bounds(seq_off, ranges, lo, o0, oe);
        const u64 n = oe - o0;
        u64 beg; u32 wl;
        window_of(n, db.winlen, db.winstride, (u32)(w - win_off[lo]), beg, wl);
        u32 m = wave_sketch(bases + o0 + beg, wl, db.k, db.s, lane, sk, sk + 64);
        if (lane < db.s) features[w * db.s + lane] = (lane < m) ? sk[64 + lane] : MCQ_EMPTY;
        if (lane == 0) n_feat[w] = m;
        wave_sync();
    }
}

// one wave per sequence, looping over its (few) windows: no search, window math in 32 bits
__global__ __launch_bounds__(256) void k_sketch_seqs(DbDev db, const char* bases, const u64* seq_off, u32 ranges, u64 n_seqs,
                                                     const u64* win_off, u32* features, u32* n_feat) {
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 nwaves = (u64)gridDim.x * 4;
    __shared__ u32 s_sk[4][128];
    u32* sk = s_sk[wave];
    for (u64 i = (u64)blockIdx.x * 4 + wave; i < n_seqs; i += nwaves) {
        u64 o0, oe; seq_bounds(seq_off, ranges, i, o0, oe);
        const u32 n = (u32)(oe - o0);
        const u64 w0 = win_off[i];
        const u32 nw = (u32)(win_off[i + 1] - w0);
        for (u32 j = 0; j < nw; ++j) {
            u32 beg, wl;
            window_of32(n, db.winlen, db.winstride, db.magic_stride, j, beg, wl);
            u32 m = wave_sketch(bases + o0 + beg, wl, db.k, db.s, lane, sk, sk + 64);
            if (lane < db.s) features[(w0 + j) * db.s + lane] = (lane < m) ? sk[64 + lane] : MCQ_EMPTY;
            if (lane == 0) n_feat[w0 + j] = m;
            wave_sync();
        }
    }
}

__global__ void k_lookup_count(DbDev db, const u32* features, u64 n, u32* list_len, u64* list_src) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 off; u32 len;
    probe(db, features[i], off, len);
    list_len[i] = len;
    if (list_src) list_src[i] = off;
}

// one wave per 64 consecutive features: (probe again unless the list starts were kept), then
// copy the lists cooperatively, in the handle's native location width
template <class KeyT>
__global__ __launch_bounds__(256) void k_lookup_gather(DbDev db, const u32* features, u64 n, const u32* list_len,
                                                       const u64* list_src, const u64* out_off, KeyT* out_locs) {
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 ngroups = (n + 63) / 64, nwaves = (u64)gridDim.x * 4;
    const KeyT* __restrict__ locs = static_cast<const KeyT*>(db.locs);
    for (u64 g = (u64)blockIdx.x * 4 + wave; g < ngroups; g += nwaves) {
        const u64 i = g * 64 + lane;
        u64 off = 0; u32 len = 0;
        if (i < n) {
            if (list_src) { off = list_src[i]; len = list_len[i]; }
            else probe(db, features[i], off, len);
        }
        const u64 obase = out_off[g * 64];
        u32 incl = wave_incl_scan_dpp(len);
        u32 pos = incl - len;
        const u32 T = bcast(incl, 63);
        for (u32 base = 0; base < T; base += 256) {           // four 64-element chunks in flight per round trip
            KeyT v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = 0;
                if (base + (u32)(u * 64) < T) {
                    const u32 t = base + u * 64 + lane;
                    const u32 tt = t < T ? t : T - 1;
                    u32 lo = 0;
#pragma unroll
                    for (u32 step = 32; step > 0; step >>= 1) {
                        u32 c = lo + step;
                        u32 pc = __shfl(pos, (int)(c & 63), 64);
                        if (c < 64 && pc <= tt) lo = c;
                    }
                    u32 pj = __shfl(pos, (int)lo, 64);
                    u32 olo = __shfl((u32)off, (int)lo, 64), ohi = __shfl((u32)(off >> 32), (int)lo, 64);
                    if (t < T) v[u] = locs[(((u64)ohi << 32) | olo) + (tt - pj)];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 t = base + u * 64 + lane;
                if (t < T) out_locs[obase + t] = v[u];
            }
        }
    }
}

// rows 8-11 from per-query location segments in the handle's native width (home GPU of the sharded path)
template <class KeyT, int E>
__device__ __forceinline__ void load_sort_store(KeyT* buf, const KeyT* src, u32 T, u32 lane) {
    KeyT r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const u32 t = e * 64 + lane; r[e] = t < T ? src[t] : key_pad<KeyT>(); }
    wave_regsort<KeyT, E>(r, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) buf[e * 64 + lane] = r[e];
}

template <class KeyT, int LCAP, bool GW = false>
__global__ __launch_bounds__(256) void k_reduce_wave(DbDev db, OptDev opt, OutDev out, CountersDev* ctr, u32* ovf_list,
                                                     u64 nq, const u64* loc_off, const KeyT* locs, const u32* query_len, GwDev gwd) {
    const typename LocOf<KeyT, GW>::type lf = loc_format<KeyT, GW>(db, gwd);
    __shared__ KeyT s_buf[4][LCAP];
    __shared__ u32 s_hits[4][LCAP];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    KeyT* buf = s_buf[wave];
    u32* hits = s_hits[wave];
    const u64 nwaves = (u64)gridDim.x * 4;
    unsigned long long st_loc = 0, st_cand = 0;
    __shared__ u32 s_ovf[4][5];
    if (lane == 0) ovf_init(s_ovf[wave]);
    for (u64 q = (u64)blockIdx.x * 4 + wave; q < nq; q += nwaves) {
        const u64 b0 = loc_off[q], T64 = loc_off[q + 1] - b0;
        if (T64 > (u64)LCAP || (opt.hooks & 4)) {   // the same two queues as in k_query_wave
            if (lane == 0) {
                if (sizeof(KeyT) == 4 && T64 > (u64)LCAP && T64 <= (u64)MCQ_LCAP_WAVE16 && !(opt.hooks & 6)) ovf_push(s_ovf[wave], 1, ctr, ovf_list, nq, (u32)q);
                else ovf_push(s_ovf[wave], 0, ctr, ovf_list, nq, (u32)q);
            }
            continue;
        }
        const u32 T = (u32)T64;
        st_loc += T;
        if (T == 0) { if (lane == 0) out.ncand[q] = 0; continue; }
        const u32 numWindows = range_width(query_len[q], opt.insert_size_max, db.tgt_winstride, db.magic_tgt_stride);
        if constexpr (sizeof(KeyT) == 4) {
            if (T <= MCQ_DEDUP_MAX_T && !(opt.hooks & 1)) {
                const u32* src = locs + b0;
                u32 D, k1, incl1, t1, tb1;
                if (T <= 64)       D = load_dedup_insert<1>(src, buf, hits, T, lane);
                else if (T <= 128) D = load_dedup_insert<2>(src, buf, hits, T, lane);
                else if (T <= 192) D = load_dedup_insert<3>(src, buf, hits, T, lane);
                else if (T <= 256) D = load_dedup_insert<4>(src, buf, hits, T, lane);
                else if (T <= 384) D = load_dedup_insert<6>(src, buf, hits, T, lane);
                else               D = load_dedup_insert<8>(src, buf, hits, T, lane);
                D = dedup_finish(D, reinterpret_cast<u32*>(buf), hits, lane, lf, k1, incl1, t1, tb1);
                if (D != ~0u) {
                    if (D <= 64 && numWindows <= 8) sweep_targets_regs(k1, incl1, tb1, reinterpret_cast<u32*>(buf), D, numWindows, lf, lane);
                    else sweep_targets_weighted(dedup_sk(hits), dedup_wp(hits), reinterpret_cast<u32*>(buf), D, numWindows, lf, lane);
                    st_cand += MCQ_TOPK_DEDUP(db, opt, out, dedup_sk(hits), reinterpret_cast<u32*>(buf), D, numWindows, lf, q, lane, t1);
                    wave_sync();
                    continue;
                }
                wave_sync();
            }
        }
        if (T <= 64)       load_sort_store<KeyT, 1>(buf, locs + b0, T, lane);
        else if (T <= 128) load_sort_store<KeyT, 2>(buf, locs + b0, T, lane);
        else if (T <= 256) load_sort_store<KeyT, 4>(buf, locs + b0, T, lane);
        else               load_sort_store<KeyT, 8>(buf, locs + b0, T, lane);
        wave_sync();
        sweep_targets_wave<KeyT>(buf, hits, T, numWindows, lf, lane);
        if constexpr (sizeof(KeyT) == 4) st_cand += topk_heads<9, LCAP>(db, opt, out, reinterpret_cast<const u32*>(buf), hits, T, numWindows, lf, q, lane);
        else st_cand += topk_heads64<KeyT>(db, opt, out, buf, hits, T, numWindows, lf, q, lane);
        wave_sync();
    }
    if (lane == 0) ovf_flush(s_ovf[wave], ctr, ovf_list, nq);
    if (lane == 0 && st_loc) { atomicAdd(&ctr->n_locations, st_loc); atomicAdd(&ctr->n_cands, st_cand); }
}

// second wave stage of the staged path (see k_query_wave16): 513..1024 locations, 16 keys per lane
template <bool GW = false>
__global__ __launch_bounds__(256, MCQ_WAVE16_OCC) void k_reduce_wave16(DbDev db, OptDev opt, OutDev out, CountersDev* ctr, const u32* ovf_list,
                                                                       u64 nq, const u64* loc_off, const u32* locs, const u32* query_len, GwDev gwd) {
    constexpr int LCAP = MCQ_LCAP_WAVE16, JB = 10;
    const typename LocOf<u32, GW>::type lf = loc_format<u32, GW>(db, gwd);
    __shared__ u32 s_buf[4][LCAP];
    __shared__ u32 s_hits[4][LCAP];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32* buf = s_buf[wave];
    u32* hits = s_hits[wave];
    const u32 nwaves = gridDim.x * 4;
    const u32 n_mid = ctr->ovf_mid_count;
    unsigned long long st_loc = 0, st_cand = 0;
    for (u32 it = blockIdx.x * 4 + wave; it < n_mid; it += nwaves) {
        const u32 q32 = ovf_list[ovf_slot(nq, 1, ovf_visit(it, n_mid))];
        if (q32 == MCQ_EMPTY) continue;
        const u64 q = q32;
        const u64 b0 = loc_off[q];
        const u32 T = (u32)(loc_off[q + 1] - b0);
        st_loc += T;
        const u32 numWindows = range_width(query_len[q], opt.insert_size_max, db.tgt_winstride, db.magic_tgt_stride);
        load_sort_store<u32, 16>(buf, locs + b0, T, lane);
        wave_sync();
        sweep_targets_wave<u32, JB>(buf, hits, T, numWindows, lf, lane);
        st_cand += topk_heads<JB, LCAP>(db, opt, out, buf, hits, T, numWindows, lf, q, lane);
        wave_sync();
    }
    if (lane == 0 && st_loc) { atomicAdd(&ctr->n_locations, st_loc); atomicAdd(&ctr->n_cands, st_cand); }
}

template <class KeyT, int LCAPB, int BIG = 0, bool GW = false>
__global__ __launch_bounds__(1024) void k_reduce_block(DbDev db, OptDev opt, OutDev out, CountersDev* ctr, const u32* ovf_list,
                                                       ScratchDev sc, const u64* loc_off, const KeyT* locs, const u32* query_len, GwDev gwd) {
    const typename LocOf<KeyT, GW>::type lf = loc_format<KeyT, GW>(db, gwd);
    __shared__ KeyT s_buf[LCAPB];
    __shared__ u32 s_hits[LCAPB];
    __shared__ u32 s_biglist[BIG == 1 ? 2 * MCQ_BIGLIST_MAX : 1];
    const u32 tid = threadIdx.x;
    KeyT* gbuf = reinterpret_cast<KeyT*>(sc.gbuf + (u64)blockIdx.x * sc.lmax);
    u64* ghits = sc.ghits + (u64)blockIdx.x * sc.lmax;
    const u32 n_ovf = ctr->ovf_count;
    DebugDev dbg; dbg.mode = 0; dbg.match_cnt = nullptr; dbg.match_off = nullptr; dbg.matches = nullptr;
    for (u32 it = blockIdx.x; it < n_ovf; it += gridDim.x) {
        const u32 q32 = ovf_list[ovf_visit(it, n_ovf)];
        if (q32 == MCQ_EMPTY) continue;
        const u64 q = q32;
        const u64 b0 = loc_off[q], T64 = loc_off[q + 1] - b0;
        if (T64 > sc.lmax) {
            if (tid == 0) { out.ncand[q] = 0; atomicAdd(&ctr->err_count, 1u); }
            continue;
        }
        const u32 T = (u32)T64;
        if (T == 0) { if (tid == 0) out.ncand[q] = 0; continue; }
        if (tid == 0) atomicAdd(&ctr->n_locations, (unsigned long long)T);
        const u32 numWindows = range_width(query_len[q], opt.insert_size_max, db.tgt_winstride, db.magic_tgt_stride);
        auto fill = [&](KeyT* B) { for (u32 t = tid; t < T; t += 1024) B[t] = locs[b0 + t]; };
        // (the sort is padded to whole 128-key chunks only, so a list fits the LDS whenever that many keys do)
        if (((T + 127u) & ~127u) <= (u32)LCAPB && pow2ceil(T) <= 8192u) block_tail<KeyT, u32, 13, BIG, true>(db, opt, out, ctr, s_buf, s_hits, T, numWindows, lf, q, tid, dbg, s_biglist, fill);
        else                           block_tail<KeyT, u64, 32, BIG, true>(db, opt, out, ctr, gbuf, ghits, T, numWindows, lf, q, tid, dbg, s_biglist, fill);
    }
}

// ------------------------------------------------------------------ sharded-path routing kernels
// Bucket features by owning shard (mcq_owner); EMPTY features are dropped.  Counting sort
// over workgroup tiles: (1) every workgroup counts its tile per shard, (2) one workgroup
// turns the [shard][workgroup] counts into start offsets (shard-major), (3) every workgroup
// places its features; waves of a workgroup claim their slice with an LDS atomic.
#define MCQ_BUCKET_MAX_SHARDS 64
__device__ __forceinline__ u32 owner_of(u32 f, u32 n_shards) {
    return f == MCQ_EMPTY ? 0xFFFFFFFFu : (u32)(((u64)tmh(f) * n_shards) >> 32);
}
__global__ __launch_bounds__(256) void k_bucket_count(const u32* features, u64 n, u32 n_shards, u64 tile,
                                                      unsigned long long* blk_counts /* [n_shards][gridDim.x] */) {
    __shared__ u32 s_c[MCQ_BUCKET_MAX_SHARDS];
    const u32 lane = threadIdx.x & 63;
    if (threadIdx.x < MCQ_BUCKET_MAX_SHARDS) s_c[threadIdx.x] = 0;
    __syncthreads();
    const u64 t0 = (u64)blockIdx.x * tile, t1 = t0 + tile < n ? t0 + tile : n;
    for (u64 i0 = t0 + (threadIdx.x & ~63u); i0 < t1; i0 += 256) {
        const u64 i = i0 + lane;
        const u32 own = owner_of(i < t1 ? features[i] : MCQ_EMPTY, n_shards);
        for (u32 o = 0; o < n_shards; ++o) {
            u32 c = (u32)__builtin_popcountll(__ballot(own == o));
            if (lane == 0 && c) atomicAdd(&s_c[o], c);
        }
    }
    __syncthreads();
    if (threadIdx.x < n_shards) blk_counts[(u64)threadIdx.x * gridDim.x + blockIdx.x] = s_c[threadIdx.x];
}
// exclusive scan over the [shard][workgroup] matrix in shard-major order; totals per shard to counts[]
__global__ __launch_bounds__(1024) void k_bucket_scan(unsigned long long* blk_counts, u32 n_shards, u32 n_blocks, unsigned long long* counts) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 n = (u64)n_shards * n_blocks;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < n; base += 1024) {
        const u64 i = base + tid;
        unsigned long long v = i < n ? blk_counts[i] : 0, x = v;
        for (int d = 1; d < 64; d <<= 1) { unsigned long long t = __shfl_up(x, d, 64); if (lane >= (u32)d) x += t; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        unsigned long long woff = 0;
        for (u32 w = 0; w < wave; ++w) woff += s_w[w];
        const unsigned long long carry = s_carry;
        if (i < n) blk_counts[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + woff + x;
        __syncthreads();
    }
    // per-shard totals = difference of the shard's first offsets
    if (tid < n_shards) {
        unsigned long long b0 = blk_counts[(u64)tid * n_blocks];
        unsigned long long b1 = (tid + 1 < n_shards) ? blk_counts[(u64)(tid + 1) * n_blocks] : s_carry;
        counts[tid] = b1 - b0;
    }
}
__global__ __launch_bounds__(256) void k_bucket_fill(const u32* features, u64 n, u32 n_shards, u64 tile,
                                                     const unsigned long long* blk_off, u32* bucketed, u32* src_index) {
    __shared__ unsigned long long s_cur[MCQ_BUCKET_MAX_SHARDS];
    const u32 lane = threadIdx.x & 63;
    if (threadIdx.x < n_shards) s_cur[threadIdx.x] = blk_off[(u64)threadIdx.x * gridDim.x + blockIdx.x];
    __syncthreads();
    const u64 t0 = (u64)blockIdx.x * tile, t1 = t0 + tile < n ? t0 + tile : n;
    for (u64 i0 = t0 + (threadIdx.x & ~63u); i0 < t1; i0 += 256) {
        const u64 i = i0 + lane;
        const u32 f = i < t1 ? features[i] : MCQ_EMPTY;
        const u32 own = owner_of(f, n_shards);
        for (u32 o = 0; o < n_shards; ++o) {
            const u64 m = __ballot(own == o);
            const u32 c = (u32)__builtin_popcountll(m);
            unsigned long long start = 0;
            if (lane == 0 && c) start = atomicAdd(&s_cur[o], (unsigned long long)c);
            start = ((unsigned long long)__builtin_amdgcn_readfirstlane((u32)(start >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)start);
            if (own == o) {
                const u64 d = start + lane_rank(m);
                bucketed[d] = f; src_index[d] = (u32)i;
            }
        }
    }
}

// list i = src_locs[src_off[i] .. src_off[i+1]) goes to dst_locs[dst_off[dst_slot[i]] ..); one wave per 64 lists
template <class KeyT>
__global__ __launch_bounds__(256) void k_scatter_lists(u64 n_lists, const u64* src_off, const u32* dst_slot, const u64* dst_off,
                                                       const KeyT* src_locs, KeyT* dst_locs) {
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 ngroups = (n_lists + 63) / 64, nwaves = (u64)gridDim.x * 4;
    for (u64 g = (u64)blockIdx.x * 4 + wave; g < ngroups; g += nwaves) {
        const u64 i = g * 64 + lane;
        u64 so = 0, d = 0; u32 len = 0;
        if (i < n_lists) { so = src_off[i]; len = (u32)(src_off[i + 1] - so); d = dst_off[dst_slot[i]]; }
        const u64 sbase = src_off[g * 64];
        u32 incl = wave_incl_scan_dpp(len);
        u32 pos = incl - len;
        const u32 T = bcast(incl, 63);
        for (u32 base = 0; base < T; base += 64) {
            const u32 t = base + lane;
            const u32 tt = t < T ? t : T - 1;
            u32 lo = 0;
#pragma unroll
            for (u32 step = 32; step > 0; step >>= 1) {
                u32 c = lo + step;
                u32 pc = __shfl(pos, (int)(c & 63), 64);
                if (c < 64 && pc <= tt) lo = c;
            }
            u32 pj = __shfl(pos, (int)lo, 64);
            u32 dlo = __shfl((u32)d, (int)lo, 64), dhi = __shfl((u32)(d >> 32), (int)lo, 64);
            if (t < T) dst_locs[(((u64)dhi << 32) | dlo) + (tt - pj)] = src_locs[sbase + t];
        }
    }
}
__global__ void k_scatter_len(const u32* list_len, const u32* slot, u64 n, u32* slot_len) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slot_len[slot[i]] = list_len[i];
}
// per query: location segment start = offset of its first feature slot; length = sum of its mates
__global__ void k_query_offsets(u64 nq, u32 qstep, u32 s, const u64* win_off, const u64* seq_off, u32 ranges, const u64* dst_off, u64 n_slots,
                                u64* loc_off, u32* query_len) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) {
        loc_off[q] = dst_off[win_off[q * qstep] * s];
        u64 len = 0;
        for (u32 m = 0; m < qstep; ++m) { u64 bg, en; seq_bounds(seq_off, ranges, q * qstep + m, bg, en); len += en - bg; }
        query_len[q] = (u32)len;
    }
    if (q == 0) loc_off[nq] = dst_off[n_slots];
}

// ------------------------------------------------------------------ row f4: FASTQ text -> sequence ranges on the GPU
// FASTQ is four lines per record and the reference reads it exactly so (fastq_reader::read_next,
// src/sequence_io.cpp:251-285: getline header, getline data, getline '+', getline qualities), so
// the sequence of record r is line 4r+1.  Workgroup tiles of 4 KiB count their newlines, a scan
// gives every newline its line number, and the newline that ends line 4r (4r+1) writes the
// begin (end) of sequence r.  The text is not copied: mcq_query reads the bases in place
// (MCQ_BATCH_RANGES).  Like getline, a '\r' before the newline stays part of the line.
#define MCQ_FQ_TILE 4096
// 16-bit mask of the newlines among text[base .. base+16): one 16-B load and exact per-byte zero detection of
// (word ^ 0x0A0A0A0A) when the address is aligned and inside the buffer, a byte loop otherwise
__device__ __forceinline__ u32 fq_newline_mask(const char* __restrict__ text, u64 base, u64 n) {
    u32 mask = 0;
    if (base + 16 <= n && ((reinterpret_cast<uintptr_t>(text) + base) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + base);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 m = w[i] ^ 0x0A0A0A0Au;
            const u32 z = ~(((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m) & 0x80808080u;     // bit 7 of every zero byte
            mask |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * i);
        }
    } else {
        for (u32 j = 0; j < 16; ++j) mask |= (u32)(base + j < n && text[base + j] == '\n') << j;
    }
    return mask;
}
__global__ __launch_bounds__(256) void k_fq_count(const char* text, u64 n, u64* tile_cnt) {
    __shared__ u32 s_c;
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * MCQ_FQ_TILE + (u64)threadIdx.x * 16;
    u32 c = (u32)__builtin_popcount(fq_newline_mask(text, base, n));
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_c;
}
// L: lines per record (4: FASTQ; 2: FASTA with the sequence on one line); the sequence is the record's second line
__global__ __launch_bounds__(256) void k_fq_ranges(const char* text, u64 n, const u64* tile_off, u64 n_tiles, u64* ranges, u64 max_seqs,
                                                   u64* n_seqs_out, u32 L) {
    __shared__ u32 s_w[4];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = (u64)blockIdx.x * MCQ_FQ_TILE + (u64)tid * 16;
    u32 mask = fq_newline_mask(text, base, n);
    const u32 c = (u32)__builtin_popcount(mask);
    u32 incl = wave_incl_scan_dpp(c);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < wave; ++w) woff += s_w[w];
    u64 line = tile_off[blockIdx.x] + woff + incl - c;         // index of the line my first newline terminates
    while (mask) {
        const u32 j = (u32)__builtin_ctz(mask);
        mask &= mask - 1;
        const u64 p = base + j;
        const u64 rec = L == 4 ? line >> 2 : line >> 1;
        const u32 li = (u32)(line & (L - 1));
        if (rec < max_seqs) {
            if (li == 0) ranges[2 * rec] = p + 1;
            else if (li == 1) ranges[2 * rec + 1] = p;
        }
        ++line;
    }
    if (blockIdx.x == 0 && tid == 0) {
        const u64 total = tile_off[n_tiles];
        u64 ns = total >= 2 ? (total - 2) / L + 1 : 0;          // records whose sequence line is complete
        *n_seqs_out = ns < max_seqs ? ns : max_seqs;
    }
}

// ------------------------------------------------------------------ host helpers
static u64 pow2ceil64(u64 x) { u64 p = 1; while (p < x) p <<= 1; return p; }

// nt: threads per workgroup, 256 or 1024.  A 1024-thread workgroup needs 16 free wave slots on ONE CU at once: enqueued
// beside a grid of smaller workgroups that fills the GPU (the sharded path's second stream) it waits until that grid has
// drained; 256-thread workgroups slip in as the others retire.
template <class InT>
static int device_exclusive_scan(const InT* in, u64* out, u64 n, hipStream_t st, u32 nt = 1024) {
    const u64 ntiles = (n + MCQ_SCAN_TILE - 1) / MCQ_SCAN_TILE;
    if (ntiles <= 1) { hipLaunchKernelGGL(k_scan_u64<InT>, dim3(1), dim3(nt), 0, st, in, out, n); return MCQ_OK; }
    u64 *sums = nullptr, *offs = nullptr;
    HIPCHK(hipMallocAsync((void**)&sums, ntiles * 8, st));
    HIPCHK(hipMallocAsync((void**)&offs, (ntiles + 1) * 8, st));
    hipLaunchKernelGGL(k_scan_tiles<InT>, dim3((u32)ntiles), dim3(nt), 0, st, in, out, n, sums);
    hipLaunchKernelGGL(k_scan_u64<u64>, dim3(1), dim3(nt), 0, st, (const u64*)sums, offs, ntiles);
    hipLaunchKernelGGL(k_scan_add, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, out, n, (const u64*)offs);
    HIPCHK(hipFreeAsync(sums, st));
    HIPCHK(hipFreeAsync(offs, st));
    return MCQ_OK;
}

// Fold schedule of the reference's merge loop (src/querying.h:867-1073): senders = odd
// ranks, receivers = even ranks, i-th sender -> i-th receiver; used senders retire, every
// second used receiver becomes a sender; floor(log2 P) rounds.
static void fold_schedule(u32 P, std::vector<std::pair<u32, u32>>& sched, std::vector<u32>* level_end = nullptr) {
    std::vector<u32> snd, rcv;
    for (u32 i = 0; i < P; ++i) (i % 2 ? snd : rcv).push_back(i);
    for (u32 k = P; k > 1; k /= 2) {
        size_t np = std::min(snd.size(), rcv.size());
        std::vector<u32> us(snd.begin(), snd.begin() + np), ur(rcv.begin(), rcv.begin() + np);
        for (size_t i = 0; i < np; ++i) sched.emplace_back(us[i], ur[i]);
        if (level_end && np) level_end->push_back((u32)sched.size());
        std::vector<u32> ns(snd.begin() + np, snd.end()), nr;
        for (size_t i = 0; i < rcv.size(); ++i) {
            if (i < np && (i % 2) == 1) ns.push_back(rcv[i]); else nr.push_back(rcv[i]);
        }
        std::sort(ns.begin(), ns.end());
        snd.swap(ns); rcv.swap(nr);
    }
}

// flags of mcq_query_opts a caller may set (anything else is rejected: a stray bit must not change results silently)
#ifdef MCQ_PROFILE_HOOKS        // profiling builds only: bits 12..15 = stop the fused kernel after stage 1..5 (results invalid)
#define MCQ_OPT_FLAGS_KNOWN (MCQ_QUIRK_SEQ_DROP | MCQ_FORCE_BLOCK_PATH | MCQ_FORCE_RAW_SORT | MCQ_NO_WAVE16 | MCQ_NO_TWO_CLASS | MCQ_FOLD_BY_LISTS | 0xF000u)
#else
#define MCQ_OPT_FLAGS_KNOWN (MCQ_QUIRK_SEQ_DROP | MCQ_FORCE_BLOCK_PATH | MCQ_FORCE_RAW_SORT | MCQ_NO_WAVE16 | MCQ_NO_TWO_CLASS | MCQ_FOLD_BY_LISTS)
#endif
static int force_bits(u32 flags) {
    int f = ((flags & MCQ_FORCE_BLOCK_PATH) ? 1 : 0) | ((flags & MCQ_FORCE_RAW_SORT) ? 2 : 0) | ((flags & MCQ_NO_WAVE16) ? 4 : 0);
#ifdef MCQ_PROFILE_HOOKS
    f |= (int)((flags >> 12) & 0xFu) << 4;
#endif
    return f;
}

static int make_opt(const mcq_query_opts* o, OptDev& d, const mcq_db* db) {
    if (!o) return fail(MCQ_E_ARG, "opts is null");
    if (o->flags & ~(u32)MCQ_OPT_FLAGS_KNOWN) return fail(MCQ_E_ARG, "unknown bits in mcq_query_opts.flags");
    u32 P = o->emulate_ranks ? o->emulate_ranks : 1;
    if (P > 64) return fail(MCQ_E_UNSUPPORTED, "emulate_ranks > 64");
    u32 p2 = (u32)pow2ceil64(P);
    if (o->max_cand < 1 || o->max_cand > 16) return fail(MCQ_E_UNSUPPORTED, "max_cand must be in 1..16");
    memset(&d, 0, sizeof(d));
    // P lists of M entries side by side in the 64 lanes of a wave when they fit; else (the reference's -n 32 / -n 64
    // runs with -maxcand 4) in the LDS of the workgroup kernel, which then takes every query
    d.big = p2 * o->max_cand > 64 ? 1 : 0;
    d.max_cand = o->max_cand; d.P = P; d.seg = d.big ? o->max_cand : 64 / p2;
    d.quirk_seq_drop = (o->flags & MCQ_QUIRK_SEQ_DROP) ? 1 : 0;
    d.hooks = ((o->flags & MCQ_FORCE_RAW_SORT) ? 1u : 0u) | ((o->flags & MCQ_NO_WAVE16) ? 2u : 0u) |
              (((o->flags & MCQ_FORCE_BLOCK_PATH) || d.big) ? 4u : 0u) | ((o->flags & MCQ_NO_TWO_CLASS) ? 8u : 0u);
    d.insert_size_max = o->insert_size_max;
    std::vector<std::pair<u32, u32>> sched;
    std::vector<u32> level_end;
    fold_schedule(P, sched, &level_end);
    if (sched.size() > MCQ_MAX_FOLD || level_end.size() > 8) return fail(MCQ_E_UNSUPPORTED, "fold schedule too long");
    d.n_fold = (u32)sched.size();
    d.n_levels = (u32)level_end.size();
    for (size_t i = 0; i < level_end.size(); ++i) d.level_end[i] = (unsigned char)level_end[i];
    for (size_t i = 0; i < sched.size(); ++i) { d.fold_snd[i] = (unsigned char)sched[i].first; d.fold_rcv[i] = (unsigned char)sched[i].second; }
    // The P lists and the tree fold as ONE selection (mcq_device.hpp, topk_lin_write) whenever that is the same thing: always
    // but under MCQ_QUIRK_SEQ_DROP on a table that has sequence-level taxa -- a dropped entry has held a slot of an
    // intermediate list.  The schedule must concatenate the ranks in ascending order and keep exactly [0, 2^floor(log2 P)).
    if (P > 1 && !(o->flags & MCQ_FOLD_BY_LISTS) && !(d.quirk_seq_drop && (!db || db->seq_taxa))) {
        std::vector<std::vector<u32>> seq(P);
        for (u32 r = 0; r < P; ++r) seq[r].push_back(r);
        for (auto& e : sched) { auto& a = seq[e.second]; auto& b = seq[e.first]; a.insert(a.end(), b.begin(), b.end()); b.clear(); }
        u32 keep = 1; while (keep * 2 <= P) keep *= 2;
        bool ok = seq[0].size() == keep;
        for (u32 i = 0; ok && i < keep; ++i) ok = seq[0][i] == i;
        if (ok) { d.lin = 1; d.keep = keep; d.big = 0; d.seg = 64; d.hooks &= ~4u; if (o->flags & MCQ_FORCE_BLOCK_PATH) d.hooks |= 4u; d.n_fold = 0; d.n_levels = 0; }
    }
    return MCQ_OK;
}

extern "C" uint32_t mcq_owner(uint32_t feature, uint32_t n_shards) {
    u32 x = feature;
    x = ((x >> 16) ^ x) * 0x45d9f3bu; x = ((x >> 16) ^ x) * 0x45d9f3bu; x = (x >> 16) ^ x;
    return (u32)(((u64)x * (n_shards ? n_shards : 1)) >> 32);
}

// ------------------------------------------------------------------ db
// temporaries are released on every way out; the handle itself by mcq_db_destroy on failure
struct DevTemps {
    std::vector<void*> p;
    ~DevTemps() { for (void* x : p) (void)hipFree(x); }
    hipError_t alloc(void** out, u64 bytes) { hipError_t e = hipMalloc(out, bytes ? bytes : 1); if (e == hipSuccess) p.push_back(*out); return e; }
    void release(void* x) { for (auto& q : p) if (q == x) { (void)hipFree(x); q = nullptr; } }
};
static int check_params(const mcq_db_desc* desc) {
    if (desc->k < 1 || desc->k > 16) return fail(MCQ_E_UNSUPPORTED, "k must be 1..16");
    if (desc->sketch_size < 1 || desc->sketch_size > 32) return fail(MCQ_E_UNSUPPORTED, "sketch_size must be 1..32");
    if (desc->winlen < desc->k || desc->winlen > 128) return fail(MCQ_E_UNSUPPORTED, "winlen must be k..128");
    if (desc->winstride < 1) return fail(MCQ_E_ARG, "winstride must be >= 1");
    if (desc->shard_id >= (desc->n_shards ? desc->n_shards : 1)) return fail(MCQ_E_ARG, "shard_id >= n_shards");
    return MCQ_OK;
}
// windows per target -> gw_off (u32 [n_targets + 1]) and the block table; ext: device, u32 [n_targets]
static int make_gw_tables(const u32* d_ext, u32 nt, DevTemps& tmp, u32** gw_off, u32** gw_blk, u32* gw_shift, u64* n_windows) {
    const u32 TB = 256;
    u64* d_off64 = nullptr;
    HIPCHK(tmp.alloc((void**)&d_off64, ((u64)nt + 1) * 8));
    { int rcs = device_exclusive_scan<u32>(d_ext, d_off64, nt, 0); if (rcs) return rcs; }
    HIPCHK(hipMemcpy(n_windows, d_off64 + nt, 8, hipMemcpyDeviceToHost));
    *gw_off = nullptr; *gw_blk = nullptr;
    if (*n_windows >= 0xFFFFFFFFull) return MCQ_OK;       // does not fit 32 bits: the caller decides
    // block table: at most 2^18 entries (2 MB: stays in L2), at least 64 windows per block
    u32 sh = 6; while ((*n_windows >> sh) > (1ull << 18)) ++sh;
    const u64 n_blk = (*n_windows >> sh) + 2;
    HIPCHK(hipMalloc(gw_off, ((u64)nt + 1) * 4));
    hipLaunchKernelGGL(k_u64_to_u32, dim3((u32)((nt + 1 + TB - 1) / TB)), dim3(TB), 0, 0, (const u64*)d_off64, *gw_off, (u64)nt + 1);
    if (hipMalloc(gw_blk, n_blk * 8) != hipSuccess) { (void)hipFree(*gw_off); *gw_off = nullptr; return fail(MCQ_E_HIP, "hipMalloc of the window block table failed"); }
    hipLaunchKernelGGL(k_gw_blocks, dim3((u32)((n_blk + TB - 1) / TB)), dim3(TB), 0, 0, (const u32*)*gw_off, nt, sh, n_blk, reinterpret_cast<uint2*>(*gw_blk));
    *gw_shift = sh;
    return MCQ_OK;
}

// the table itself, from one or several parts (device memory); format decided by the caller
static int create_table(const mcq_db_desc* desc, const std::vector<PartView>& parts, u32 compact, u32 wb, u32 gw,
                        u32* d_gwoff, u32* d_gwblk, u32 gw_shift, u64 n_windows, mcq_db** out) {
    const u32 TB = 256;
    const bool dev = (desc->flags & MCQ_DEVICE_PTRS) != 0;
    const u32 n_shards = desc->n_shards ? desc->n_shards : 1;
    const u64 locsz = compact ? 4 : 8;
    mcq_db* db = new mcq_db();
    memset(db, 0, sizeof(*db));
    db->device = desc->device; db->n_shards = n_shards; db->shard_id = desc->shard_id;
    db->gw_off = d_gwoff; db->gw_blk = d_gwblk;           // (released by mcq_db_destroy from here on)
    DevTemps tmp;
#define DBCHK(expr) HIPCHK_OR(expr, (void)mcq_db_destroy(db))
#define DBRC(expr) do { int rc_ = (expr); if (rc_) { (void)mcq_db_destroy(db); return rc_; } } while (0)

    // ---- pass 1 over the parts: owned non-empty keys, owned locations, locations of lists too long for a 64-B bucket
    u64 nk_max = 0;
    for (const auto& pv : parts) nk_max = std::max(nk_max, pv.n_keys);
    u64 *d_len = nullptr, *d_ext = nullptr, *d_new = nullptr; unsigned long long* d_tot = nullptr;
    DBCHK(tmp.alloc((void**)&d_len, std::max<u64>(1, nk_max) * 8));
    DBCHK(tmp.alloc((void**)&d_tot, 24));
    DBCHK(hipMemset(d_tot, 0, 24));
    const u32 inl64 = bucket_inline_max(2u, compact);
    for (const auto& pv : parts) {
        if (!pv.n_keys) continue;
        hipLaunchKernelGGL(k_owned_len, dim3((u32)((pv.n_keys + TB - 1) / TB)), dim3(TB), 0, 0, pv, n_shards, desc->shard_id, d_len);
        hipLaunchKernelGGL(k_owned_totals, dim3(1024), dim3(TB), 0, 0, (const u64*)d_len, pv.n_keys, inl64, d_tot);
    }
    unsigned long long tot[3] = {0, 0, 0};
    DBCHK(hipMemcpy(tot, d_tot, 24, hipMemcpyDeviceToHost));
    const u64 nk_local = tot[0], nl_local = tot[1];

    // ---- layout, per table.  64-B buckets hold a list of up to 14 (7) locations next to its key -- in the sector the
    // probe has just brought in -- and pay with 64 B per slot; worth it while most lists are that short (2 Gbp: mean
    // 2.9 locations per key).  On larger tables (>= 10 Gbp: mean >= 4.9) most lists sit behind the array anyway, and
    // the r01 layout -- 16-B slots, every list behind them -- is 27-30 GB smaller and 1-3 % faster (profiles/r02_db_scale.txt).
    u32 bucket_bytes = (nk_local == 0 || (double)nl_local / (double)nk_local <= 4.0) ? 64u : 16u;
    if (desc->flags & MCQ_DB_SLOTS_16) bucket_bytes = 16;
    if (desc->flags & MCQ_DB_BUCKETS_64) bucket_bytes = 64;
    if (const char* e = getenv("MCQ_BUCKET_BYTES")) { const int v = atoi(e); if (v == 16 || v == 64) bucket_bytes = (u32)v; }   // tuning knob
    const u32 bsh = bucket_bytes == 64 ? 2u : 0u;
    const u32 inl = bucket_inline_max(bsh, compact);
    const u64 nl_ext = bucket_bytes == 64 ? tot[2] : nl_local;

    // load factor <= 0.25 (43 % of a read's features are not in the table, and every step of a linear probe is a new
    // sector) while the slot array stays below 48 GB and a third of the free memory, else <= 0.5 -- also when the
    // allocation at 0.25 fails.  MCQ_SLOTS_PER_KEY overrides.
    size_t mem_free = 0, mem_total = 0;
    DBCHK(hipMemGetInfo(&mem_free, &mem_total));
    const u64 ext_bytes = std::max<u64>(1, nl_ext) * locsz;
    u64 slots_per_key = 4;
    { const u64 b4 = pow2ceil64(nk_local * 4) * bucket_bytes; if (b4 > (48ull << 30) || b4 + ext_bytes > mem_free / 3) slots_per_key = 2; }
    bool spk_forced = false;
    if (const char* e = getenv("MCQ_SLOTS_PER_KEY")) { slots_per_key = std::max<u64>(1, strtoull(e, nullptr, 10)); spk_forced = true; }   // tuning knob
    u64 nslots = 0, table_bytes = 0;
    for (;;) {
        nslots = std::max<u64>(1024, pow2ceil64(nk_local * slots_per_key));
        if (nslots > (1ull << 32)) { (void)mcq_db_destroy(db); return fail(MCQ_E_UNSUPPORTED, "table too large"); }
        table_bytes = nslots * bucket_bytes + ext_bytes;
        const hipError_t e = hipMalloc(&db->slots, table_bytes);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        db->slots = nullptr;
        if (slots_per_key > 2 && !spk_forced) { slots_per_key = 2; continue; }
        (void)mcq_db_destroy(db);
        return fail(MCQ_E_HIP, std::string("hipMalloc of the table (") + std::to_string(table_bytes >> 20) + " MiB): " + hipGetErrorString(e));
    }
    db->n_keys_local = nk_local; db->n_locs_local = nl_local; db->nslots = nslots;
    db->bucket_bytes = bucket_bytes; db->slots_per_key = (u32)slots_per_key; db->n_ext = nl_ext; db->n_windows = n_windows;
    DBCHK(hipMalloc(&db->tgt2tax, std::max<u32>(1, desc->n_targets) * 4));
    if (desc->n_targets)
        DBCHK(hipMemcpy(db->tgt2tax, desc->tgt2tax, (u64)desc->n_targets * 4, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    {
        std::vector<u32> t2t(desc->n_targets);
        if (desc->n_targets) DBCHK(hipMemcpy(t2t.data(), db->tgt2tax, (u64)desc->n_targets * 4, hipMemcpyDeviceToHost));
        db->seq_taxa = false;
        for (u32 x : t2t) if (x != MCQ_EMPTY && (x & 0x80000000u)) { db->seq_taxa = true; break; }
    }
    const u32 bq = bucket_bytes / 16;
    const u64 n_uint4 = nslots * bq;
    hipLaunchKernelGGL(k_fill_slots, dim3((u32)std::min<u64>((n_uint4 + TB - 1) / TB, 1u << 20)), dim3(TB), 0, 0, db->slots, n_uint4);

    // ---- pass 2: insert part by part; the long lists of part p start where those of part p - 1 end
    DBCHK(tmp.alloc((void**)&d_ext, std::max<u64>(1, nk_max) * 8));
    DBCHK(tmp.alloc((void**)&d_new, (nk_max + 1) * 8));
    char* ext = reinterpret_cast<char*>(db->slots) + nslots * bucket_bytes;
    u64 ext_base = 0;
    for (const auto& pv : parts) {
        if (!pv.n_keys) continue;
        const dim3 ig((u32)((pv.n_keys + TB - 1) / TB));
        hipLaunchKernelGGL(k_owned_len, ig, dim3(TB), 0, 0, pv, n_shards, desc->shard_id, d_len);
        hipLaunchKernelGGL(k_ext_len, ig, dim3(TB), 0, 0, (const u64*)d_len, pv.n_keys, inl, d_ext);
        DBRC(device_exclusive_scan<u64>(d_ext, d_new, pv.n_keys, 0));
        u64 part_ext = 0;
        DBCHK(hipMemcpy(&part_ext, d_new + pv.n_keys, 8, hipMemcpyDeviceToHost));
        if (ext_base + part_ext > std::max<u64>(1, nl_ext)) { (void)mcq_db_destroy(db); return fail(MCQ_E_ARG, "the parts changed between the two passes"); }
        if (compact) hipLaunchKernelGGL(k_insert_keys<u32>, ig, dim3(TB), 0, 0, db->slots, (u32)(nslots - 1), bq, inl, pv, (const u64*)d_len, (const u64*)d_new, ext_base, wb, (const u32*)d_gwoff);
        else         hipLaunchKernelGGL(k_insert_keys<u64>, ig, dim3(TB), 0, 0, db->slots, (u32)(nslots - 1), bq, inl, pv, (const u64*)d_len, (const u64*)d_new, ext_base, 32u, (const u32*)nullptr);
        DBCHK(hipGetLastError());
        const dim3 cg((u32)std::min<u64>((pv.n_keys * 64 + TB - 1) / TB, 1u << 20));
        if (compact) hipLaunchKernelGGL(k_copy_lists<u32>, cg, dim3(TB), 0, 0, pv, (const u64*)d_new, reinterpret_cast<u32*>(ext) + ext_base, wb, (const u32*)d_gwoff);
        else         hipLaunchKernelGGL(k_copy_lists<u64>, cg, dim3(TB), 0, 0, pv, (const u64*)d_new, reinterpret_cast<u64*>(ext) + ext_base, 32u, (const u32*)nullptr);
        DBCHK(hipGetLastError());
        ext_base += part_ext;
    }
    DBCHK(hipDeviceSynchronize());
#undef DBCHK
#undef DBRC

    db->d.slots = db->slots; db->d.slot_mask = (u32)(db->nslots - 1); db->d.locs = db->slots;
    db->d.bsh = bsh;
    db->d.wb = wb; db->d.compact = compact;
    db->g.on = gw; db->g.shift = gw_shift; db->g.off = d_gwoff; db->g.blk = d_gwblk;
    db->d.tgt2tax = db->tgt2tax; db->d.n_targets = desc->n_targets;
    db->d.k = desc->k; db->d.s = desc->sketch_size; db->d.winlen = desc->winlen; db->d.winstride = desc->winstride;
    db->d.tgt_winstride = desc->tgt_winstride ? desc->tgt_winstride : desc->winstride;
    db->d.magic_stride = (u32)std::min<u64>((1ull << 32) / db->d.winstride, 0xFFFFFFFFull);
    db->d.magic_tgt_stride = (u32)std::min<u64>((1ull << 32) / db->d.tgt_winstride, 0xFFFFFFFFull);
    db->bytes = table_bytes + (u64)desc->n_targets * 4 + (gw ? ((u64)desc->n_targets + 1) * 4 + ((n_windows >> gw_shift) + 2) * 4 : 0);
    {   // FNV-1a over everything a peer must agree on to read this handle's location words
        u64 h = 1469598103934665603ull;
        auto mix = [&h](u64 v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xFF; h *= 1099511628211ull; } };
        mix(compact); mix(gw); mix(wb); mix(gw ? n_windows : 0); mix(desc->n_targets); mix(db->d.k); mix(db->d.s); mix(db->d.winlen); mix(db->d.winstride);
        mix(db->d.tgt_winstride); mix(n_shards);
        if (gw && desc->n_targets) {
            std::vector<u32> go((u64)desc->n_targets + 1);
            if (hipMemcpy(go.data(), d_gwoff, go.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)mcq_db_destroy(db); return fail(MCQ_E_HIP, "reading back the window offsets failed"); }
            for (u32 v : go) { h ^= v; h *= 1099511628211ull; }
        }
        db->fmt_sig = h;
    }
    *out = db;
    return MCQ_OK;
}

extern "C" int mcq_db_create(const mcq_db_desc* desc, mcq_db** out) {
    if (!desc || !out) return fail(MCQ_E_ARG, "null argument");
    { int rc = check_params(desc); if (rc) return rc; }
    HIPCHK(hipSetDevice(desc->device));

    const bool dev = (desc->flags & MCQ_DEVICE_PTRS) != 0;
    const u64 nk = desc->n_keys, nl = desc->n_locs;
    const u32* d_keys = desc->keys; const u64* d_off = desc->list_off; const u64* d_locs = desc->locs;
    DevTemps tmp;
    u32* t_keys = nullptr; u64* t_off = nullptr; u64* t_locs = nullptr;
    if (!dev) {
        HIPCHK(tmp.alloc((void**)&t_keys, std::max<u64>(1, nk) * 4));
        HIPCHK(tmp.alloc((void**)&t_off, (nk + 1) * 8));
        HIPCHK(tmp.alloc((void**)&t_locs, std::max<u64>(1, nl) * 8));
        if (nk) HIPCHK(hipMemcpy(t_keys, desc->keys, nk * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(t_off, desc->list_off, (nk + 1) * 8, hipMemcpyHostToDevice));
        if (nl) HIPCHK(hipMemcpy(t_locs, desc->locs, nl * 8, hipMemcpyHostToDevice));
        d_keys = t_keys; d_off = t_off; d_locs = t_locs;
    }

    // ---- location format: 32-bit bit fields (tgt << wb) | win when target and window ids fit; else the 32-bit global
    // window index (any table of fewer than 2^32 - 1 windows); else 64-bit words
    u32 wb = 32, compact = 0, gw = 0;
    u32 *d_gwoff = nullptr, *d_gwblk = nullptr; u32 gw_shift = 0; u64 n_windows = 0;
    if ((desc->flags & MCQ_DB_LOCS_64) && (desc->flags & MCQ_DB_LOCS_GW)) return fail(MCQ_E_ARG, "MCQ_DB_LOCS_64 and MCQ_DB_LOCS_GW exclude each other");
    if (!(desc->flags & MCQ_DB_LOCS_64)) {
        u32* d_mw = nullptr; u32 maxwin = 0;
        HIPCHK(tmp.alloc((void**)&d_mw, 4));
        HIPCHK(hipMemset(d_mw, 0, 4));
        if (nl) hipLaunchKernelGGL(k_max_win, dim3(1024), dim3(256), 0, 0, d_locs, nl, d_mw);
        HIPCHK(hipMemcpy(&maxwin, d_mw, 4, hipMemcpyDeviceToHost));
        u32 winbits = 1; while (winbits < 32 && (maxwin >> winbits)) ++winbits;
        u32 maxtgt = desc->n_targets ? desc->n_targets - 1 : 0;
        u32 tgtbits = 1; while (tgtbits < 32 && (maxtgt >> tgtbits)) ++tgtbits;
        if (desc->loc_win_bits > winbits) winbits = desc->loc_win_bits;
        if (!(desc->flags & MCQ_DB_LOCS_GW) && winbits + tgtbits <= 32 && winbits <= 31 &&
            ((((u64)maxtgt << winbits) | maxwin) < 0xFFFFFFFFull)) { compact = 1; wb = winbits; }
        else if (desc->n_targets) {
            // global-window form: windows per target (given, or 1 + the largest window id among the locations), offsets
            const u32 nt = desc->n_targets;
            u32* d_ext = nullptr;
            HIPCHK(tmp.alloc((void**)&d_ext, (u64)nt * 4));
            if (desc->tgt_windows) HIPCHK(hipMemcpy(d_ext, desc->tgt_windows, (u64)nt * 4, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
            else {
                HIPCHK(hipMemset(d_ext, 0, (u64)nt * 4));
                if (nl) hipLaunchKernelGGL(k_tgt_extent, dim3(2048), dim3(256), 0, 0, d_locs, nl, nt, d_ext);
            }
            int rc = make_gw_tables(d_ext, nt, tmp, &d_gwoff, &d_gwblk, &gw_shift, &n_windows); if (rc) return rc;
            if (d_gwoff) { compact = 1; gw = 1; wb = 0; }
            else if (desc->flags & MCQ_DB_LOCS_GW) return fail(MCQ_E_UNSUPPORTED, "MCQ_DB_LOCS_GW: the table has 2^32 - 1 windows or more");
        }
    }
    PartView pv; pv.n_keys = nk; pv.n_locs = nl; pv.keys = d_keys; pv.off = d_off; pv.locs = d_locs; pv.gw_words = 0;
    return create_table(desc, std::vector<PartView>{pv}, compact, wb, gw, d_gwoff, d_gwblk, gw_shift, n_windows, out);
}

// The same for a table that is larger than the memory for its one-piece description (RefSeq scale: the 64-bit locations
// alone would be 8 B x 1.7e10): handed over in parts -- e.g. one per feature-hash range, as mcq_build_parts makes them --
// whose locations are 32-bit global-window words already.  Device memory only.
extern "C" int mcq_db_create_parts(const mcq_db_desc* desc, const mcq_db_part* parts, uint32_t n_parts, mcq_db** out) {
    if (!desc || !out || (n_parts && !parts)) return fail(MCQ_E_ARG, "null argument");
    { int rc = check_params(desc); if (rc) return rc; }
    if (!(desc->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "mcq_db_create_parts takes device pointers");
    if (desc->flags & MCQ_DB_LOCS_64) return fail(MCQ_E_UNSUPPORTED, "parts hold global-window words: the handle keeps that form");
    if (!desc->tgt_windows || !desc->n_targets) return fail(MCQ_E_ARG, "mcq_db_create_parts needs tgt_windows (the words of the parts are defined by it)");
    HIPCHK(hipSetDevice(desc->device));
    DevTemps tmp;
    u32 *d_gwoff = nullptr, *d_gwblk = nullptr; u32 gw_shift = 0; u64 n_windows = 0;
    int rc = make_gw_tables(desc->tgt_windows, desc->n_targets, tmp, &d_gwoff, &d_gwblk, &gw_shift, &n_windows); if (rc) return rc;
    if (!d_gwoff) return fail(MCQ_E_UNSUPPORTED, "the table has 2^32 - 1 windows or more");
    std::vector<PartView> pvs;
    std::vector<u64*> offs;
    for (u32 i = 0; i < n_parts; ++i) {
        const mcq_db_part& p = parts[i];
        if (p.n_keys && (!p.keys || !p.list_len || (p.n_locs && !p.locs))) { (void)hipFree(d_gwoff); (void)hipFree(d_gwblk); return fail(MCQ_E_ARG, "null pointer in a part"); }
        u64* off = nullptr;
        if (tmp.alloc((void**)&off, (p.n_keys + 1) * 8) != hipSuccess) { (void)hipFree(d_gwoff); (void)hipFree(d_gwblk); return fail(MCQ_E_HIP, "hipMalloc of a part's list offsets failed"); }
        rc = device_exclusive_scan<u32>(p.list_len, off, p.n_keys, 0);
        if (rc) { (void)hipFree(d_gwoff); (void)hipFree(d_gwblk); return rc; }
        PartView pv; pv.n_keys = p.n_keys; pv.n_locs = p.n_locs; pv.keys = p.keys; pv.off = off; pv.locs = p.locs; pv.gw_words = 1;
        pvs.push_back(pv);
    }
    return create_table(desc, pvs, 1, 0, 1, d_gwoff, d_gwblk, gw_shift, n_windows, out);
}

extern "C" int mcq_db_destroy(mcq_db* db) {
    if (!db) return MCQ_OK;
    (void)hipSetDevice(db->device);
    (void)hipFree(db->slots); (void)hipFree(db->tgt2tax); (void)hipFree(db->gw_off); (void)hipFree(db->gw_blk);
    delete db;
    return MCQ_OK;
}

extern "C" uint64_t mcq_db_bytes(const mcq_db* db) { return db ? db->bytes : 0; }

#ifndef MCQ_BLOCK_NT            // tuning knobs: shape of the 32-bit workgroup kernel (threads, keys of LDS, workgroups)
#define MCQ_BLOCK_NT 1024
#define MCQ_BLOCK_LCAP 8192
#define MCQ_BLOCK_WGS_DEFAULT 512
#endif
// ------------------------------------------------------------------ workspace
extern "C" int mcq_ws_create(const mcq_db* db, uint64_t max_queries, uint64_t max_bases,
                             uint64_t max_locs_per_query, mcq_ws** out) {
    if (!db || !out) return fail(MCQ_E_ARG, "null argument");
    if (max_queries >= (1ull << 31)) return fail(MCQ_E_UNSUPPORTED, "max_queries must be < 2^31 per batch");
    u64 lmax = max_locs_per_query ? pow2ceil64(max_locs_per_query) : (1ull << 18);
    if (lmax > (1ull << 30)) return fail(MCQ_E_UNSUPPORTED, "max_locs_per_query too large");
    HIPCHK(hipSetDevice(db->device));
    mcq_ws* ws = new mcq_ws();
    memset(ws, 0, sizeof(*ws));
    ws->device = db->device; ws->max_queries = max_queries; ws->max_bases = max_bases;
    ws->sc.lmax = (u32)lmax;
    ws->sc.fmax = 1u << 15;
    ws->n_block_wgs = MCQ_BLOCK_WGS_DEFAULT;   // two resident workgroups per CU (32-bit keys: 64 KB of LDS, 64 VGPRs)
    if (const char* e = getenv("MCQ_BLOCK_WGS")) ws->n_block_wgs = std::max(1, atoi(e));      // tuning knob
    ws->ev_used = new std::vector<TimedLaunch>();
    ws->ev_free = new std::vector<TimedLaunch>();
    ws->cap_wave = db->g.on ? resident_blocks(k_query_wave<u32, 512, false, false, true, 2>, 256, db->device)
                 : db->d.compact ? resident_blocks(k_query_wave<u32, 512, false, false, false, 2>, 256, db->device)
                                 : resident_blocks(k_query_wave<u64, 512, false, false, false, 2>, 256, db->device);
    ws->cap_wave16 = db->g.on ? resident_blocks(k_query_wave16<false, false, true, 2>, 256, db->device)
                              : resident_blocks(k_query_wave16<false, false, false, 2>, 256, db->device);
    ws->cap_reduce16 = db->g.on ? resident_blocks(k_reduce_wave16<true>, 256, db->device) : resident_blocks(k_reduce_wave16<false>, 256, db->device);
    ws->cap_wave_many = db->g.on ? resident_blocks(k_query_wave<u32, 512, false, false, true, 2, 4>, 256, db->device)
                               : resident_blocks(k_query_wave<u32, 512, false, false, false, 2, 4>, 256, db->device);
    ws->cap_wave32 = db->g.on ? resident_blocks(k_query_wave32<false, true, 2>, 256, db->device) : resident_blocks(k_query_wave32<false, false, 2>, 256, db->device);
    const u64 nb = (u64)ws->n_block_wgs;
#define WSCHK(expr) HIPCHK_OR(expr, (void)mcq_ws_destroy(ws))
    WSCHK(hipMalloc(&ws->ctr, sizeof(CountersDev)));
    WSCHK(hipHostMalloc(&ws->ctr_host, sizeof(CountersDev)));
    WSCHK(hipMalloc(&ws->ovf_list, ovf_capacity(max_queries) * 4));
    // back-queue rows (first-stage pushes: below max_queries + MCQ_OVF_TAIL), then front-queue rows (first-stage front pushes AND the
    // second stage's hand-ons, each set of waves with its reservation tails: below max_queries + 2 x MCQ_OVF_TAIL)
    const u64 probe_rows = 2 * max_queries + 3 * (u64)MCQ_OVF_TAIL;
    WSCHK(hipMalloc(&ws->probe_buf, probe_rows * 64 * 8));
    WSCHK(hipMemset(ws->ctr, 0, sizeof(CountersDev)));
    WSCHK(hipMemcpy(&ws->ctr->probe_buf, &ws->probe_buf, sizeof(ws->probe_buf), hipMemcpyHostToDevice));
    WSCHK(hipMemset(ws->probe_buf, 0, probe_rows * 64 * 8));        // (a row never written reads as 64 empty lists)
    {
        unsigned long long* front = ws->probe_buf + (max_queries + (u64)MCQ_OVF_TAIL) * 64;
        WSCHK(hipMemcpy(&ws->ctr->probe_front, &front, sizeof(front), hipMemcpyHostToDevice));
    }
    WSCHK(hipMalloc(&ws->sc.feat, nb * ws->sc.fmax * 4));
    WSCHK(hipMalloc(&ws->sc.fpos, nb * ((u64)ws->sc.fmax + 1) * 4));
    WSCHK(hipMalloc(&ws->sc.foff, nb * ws->sc.fmax * 8));
    WSCHK(hipMalloc(&ws->sc.gbuf, nb * lmax * 8));
    WSCHK(hipMalloc(&ws->sc.ghits, nb * lmax * 8));
#undef WSCHK
    *out = ws;
    return MCQ_OK;
}

extern "C" int mcq_ws_destroy(mcq_ws* ws) {
    if (!ws) return MCQ_OK;
    (void)hipSetDevice(ws->device);
    (void)hipFree(ws->ctr); (void)hipHostFree(ws->ctr_host); (void)hipFree(ws->ovf_list); (void)hipFree(ws->probe_buf);
    (void)hipFree(ws->sc.feat); (void)hipFree(ws->sc.fpos); (void)hipFree(ws->sc.foff); (void)hipFree(ws->sc.gbuf); (void)hipFree(ws->sc.ghits);
    if (ws->d_bases) (void)hipFree(ws->d_bases);
    if (ws->d_seq_off) (void)hipFree(ws->d_seq_off);
    if (ws->d_cands) (void)hipFree(ws->d_cands);
    if (ws->d_ncand) (void)hipFree(ws->d_ncand);
    if (ws->pipe.ready) {
        for (int k = 0; k < 2; ++k) {
            (void)hipFree(ws->pipe.d_bases[k]); (void)hipFree(ws->pipe.d_seq_off[k]); (void)hipFree(ws->pipe.d_cands[k]); (void)hipFree(ws->pipe.d_ncand[k]);
            (void)hipEventDestroy(ws->pipe.ev_in[k]); (void)hipEventDestroy(ws->pipe.ev_k[k]); (void)hipEventDestroy(ws->pipe.ev_out[k]);
        }
        (void)hipStreamDestroy(ws->pipe.s_in); (void)hipStreamDestroy(ws->pipe.s_k); (void)hipStreamDestroy(ws->pipe.s_out);
    }
    for (auto* v : {ws->ev_used, ws->ev_free}) {
        if (!v) continue;
        for (auto& t : *v) for (auto e : t.ev) (void)hipEventDestroy(e);
        delete v;
    }
    delete ws;
    return MCQ_OK;
}

static int ensure_staging(mcq_ws* ws) {
    if (ws->d_bases) return MCQ_OK;
    HIPCHK(hipMalloc(&ws->d_bases, std::max<u64>(1, ws->max_bases) + 16));
    HIPCHK(hipMalloc(&ws->d_seq_off, (2 * ws->max_queries + 2) * 8));
    HIPCHK(hipMalloc(&ws->d_cands, std::max<u64>(1, ws->max_queries) * 16 * 16));
    HIPCHK(hipMalloc(&ws->d_ncand, std::max<u64>(1, ws->max_queries) * 4));
    return MCQ_OK;
}

// layout of an MCQ_BATCH_PACKED buffer for n bases (u32 words): [ceil(n/16) words of 2-bit codes][1 zero pad word]
// [ceil(n/32) words of ambiguity bits][1 zero pad word]
static u64 packed_words2(u64 n) { return (n + 15) / 16; }
static u64 packed_wordsA(u64 n) { return (n + 31) / 32; }
extern "C" uint64_t mcq_packed_bytes(uint64_t n_bases) { return (packed_words2(n_bases) + 1 + packed_wordsA(n_bases) + 1) * 4; }

// device-side view of a batch whose buffers are (already) in device memory
static int batch_dev(const mcq_batch* in, const char* d_bases, const u64* d_seq_off, BatchDev& b) {
    memset(&b, 0, sizeof(b));
    b.bases = d_bases; b.seq_off = d_seq_off; b.n_seq = in->n_seqs; b.nq = in->paired ? in->n_seqs / 2 : in->n_seqs;
    b.paired = in->paired ? 1 : 0;
    b.ranges = (in->flags & MCQ_BATCH_RANGES) ? 1 : 0;
    if (in->flags & MCQ_BATCH_PACKED) {
        if (b.ranges) return fail(MCQ_E_ARG, "MCQ_BATCH_PACKED and MCQ_BATCH_RANGES exclude each other");
        if (in->n_bases >= (1ull << 35)) return fail(MCQ_E_UNSUPPORTED, "packed batches hold fewer than 2^35 bases");
        b.packed = 1;
        b.last_word = (u32)packed_words2(in->n_bases);
        b.amb_off = b.last_word + 1;
        b.amb_last = (u32)packed_wordsA(in->n_bases);
    }
    return MCQ_OK;
}

// ASCII bases -> MCQ_BATCH_PACKED words; one thread per 32 bases
__global__ void k_pack_bases(const char* __restrict__ src, u64 n, u32* __restrict__ dst, u64 n2, u64 amb_off, u64 nA) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > nA) return;
    u32 w0 = 0, w1 = 0, am = 0;
    for (u32 j = 0; j < 32; ++j) {
        const u64 i = g * 32 + j;
        u32 code = 0, amb = 0;                                  // behind the end: code 0, not ambiguous (never looked at)
        if (i < n) {
            const u32 u = (u32)(unsigned char)src[i] & 0xDFu;
            code = (u >> 1) & 3u; code ^= code >> 1;
            amb = !(u == 'A' || u == 'C' || u == 'G' || u == 'T');
            if (amb) code = 0;
        }
        if (j < 16) w0 |= code << (30 - 2 * j); else w1 |= code << (30 - 2 * (j - 16));
        am |= amb << (31 - j);
    }
    if (2 * g <= n2) dst[2 * g] = (2 * g < n2) ? w0 : 0u;      // index n2 is the zero pad word
    if (2 * g + 1 <= n2) dst[2 * g + 1] = (2 * g + 1 < n2) ? w1 : 0u;
    dst[amb_off + g] = g < nA ? am : 0u;
}

extern "C" int mcq_pack_bases(const char* bases, uint64_t n_bases, void* out, uint32_t flags, void* stream) {
    if (!out || (n_bases && !bases)) return fail(MCQ_E_ARG, "null argument");
    const u64 n2 = packed_words2(n_bases), nA = packed_wordsA(n_bases), amb_off = n2 + 1;
    if (flags & MCQ_DEVICE_PTRS) {
        hipLaunchKernelGGL(k_pack_bases, dim3((u32)((nA + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bases, n_bases, (u32*)out, n2, amb_off, nA);
        HIPCHK(hipGetLastError());
        return MCQ_OK;
    }
    u32* dst = (u32*)out;
    memset(dst, 0, mcq_packed_bytes(n_bases));
    for (u64 i = 0; i < n_bases; ++i) {
        const u32 u = (u32)(unsigned char)bases[i] & 0xDFu;
        u32 code = (u >> 1) & 3u; code ^= code >> 1;
        const bool amb = !(u == 'A' || u == 'C' || u == 'G' || u == 'T');
        if (amb) dst[amb_off + (i >> 5)] |= 1u << (31 - (i & 31));
        else dst[i >> 4] |= code << (30 - 2 * (i & 15));
    }
    return MCQ_OK;
}

static const int kLcapWave = 512;
static const int kLcapBlock = 8192;

// events between the kernels of one batch (timing enabled only): begin(), then mark() after each kernel
struct LaunchTimer {
    mcq_ws* ws; hipStream_t st; TimedLaunch t; int n; bool on;
    LaunchTimer(mcq_ws* w, hipStream_t s) : ws(w), st(s), n(0), on(w->timing != 0) {}
    int begin() {
        if (!on) return MCQ_OK;
        if (!ws->ev_free->empty()) { t = ws->ev_free->back(); ws->ev_free->pop_back(); }
        else for (auto& e : t.ev) HIPCHK(hipEventCreate(&e));
        return mark();
    }
    int mark() { if (on && n <= MCQ_N_TIMED) HIPCHK(hipEventRecord(t.ev[n++], st)); return MCQ_OK; }
    int end() {                                 // kernels that were not launched take no time: repeat the last event
        if (!on) return MCQ_OK;
        while (n <= MCQ_N_TIMED) { int rc = mark(); if (rc) return rc; }
        ws->ev_used->push_back(t);
        return MCQ_OK;
    }
};

// how the next batch on this workspace enters (CountersDev::direct_mode): after the last kernel of a batch.  In the direct mode every
// query counts as queued; those the second stage found short (n_short) would not have been.  Enter at 3/4, leave at 1/2.
__global__ void k_next_mode(CountersDev* ctr, u64 nq) {
    const unsigned long long would = (unsigned long long)ctr->n_ovf - ctr->n_short, geom = ctr->n_geom;
    const u32 was = ctr->direct_mode;
    const u32 b0 = (was & 1u) ? (would * 2 > nq ? 1u : 0u) : (would * 4 > nq * 3 ? 1u : 0u);
    const u32 b1 = (was & 2u) ? (geom * 2 > nq ? 2u : 0u) : (geom * 4 > nq * 3 ? 2u : 0u);
    ctr->direct_mode = b0 | b1;
}

// sh != nullptr: the feature-sharded home side (SH instantiations; dbd = the handle's DbDev with `locs` pointing at the
// received location buffer); the counters are then zeroed by the caller (the sketch kernel has already counted)
static int launch_query(const mcq_db* db, mcq_ws* ws, const BatchDev& b, const OptDev& od_in, const OutDev& o,
                        hipStream_t st, int force_block_in, const DebugDev& dbg, const ShardDev* shp = nullptr, const DbDev* dbd = nullptr) {
    if (!shp) HIPCHK(hipMemsetAsync(ws->ctr, 0, MCQ_CTR_ZEROED, st));
    if (b.nq == 0) return MCQ_OK;
    ShardDev sh; memset(&sh, 0, sizeof(sh));
    if (shp) sh = *shp;
    const DbDev& D = dbd ? *dbd : db->d;
    const u64 want = (b.nq + 3) / 4;
    // (sharded home side too: two to four times the resident workgroups, so that they retire all along beside the other
    // streams' kernels, measured 1-7 % slower per batch than the resident grid)
    const u32 grid = grid_for(ws->cap_wave, want);
    LaunchTimer tm(ws, st);
    int rc = tm.begin(); if (rc) return rc;
    const bool tap = dbg.mode != 0;     // mcq_debug_matches: the instantiations that also write the sorted match lists
    // The workgroup kernel exists twice: plain, and with the two-class tail for queries with narrow window ranges (short reads
    // whose lists outgrew the wave stages; see k_query_block).  The second one is launched when its lists can be proven (32-bit
    // words, P x M <= 16): range_width = 2 + max(len, insert_size_max) / stride <= MCQ_TC_MAX_WINDOWS  <=>  max(len,
    // insert_size_max) < (MCQ_TC_MAX_WINDOWS - 1) x stride = tc_limit; the queueing kernels count the queries below it.
    const u64 tc_len = (u64)(MCQ_TC_MAX_WINDOWS - 1) * D.tgt_winstride;
    const bool tc_lists = od_in.lin ? od_in.P <= 8 : od_in.P * od_in.max_cand <= MCQ_TWO_CLASS_MAX_PM;     // (one list: ~40 / P of the light prefix are rank-0 words)
    const bool with_tc = db->d.compact && !od_in.big && !tap && tc_lists && !(od_in.hooks & 8u) && od_in.insert_size_max < tc_len;
    OptDev od = od_in;
    od.tc_limit = with_tc ? tc_len : 0;
    // P x M beyond a wave's 64 lanes (the reference's mpiexec -n 32 / -n 64 with -maxcand 4): up to pow2ceil(P) x pow2ceil(M) = 256
    // the first wave stage keeps the lists in four registers per lane (NL instantiation; 32-bit words); what overflows it, and
    // every other case, takes the workgroup kernel with the lists in its LDS (`big`)
    const u32 pm_slots = (u32)pow2ceil64(od_in.P) * (u32)pow2ceil64(od_in.max_cand);
    const bool many = od_in.big && db->d.compact && !tap && !shp && pm_slots <= 256 && !(force_block_in & 1);
    const int force_block = force_block_in | ((od_in.big && !many) ? 1 : 0) | (many ? 4 : 0);
    OptDev od_many = od;
    if (many) { od_many.big = 0; od_many.seg = (u32)pow2ceil64(od_in.max_cand); }
    const bool gw = db->g.on != 0;      // 32-bit locations in the global-window form: the GW instantiations
    const bool b64 = db->d.bsh != 0;    // table layout: the wave kernels are instantiated per layout (taps and the sharded home side: run-time / unused)
#define MCQ_LAUNCH_WAVE(KT, TAPV, SHV, GWV, BSHV) hipLaunchKernelGGL((k_query_wave<KT, kLcapWave, TAPV, SHV, GWV, BSHV>), dim3(grid), dim3(256), 0, st, D, b, od, o, ws->ctr, ws->ovf_list, force_block, dbg, sh, db->g)
#define MCQ_LAUNCH_WAVE_L(KT, GWV) do { if (b64) MCQ_LAUNCH_WAVE(KT, false, false, GWV, 2); else MCQ_LAUNCH_WAVE(KT, false, false, GWV, 0); } while (0)
    if (shp)                { if (!db->d.compact) MCQ_LAUNCH_WAVE(u64, false, true, false, -1); else if (gw) MCQ_LAUNCH_WAVE(u32, false, true, true, -1); else MCQ_LAUNCH_WAVE(u32, false, true, false, -1); }
    else if (tap)           { if (!db->d.compact) MCQ_LAUNCH_WAVE(u64, true, false, false, -1); else if (gw) MCQ_LAUNCH_WAVE(u32, true, false, true, -1); else MCQ_LAUNCH_WAVE(u32, true, false, false, -1); }
    else if (many) {
#define MCQ_LAUNCH_MANY(GWV, BSHV) hipLaunchKernelGGL((k_query_wave<u32, kLcapWave, false, false, GWV, BSHV, 4>), dim3(grid_for(ws->cap_wave_many, want)), dim3(256), 0, st, D, b, od_many, o, ws->ctr, ws->ovf_list, force_block, dbg, sh, db->g)
        if (gw) { if (b64) MCQ_LAUNCH_MANY(true, 2); else MCQ_LAUNCH_MANY(true, 0); }
        else    { if (b64) MCQ_LAUNCH_MANY(false, 2); else MCQ_LAUNCH_MANY(false, 0); }
#undef MCQ_LAUNCH_MANY
    }
    else if (db->d.compact) { if (gw) MCQ_LAUNCH_WAVE_L(u32, true); else MCQ_LAUNCH_WAVE_L(u32, false); }
    else                    MCQ_LAUNCH_WAVE_L(u64, false);
#undef MCQ_LAUNCH_WAVE_L
#undef MCQ_LAUNCH_WAVE
    rc = tm.mark(); if (rc) return rc;
    if (db->d.compact) {   // second wave stage (back queue); no queue for 64-bit keys
        const dim3 g16(grid_for(ws->cap_wave16, want));
#define MCQ_LAUNCH_WAVE16(TAPV, SHV, GWV, BSHV) hipLaunchKernelGGL((k_query_wave16<TAPV, SHV, GWV, BSHV>), g16, dim3(256), 0, st, D, b, od, o, ws->ctr, ws->ovf_list, dbg, sh, db->g)
        if (shp)      { if (gw) MCQ_LAUNCH_WAVE16(false, true, true, -1); else MCQ_LAUNCH_WAVE16(false, true, false, -1); }
        else if (tap) { if (gw) MCQ_LAUNCH_WAVE16(true, false, true, -1); else MCQ_LAUNCH_WAVE16(true, false, false, -1); }
        else if (gw)  { if (b64) MCQ_LAUNCH_WAVE16(false, false, true, 2); else MCQ_LAUNCH_WAVE16(false, false, true, 0); }
        else          { if (b64) MCQ_LAUNCH_WAVE16(false, false, false, 2); else MCQ_LAUNCH_WAVE16(false, false, false, 0); }
#undef MCQ_LAUNCH_WAVE16
    }
    if (with_tc) {          // third wave stage: front-queue entries of up to 2048 locations (see k_query_wave32); counts the narrow ones it leaves
        const dim3 g32(grid_for(ws->cap_wave32, want));
#define MCQ_LAUNCH_WAVE32(SHV, GWV, BSHV) hipLaunchKernelGGL((k_query_wave32<SHV, GWV, BSHV>), g32, dim3(256), 0, st, D, b, od, o, ws->ctr, ws->ovf_list, force_block, sh, db->g)
        if (shp)     { if (gw) MCQ_LAUNCH_WAVE32(true, true, -1); else MCQ_LAUNCH_WAVE32(true, false, -1); }
        else if (gw) { if (b64) MCQ_LAUNCH_WAVE32(false, true, 2); else MCQ_LAUNCH_WAVE32(false, true, 0); }
        else         { if (b64) MCQ_LAUNCH_WAVE32(false, false, 2); else MCQ_LAUNCH_WAVE32(false, false, 0); }
#undef MCQ_LAUNCH_WAVE32
    }
    rc = tm.mark(); if (rc) return rc;
#define MCQ_LAUNCH_BLOCK(KT, LC, NTH, BIGV, SHV, GWV) hipLaunchKernelGGL((k_query_block<KT, LC, NTH, BIGV, SHV, GWV, false>), dim3(ws->n_block_wgs), dim3(NTH), 0, st, D, b, od, o, ws->ctr, \
                                                             (const u32*)ws->ovf_list, ws->sc, dbg, sh, db->g)
#define MCQ_LAUNCH_BLOCK2(KT, LC, NTH, SHV, GWV) do { if (od.lin) MCQ_LAUNCH_BLOCK(KT, LC, NTH, 2, SHV, GWV); else if (od.big) MCQ_LAUNCH_BLOCK(KT, LC, NTH, 1, SHV, GWV); \
                                                      else MCQ_LAUNCH_BLOCK(KT, LC, NTH, 0, SHV, GWV); } while (0)
#define MCQ_LAUNCH_BLOCK32(SHV) do { if (gw) MCQ_LAUNCH_BLOCK2(u32, MCQ_BLOCK_LCAP, MCQ_BLOCK_NT, SHV, true); else MCQ_LAUNCH_BLOCK2(u32, MCQ_BLOCK_LCAP, MCQ_BLOCK_NT, SHV, false); } while (0)
    if (db->d.compact) { if (shp) MCQ_LAUNCH_BLOCK32(true); else MCQ_LAUNCH_BLOCK32(false); }
    else               { if (shp) MCQ_LAUNCH_BLOCK2(u64, kLcapBlock, 1024, true, false); else MCQ_LAUNCH_BLOCK2(u64, kLcapBlock, 1024, false, false); }
#undef MCQ_LAUNCH_BLOCK32
#undef MCQ_LAUNCH_BLOCK2
#undef MCQ_LAUNCH_BLOCK
    if (with_tc) {
#define MCQ_LAUNCH_TC1(FORMV, SHV, GWV) hipLaunchKernelGGL((k_query_block<u32, MCQ_BLOCK_LCAP, MCQ_BLOCK_NT, FORMV, SHV, GWV, true>), dim3(ws->n_block_wgs), dim3(MCQ_BLOCK_NT), 0, st, \
                                                   D, b, od, o, ws->ctr, (const u32*)ws->ovf_list, ws->sc, dbg, sh, db->g)
#define MCQ_LAUNCH_TC(SHV, GWV) do { if (od.lin) MCQ_LAUNCH_TC1(2, SHV, GWV); else MCQ_LAUNCH_TC1(0, SHV, GWV); } while (0)
        if (shp) { if (gw) MCQ_LAUNCH_TC(true, true); else MCQ_LAUNCH_TC(true, false); }
        else     { if (gw) MCQ_LAUNCH_TC(false, true); else MCQ_LAUNCH_TC(false, false); }
#undef MCQ_LAUNCH_TC1
#undef MCQ_LAUNCH_TC
    }
    rc = tm.end(); if (rc) return rc;
#ifndef MCQ_NO_DIRECT_MODE
    if (db->d.compact && !tap && !many && !(force_block & 7) && !getenv("MCQ_NO_DIRECT_MODE")) hipLaunchKernelGGL(k_next_mode, dim3(1), dim3(1), 0, st, ws->ctr, b.nq);
    else HIPCHK(hipMemsetAsync(&ws->ctr->direct_mode, 0, 4, st));
#endif
    HIPCHK(hipGetLastError());
    ws->last_nq = b.nq;
    return MCQ_OK;
}

extern "C" int mcq_query(const mcq_db* db, mcq_ws* ws, const mcq_batch* in, const mcq_query_opts* opt,
                         mcq_result* out, void* stream) {
    if (!db || !ws || !in || !opt || !out) return fail(MCQ_E_ARG, "null argument");
    OptDev od;
    int rc = make_opt(opt, od, db);
    if (rc) return rc;
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = (hipStream_t)stream;
    const u64 nq = in->paired ? in->n_seqs / 2 : in->n_seqs;
    if (nq > ws->max_queries) return fail(MCQ_E_ARG, "batch has more queries than the workspace allows");
    const bool dev_in = (in->flags & MCQ_DEVICE_PTRS) != 0, dev_out = (out->flags & MCQ_DEVICE_PTRS) != 0;
    const bool packed = (in->flags & MCQ_BATCH_PACKED) != 0;
    if ((in->flags & MCQ_BATCH_RANGES) && !dev_in) return fail(MCQ_E_ARG, "MCQ_BATCH_RANGES needs device pointers");
    OutDev o;
    u64 nbases = 0;
    if (!dev_in) {
        nbases = in->n_seqs ? in->seq_off[in->n_seqs] - in->seq_off[0] : 0;
        if (nbases > ws->max_bases) return fail(MCQ_E_ARG, "batch has more bases than the workspace allows");
        if (in->n_seqs && in->seq_off[0] != 0) return fail(MCQ_E_ARG, "host batches must start at offset 0");
        if (packed && in->n_bases && in->n_bases != nbases) return fail(MCQ_E_ARG, "mcq_batch.n_bases must equal seq_off[n_seqs] for a packed batch");
    }
    if (!dev_in || !dev_out) { rc = ensure_staging(ws); if (rc) return rc; }
    BatchDev b;
    if (!dev_in) {
        const u64 bytes = packed ? mcq_packed_bytes(nbases) : nbases;      // (a packed batch is at most as large as its ASCII form + 16 B)
        if (bytes) HIPCHK(hipMemcpyAsync(ws->d_bases, in->bases, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ws->d_seq_off, in->seq_off, (in->n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
        mcq_batch hin = *in; hin.n_bases = nbases;
        rc = batch_dev(&hin, ws->d_bases, ws->d_seq_off, b);
    } else rc = batch_dev(in, in->bases, in->seq_off, b);
    if (rc) return rc;
    if (!dev_out) { o.cands = ws->d_cands; o.ncand = ws->d_ncand; }
    else { o.cands = (u32*)out->cands; o.ncand = out->n_cand; }
    DebugDev dbg; memset(&dbg, 0, sizeof(dbg));
    rc = launch_query(db, ws, b, od, o, st, force_bits(opt->flags), dbg);
    if (rc) return rc;
    if (!dev_out && nq) {
        HIPCHK(hipMemcpyAsync(out->cands, ws->d_cands, nq * od.max_cand * 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(out->n_cand, ws->d_ncand, nq * 4, hipMemcpyDeviceToHost, st));
    }
    if (!dev_in || !dev_out) return mcq_ws_sync(ws, stream, nullptr);
    return MCQ_OK;
}

// ------------------------------------------------------------------ host buffers at rate: copy in / compute / copy out overlapped
// mcq_query with host pointers is synchronous (copy, run, copy, wait): 1.87e8 reads/s on configs[1], the copies being
// 3.3 ms of the 5.6.  The pipelined form keeps two batches in flight: batch i+1 is copied in and batch i-1 copied out
// on their own streams while batch i computes -- 3.2e8 reads/s with ASCII bases (then bound by PCIe: 157 MB per 1 M reads
// at 47 GB/s), kernel-bound with MCQ_BATCH_PACKED.
static int pipe_init(mcq_ws* ws) {
    auto& p = ws->pipe;
    if (p.ready) return MCQ_OK;
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipMalloc(&p.d_bases[k], std::max<u64>(1, ws->max_bases) + 16));
        HIPCHK(hipMalloc(&p.d_seq_off[k], (2 * ws->max_queries + 2) * 8));
        HIPCHK(hipMalloc(&p.d_cands[k], std::max<u64>(1, ws->max_queries) * 16 * 16));
        HIPCHK(hipMalloc(&p.d_ncand[k], std::max<u64>(1, ws->max_queries) * 4));
        HIPCHK(hipEventCreateWithFlags(&p.ev_in[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p.ev_k[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p.ev_out[k], hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&p.s_k, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking));
    p.issued = 0; p.ready = true;
    return MCQ_OK;
}

extern "C" int mcq_query_pipelined(const mcq_db* db, mcq_ws* ws, const mcq_batch* in, const mcq_query_opts* opt,
                                   mcq_result* out, uint64_t* ticket) {
    if (!db || !ws || !in || !opt || !out || !ticket) return fail(MCQ_E_ARG, "null argument");
    if ((in->flags & MCQ_DEVICE_PTRS) || (out->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "the pipelined call takes host buffers");
    if (in->flags & MCQ_BATCH_RANGES) return fail(MCQ_E_ARG, "MCQ_BATCH_RANGES needs device pointers");
    OptDev od;
    int rc = make_opt(opt, od, db); if (rc) return rc;
    HIPCHK(hipSetDevice(db->device));
    const u64 nq = in->paired ? in->n_seqs / 2 : in->n_seqs;
    if (nq > ws->max_queries) return fail(MCQ_E_ARG, "batch has more queries than the workspace allows");
    const u64 nbases = in->n_seqs ? in->seq_off[in->n_seqs] - in->seq_off[0] : 0;
    if (nbases > ws->max_bases) return fail(MCQ_E_ARG, "batch has more bases than the workspace allows");
    if (in->n_seqs && in->seq_off[0] != 0) return fail(MCQ_E_ARG, "host batches must start at offset 0");
    const bool packed = (in->flags & MCQ_BATCH_PACKED) != 0;
    if (packed && in->n_bases && in->n_bases != nbases) return fail(MCQ_E_ARG, "mcq_batch.n_bases must equal seq_off[n_seqs] for a packed batch");
    rc = pipe_init(ws); if (rc) return rc;
    auto& p = ws->pipe;
    const u64 i = p.issued;
    const int k = (int)(i & 1);
    // in: this staging set was last read by the kernels of call i - 2
    if (i >= 2) HIPCHK(hipStreamWaitEvent(p.s_in, p.ev_k[k], 0));
    const u64 bytes = packed ? mcq_packed_bytes(nbases) : nbases;
    if (bytes) HIPCHK(hipMemcpyAsync(p.d_bases[k], in->bases, bytes, hipMemcpyHostToDevice, p.s_in));
    HIPCHK(hipMemcpyAsync(p.d_seq_off[k], in->seq_off, (in->n_seqs + 1) * 8, hipMemcpyHostToDevice, p.s_in));
    HIPCHK(hipEventRecord(p.ev_in[k], p.s_in));
    // compute: after its input arrived and its result set was copied out (call i - 2)
    HIPCHK(hipStreamWaitEvent(p.s_k, p.ev_in[k], 0));
    if (i >= 2) HIPCHK(hipStreamWaitEvent(p.s_k, p.ev_out[k], 0));
    mcq_batch hin = *in; hin.n_bases = nbases;
    BatchDev b; rc = batch_dev(&hin, p.d_bases[k], p.d_seq_off[k], b); if (rc) return rc;
    OutDev o; o.cands = p.d_cands[k]; o.ncand = p.d_ncand[k];
    DebugDev dbg; memset(&dbg, 0, sizeof(dbg));
    rc = launch_query(db, ws, b, od, o, p.s_k, force_bits(opt->flags), dbg); if (rc) return rc;
    HIPCHK(hipEventRecord(p.ev_k[k], p.s_k));
    // out
    HIPCHK(hipStreamWaitEvent(p.s_out, p.ev_k[k], 0));
    if (nq) {
        HIPCHK(hipMemcpyAsync(out->cands, p.d_cands[k], nq * od.max_cand * 16, hipMemcpyDeviceToHost, p.s_out));
        HIPCHK(hipMemcpyAsync(out->n_cand, p.d_ncand[k], nq * 4, hipMemcpyDeviceToHost, p.s_out));
    }
    HIPCHK(hipEventRecord(p.ev_out[k], p.s_out));
    *ticket = i;
    p.issued = i + 1;
    return MCQ_OK;
}

extern "C" int mcq_ws_wait(mcq_ws* ws, uint64_t ticket) {
    if (!ws || !ws->pipe.ready) return fail(MCQ_E_ARG, "no pipelined call on this workspace");
    auto& p = ws->pipe;
    if (ticket >= p.issued) return fail(MCQ_E_ARG, "unknown ticket");
    // s_out is in order: the event of a later call on the same staging set is recorded behind this ticket's copy, so waiting
    // for whatever was recorded last on it covers the ticket (a device-side hipStreamWaitEvent of a later call is NOT a
    // host wait: the header promises that the results are in `out` when this returns)
    HIPCHK(hipSetDevice(ws->device));
    HIPCHK(hipEventSynchronize(p.ev_out[ticket & 1]));
    return MCQ_OK;
}

extern "C" int mcq_ws_sync(mcq_ws* ws, void* stream, mcq_stats* stats) {
    if (!ws) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(ws->device));
    hipStream_t st = (hipStream_t)stream;
    if (ws->pipe.ready && ws->pipe.issued) { HIPCHK(hipStreamSynchronize(ws->pipe.s_k)); HIPCHK(hipStreamSynchronize(ws->pipe.s_out)); }
    HIPCHK(hipMemcpyAsync(ws->ctr_host, ws->ctr, sizeof(CountersDev), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (stats) {
        stats->n_queries = ws->last_nq;
        stats->n_features = ws->ctr_host->n_features; stats->n_hit_features = ws->ctr_host->n_hit_features;
        stats->n_locations = ws->ctr_host->n_locations; stats->n_cands = ws->ctr_host->n_cands;
        stats->n_overflow = ws->ctr_host->n_ovf - ws->ctr_host->n_short;      // (direct mode: every query is queued; those the first stage would have kept are not counted)
        stats->n_two_class = ws->ctr_host->n_two_class; stats->n_two_class_retry = ws->ctr_host->n_two_class_retry;
        stats->n_narrow_queued = ws->ctr_host->n_narrow;
    }
    if (ws->ctr_host->err_count)
        return fail(MCQ_E_CAPACITY, std::to_string(ws->ctr_host->err_count) + " queries exceeded the workspace's per-query capacity");
    return MCQ_OK;
}

// diagnostic builds (-DMCQ_PHASE_CLOCK): shader clocks per phase of the workgroup kernel, summed over its workgroups, of the last call
// synchronised by mcq_ws_sync (zeros in a normal build)
extern "C" int mcq_debug_phase_clocks(mcq_ws* ws, uint64_t* out22) {
    if (!ws || !out22) return fail(MCQ_E_ARG, "null argument");
    for (int i = 0; i < 22; ++i) out22[i] = i < 17 ? ws->ctr_host->pad_[i] : 0;
    return MCQ_OK;
}

// ------------------------------------------------------------------ debug tap: sorted match lists
extern "C" int mcq_debug_matches(const mcq_db* db, mcq_ws* ws, const mcq_batch* in, uint32_t path_flags,
                                 uint64_t* match_off, uint64_t* matches, uint64_t cap) {
    if (!db || !ws || !in || !match_off) return fail(MCQ_E_ARG, "null argument");
    if (path_flags & ~(u32)(MCQ_FORCE_BLOCK_PATH | MCQ_FORCE_RAW_SORT | MCQ_NO_WAVE16)) return fail(MCQ_E_ARG, "path_flags: test hooks only");
    if (in->flags & MCQ_DEVICE_PTRS) return fail(MCQ_E_ARG, "debug tap takes host batches");
    HIPCHK(hipSetDevice(db->device));
    const u64 nq = in->paired ? in->n_seqs / 2 : in->n_seqs;
    if (nq > ws->max_queries) return fail(MCQ_E_ARG, "batch too large");
    int rc = ensure_staging(ws); if (rc) return rc;
    const u64 nbases = in->n_seqs ? in->seq_off[in->n_seqs] : 0;
    if (nbases > ws->max_bases) return fail(MCQ_E_ARG, "batch too large");
    if (nbases) HIPCHK(hipMemcpy(ws->d_bases, in->bases, nbases, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ws->d_seq_off, in->seq_off, (in->n_seqs + 1) * 8, hipMemcpyHostToDevice));
    mcq_query_opts qo; qo.max_cand = 1; qo.emulate_ranks = 1; qo.insert_size_max = 0; qo.flags = 0;
    OptDev od; rc = make_opt(&qo, od, db); if (rc) return rc;
    if (in->flags & (MCQ_BATCH_PACKED | MCQ_BATCH_RANGES)) return fail(MCQ_E_ARG, "debug tap takes plain ASCII batches");
    BatchDev b; rc = batch_dev(in, ws->d_bases, ws->d_seq_off, b); if (rc) return rc;
    OutDev o; o.cands = ws->d_cands; o.ncand = ws->d_ncand;
    u64 *d_cnt = nullptr, *d_off = nullptr, *d_m = nullptr;
    HIPCHK(hipMalloc(&d_cnt, std::max<u64>(1, nq) * 8));
    HIPCHK(hipMemset(d_cnt, 0, std::max<u64>(1, nq) * 8));
    HIPCHK(hipMalloc(&d_off, (nq + 1) * 8));
    DebugDev dbg; memset(&dbg, 0, sizeof(dbg));
    dbg.mode = 1; dbg.match_cnt = d_cnt;
    rc = launch_query(db, ws, b, od, o, 0, force_bits(path_flags), dbg); if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    std::vector<u64> cnt(nq);
    if (nq) HIPCHK(hipMemcpy(cnt.data(), d_cnt, nq * 8, hipMemcpyDeviceToHost));
    match_off[0] = 0;
    for (u64 q = 0; q < nq; ++q) match_off[q + 1] = match_off[q] + cnt[q];
    if (matches && match_off[nq] <= cap && match_off[nq] > 0) {
        HIPCHK(hipMalloc(&d_m, match_off[nq] * 8));
        HIPCHK(hipMemcpy(d_off, match_off, (nq + 1) * 8, hipMemcpyHostToDevice));
        dbg.mode = 2; dbg.match_off = d_off; dbg.matches = d_m;
        rc = launch_query(db, ws, b, od, o, 0, force_bits(path_flags), dbg); if (rc) return rc;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(matches, d_m, match_off[nq] * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_m);
    }
    (void)hipFree(d_cnt); (void)hipFree(d_off);
    return MCQ_OK;
}

// ------------------------------------------------------------------ staged entry points (device pointers only)
extern "C" int mcq_count_windows(const mcq_db* db, const mcq_batch* in, uint64_t* win_off, void* stream) {
    if (!db || !in || !win_off) return fail(MCQ_E_ARG, "null argument");
    if (!(in->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "staged entry points take device pointers");
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = (hipStream_t)stream;
    const u64 n = in->n_seqs;
    u64* cnt = nullptr;
    HIPCHK(hipMallocAsync((void**)&cnt, std::max<u64>(1, n) * 8, st));
    if (n) hipLaunchKernelGGL(k_count_windows, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, in->seq_off, (in->flags & MCQ_BATCH_RANGES) ? 1u : 0u, n, db->d.winlen, db->d.winstride, cnt);
    { int rcs = device_exclusive_scan<u64>((const u64*)cnt, win_off, n, st, 256); if (rcs) return rcs; }
    HIPCHK(hipFreeAsync(cnt, st));
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

extern "C" int mcq_sketch(const mcq_db* db, const mcq_batch* in, const uint64_t* win_off,
                          uint32_t* features, uint32_t* n_feat, void* stream) {
    if (!db || !in || !win_off || !features || !n_feat) return fail(MCQ_E_ARG, "null argument");
    if (!(in->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "staged entry points take device pointers");
    if (in->flags & MCQ_BATCH_PACKED) return fail(MCQ_E_ARG, "mcq_sketch takes ASCII batches");
    HIPCHK(hipSetDevice(db->device));
    if (in->n_seqs == 0) return MCQ_OK;
    // many short sequences (reads): one wave per sequence; few long ones (genomes): one wave per window
    if (in->n_seqs >= 4096) {
        const u32 grid = (u32)std::min<u64>((in->n_seqs + 3) / 4, 256ull * 32);    // short items: several rounds balance better
        hipLaunchKernelGGL(k_sketch_seqs, dim3(grid), dim3(256), 0, (hipStream_t)stream, db->d, in->bases, in->seq_off,
                           (in->flags & MCQ_BATCH_RANGES) ? 1u : 0u, in->n_seqs, win_off, features, n_feat);
    } else {
        hipLaunchKernelGGL(k_sketch_windows, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, db->d, in->bases, in->seq_off,
                           (in->flags & MCQ_BATCH_RANGES) ? 1u : 0u, in->n_seqs, win_off, features, n_feat);
    }
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

extern "C" uint32_t mcq_db_loc_bytes(const mcq_db* db) { return db && db->d.compact ? 4u : 8u; }
extern "C" uint32_t mcq_db_win_bits(const mcq_db* db) { return db ? db->d.wb : 32u; }
extern "C" int mcq_db_layout_get(const mcq_db* db, mcq_db_layout* out) {
    if (!db || !out) return fail(MCQ_E_ARG, "null argument");
    memset(out, 0, sizeof(*out));
    out->loc_bytes = db->d.compact ? 4u : 8u;
    out->loc_format = db->g.on ? MCQ_LOC_GLOBAL_WINDOW : db->d.compact ? MCQ_LOC_FIELDS32 : MCQ_LOC_FIELDS64;
    out->win_bits = db->d.wb; out->bucket_bytes = db->bucket_bytes; out->slots_per_key = db->slots_per_key;
    out->n_slots = db->nslots; out->n_keys = db->n_keys_local; out->n_locs = db->n_locs_local; out->n_ext_locs = db->n_ext;
    out->n_windows = db->n_windows; out->bytes = db->bytes;
    out->gw_offsets = db->gw_off;
    return MCQ_OK;
}

extern "C" int mcq_lookup_count(const mcq_db* db, const uint32_t* features, uint64_t n_features,
                                uint32_t* list_len, uint64_t* list_src, void* stream) {
    if (!db || (n_features && (!features || !list_len))) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(db->device));
    if (n_features == 0) return MCQ_OK;
    hipLaunchKernelGGL(k_lookup_count, dim3((u32)((n_features + 255) / 256)), dim3(256), 0, (hipStream_t)stream, db->d, features,
                       n_features, list_len, list_src);
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

extern "C" int mcq_lookup_gather(const mcq_db* db, const uint32_t* features, uint64_t n_features,
                                 const uint32_t* list_len, const uint64_t* list_src,
                                 const uint64_t* out_off, void* out_locs, void* stream) {
    if (!db || (n_features && (!features || !out_off))) return fail(MCQ_E_ARG, "null argument");
    if (list_src && !list_len) return fail(MCQ_E_ARG, "list_src needs list_len");
    HIPCHK(hipSetDevice(db->device));
    if (n_features == 0) return MCQ_OK;
    u64 groups = (n_features + 63) / 64;
    const u32 grid = (u32)std::min<u64>((groups + 3) / 4, 256ull * 32);
    if (db->d.compact) hipLaunchKernelGGL(k_lookup_gather<u32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, db->d, features, n_features, list_len, list_src, out_off, (u32*)out_locs);
    else               hipLaunchKernelGGL(k_lookup_gather<u64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, db->d, features, n_features, list_len, list_src, out_off, (u64*)out_locs);
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

extern "C" int mcq_assemble(const mcq_db* db, uint64_t n_lists, const uint32_t* list_len, const uint32_t* src_slot,
                            uint64_t n_slots, const void* src_locs, const mcq_batch* in, const uint64_t* win_off,
                            uint64_t* loc_off, uint32_t* query_len, void* dst_locs, void* stream) {
    if (!db || !in || !win_off || !loc_off || !query_len) return fail(MCQ_E_ARG, "null argument");
    if (!(in->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "staged entry points take device pointers");
    if (n_lists && (!list_len || !src_slot)) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = (hipStream_t)stream;
    const u64 nq = in->paired ? in->n_seqs / 2 : in->n_seqs;
    u32* slot_len = nullptr; u64 *dst_off = nullptr, *src_off = nullptr;
    HIPCHK(hipMallocAsync((void**)&slot_len, std::max<u64>(1, n_slots) * 4, st));
    HIPCHK(hipMallocAsync((void**)&dst_off, (n_slots + 1) * 8, st));
    HIPCHK(hipMallocAsync((void**)&src_off, (n_lists + 1) * 8, st));
    HIPCHK(hipMemsetAsync(slot_len, 0, std::max<u64>(1, n_slots) * 4, st));
    if (n_lists) hipLaunchKernelGGL(k_scatter_len, dim3((u32)((n_lists + 255) / 256)), dim3(256), 0, st, list_len, src_slot, n_lists, slot_len);
    int rc = device_exclusive_scan<u32>(slot_len, dst_off, n_slots, st); if (rc) return rc;
    rc = device_exclusive_scan<u32>(list_len, src_off, n_lists, st); if (rc) return rc;
    if (n_lists) {
        u64 groups = (n_lists + 63) / 64;
        const u32 grid = (u32)std::min<u64>((groups + 3) / 4, 256ull * 32);
        if (db->d.compact) hipLaunchKernelGGL(k_scatter_lists<u32>, dim3(grid), dim3(256), 0, st, n_lists, (const u64*)src_off, src_slot, (const u64*)dst_off, (const u32*)src_locs, (u32*)dst_locs);
        else               hipLaunchKernelGGL(k_scatter_lists<u64>, dim3(grid), dim3(256), 0, st, n_lists, (const u64*)src_off, src_slot, (const u64*)dst_off, (const u64*)src_locs, (u64*)dst_locs);
    }
    hipLaunchKernelGGL(k_query_offsets, dim3((u32)((nq + 256) / 256)), dim3(256), 0, st, nq, in->paired ? 2u : 1u, db->d.s, win_off,
                       in->seq_off, (in->flags & MCQ_BATCH_RANGES) ? 1u : 0u, (const u64*)dst_off, n_slots, loc_off, query_len);
    HIPCHK(hipFreeAsync(slot_len, st)); HIPCHK(hipFreeAsync(dst_off, st)); HIPCHK(hipFreeAsync(src_off, st));
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

extern "C" int mcq_reduce(const mcq_db* db, mcq_ws* ws, uint64_t n_queries, const uint64_t* loc_off,
                          const void* locs, const uint32_t* query_len, const mcq_query_opts* opt, mcq_result* out, void* stream) {
    if (!db || !ws || !opt || !out || (n_queries && (!loc_off || !query_len))) return fail(MCQ_E_ARG, "null argument");
    if (!(out->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "staged entry points take device pointers");
    if (n_queries > ws->max_queries) return fail(MCQ_E_ARG, "more queries than the workspace allows");
    OptDev od;
    int rc = make_opt(opt, od, db);
    if (rc) return rc;
    HIPCHK(hipSetDevice(db->device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(ws->ctr, 0, MCQ_CTR_ZEROED, st));
    ws->last_nq = n_queries;
    if (n_queries == 0) return MCQ_OK;
    OutDev o; o.cands = (u32*)out->cands; o.ncand = out->n_cand;
    const u32 grid = (u32)std::min<u64>((n_queries + 3) / 4, 256ull * 24);     // measured: 1.07 ms vs 1.25 ms at the resident 8 per CU
    LaunchTimer tm(ws, st);
    rc = tm.begin(); if (rc) return rc;
#define MCQ_REDUCE32(GWV) do { \
        hipLaunchKernelGGL((k_reduce_wave<u32, kLcapWave, GWV>), dim3(grid), dim3(256), 0, st, db->d, od, o, ws->ctr, ws->ovf_list, \
                           n_queries, loc_off, (const u32*)locs, query_len, db->g); \
        rc = tm.mark(); if (rc) return rc; \
        hipLaunchKernelGGL(k_reduce_wave16<GWV>, dim3(grid_for(ws->cap_reduce16, (n_queries + 3) / 4)), dim3(256), 0, st, db->d, od, o, ws->ctr, \
                           (const u32*)ws->ovf_list, n_queries, loc_off, (const u32*)locs, query_len, db->g); \
        rc = tm.mark(); if (rc) return rc; \
        if (od.big) hipLaunchKernelGGL((k_reduce_block<u32, kLcapBlock, 1, GWV>), dim3(ws->n_block_wgs), dim3(1024), 0, st, db->d, od, o, ws->ctr, \
                                       (const u32*)ws->ovf_list, ws->sc, loc_off, (const u32*)locs, query_len, db->g); \
        else        hipLaunchKernelGGL((k_reduce_block<u32, kLcapBlock, 0, GWV>), dim3(ws->n_block_wgs), dim3(1024), 0, st, db->d, od, o, ws->ctr, \
                                       (const u32*)ws->ovf_list, ws->sc, loc_off, (const u32*)locs, query_len, db->g); } while (0)
    if (db->d.compact) {
        if (db->g.on) MCQ_REDUCE32(true); else MCQ_REDUCE32(false);
    } else {
        hipLaunchKernelGGL((k_reduce_wave<u64, kLcapWave>), dim3(grid), dim3(256), 0, st, db->d, od, o, ws->ctr, ws->ovf_list,
                           n_queries, loc_off, (const u64*)locs, query_len, db->g);
        rc = tm.mark(); if (rc) return rc;
        rc = tm.mark(); if (rc) return rc;
        if (od.big) hipLaunchKernelGGL((k_reduce_block<u64, kLcapBlock, 1>), dim3(ws->n_block_wgs), dim3(1024), 0, st, db->d, od, o, ws->ctr,
                                       (const u32*)ws->ovf_list, ws->sc, loc_off, (const u64*)locs, query_len, db->g);
        else        hipLaunchKernelGGL((k_reduce_block<u64, kLcapBlock, 0>), dim3(ws->n_block_wgs), dim3(1024), 0, st, db->d, od, o, ws->ctr,
                                       (const u32*)ws->ovf_list, ws->sc, loc_off, (const u64*)locs, query_len, db->g);
    }
#undef MCQ_REDUCE32
    rc = tm.end(); if (rc) return rc;
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

// ------------------------------------------------------------------ per-kernel timing
static int drain_events(mcq_ws* ws) {
    for (auto& t : *ws->ev_used) {
        HIPCHK(hipEventSynchronize(t.ev[MCQ_N_TIMED]));
        for (int i = 0; i < MCQ_N_TIMED; ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
            ws->timed_ms[i] += ms;
        }
        ws->timed_launches += 1;
        ws->ev_free->push_back(t);
    }
    ws->ev_used->clear();
    return MCQ_OK;
}

extern "C" int mcq_ws_timing(mcq_ws* ws, int enable) {
    if (!ws) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(ws->device));
    int rc = drain_events(ws); if (rc) return rc;
    ws->timing = enable ? 1 : 0;
    if (enable) { for (auto& m : ws->timed_ms) m = 0; ws->timed_launches = 0; }
    return MCQ_OK;
}

extern "C" int mcq_ws_kernel_times(mcq_ws* ws, double* ms, uint64_t* n_batches) {
    if (!ws) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(ws->device));
    int rc = drain_events(ws); if (rc) return rc;
    if (ms) for (int i = 0; i < MCQ_N_TIMED; ++i) ms[i] = ws->timed_ms[i];
    if (n_batches) *n_batches = ws->timed_launches;
    return MCQ_OK;
}

extern "C" int mcq_ws_kernel_time(mcq_ws* ws, double* total_ms, uint64_t* n_batches) {
    double ms[MCQ_N_TIMED];
    int rc = mcq_ws_kernel_times(ws, ms, n_batches); if (rc) return rc;
    if (total_ms) { *total_ms = 0; for (double m : ms) *total_ms += m; }
    return MCQ_OK;
}

// ------------------------------------------------------------------ sharded-path routing entry points
extern "C" int mcq_bucket_features(const uint32_t* features, uint64_t n, uint32_t n_shards,
                                   uint64_t* counts, uint32_t* bucketed, uint32_t* src_index, void* stream) {
    if (!counts || (n && (!features || !bucketed || !src_index))) return fail(MCQ_E_ARG, "null argument");
    if (n_shards < 1 || n_shards > MCQ_BUCKET_MAX_SHARDS) return fail(MCQ_E_ARG, "n_shards must be 1..64");
    if (n >= (1ull << 32)) return fail(MCQ_E_UNSUPPORTED, "more than 2^32 feature slots in one batch");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { HIPCHK(hipMemsetAsync(counts, 0, (u64)n_shards * 8, st)); return MCQ_OK; }
    const u32 grid = (u32)std::min<u64>((n + 4095) / 4096, 2048);
    const u64 tile = ((n + grid - 1) / grid + 255) / 256 * 256;
    unsigned long long* blk = nullptr;
    HIPCHK(hipMallocAsync((void**)&blk, (u64)n_shards * grid * 8, st));
    hipLaunchKernelGGL(k_bucket_count, dim3(grid), dim3(256), 0, st, features, n, n_shards, tile, blk);
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, st, blk, n_shards, grid, (unsigned long long*)counts);
    hipLaunchKernelGGL(k_bucket_fill, dim3(grid), dim3(256), 0, st, features, n, n_shards, tile, (const unsigned long long*)blk, bucketed, src_index);
    HIPCHK(hipFreeAsync(blk, st));
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

// ------------------------------------------------------------------ row f4 entry point
static int text_index(const char* text, uint64_t n_bytes, uint64_t* seq_ranges, uint64_t max_seqs, uint64_t* n_seqs_out, void* stream, u32 L);
extern "C" int mcq_fastq_index(const char* text, uint64_t n_bytes, uint64_t* seq_ranges, uint64_t max_seqs,
                               uint64_t* n_seqs_out, void* stream) {
    return text_index(text, n_bytes, seq_ranges, max_seqs, n_seqs_out, stream, 4);
}
extern "C" int mcq_fasta_index(const char* text, uint64_t n_bytes, uint64_t* seq_ranges, uint64_t max_seqs,
                               uint64_t* n_seqs_out, void* stream) {
    return text_index(text, n_bytes, seq_ranges, max_seqs, n_seqs_out, stream, 2);
}
static int text_index(const char* text, uint64_t n_bytes, uint64_t* seq_ranges, uint64_t max_seqs, uint64_t* n_seqs_out, void* stream, u32 L) {
    if (!seq_ranges || !n_seqs_out || (n_bytes && !text)) return fail(MCQ_E_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const u64 n_tiles = std::max<u64>(1, (n_bytes + MCQ_FQ_TILE - 1) / MCQ_FQ_TILE);
    if (n_tiles >= (1ull << 31)) return fail(MCQ_E_UNSUPPORTED, "text too large for one call");
    u64 *cnt = nullptr, *off = nullptr;
    HIPCHK(hipMallocAsync((void**)&cnt, n_tiles * 8, st));
    HIPCHK(hipMallocAsync((void**)&off, (n_tiles + 1) * 8, st));
    hipLaunchKernelGGL(k_fq_count, dim3((u32)n_tiles), dim3(256), 0, st, text, n_bytes, cnt);
    int rc = device_exclusive_scan<u64>((const u64*)cnt, off, n_tiles, st); if (rc) return rc;
    hipLaunchKernelGGL(k_fq_ranges, dim3((u32)n_tiles), dim3(256), 0, st, text, n_bytes, (const u64*)off, n_tiles, seq_ranges, max_seqs, n_seqs_out, L);
    HIPCHK(hipFreeAsync(cnt, st)); HIPCHK(hipFreeAsync(off, st));
    HIPCHK(hipGetLastError());
    return MCQ_OK;
}

// ------------------------------------------------------------------ feature-sharded multi-GPU path (mcq_shard_*)
#include "mcq_shard.hpp"
