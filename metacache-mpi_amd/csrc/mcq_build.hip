// mcq_build.hip -- GPU construction of the feature -> locations table from reference sequences
// (row f2 of SURVEY.md section 8): mcq_build_table / mcq_db_build of include/mcq.h.
//
// Restates the reference's build-side insertion (add_all_window_sketches,
// src/sketch_database.h:1079-1097, with target t sketched on rank t % P, :540-542): every window of
// every target is sketched (same kernel as the query path, through mcq_sketch); per (feature,
// virtual rank) only the first max_locs = 254 locations in (target, window) order survive
// (:1090-1092); the table is the union of the P rank tables, lists in (target, window) order.
//
// Only the public C ABI of mcq_engine.hip is used from here (mcq_count_windows, mcq_sketch,
// mcq_db_create), plus rocPRIM's radix sort for the two global sorts -- a plain library sort of
// ~3e8 pairs, run once per database, outside any timed region.
#include <cstring>
#include <chrono>
#include <cstdio>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/mcq.h"

typedef uint32_t u32;
typedef uint64_t u64;

namespace {
thread_local std::string g_berr;
int bfail(int code, const std::string& m) { g_berr = m; return code; }
#define BCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return bfail(MCQ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define MCHK(expr) do { int r_ = (expr); if (r_ != MCQ_OK) return r_; } while (0)

// Temporaries of one build come from the device's default memory pool with its release threshold lifted, so a buffer
// freed by one phase is handed to the next without unmapping and remapping HBM (hipMalloc / hipFree of tens of GB were
// most of the build time of a 16 Gbp input).  The destructor frees what an error path left behind, restores the
// threshold and trims the pool, so nothing stays reserved after the call.
struct Scratch {
    hipMemPool_t pool = nullptr;
    uint64_t old_threshold = 0;
    std::vector<void*> live;
    hipError_t init(int device) {
        hipError_t e = hipDeviceGetDefaultMemPool(&pool, device);
        if (e != hipSuccess) return e;
        e = hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &old_threshold);
        if (e != hipSuccess) return e;
        uint64_t keep = ~0ull;
        return hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    template <class T> hipError_t get(T** p, u64 bytes) {
        void* q = nullptr;
        hipError_t e = hipMallocAsync(&q, bytes ? bytes : 1, 0);
        if (e == hipSuccess) { live.push_back(q); *p = static_cast<T*>(q); }
        return e;
    }
    void put(void* p) {
        if (!p) return;
        for (size_t i = 0; i < live.size(); ++i) if (live[i] == p) { live[i] = live.back(); live.pop_back(); break; }
        (void)hipFreeAsync(p, 0);
    }
    ~Scratch() {
        for (void* p : live) (void)hipFreeAsync(p, 0);
        (void)hipStreamSynchronize(0);
        if (pool) {
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &old_threshold);
            (void)hipMemPoolTrimTo(pool, 0);
        }
    }
};

const u32 TB = 256;
inline dim3 grid_for(u64 n) { u64 g = (n + TB - 1) / TB; return dim3((u32)(g < (1u << 22) ? (g ? g : 1) : (1u << 22))); }

// target of global window w: last t with win_off[t] <= w
__device__ __forceinline__ u32 target_of(const u64* win_off, u32 n_targets, u64 w) {
    u32 lo = 0, hi = n_targets;
    while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (win_off[mid] <= w) lo = mid; else hi = mid; }
    return lo;
}

// slot i = window i / s, sketch position i % s  ->  key = feature * P + (target % P), value = global window
__global__ void k_make_pairs(const u32* feat, u64 n_slots, u32 s, const u64* win_off, u32 n_targets, u32 P, u64* key, u32* val) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
        const u32 f = feat[i];
        const u64 w = i / s;
        if (f == 0xFFFFFFFFu) { key[i] = ~0ull; val[i] = 0; continue; }
        const u32 t = target_of(win_off, n_targets, w);
        key[i] = (u64)f * P + (t % P);
        val[i] = (u32)w;
    }
}
// head[i] = 1 where a new key group starts (sorted keys)
__global__ void k_heads(const u64* key, u64 n, u32* head) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}
__global__ void k_group_start(const u32* head, const u64* gid_excl, u64 n, u64* gstart) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) if (head[i]) gstart[gid_excl[i]] = i;
}
// keep the first max_locs entries of every (feature, rank) group
__global__ void k_keep(const u64* key, const u32* head, const u64* gid_excl, const u64* gstart, u64 n, u32 max_locs, u32* keep) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 g = gid_excl[i] + head[i] - 1;          // heads before me, plus my own, minus one
        keep[i] = (key[i] != ~0ull && (i - gstart[g]) < max_locs) ? 1u : 0u;
    }
}
__global__ void k_compact(const u64* key, const u32* val, const u32* keep, const u64* pos, u64 n, u32 P, u64* out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (keep[i]) out[pos[i]] = ((key[i] / P) << 32) | val[i];               // (feature << 32) | global window
}
// -remove-overpopulated-features: entries of features with more than `limit` locations are dropped
__global__ void k_keep_small(const u64* fw, const u32* head, const u64* kid_excl, const u64* first, u64 n, u64 n_keys, u64 limit, u32* keep) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 kx = kid_excl[i] + head[i] - 1;
        const u64 end = (kx + 1 < n_keys) ? first[kx + 1] : n;
        keep[i] = (end - first[kx] <= limit) ? 1u : 0u;
    }
}
__global__ void k_first_of_key(const u32* head, const u64* kid_excl, u64 n, u64* first) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) if (head[i]) first[kid_excl[i]] = i;
}
__global__ void k_compact_u64(const u64* in, const u32* keep, const u64* pos, u64 n, u64* out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) if (keep[i]) out[pos[i]] = in[i];
}
__global__ void k_feat_heads(const u64* fw, u64 n, u32* head) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        head[i] = (i == 0 || (fw[i] >> 32) != (fw[i - 1] >> 32)) ? 1u : 0u;
}
__global__ void k_emit(const u64* fw, const u32* head, const u64* kid_excl, u64 n, u64 n_keys, const u64* win_off, u32 n_targets,
                       u32* keys, u64* list_off, u64* locs) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 w = fw[i] & 0xFFFFFFFFull;
        const u32 t = target_of(win_off, n_targets, w);
        locs[i] = ((u64)t << 32) | (w - win_off[t]);
        if (head[i]) { keys[kid_excl[i]] = (u32)(fw[i] >> 32); list_off[kid_excl[i]] = i; }
        if (i == 0) list_off[n_keys] = n;
    }
}

// exclusive sum scan u32 -> u64 via rocPRIM (out has n entries; total returned through *total on the host)
int excl_scan(const u32* in, u64* out, u64 n, u64* total) {
    if (n == 0) { *total = 0; return MCQ_OK; }
    size_t tmp = 0;
    auto first = rocprim::make_transform_iterator(in, [] __device__(u32 x) { return (u64)x; });
    BCHK(rocprim::exclusive_scan(nullptr, tmp, first, out, (u64)0, n, rocprim::plus<u64>()));
    void* t = nullptr;
    BCHK(hipMalloc(&t, tmp ? tmp : 1));
    BCHK(rocprim::exclusive_scan(t, tmp, first, out, (u64)0, n, rocprim::plus<u64>()));
    u64 last_out = 0; u32 last_in = 0;
    BCHK(hipMemcpy(&last_out, out + n - 1, 8, hipMemcpyDeviceToHost));
    BCHK(hipMemcpy(&last_in, in + n - 1, 4, hipMemcpyDeviceToHost));
    BCHK(hipFree(t));
    *total = last_out + last_in;
    return MCQ_OK;
}
}  // namespace

struct mcq_table {            // device arrays, freed by mcq_table_free
    u64 n_keys, n_locs;
    u32* keys; u64* list_off; u64* locs;
    u64* win_off; u32 n_targets;
    int device;
};

extern "C" const char* mcq_build_last_error(void) { return g_berr.c_str(); }

extern "C" int mcq_table_free(mcq_table* t) {
    if (!t) return MCQ_OK;
    (void)hipSetDevice(t->device);
    (void)hipFree(t->keys); (void)hipFree(t->list_off); (void)hipFree(t->locs); (void)hipFree(t->win_off);
    delete t;
    return MCQ_OK;
}
extern "C" int mcq_table_info(const mcq_table* t, uint64_t* n_keys, uint64_t* n_locs, const uint32_t** keys,
                              const uint64_t** list_off, const uint64_t** locs, const uint64_t** win_off) {
    if (!t) return bfail(MCQ_E_ARG, "null argument");
    if (n_keys) *n_keys = t->n_keys;
    if (n_locs) *n_locs = t->n_locs;
    if (keys) *keys = t->keys;
    if (list_off) *list_off = t->list_off;
    if (locs) *locs = t->locs;
    if (win_off) *win_off = t->win_off;
    return MCQ_OK;
}

// ---- the pipeline behind both forms of the build -----------------------------------------------------------------
// (feature * P + rank, global window) pairs in (target, window) order -> stable sort by key, first max_locs of every
// (feature, rank) group, lists of a feature merged over the ranks into (target, window) order, optionally
// -remove-overpopulated-features; leaves fw[0..n_kept) = (feature << 32) | global window sorted, head[i] = 1 where a
// feature's list starts, kid[i] = exclusive count of heads, n_keys.  Takes over key / val (both from tmpbuf).
struct Lists { u64* fw = nullptr; u32* head = nullptr; u64* kid = nullptr; u64 n_kept = 0, n_keys = 0; };
template <class Phase>
static int sort_truncate(Scratch& tmpbuf, u64* key, u32* val, u64 n, u32 P, u32 max_locs, bool remove_overpop, Lists& L, Phase phase) {
    u64* key2 = nullptr; u32* val2 = nullptr;
    BCHK(tmpbuf.get(&key2, (n ? n : 1) * 8)); BCHK(tmpbuf.get(&val2, (n ? n : 1) * 4));
    if (n) {
        // keys are feature * P + rank < 2^32 * P: only these bits take part in the sort
        u32 key_bits = 32; while (key_bits < 64 && (((u64)P - 1) >> (key_bits - 32))) ++key_bits;
        size_t tmp = 0;
        BCHK(rocprim::radix_sort_pairs(nullptr, tmp, key, key2, val, val2, n, 0, key_bits));
        void* t = nullptr; BCHK(tmpbuf.get(&t, tmp ? tmp : 1));
        BCHK(rocprim::radix_sort_pairs(t, tmp, key, key2, val, val2, n, 0, key_bits));
        BCHK(hipDeviceSynchronize());
        tmpbuf.put(t);
    }
    tmpbuf.put(key); tmpbuf.put(val);
    phase("sort by (feature, rank)");

    // rank inside each group, keep the first max_locs
    u32 *head = nullptr, *keep = nullptr; u64 *gid = nullptr, *gstart = nullptr, *pos = nullptr;
    BCHK(tmpbuf.get(&head, (n ? n : 1) * 4)); BCHK(tmpbuf.get(&keep, (n ? n : 1) * 4));
    BCHK(tmpbuf.get(&gid, (n ? n : 1) * 8)); BCHK(tmpbuf.get(&pos, (n ? n : 1) * 8));
    u64 n_groups = 0, n_kept = 0;
    if (n) hipLaunchKernelGGL(k_heads, grid_for(n), dim3(TB), 0, 0, key2, n, head);
    MCHK(excl_scan(head, gid, n, &n_groups));
    BCHK(tmpbuf.get(&gstart, (n_groups ? n_groups : 1) * 8));
    if (n) {
        hipLaunchKernelGGL(k_group_start, grid_for(n), dim3(TB), 0, 0, head, gid, n, gstart);
        hipLaunchKernelGGL(k_keep, grid_for(n), dim3(TB), 0, 0, key2, head, gid, gstart, n, max_locs, keep);
    }
    MCHK(excl_scan(keep, pos, n, &n_kept));
    u64* fw = nullptr;
    BCHK(tmpbuf.get(&fw, (n_kept ? n_kept : 1) * 8));
    if (n) hipLaunchKernelGGL(k_compact, grid_for(n), dim3(TB), 0, 0, key2, val2, keep, pos, n, P, fw);
    BCHK(hipDeviceSynchronize());
    tmpbuf.put(key2); tmpbuf.put(val2); tmpbuf.put(keep); tmpbuf.put(gid); tmpbuf.put(gstart); tmpbuf.put(pos);
    phase("truncate to max_locs");

    // merge the virtual ranks' lists of a feature into (target, window) order
    if (P > 1 && n_kept) {
        u64* fw2 = nullptr; BCHK(tmpbuf.get(&fw2, n_kept * 8));
        size_t tmp = 0;
        BCHK(rocprim::radix_sort_keys(nullptr, tmp, fw, fw2, n_kept, 0, 64));
        void* t = nullptr; BCHK(tmpbuf.get(&t, tmp ? tmp : 1));
        BCHK(rocprim::radix_sort_keys(t, tmp, fw, fw2, n_kept, 0, 64));
        BCHK(hipDeviceSynchronize());
        tmpbuf.put(t); tmpbuf.put(fw);
        fw = fw2;
    }
    phase("merge ranks");
    u64* kid = nullptr; u64 n_keys = 0;
    tmpbuf.put(head);
    BCHK(tmpbuf.get(&head, (n_kept ? n_kept : 1) * 4)); BCHK(tmpbuf.get(&kid, (n_kept ? n_kept : 1) * 8));
    if (n_kept) hipLaunchKernelGGL(k_feat_heads, grid_for(n_kept), dim3(TB), 0, 0, fw, n_kept, head);
    MCHK(excl_scan(head, kid, n_kept, &n_keys));
    if (remove_overpop && n_kept && max_locs > 1) {
        u64 *first = nullptr, *pos2 = nullptr, *fw2 = nullptr; u32* keep2 = nullptr; u64 n2 = 0;
        BCHK(tmpbuf.get(&first, (n_keys ? n_keys : 1) * 8)); BCHK(tmpbuf.get(&pos2, n_kept * 8)); BCHK(tmpbuf.get(&keep2, n_kept * 4));
        hipLaunchKernelGGL(k_first_of_key, grid_for(n_kept), dim3(TB), 0, 0, head, kid, n_kept, first);
        hipLaunchKernelGGL(k_keep_small, grid_for(n_kept), dim3(TB), 0, 0, fw, head, kid, first, n_kept, n_keys, (u64)max_locs - 1, keep2);
        MCHK(excl_scan(keep2, pos2, n_kept, &n2));
        BCHK(tmpbuf.get(&fw2, (n2 ? n2 : 1) * 8));
        hipLaunchKernelGGL(k_compact_u64, grid_for(n_kept), dim3(TB), 0, 0, fw, keep2, pos2, n_kept, fw2);
        BCHK(hipDeviceSynchronize());
        tmpbuf.put(first); tmpbuf.put(pos2); tmpbuf.put(keep2); tmpbuf.put(fw);
        fw = fw2; n_kept = n2;
        if (n_kept) hipLaunchKernelGGL(k_feat_heads, grid_for(n_kept), dim3(TB), 0, 0, fw, n_kept, head);
        MCHK(excl_scan(head, kid, n_kept, &n_keys));
    }
    L.fw = fw; L.head = head; L.kid = kid; L.n_kept = n_kept; L.n_keys = n_keys;
    return MCQ_OK;
}

struct PhaseTrace {
    bool on; std::chrono::steady_clock::time_point last;
    PhaseTrace() : on(getenv("MCQ_BUILD_TRACE") != nullptr), last(std::chrono::steady_clock::now()) {}
    void operator()(const char* name) {        // MCQ_BUILD_TRACE=1: phase times on stderr (adds a device synchronisation per phase)
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mcq_build] %-28s %8.3f s\n", name, std::chrono::duration<double>(now - last).count());
        last = now;
    }
};

extern "C" int mcq_build_table(const mcq_build_desc* d, mcq_table** out) {
    if (!d || !out || !d->seq_off || (d->n_targets && !d->bases)) return bfail(MCQ_E_ARG, "null argument");
    if (d->n_targets < 1) return bfail(MCQ_E_ARG, "no targets");
    const u32 P = d->emulate_ranks ? d->emulate_ranks : 1;
    const u32 max_locs = d->max_locs ? d->max_locs : 254;
    BCHK(hipSetDevice(d->device));
    Scratch tmpbuf;
    BCHK(tmpbuf.init(d->device));
    const bool dev = (d->flags & MCQ_DEVICE_PTRS) != 0;
    const u32 nt = d->n_targets;
    PhaseTrace phase;

    // inputs on the device
    const char* bases = d->bases; const u64* seq_off = d->seq_off;
    char* t_bases = nullptr; u64* t_off = nullptr;
    if (!dev) {
        const u64 nb = d->seq_off[nt];
        BCHK(tmpbuf.get(&t_bases, nb ? nb : 1)); BCHK(tmpbuf.get(&t_off, (u64)(nt + 1) * 8));
        if (nb) BCHK(hipMemcpy(t_bases, d->bases, nb, hipMemcpyHostToDevice));
        BCHK(hipMemcpy(t_off, d->seq_off, (u64)(nt + 1) * 8, hipMemcpyHostToDevice));
        bases = t_bases; seq_off = t_off;
    }
    // a key-less handle carries the sketching parameters
    mcq_db_desc sd; std::memset(&sd, 0, sizeof(sd));
    sd.k = d->k; sd.sketch_size = d->sketch_size; sd.winlen = d->winlen; sd.winstride = d->winstride;
    const u64 zero = 0; sd.list_off = &zero; sd.n_shards = 1; sd.device = d->device;
    mcq_db* sk = nullptr;
    if (mcq_db_create(&sd, &sk) != MCQ_OK) return bfail(MCQ_E_ARG, std::string("sketch parameters: ") + mcq_last_error());

    mcq_table* T = new mcq_table();
    std::memset(T, 0, sizeof(*T));
    T->device = d->device; T->n_targets = nt;
    struct TableGuard { mcq_table*& t; ~TableGuard() { if (t) mcq_table_free(t); } } guard{T};     // error paths
    BCHK(hipMalloc(&T->win_off, (u64)(nt + 1) * 8));
    mcq_batch b; b.n_seqs = nt; b.bases = bases; b.seq_off = seq_off; b.paired = 0; b.flags = MCQ_DEVICE_PTRS;
    MCHK(mcq_count_windows(sk, &b, T->win_off, nullptr));
    u64 n_win = 0;
    BCHK(hipMemcpy(&n_win, T->win_off + nt, 8, hipMemcpyDeviceToHost));
    if (n_win >= (1ull << 32)) return bfail(MCQ_E_UNSUPPORTED, "more than 2^32 windows");
    const u32 s = d->sketch_size;
    const u64 n = n_win * s;
    u32 *feat = nullptr, *nfeat = nullptr;
    BCHK(tmpbuf.get(&feat, (n ? n : 1) * 4)); BCHK(tmpbuf.get(&nfeat, (n_win ? n_win : 1) * 4));
    MCHK(mcq_sketch(sk, &b, T->win_off, feat, nfeat, nullptr));
    BCHK(hipDeviceSynchronize());
    mcq_db_destroy(sk);
    tmpbuf.put(nfeat);
    phase("windows + sketches");

    // (feature * P + rank, global window) in window order
    u64* key = nullptr; u32* val = nullptr;
    BCHK(tmpbuf.get(&key, (n ? n : 1) * 8)); BCHK(tmpbuf.get(&val, (n ? n : 1) * 4));
    if (n) hipLaunchKernelGGL(k_make_pairs, grid_for(n), dim3(TB), 0, 0, feat, n, s, T->win_off, nt, P, key, val);
    tmpbuf.put(feat);
    phase("pairs");
    Lists L;
    MCHK(sort_truncate(tmpbuf, key, val, n, P, max_locs, (d->flags & MCQ_BUILD_REMOVE_OVERPOPULATED) != 0, L, phase));
    // keys, offsets, (target, window) locations
    T->n_keys = L.n_keys; T->n_locs = L.n_kept;
    BCHK(hipMalloc(&T->keys, (L.n_keys ? L.n_keys : 1) * 4)); BCHK(hipMalloc(&T->list_off, (L.n_keys + 1) * 8));
    BCHK(hipMalloc(&T->locs, (L.n_kept ? L.n_kept : 1) * 8));
    if (L.n_kept) hipLaunchKernelGGL(k_emit, grid_for(L.n_kept), dim3(TB), 0, 0, L.fw, L.head, L.kid, L.n_kept, L.n_keys, T->win_off, nt, T->keys, T->list_off, T->locs);
    else BCHK(hipMemset(T->list_off, 0, 8));
    BCHK(hipDeviceSynchronize());
    BCHK(hipGetLastError());
    tmpbuf.put(L.fw); tmpbuf.put(L.head); tmpbuf.put(L.kid);
    phase("emit keys / offsets / locations");
    if (t_bases) tmpbuf.put(t_bases);
    if (t_off) tmpbuf.put(t_off);
    *out = T;
    T = nullptr;
    return MCQ_OK;
}

// ---- the build in parts: tables whose one-piece temporaries do not fit (RefSeq scale) -------------------------------
// The features are cut into R ranges of h2(feature) (sub-ranges of the shard's range when the table is sharded: the same
// hash as mcq_owner, so a part never crosses a shard).  Pass p sketches every target again (the sketch is the cheap part:
// ~40 Gbp/s), in chunks of whole targets, keeps the features of range p -- flags, scan, scatter: (target, window) order is
// kept, which the stable sort relies on -- and runs the pipeline above on them; what is left of a pass is its part of the
// table in the form the query side stores: keys, list lengths, 32-bit global-window words.
struct mcq_parts {
    int device; u32 n_targets;
    u32 k, s, winlen, winstride;      // what the handle sketches QUERIES with
    u32 tgt_winstride;                // window stride of the targets (range width of the candidates); 0 = winstride
    u64 n_windows;
    u32* tgt_windows;         // device [n_targets]
    std::vector<mcq_db_part> parts;
    u64 n_keys, n_locs, bytes;
};
namespace {
__device__ __forceinline__ u32 tmh_dev(u32 x) {        // thomas_mueller_hash (src/hash_int.h:39-45), as mcq_owner
    x = ((x >> 16) ^ x) * 0x45d9f3bu; x = ((x >> 16) ^ x) * 0x45d9f3bu; return (x >> 16) ^ x;
}
__global__ void k_part_flags(const u32* feat, u64 n, u32 n_ranges, u32 range, u32* flag) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u32 f = feat[i];
        flag[i] = (f != 0xFFFFFFFFu && (u32)(((u64)tmh_dev(f) * n_ranges) >> 32) == range) ? 1u : 0u;
    }
}
// chunk slot i = window w0 + i / s; kept slots go to key / val at cursor + pos[i]
__global__ void k_part_scatter(const u32* feat, const u32* flag, const u64* pos, u64 n, u32 s, u64 w0, const u64* win_off, u32 n_targets, u32 P,
                               u64 cursor, u64 cap, u64* key, u32* val) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!flag[i]) continue;
        const u64 o = cursor + pos[i];
        if (o >= cap) continue;                       // (reported by the host from the counts)
        const u64 w = w0 + i / s;
        const u32 t = P > 1 ? target_of(win_off, n_targets, w) : 0u;
        key[o] = (u64)feat[i] * P + (P > 1 ? t % P : 0u);
        val[o] = (u32)w;
    }
}
__global__ void k_shift_off(const u64* in, u64 n, u64 base, u64* out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] - base;
}
__global__ void k_windows_of(const u64* win_off, u32 n_targets, u32* out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_targets) out[i] = (u32)(win_off[i + 1] - win_off[i]);
}
// part arrays from the sorted lists: the key of every list and the words themselves (lengths: k_part_first + k_part_len)
__global__ void k_emit_part(const u64* fw, const u32* head, const u64* kid, u64 n, u32* keys, u32* locs) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        locs[i] = (u32)fw[i];
        if (head[i]) keys[kid[i]] = (u32)(fw[i] >> 32);
    }
}
__global__ void k_part_first(const u32* head, const u64* kid, u64 n, u64* first) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) if (head[i]) first[kid[i]] = i;
}
__global__ void k_part_len(const u64* first, u64 n_keys, u64 n, u32* list_len) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_keys) list_len[i] = (u32)(((i + 1 < n_keys) ? first[i + 1] : n) - first[i]);
}
}  // namespace

extern "C" int mcq_parts_free(mcq_parts* p) {
    if (!p) return MCQ_OK;
    (void)hipSetDevice(p->device);
    for (auto& q : p->parts) { (void)hipFree((void*)q.keys); (void)hipFree((void*)q.list_len); (void)hipFree((void*)q.locs); }
    (void)hipFree(p->tgt_windows);
    delete p;
    return MCQ_OK;
}
extern "C" int mcq_parts_info(const mcq_parts* p, uint64_t* n_keys, uint64_t* n_locs, uint64_t* n_windows, uint32_t* n_parts, uint64_t* bytes) {
    if (!p) return bfail(MCQ_E_ARG, "null argument");
    if (n_keys) *n_keys = p->n_keys;
    if (n_locs) *n_locs = p->n_locs;
    if (n_windows) *n_windows = p->n_windows;
    if (n_parts) *n_parts = (u32)p->parts.size();
    if (bytes) *bytes = p->bytes;
    return MCQ_OK;
}

extern "C" int mcq_build_parts(const mcq_build_desc* d, mcq_parts** out) {
    if (!d || !out || !d->seq_off || (d->n_targets && !d->bases)) return bfail(MCQ_E_ARG, "null argument");
    if (d->n_targets < 1) return bfail(MCQ_E_ARG, "no targets");
    if (!(d->flags & MCQ_DEVICE_PTRS)) return bfail(MCQ_E_ARG, "mcq_build_parts takes device pointers (the sequences of such a table do not fit a staging copy)");
    const u32 P = d->emulate_ranks ? d->emulate_ranks : 1;
    const u32 max_locs = d->max_locs ? d->max_locs : 254;
    const u32 n_shards = d->n_shards ? d->n_shards : 1;
    if (d->shard_id >= n_shards) return bfail(MCQ_E_ARG, "shard_id >= n_shards");
    BCHK(hipSetDevice(d->device));
    Scratch tmpbuf;
    BCHK(tmpbuf.init(d->device));
    const u32 nt = d->n_targets, s = d->sketch_size;
    PhaseTrace phase;
    mcq_db_desc sd; std::memset(&sd, 0, sizeof(sd));
    sd.k = d->k; sd.sketch_size = d->sketch_size; sd.winlen = d->winlen; sd.winstride = d->winstride;
    const u64 zero = 0; sd.list_off = &zero; sd.n_shards = 1; sd.device = d->device;
    mcq_db* sk = nullptr;
    if (mcq_db_create(&sd, &sk) != MCQ_OK) return bfail(MCQ_E_ARG, std::string("sketch parameters: ") + mcq_last_error());
    struct SkGuard { mcq_db* h; ~SkGuard() { mcq_db_destroy(h); } } skg{sk};

    mcq_parts* R = new mcq_parts();
    R->device = d->device; R->n_targets = nt; R->k = d->k; R->s = s; R->winlen = d->winlen; R->winstride = d->winstride; R->tgt_winstride = 0;
    R->n_windows = 0; R->tgt_windows = nullptr; R->n_keys = R->n_locs = R->bytes = 0;
    struct PartsGuard { mcq_parts*& p; ~PartsGuard() { if (p) mcq_parts_free(p); } } guard{R};

    // windows of every target (device + host copy: the chunks are cut at target boundaries)
    u64* win_off = nullptr;
    BCHK(tmpbuf.get(&win_off, (u64)(nt + 1) * 8));
    mcq_batch b; b.n_seqs = nt; b.bases = d->bases; b.seq_off = d->seq_off; b.paired = 0; b.flags = MCQ_DEVICE_PTRS; b.n_bases = 0;
    MCHK(mcq_count_windows(sk, &b, win_off, nullptr));
    std::vector<u64> h_win(nt + 1);
    BCHK(hipMemcpy(h_win.data(), win_off, (u64)(nt + 1) * 8, hipMemcpyDeviceToHost));
    const u64 n_win = h_win[nt];
    if (n_win >= 0xFFFFFFFFull) return bfail(MCQ_E_UNSUPPORTED, "2^32 - 1 windows or more");
    R->n_windows = n_win;
    BCHK(hipMalloc(&R->tgt_windows, (u64)nt * 4));
    hipLaunchKernelGGL(k_windows_of, dim3((nt + TB - 1) / TB), dim3(TB), 0, 0, (const u64*)win_off, nt, R->tgt_windows);
    phase("windows");

    // chunks of whole targets for the sketch (MCQ_BUILD_CHUNK_WINDOWS: tuning knob / test hook)
    u64 chunk_win = 1ull << 24;
    if (const char* e = getenv("MCQ_BUILD_CHUNK_WINDOWS")) chunk_win = std::max<u64>(1, strtoull(e, nullptr, 10));
    std::vector<u32> cut{0};
    u64 largest = 0;
    for (u32 t = 0; t < nt;) {
        u32 e = t + 1;
        while (e < nt && h_win[e + 1] - h_win[t] <= chunk_win) ++e;
        largest = std::max(largest, h_win[e] - h_win[t]);
        cut.push_back(e); t = e;
    }
    // number of parts: what one pass keeps in flight (44 B per pair at its peak, plus the chunk's buffers) against a third of the
    // free memory (MCQ_BUILD_PARTS: tuning knob / test hook)
    size_t mem_free = 0, mem_total = 0;
    BCHK(hipMemGetInfo(&mem_free, &mem_total));
    const u64 n_slots_mine = (n_win * s + n_shards - 1) / n_shards;
    const u64 chunk_bytes = largest * s * 16 + largest * 4;
    u64 budget = mem_free / 3 > chunk_bytes + (1ull << 30) ? mem_free / 3 - chunk_bytes : (1ull << 30);
    u32 n_parts = (u32)std::max<u64>(1, (n_slots_mine * 44 + budget - 1) / budget);
    if (const char* e = getenv("MCQ_BUILD_PARTS")) n_parts = (u32)std::max<u64>(1, strtoull(e, nullptr, 10));
    if ((u64)n_parts * n_shards > (1u << 20)) return bfail(MCQ_E_UNSUPPORTED, "too many parts");
    const u32 n_ranges = n_parts * n_shards;
    // a range's share of the slots to start with: the hash spreads distinct features evenly (3 % and 2^20 slack; a range that holds
    // more -- repeats -- grows, see below)
    const u64 cap = (u64)((double)n_slots_mine / n_parts * 1.03) + (1ull << 20);

    u32 *feat = nullptr, *nfeat = nullptr, *flag = nullptr; u64 *pos = nullptr, *woff = nullptr;
    BCHK(tmpbuf.get(&feat, std::max<u64>(1, largest * s) * 4)); BCHK(tmpbuf.get(&nfeat, std::max<u64>(1, largest) * 4));
    BCHK(tmpbuf.get(&flag, std::max<u64>(1, largest * s) * 4)); BCHK(tmpbuf.get(&pos, std::max<u64>(1, largest * s) * 8));
    BCHK(tmpbuf.get(&woff, ((u64)nt + 1) * 8));
    for (u32 p = 0; p < n_parts; ++p) {
        const u32 range = d->shard_id * n_parts + p;
        u64* key = nullptr; u32* val = nullptr;
        u64 cap_p = cap;
        if (const char* e = getenv("MCQ_BUILD_PART_CAP")) cap_p = std::max<u64>(1, strtoull(e, nullptr, 10));      // test hook: start small, grow
        BCHK(tmpbuf.get(&key, cap_p * 8)); BCHK(tmpbuf.get(&val, cap_p * 4));
        u64 cursor = 0;
        for (size_t c = 0; c + 1 < cut.size(); ++c) {
            const u32 t0 = cut[c], t1 = cut[c + 1];
            const u64 w0 = h_win[t0], nw = h_win[t1] - w0, ns = nw * s;
            if (!nw) continue;
            hipLaunchKernelGGL(k_shift_off, dim3((t1 - t0 + 1 + TB - 1) / TB), dim3(TB), 0, 0, (const u64*)(win_off + t0), (u64)(t1 - t0) + 1, w0, woff);
            mcq_batch cb; cb.n_seqs = t1 - t0; cb.bases = d->bases; cb.seq_off = d->seq_off + t0; cb.paired = 0; cb.flags = MCQ_DEVICE_PTRS; cb.n_bases = 0;
            MCHK(mcq_sketch(sk, &cb, woff, feat, nfeat, nullptr));
            hipLaunchKernelGGL(k_part_flags, grid_for(ns), dim3(TB), 0, 0, (const u32*)feat, ns, n_ranges, range, flag);
            u64 kept = 0;
            MCHK(excl_scan(flag, pos, ns, &kept));
            if (cursor + kept > cap_p) {
                // more than the even share + 3 %: real data is not uniform -- features that repeat millions of times (low-complexity
                // minimisers, rRNA, IS copies) all land in ONE range before the 254-per-feature limit applies.  The pair arrays of
                // this part grow (half again, at least what is needed) instead of failing (until r04: MCQ_E_CAPACITY and the advice
                // to ask for more parts, which shrinks the cap as well)
                const u64 ncap = std::max<u64>(cursor + kept, cap_p + cap_p / 2);
                u64* key2 = nullptr; u32* val2 = nullptr;
                BCHK(tmpbuf.get(&key2, ncap * 8)); BCHK(tmpbuf.get(&val2, ncap * 4));
                if (cursor) { BCHK(hipMemcpyAsync(key2, key, cursor * 8, hipMemcpyDeviceToDevice, 0)); BCHK(hipMemcpyAsync(val2, val, cursor * 4, hipMemcpyDeviceToDevice, 0)); }
                BCHK(hipStreamSynchronize(0));
                tmpbuf.put(key); tmpbuf.put(val);
                key = key2; val = val2; cap_p = ncap;
            }
            hipLaunchKernelGGL(k_part_scatter, grid_for(ns), dim3(TB), 0, 0, (const u32*)feat, (const u32*)flag, (const u64*)pos, ns, s, w0,
                               (const u64*)win_off, nt, P, cursor, cap_p, key, val);
            cursor += kept;
        }
        BCHK(hipDeviceSynchronize());
        phase("  sketch + keep the part's features");
        Lists L;
        MCHK(sort_truncate(tmpbuf, key, val, cursor, P, max_locs, (d->flags & MCQ_BUILD_REMOVE_OVERPOPULATED) != 0, L, phase));
        mcq_db_part q; q.n_keys = L.n_keys; q.n_locs = L.n_kept; q.keys = nullptr; q.list_len = nullptr; q.locs = nullptr;
        u32 *pk = nullptr, *pl = nullptr, *pw = nullptr; u64* first = nullptr;
        BCHK(hipMalloc(&pk, (L.n_keys ? L.n_keys : 1) * 4)); q.keys = pk;
        R->parts.push_back(q);                                   // (owned by R from here: freed by the guard on an error)
        BCHK(hipMalloc(&pl, (L.n_keys ? L.n_keys : 1) * 4)); R->parts.back().list_len = pl;
        BCHK(hipMalloc(&pw, (L.n_kept ? L.n_kept : 1) * 4)); R->parts.back().locs = pw;
        BCHK(tmpbuf.get(&first, (L.n_keys ? L.n_keys : 1) * 8));
        if (L.n_kept) {
            hipLaunchKernelGGL(k_part_first, grid_for(L.n_kept), dim3(TB), 0, 0, (const u32*)L.head, (const u64*)L.kid, L.n_kept, first);
            hipLaunchKernelGGL(k_part_len, grid_for(L.n_keys), dim3(TB), 0, 0, (const u64*)first, L.n_keys, L.n_kept, pl);
            hipLaunchKernelGGL(k_emit_part, grid_for(L.n_kept), dim3(TB), 0, 0, (const u64*)L.fw, (const u32*)L.head, (const u64*)L.kid, L.n_kept, pk, pw);
        }
        BCHK(hipDeviceSynchronize());
        BCHK(hipGetLastError());
        tmpbuf.put(first); tmpbuf.put(L.fw); tmpbuf.put(L.head); tmpbuf.put(L.kid);
        R->n_keys += L.n_keys; R->n_locs += L.n_kept; R->bytes += L.n_keys * 8 + L.n_kept * 4;
        phase("  part emitted");
    }
    tmpbuf.put(feat); tmpbuf.put(nfeat); tmpbuf.put(flag); tmpbuf.put(pos); tmpbuf.put(woff); tmpbuf.put(win_off);
    *out = R;
    R = nullptr;
    return MCQ_OK;
}


// ---- parts from streamed (feature, target, window) triples: the reference's shard files without a host-side union --------------
// (include/mcq.h, mcq_parts_builder_*; the host side is mcq_refdb_open_meta + mcq_shard_stream_* of include/mcq_host.h.)
// A chunk of triples becomes (feature << 32 | global window) words on the device and is scattered to the buffer of its feature-hash
// range -- the ranges of mcq_build_parts, sub-ranges of this shard's range: foreign features are dropped --; finish() sorts
// every range (that is the merge of the reference's P per-rank lists of a feature into (target, window) order: each is
// already truncated to 254 per rank, src/sketch_database.h:1090-1092) and cuts it into keys / list lengths / words.
struct mcq_parts_builder {
    int device = 0; u32 nt = 0, n_ranges = 1, n_shards = 1, shard_id = 0;
    u32 k = 0, s = 0, winlen = 0, winstride = 0, tgt_winstride = 0;
    u32* tgt_windows = nullptr;       // device [nt] (moves into the parts)
    u32* gw_off = nullptr;            // device [nt + 1]
    u64 n_windows = 0;
    std::vector<u64*> buf; std::vector<u64> cnt, cap;
    u64** d_buf = nullptr;            // device copy of the buffer pointers
    unsigned long long* d_cur = nullptr;   // device write cursors [n_ranges]
    u32* d_hist = nullptr;            // device [n_ranges + 1]: chunk histogram, [n_ranges] = bad triples
    u32 *d_feat = nullptr, *d_tgt = nullptr, *d_win = nullptr; u64 stage_cap = 0;
    u64 n_added = 0, n_kept = 0;
};
namespace {
__global__ void k_gwoff(const u32* tgt_windows, u32 nt, u32* gw_off) {       // (one thread: nt is at most a few 1e5)
    if (blockIdx.x || threadIdx.x) return;
    u64 acc = 0;
    for (u32 t = 0; t < nt; ++t) { gw_off[t] = (u32)acc; acc += tgt_windows[t]; }
    gw_off[nt] = (u32)acc;
}
// range of every triple of a chunk (0xFFFFFFFF: another shard's feature) and the chunk's histogram
__global__ void k_triple_range(const u32* feat, const u32* tgt, const u32* win, u64 n, u32 n_ranges, u32 n_shards, u32 shard_id,
                               const u32* tgt_windows, u32 nt, u32* rng, u32* hist) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u32 f = feat[i];
        const u32 r = (u32)(((u64)tmh_dev(f) * ((u64)n_ranges * n_shards)) >> 32);
        u32 mine = (r / n_ranges == shard_id) ? r - shard_id * n_ranges : 0xFFFFFFFFu;
        if (f == 0xFFFFFFFFu || tgt[i] >= nt || win[i] >= tgt_windows[tgt[i]]) { mine = 0xFFFFFFFFu; atomicAdd(&hist[n_ranges], 1u); }
        rng[i] = mine;
        if (mine != 0xFFFFFFFFu) atomicAdd(&hist[mine], 1u);
    }
}
__global__ void k_triple_scatter(const u32* feat, const u32* tgt, const u32* win, const u32* rng, u64 n, const u32* gw_off,
                                 u64* const* bufs, unsigned long long* cur) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u32 lane = threadIdx.x & 63;
    for (u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x - lane; i0 < n; i0 += stride) {       // (whole waves iterate together)
        const u64 i = i0 + lane;
        const u32 r = i < n ? rng[i] : 0xFFFFFFFFu;
        unsigned long long todo = __ballot(r != 0xFFFFFFFFu);
        while (todo) {                                        // one atomic per distinct range among the wave's triples
            const u32 rr = __shfl(r, (int)__builtin_ctzll(todo), 64);
            const unsigned long long mm = __ballot(r == rr);
            unsigned long long base = 0;
            if (lane == (u32)__builtin_ctzll(mm)) base = atomicAdd(&cur[rr], (unsigned long long)__builtin_popcountll(mm));
            base = __shfl(base, (int)__builtin_ctzll(mm), 64);
            if (r == rr) bufs[rr][base + __builtin_popcountll(mm & ((1ull << lane) - 1))] = ((u64)feat[i] << 32) | (u64)(gw_off[tgt[i]] + win[i]);
            todo &= ~mm;
        }
    }
}
}  // namespace

extern "C" int mcq_parts_builder_free(mcq_parts_builder* b) {
    if (!b) return MCQ_OK;
    (void)hipSetDevice(b->device);
    for (u64* p : b->buf) (void)hipFree(p);
    (void)hipFree(b->tgt_windows); (void)hipFree(b->gw_off); (void)hipFree(b->d_buf); (void)hipFree(b->d_cur); (void)hipFree(b->d_hist);
    (void)hipFree(b->d_feat); (void)hipFree(b->d_tgt); (void)hipFree(b->d_win);
    delete b;
    return MCQ_OK;
}

extern "C" int mcq_parts_builder_create(const mcq_parts_builder_desc* d, mcq_parts_builder** out) {
    if (!d || !out || !d->tgt_windows || d->n_targets < 1) return bfail(MCQ_E_ARG, "null argument");
    const u32 n_shards = d->n_shards ? d->n_shards : 1;
    if (d->shard_id >= n_shards) return bfail(MCQ_E_ARG, "shard_id >= n_shards");
    BCHK(hipSetDevice(d->device));
    mcq_parts_builder* b = new mcq_parts_builder();
    struct Guard { mcq_parts_builder*& p; ~Guard() { if (p) mcq_parts_builder_free(p); } } guard{b};
    b->device = d->device; b->nt = d->n_targets; b->n_shards = n_shards; b->shard_id = d->shard_id;
    b->k = d->k; b->s = d->sketch_size; b->winlen = d->winlen; b->winstride = d->winstride; b->tgt_winstride = d->tgt_winstride;
    u64 total = 0;
    for (u32 t = 0; t < d->n_targets; ++t) total += d->tgt_windows[t];
    if (total >= 0xFFFFFFFFull) return bfail(MCQ_E_UNSUPPORTED, "2^32 - 1 windows or more");
    b->n_windows = total;
    BCHK(hipMalloc(&b->tgt_windows, (u64)d->n_targets * 4));
    BCHK(hipMemcpy(b->tgt_windows, d->tgt_windows, (u64)d->n_targets * 4, hipMemcpyHostToDevice));
    BCHK(hipMalloc(&b->gw_off, ((u64)d->n_targets + 1) * 4));
    hipLaunchKernelGGL(k_gwoff, dim3(1), dim3(1), 0, 0, (const u32*)b->tgt_windows, d->n_targets, b->gw_off);
    // ranges: what the sort of one range needs at its peak (~30 B per location: the words twice, heads, key ids) against a
    // quarter of the free memory, from the caller's estimate of the locations this shard will receive
    size_t mem_free = 0, mem_total = 0;
    BCHK(hipMemGetInfo(&mem_free, &mem_total));
    const u64 expect = d->expected_locations / n_shards + 1;
    u32 n_ranges = (u32)std::max<u64>(1, (expect * 30 + mem_free / 4 - 1) / (mem_free / 4 ? mem_free / 4 : 1));
    if (d->n_ranges) n_ranges = d->n_ranges;
    if (const char* e = getenv("MCQ_BUILD_PARTS")) n_ranges = (u32)std::max<u64>(1, strtoull(e, nullptr, 10));      // (test hook, as mcq_build_parts)
    if ((u64)n_ranges * n_shards > (1u << 20)) return bfail(MCQ_E_UNSUPPORTED, "too many ranges");
    b->n_ranges = n_ranges;
    b->buf.assign(n_ranges, nullptr); b->cnt.assign(n_ranges, 0); b->cap.assign(n_ranges, 0);
    BCHK(hipMalloc(&b->d_buf, (u64)n_ranges * 8));
    BCHK(hipMalloc(&b->d_cur, (u64)n_ranges * 8));
    BCHK(hipMalloc(&b->d_hist, ((u64)n_ranges + 1) * 4));
    BCHK(hipDeviceSynchronize());
    *out = b;
    b = nullptr;
    return MCQ_OK;
}

extern "C" int mcq_parts_builder_add(mcq_parts_builder* b, const uint32_t* feat, const uint32_t* tgt, const uint32_t* win, uint64_t n, uint32_t flags) {
    if (!b || (n && (!feat || !tgt || !win))) return bfail(MCQ_E_ARG, "null argument");
    if (!n) return MCQ_OK;
    if (n >= (1ull << 31)) return bfail(MCQ_E_ARG, "a chunk holds fewer than 2^31 triples");
    BCHK(hipSetDevice(b->device));
    const u32 *df = feat, *dt = tgt, *dw = win;
    if (!(flags & MCQ_DEVICE_PTRS)) {
        if (n > b->stage_cap) {
            (void)hipFree(b->d_feat); (void)hipFree(b->d_tgt); (void)hipFree(b->d_win); b->d_feat = b->d_tgt = b->d_win = nullptr; b->stage_cap = 0;
            BCHK(hipMalloc(&b->d_feat, n * 4)); BCHK(hipMalloc(&b->d_tgt, n * 4)); BCHK(hipMalloc(&b->d_win, n * 4));
            b->stage_cap = n;
        }
        BCHK(hipMemcpyAsync(b->d_feat, feat, n * 4, hipMemcpyHostToDevice, 0));
        BCHK(hipMemcpyAsync(b->d_tgt, tgt, n * 4, hipMemcpyHostToDevice, 0));
        BCHK(hipMemcpyAsync(b->d_win, win, n * 4, hipMemcpyHostToDevice, 0));
        df = b->d_feat; dt = b->d_tgt; dw = b->d_win;
    }
    u32* rng = nullptr;
    BCHK(hipMalloc(&rng, n * 4));
    struct Free { void* p; ~Free() { (void)hipFree(p); } } fr{rng};
    BCHK(hipMemsetAsync(b->d_hist, 0, ((u64)b->n_ranges + 1) * 4, 0));
    hipLaunchKernelGGL(k_triple_range, grid_for(n), dim3(TB), 0, 0, df, dt, dw, n, b->n_ranges, b->n_shards, b->shard_id,
                       (const u32*)b->tgt_windows, b->nt, rng, b->d_hist);
    std::vector<u32> hist(b->n_ranges + 1);
    BCHK(hipMemcpy(hist.data(), b->d_hist, hist.size() * 4, hipMemcpyDeviceToHost));
    if (hist[b->n_ranges]) return bfail(MCQ_E_ARG, std::to_string(hist[b->n_ranges]) + " triples name a target or a window the database does not have");
    // room in every range's buffer (grown by half when it runs out: a copy of what is there)
    bool moved = false;
    for (u32 r = 0; r < b->n_ranges; ++r) {
        const u64 need = b->cnt[r] + hist[r];
        if (need <= b->cap[r]) continue;
        const u64 ncap = std::max<u64>(need + need / 2, 1u << 20);
        u64* nb = nullptr;
        BCHK(hipMalloc(&nb, ncap * 8));
        if (b->cnt[r]) BCHK(hipMemcpy(nb, b->buf[r], b->cnt[r] * 8, hipMemcpyDeviceToDevice));
        (void)hipFree(b->buf[r]);
        b->buf[r] = nb; b->cap[r] = ncap; moved = true;
    }
    if (moved) BCHK(hipMemcpy(b->d_buf, b->buf.data(), (u64)b->n_ranges * 8, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(b->d_cur, b->cnt.data(), (u64)b->n_ranges * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_triple_scatter, grid_for(n), dim3(TB), 0, 0, df, dt, dw, (const u32*)rng, n, (const u32*)b->gw_off, (u64* const*)b->d_buf, b->d_cur);
    BCHK(hipDeviceSynchronize());
    BCHK(hipGetLastError());
    for (u32 r = 0; r < b->n_ranges; ++r) { b->cnt[r] += hist[r]; b->n_kept += hist[r]; }
    b->n_added += n;
    return MCQ_OK;
}

extern "C" int mcq_parts_builder_finish(mcq_parts_builder* b, mcq_parts** out) {
    if (!b || !out) return bfail(MCQ_E_ARG, "null argument");
    BCHK(hipSetDevice(b->device));
    Scratch tmpbuf;
    BCHK(tmpbuf.init(b->device));
    PhaseTrace phase;
    mcq_parts* R = new mcq_parts();
    R->device = b->device; R->n_targets = b->nt; R->k = b->k; R->s = b->s; R->winlen = b->winlen; R->winstride = b->winstride;
    R->tgt_winstride = b->tgt_winstride;
    R->n_windows = b->n_windows; R->tgt_windows = b->tgt_windows; b->tgt_windows = nullptr;
    R->n_keys = R->n_locs = R->bytes = 0;
    struct PartsGuard { mcq_parts*& p; ~PartsGuard() { if (p) mcq_parts_free(p); } } guard{R};
    (void)hipFree(b->d_feat); (void)hipFree(b->d_tgt); (void)hipFree(b->d_win); b->d_feat = b->d_tgt = b->d_win = nullptr;
    for (u32 r = 0; r < b->n_ranges; ++r) {
        const u64 n = b->cnt[r];
        mcq_db_part q; q.n_keys = 0; q.n_locs = n; q.keys = nullptr; q.list_len = nullptr; q.locs = nullptr;
        u64* fw = nullptr; u32* head = nullptr; u64 *kid = nullptr, *first = nullptr; u64 n_keys = 0;
        if (n) {
            BCHK(tmpbuf.get(&fw, n * 8));
            size_t tmp = 0;
            BCHK(rocprim::radix_sort_keys(nullptr, tmp, b->buf[r], fw, n, 0, 64));
            void* t = nullptr; BCHK(tmpbuf.get(&t, tmp ? tmp : 1));
            BCHK(rocprim::radix_sort_keys(t, tmp, b->buf[r], fw, n, 0, 64));
            BCHK(hipDeviceSynchronize());
            tmpbuf.put(t);
            (void)hipFree(b->buf[r]); b->buf[r] = nullptr; b->cap[r] = 0;
            BCHK(tmpbuf.get(&head, n * 4)); BCHK(tmpbuf.get(&kid, n * 8));
            hipLaunchKernelGGL(k_feat_heads, grid_for(n), dim3(TB), 0, 0, (const u64*)fw, n, head);
            MCHK(excl_scan(head, kid, n, &n_keys));
        }
        q.n_keys = n_keys;
        u32 *pk = nullptr, *pl = nullptr, *pw = nullptr;
        BCHK(hipMalloc(&pk, (n_keys ? n_keys : 1) * 4)); q.keys = pk;
        R->parts.push_back(q);                                   // (owned by R from here)
        BCHK(hipMalloc(&pl, (n_keys ? n_keys : 1) * 4)); R->parts.back().list_len = pl;
        BCHK(hipMalloc(&pw, (n ? n : 1) * 4)); R->parts.back().locs = pw;
        if (n) {
            BCHK(tmpbuf.get(&first, (n_keys ? n_keys : 1) * 8));
            hipLaunchKernelGGL(k_part_first, grid_for(n), dim3(TB), 0, 0, (const u32*)head, (const u64*)kid, n, first);
            hipLaunchKernelGGL(k_part_len, grid_for(n_keys), dim3(TB), 0, 0, (const u64*)first, n_keys, n, pl);
            hipLaunchKernelGGL(k_emit_part, grid_for(n), dim3(TB), 0, 0, (const u64*)fw, (const u32*)head, (const u64*)kid, n, pk, pw);
            BCHK(hipDeviceSynchronize());
            BCHK(hipGetLastError());
            tmpbuf.put(first); tmpbuf.put(fw); tmpbuf.put(head); tmpbuf.put(kid);
        }
        R->n_keys += n_keys; R->n_locs += n; R->bytes += n_keys * 8 + n * 4;
        phase("  range sorted + emitted");
    }
    *out = R;
    R = nullptr;
    mcq_parts_builder_free(b);
    return MCQ_OK;
}

extern "C" int mcq_db_from_parts(const mcq_parts* p, const uint32_t* tgt2tax, uint32_t n_shards, uint32_t shard_id, uint32_t flags, mcq_db** out) {
    if (!p || !tgt2tax || !out) return bfail(MCQ_E_ARG, "null argument");
    BCHK(hipSetDevice(p->device));
    u32* t2t = nullptr;
    if (!(flags & MCQ_DEVICE_PTRS)) {
        BCHK(hipMalloc(&t2t, (u64)p->n_targets * 4));
        BCHK(hipMemcpy(t2t, tgt2tax, (u64)p->n_targets * 4, hipMemcpyHostToDevice));
    }
    mcq_db_desc c; std::memset(&c, 0, sizeof(c));
    c.k = p->k; c.sketch_size = p->s; c.winlen = p->winlen; c.winstride = p->winstride; c.tgt_winstride = p->tgt_winstride ? p->tgt_winstride : p->winstride;
    c.n_targets = p->n_targets; c.tgt2tax = t2t ? t2t : tgt2tax; c.tgt_windows = p->tgt_windows;
    c.n_shards = n_shards ? n_shards : 1; c.shard_id = shard_id; c.device = p->device;
    c.flags = MCQ_DEVICE_PTRS | (flags & (MCQ_DB_SLOTS_16 | MCQ_DB_BUCKETS_64));
    const int rc = mcq_db_create_parts(&c, p->parts.data(), (u32)p->parts.size(), out);
    if (rc != MCQ_OK) g_berr = mcq_last_error();
    (void)hipFree(t2t);
    return rc;
}

extern "C" int mcq_db_build(const mcq_build_desc* d, mcq_db** out) {
    if (!d || !out) return bfail(MCQ_E_ARG, "null argument");
    // in parts when the one-piece temporaries (~60 B per feature slot of the sequences) would not leave room for the table --
    // or when asked to (MCQ_BUILD_PARTS: test hook); needs the sequences in device memory.
    // The two routes must not give different location WORDS for the same data when the words travel (a sharded table: the home
    // rank decodes what the owners send with its own tables): with n_shards > 1 the route is chosen by the size of the data alone
    // (not by what this rank happens to have free), and both routes give the global-window form over the true window counts of
    // the targets.  (A one-rank table keeps the faster bit fields where they fit, and the memory-driven choice.)
    const bool sharded = d->n_shards > 1;
    if ((d->flags & MCQ_DEVICE_PTRS) && !(d->flags & MCQ_DB_LOCS_64) && d->seq_off && d->n_targets) {
        bool in_parts = getenv("MCQ_BUILD_PARTS") != nullptr;
        if (!in_parts) {
            u64 ends[1] = {0};
            BCHK(hipSetDevice(d->device));
            BCHK(hipMemcpy(ends, d->seq_off + d->n_targets, 8, hipMemcpyDeviceToHost));
            const u64 slots = ends[0] / (d->winstride ? d->winstride : 1) * d->sketch_size;
            if (sharded) in_parts = slots * 60 > (96ull << 30);
            else {
                size_t mem_free = 0, mem_total = 0;
                BCHK(hipMemGetInfo(&mem_free, &mem_total));
                in_parts = slots * 60 > mem_free / 2;
            }
        }
        if (in_parts) {
            mcq_parts* parts = nullptr;
            MCHK(mcq_build_parts(d, &parts));
            const int rc = mcq_db_from_parts(parts, d->tgt2tax, d->n_shards, d->shard_id, d->flags & (MCQ_DEVICE_PTRS | MCQ_DB_SLOTS_16 | MCQ_DB_BUCKETS_64), out);
            mcq_parts_free(parts);
            return rc;
        }
    }
    mcq_table* T = nullptr;
    MCHK(mcq_build_table(d, &T));
    u32* t2t = nullptr;
    BCHK(hipMalloc(&t2t, (u64)d->n_targets * 4));
    BCHK(hipMemcpy(t2t, d->tgt2tax, (u64)d->n_targets * 4, (d->flags & MCQ_DEVICE_PTRS) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    mcq_db_desc c; std::memset(&c, 0, sizeof(c));
    c.k = d->k; c.sketch_size = d->sketch_size; c.winlen = d->winlen; c.winstride = d->winstride; c.tgt_winstride = d->winstride;
    c.n_targets = d->n_targets; c.n_keys = T->n_keys; c.n_locs = T->n_locs;
    c.keys = T->keys; c.list_off = T->list_off; c.locs = T->locs; c.tgt2tax = t2t;
    c.n_shards = d->n_shards ? d->n_shards : 1; c.shard_id = d->shard_id; c.flags = MCQ_DEVICE_PTRS | (d->flags & (MCQ_DB_LOCS_64 | MCQ_DB_LOCS_GW | MCQ_DB_SLOTS_16 | MCQ_DB_BUCKETS_64)); c.device = d->device;
    // the true window counts of the targets (what the parts route defines its words by), and for a sharded table that form
    u32* tw = nullptr;
    BCHK(hipMalloc(&tw, (u64)d->n_targets * 4));
    hipLaunchKernelGGL(k_windows_of, dim3((d->n_targets + TB - 1) / TB), dim3(TB), 0, 0, (const u64*)T->win_off, d->n_targets, tw);
    c.tgt_windows = tw;
    if (sharded && !(d->flags & MCQ_DB_LOCS_64)) c.flags |= MCQ_DB_LOCS_GW;
    int rc = mcq_db_create(&c, out);
    (void)hipFree(tw);
    if (rc != MCQ_OK) g_berr = mcq_last_error();
    (void)hipFree(t2t);
    mcq_table_free(T);
    return rc;
}
