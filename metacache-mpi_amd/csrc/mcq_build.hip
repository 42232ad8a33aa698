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
    // MCQ_BUILD_TRACE=1: phase times on stderr (diagnostic; adds a device synchronisation per phase)
    const bool trace = getenv("MCQ_BUILD_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mcq_build] %-28s %8.3f s\n", name, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };

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

    // (feature * P + rank, global window), stable sort: groups in (target, window) order
    u64 *key = nullptr, *key2 = nullptr; u32 *val = nullptr, *val2 = nullptr;
    BCHK(tmpbuf.get(&key, (n ? n : 1) * 8)); BCHK(tmpbuf.get(&key2, (n ? n : 1) * 8));
    BCHK(tmpbuf.get(&val, (n ? n : 1) * 4)); BCHK(tmpbuf.get(&val2, (n ? n : 1) * 4));
    phase("  (allocations)");
    if (n) hipLaunchKernelGGL(k_make_pairs, grid_for(n), dim3(TB), 0, 0, feat, n, s, T->win_off, nt, P, key, val);
    phase("  (k_make_pairs)");
    tmpbuf.put(feat);
    phase("pairs");
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
    phase("  (k_heads)");
    MCHK(excl_scan(head, gid, n, &n_groups));
    phase("  (scan heads)");
    BCHK(tmpbuf.get(&gstart, (n_groups ? n_groups : 1) * 8));
    if (n) {
        hipLaunchKernelGGL(k_group_start, grid_for(n), dim3(TB), 0, 0, head, gid, n, gstart);
        hipLaunchKernelGGL(k_keep, grid_for(n), dim3(TB), 0, 0, key2, head, gid, gstart, n, max_locs, keep);
    }
    phase("  (k_group_start, k_keep)");
    MCHK(excl_scan(keep, pos, n, &n_kept));
    phase("  (scan keep)");
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
    // keys, offsets, (target, window) locations
    u64* kid = nullptr; u64 n_keys = 0;
    tmpbuf.put(head);
    BCHK(tmpbuf.get(&head, (n_kept ? n_kept : 1) * 4)); BCHK(tmpbuf.get(&kid, (n_kept ? n_kept : 1) * 8));
    if (n_kept) hipLaunchKernelGGL(k_feat_heads, grid_for(n_kept), dim3(TB), 0, 0, fw, n_kept, head);
    MCHK(excl_scan(head, kid, n_kept, &n_keys));
    if ((d->flags & MCQ_BUILD_REMOVE_OVERPOPULATED) && n_kept && max_locs > 1) {
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
    T->n_keys = n_keys; T->n_locs = n_kept;
    BCHK(hipMalloc(&T->keys, (n_keys ? n_keys : 1) * 4)); BCHK(hipMalloc(&T->list_off, (n_keys + 1) * 8));
    BCHK(hipMalloc(&T->locs, (n_kept ? n_kept : 1) * 8));
    if (n_kept) hipLaunchKernelGGL(k_emit, grid_for(n_kept), dim3(TB), 0, 0, fw, head, kid, n_kept, n_keys, T->win_off, nt, T->keys, T->list_off, T->locs);
    else BCHK(hipMemset(T->list_off, 0, 8));
    BCHK(hipDeviceSynchronize());
    BCHK(hipGetLastError());
    tmpbuf.put(fw); tmpbuf.put(head); tmpbuf.put(kid);
    phase("emit keys / offsets / locations");
    if (t_bases) tmpbuf.put(t_bases);
    if (t_off) tmpbuf.put(t_off);
    *out = T;
    T = nullptr;
    return MCQ_OK;
}

extern "C" int mcq_db_build(const mcq_build_desc* d, mcq_db** out) {
    if (!d || !out) return bfail(MCQ_E_ARG, "null argument");
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
    int rc = mcq_db_create(&c, out);
    if (rc != MCQ_OK) g_berr = mcq_last_error();
    (void)hipFree(t2t);
    mcq_table_free(T);
    return rc;
}
