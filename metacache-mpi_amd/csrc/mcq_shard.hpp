// mcq_shard.hpp -- the feature-sharded multi-GPU query path behind the C ABI (mcq_shard_* in include/mcq.h).
// Included at the end of mcq_engine.hip (same translation unit: it launches the SH instantiations of the query kernels).
//
// One process per GPU, n_ranks of them.  The feature -> locations table is partitioned by hash range of h2(feature)
// (mcq_owner); every rank keeps its shard (mcq_db with n_shards = n_ranks, shard_id = rank) and its own reads.  It
// replaces the reference's per-rank lookups + MPI tree merge (src/querying.h:792-825, :867-1073; the reference
// partitions by target, tgt % P, src/sketch_database.h:540-542): all locations of a read reach its home rank, so
// per-target hit counts are the reference's, and the fold order of its P ranks is emulated on the home rank.
//
// Per batch, on every rank (all stages enqueued on streams; no host round trip in the padded mode):
//   S1  k_shard_sketch     home   sketch every window (rows 1-5), route each feature to the block of its owner
//                                 (per-wave chunk reservations), remember slot -> (owner, position)
//   X1  exchange           feature blocks to their owners
//   S2  k_shard_lookup     owner  probe (row 6) + copy the lists (row 7) into the requester's location block,
//                                 per feature the end of its list inside its 1024-feature tile, per tile its start
//   X2  exchange           list ends / tile starts and location blocks back
//   S3  k_query_wave<SH> / k_query_wave16<SH> / k_query_block<SH>   home: the fused kernels with the probe results
//                                 fetched from the exchange instead of sketch + probe (rows 8-11), lists gathered
//                                 straight out of the received blocks
// Exchange = ncclSend/ncclRecv groups over RCCL (xGMI), a device copy at n_ranks = 1, or a caller-supplied function
// (tests: host-staged gloo).  Blocks travel either at fixed, learned sizes with the counts inside them (padded mode)
// or at exact sizes after two small count exchanges through the host (exact mode, also the fallback after
// MCQ_E_CAPACITY).
#include <dlfcn.h>
#include <rccl/rccl.h>

#define MCQ_SHARD_HDR 4u                  // u32 words in front of a feature block: [0] = number of feature positions in it

// ------------------------------------------------------------------ S1: sketch + route to owners
// One wave per sequence (grid-stride), no workgroup barriers.  A wave keeps, per owner, a reservation of `chunk`
// positions in that owner's feature block (next / left in its LDS words; the block's header word is the global
// cursor, one atomic per chunk), hands the features of a window out of it -- one round per distinct owner among
// the window's <= s features: ballot of the lanes with that owner, ranks by mbcnt -- and records slot -> (owner,
// position).  What is left of a chunk when a window does not fit, or when the wave ends, is filled with MCQ_EMPTY:
// the owner's lookup skips such positions without a memory access.  chunk = the wave's expected share of an owner's
// block / 8, so the filler stays below an eighth of the block and the atomics on one cursor at ~8 per wave.
__global__ __launch_bounds__(256) void k_shard_sketch(DbDev db, BatchDev b, const u64* win_off, u32 n_ranks, u32 chunk,
                                                      u32* sendF, u32 capF, u32* slot_pos, u64 n_slots, unsigned long long* feat_cnt, u32* err) {
    __shared__ u32 s_sk[4][128];
    __shared__ u32 s_res[4][2 * MCQ_SHARD_MAX_RANKS];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32* sk = s_sk[wave];
    u32* res = s_res[wave];                          // [2 o] next position, [2 o + 1] positions left of owner o's chunk
    if (lane < 2 * n_ranks) res[lane] = 0;
    wave_sync();
    const u64 nwaves = (u64)gridDim.x * 4;
    const u64 blk = (u64)capF + MCQ_SHARD_HDR;
    unsigned long long st_feat = 0;
    for (u64 i = (u64)blockIdx.x * 4 + wave; i < b.n_seq; i += nwaves) {
        u64 o0, oe; seq_bounds(b.seq_off, b.ranges, i, o0, oe);
        const u32 n = (u32)(oe - o0);
        const u64 w0 = win_off[i];
        const u32 nw = (u32)(win_off[i + 1] - w0);
        for (u32 j = 0; j < nw; ++j) {
            u32 beg, wl;
            window_of32(n, db.winlen, db.winstride, db.magic_stride, j, beg, wl);
            const u32 m = wave_sketch_b(b, o0 + beg, wl, db.k, db.s, lane, sk, sk + 64);
            st_feat += m;
            const bool mine = lane < db.s;
            const u32 f = (mine && lane < m) ? sk[64 + lane] : MCQ_EMPTY;
            const u64 slot0 = (w0 + j) * db.s;
            if (slot0 + db.s > n_slots) {                    // more windows than the context was created for (max_bases)
                if (lane == 0) atomicOr(err, 16u);
                continue;
            }
            const u32 slot = (u32)slot0 + lane;
            const bool valid = f != MCQ_EMPTY;
            if (mine && !valid) slot_pos[slot] = MCQ_EMPTY;
            const u32 own = valid ? (u32)(((u64)tmh(f) * n_ranks) >> 32) : MCQ_EMPTY;
            u64 todo = __ballot(valid);
            while (todo) {                                   // wave-uniform: one round per distinct owner
                const u32 ol = bcast(own, (u32)__builtin_ctzll(todo));
                const u64 mm = __ballot(own == ol);
                const u32 c = (u32)__builtin_popcountll(mm);
                u32 next = res[2 * ol], left = res[2 * ol + 1];
                u32* F = sendF + (u64)ol * blk + MCQ_SHARD_HDR;
                if (left < c) {                              // the rest of the chunk (< 16 positions) stays unused
                    if (lane < left && next + lane < capF) F[next + lane] = MCQ_EMPTY;
                    u32 nb = 0;
                    if (lane == 0) nb = atomicAdd(&sendF[(u64)ol * blk], chunk);
                    next = bcast(nb, 0); left = chunk;
                }
                if (own == ol) {
                    const u32 pos = next + lane_rank(mm);
                    if (pos < capF) { F[pos] = f; slot_pos[slot] = (ol << MCQ_SHARD_POS_BITS) | pos; }
                    else { slot_pos[slot] = MCQ_EMPTY; atomicOr(err, 1u); }       // the owner's block is full
                }
                wave_sync();
                if (lane == 0) { res[2 * ol] = next + c; res[2 * ol + 1] = left - c; }
                wave_sync();
                todo &= ~mm;
            }
        }
    }
    for (u32 o = 0; o < n_ranks; ++o) {                      // what is left of the last chunks
        const u32 next = res[2 * o], left = res[2 * o + 1];
        u32* F = sendF + (u64)o * blk + MCQ_SHARD_HDR;
        for (u32 t = lane; t < left; t += 64) if (next + t < capF) F[next + t] = MCQ_EMPTY;
    }
    if (lane == 0 && st_feat) atomicAdd(feat_cnt, st_feat);
}
__global__ void k_shard_add_count(unsigned long long* dst, const unsigned long long* src) { atomicAdd(dst, *src); }
// S1 flags its errors in a word of its buffer set (it may run on the side stream while an earlier batch's flags are read and
// cleared); the query that consumes the set folds them into the word mcq_shard_sync reports
__global__ void k_shard_fold_err(u32* err, const u32* s1_err) { if (*s1_err) atomicOr(err, *s1_err); }

// ------------------------------------------------------------------ S2: owner side, lookup + gather
// grid = n_ranks x tiles per block; workgroup (p, t) serves features [t*1024, +1024) of the block that came from rank p:
// four probes in flight per thread, inclusive scan of the list lengths over the tile (feature order), one global
// atomic per tile for its room in p's location block, then every wave copies the lists of its 64-feature groups.
// R block of p: [0] = locations served to p so far (the cursor), [1] = features of p seen, then capT tile starts,
// then capF list ends.  (One wave per 256-feature tile -- no barriers, no LDS -- measured 5 % slower per batch.  r04: FOUR tiles per
// 1024-thread workgroup with one cursor atomic for all four -- on the theory that ~33 000 atomics on one address were what this
// stage waits for -- 1.51 against 1.35 ms alone and 3.36 against 2.74 ms per step overlapped: a 1024-thread workgroup needs 16 free
// wave slots on ONE CU at once and does not start beside the resident reduce grid.  Removed.)
template <class KeyT>
__global__ __launch_bounds__(256) void k_shard_lookup(DbDev db, u32 n_ranks, const u32* recvF, u32 capF, u32 capFx, u32 capT,
                                                      u32* sendR, KeyT* sendL, u64 capL, u32* err, int count_only) {
    __shared__ u32 s_wt[4][4];
    __shared__ u32 s_tbase;
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 p = blockIdx.x / capT, t = blockIdx.x - p * capT;
    const u64 fblk = (u64)capF + MCQ_SHARD_HDR, rblk = (u64)MCQ_SHARD_HDR + capT + capF;
    const u32* F = recvF + (u64)p * fblk;
    u32 cnt = F[0];
    if (cnt > capFx) { cnt = capFx; if (t == 0 && tid == 0) atomicOr(err, 2u); }      // more than a block carries: the sender's error
    u32* R = sendR + (u64)p * rblk;
    if (t == 0 && tid == 0) R[1] = cnt;
    if (t * MCQ_SHARD_TILE >= cnt) return;
    const KeyT* __restrict__ locs = static_cast<const KeyT*>(db.locs);
    u64 off[4]; u32 len[4], incl[4];
    {   // the first slot of all four probes in one round trip; a probe that has to walk on (load 0.25: one in eight) goes alone
        u32 f[4], idx[4]; uint4 sl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 i = t * MCQ_SHARD_TILE + k * 256 + tid;
            f[k] = i < cnt ? F[MCQ_SHARD_HDR + i] : MCQ_EMPTY;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            idx[k] = tmh(f[k]) & db.slot_mask;
            sl[k] = make_uint4(MCQ_EMPTY, 0, 0, 0);
            if (f[k] != MCQ_EMPTY) sl[k] = bucket_head(db, idx[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            while (sl[k].x != f[k] && sl[k].x != MCQ_EMPTY) { idx[k] = (idx[k] + 1) & db.slot_mask; sl[k] = bucket_head(db, idx[k]); }
            len[k] = 0; off[k] = 0;
            if (f[k] != MCQ_EMPTY && sl[k].x == f[k]) bucket_list(db, idx[k], sl[k], off[k], len[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        incl[k] = wave_incl_scan_dpp(len[k]);
        if (lane == 63) s_wt[k][wave] = incl[k];
    }
    __syncthreads();
    u32 before[4], total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int w = 0; w < 4; ++w) { if ((u32)w == wave) before[k] = total; total += s_wt[k][w]; }
    if (tid == 0) {
        const u32 tb = total ? atomicAdd(&R[0], total) : 0u;
        s_tbase = tb;
        R[MCQ_SHARD_HDR + t] = tb;
        if (!count_only && (u64)tb + total > capL) atomicOr(err, 4u);             // p's location block is full
    }
    if (count_only) return;                                   // (sizing pass of the exact mode: only the cursors are wanted)
    __syncthreads();
    const u32 tbase = s_tbase;
    const bool fits = (u64)tbase + total <= capL;
    KeyT* L = sendL + (u64)p * capL + tbase;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 i = t * MCQ_SHARD_TILE + k * 256 + tid;
        if (i < cnt) R[MCQ_SHARD_HDR + capT + i] = before[k] + incl[k];
        if (!fits) continue;
        // the 64 lists of this wave's group k, copied cooperatively (as k_lookup_gather)
        const u32 pos = incl[k] - len[k];
        const u32 Tg = bcast(incl[k], 63);
        for (u32 base = 0; base < Tg; base += 256) {
            KeyT v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = 0;
                if (base + (u32)(u * 64) < Tg) {
                    const u32 x = base + u * 64 + lane;
                    const u32 xx = x < Tg ? x : Tg - 1;
                    u32 lo = 0;
#pragma unroll
                    for (u32 step = 32; step > 0; step >>= 1) {
                        const u32 c = lo + step;
                        const u32 pc = __shfl(pos, (int)(c & 63), 64);
                        if (c < 64 && pc <= xx) lo = c;
                    }
                    const u32 pj = __shfl(pos, (int)lo, 64);
                    const u32 olo = __shfl((u32)off[k], (int)lo, 64), ohi = __shfl((u32)(off[k] >> 32), (int)lo, 64);
                    if (x < Tg) v[u] = locs[(((u64)ohi << 32) | olo) + (xx - pj)];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 x = base + u * 64 + lane;
                if (x < Tg) L[before[k] + x] = v[u];
            }
        }
    }
}

// zero the cursors / headers of the n_ranks blocks (stride in u32 words)
__global__ void k_shard_zero_headers(u32* blocks, u64 stride, u32 n_ranks) {
    const u32 i = threadIdx.x;
    if (i < n_ranks) for (u32 k = 0; k < MCQ_SHARD_HDR; ++k) blocks[(u64)i * stride + k] = 0;
}
// home side after X2, in both modes: the reads that lose data belong to THIS rank, so the loss is flagged here and not only
// on the owner that noticed it.  A peer that served more locations than the block that travelled carries (capLx: the padded
// size, or the buffer's capacity in the exact mode) has truncated this rank's lists (bit 8); positions this rank handed
// out beyond what a feature block carries (capFx; the header word is the cursor of its reservations) never reached their
// owner (bit 2: the same test the owner makes on the header it received).
__global__ void k_shard_check(const u32* recvR, u64 rblk, const u32* sendF, u64 fblk, u32 n_ranks, u64 capLx, u32 capFx, u32* err) {
    const u32 i = threadIdx.x;
    if (i >= n_ranks) return;
    if ((u64)recvR[(u64)i * rblk] > capLx) atomicOr(err, 8u);
    if (sendF[(u64)i * fblk] > capFx) atomicOr(err, 2u);
}

// ------------------------------------------------------------------ transports
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return MCQ_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);      // the one torch has loaded, if any; else ROCm's
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MCQ_E_UNSUPPORTED, std::string("librccl not found: ") + dlerror());
#define MCQ_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
    if (!g_rccl.field) return fail(MCQ_E_UNSUPPORTED, std::string("librccl lacks ") + name)
    MCQ_SYM(GetUniqueId, "ncclGetUniqueId"); MCQ_SYM(CommInitRank, "ncclCommInitRank"); MCQ_SYM(CommDestroy, "ncclCommDestroy");
    MCQ_SYM(Send, "ncclSend"); MCQ_SYM(Recv, "ncclRecv"); MCQ_SYM(GroupStart, "ncclGroupStart"); MCQ_SYM(GroupEnd, "ncclGroupEnd");
    MCQ_SYM(GetErrorString, "ncclGetErrorString");
#undef MCQ_SYM
    g_rccl.lib = h;
    return MCQ_OK;
}
#define NCCLCHK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
    return fail(MCQ_E_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); } while (0)

// ------------------------------------------------------------------ context
#define MCQ_SHARD_SETS 3      // S1 of batch j+1 may start as soon as the S3 of batch j-2 is done, i.e. under S3 of j-1 and j
struct ShardBuf {             // the device buffers of one batch in flight
    u32* sendF = nullptr;     // [n][HDR + capF]
    u32* slot_pos = nullptr;  // [max_slots]
    u64* win_off = nullptr;   // [max_seqs + 1]
    unsigned long long* feat_cnt = nullptr;   // features sketched (for the statistics)
    u32* recvF = nullptr; u32* sendR = nullptr; u32* recvR = nullptr;       // [n][HDR + capF], [n][HDR + capT + capF] x 2
    void* sendL = nullptr; void* recvL = nullptr;                           // [n][capL] locations
};
struct mcq_shard {
    const mcq_db* db; mcq_ws* ws;
    int device; u32 n, rank;
    u64 max_queries, max_seqs, max_slots;
    u32 capF, capT; u64 capL;             // block capacities (buffers)
    u32 capFx; u64 capLx;                 // what travels per peer in the padded mode (0 = not learned yet)
    u32 locb;
    ShardBuf sb[MCQ_SHARD_SETS]; int cur; // batches in flight, one buffer set each (round robin)
    u32* err; u32* err_host;              // device flag words ([0] reported by sync, [1 + k] = S1 of buffer set k), pinned copy of [0]
    u32* cnt_dev; u32* cnt_host;          // staging of the exact mode's count exchanges (2 x n u64 on the device, n u32 / u64 pinned)
    hipStream_t side, xs; hipEvent_t ev_prep[MCQ_SHARD_SETS], ev_done[MCQ_SHARD_SETS], ev_x[MCQ_SHARD_SETS], ev_in;
    bool prepared[MCQ_SHARD_SETS]; const void* prep_key[MCQ_SHARD_SETS][3];
    // transport
    ncclComm_t comm; bool have_comm;
    mcq_exchange_fn xfn; void* xuser;
    u64 last_nq;
    bool fmt_checked;                     // the ranks have compared what their location words mean (first exact batch)
    u64 seen_features, seen_locations;    // exact mode: the largest per-peer counts of the batch (sizes the padded mode's blocks)
    // accounting of the exchanges (mcq_shard_exchange_bytes): bytes handed to the transport for OTHER ranks, and of the rank's own blocks
    u64 xb_batches, xb_x1, xb_x2r, xb_x2l, xb_self;
    // stage timing (mcq_shard_timing): per buffer set the events around S1 (on the stream it ran on) and X1 / S2 / X2 (on xs)
    hipEvent_t tv[MCQ_SHARD_SETS][6]; bool tv_pending[MCQ_SHARD_SETS]; bool tv_s1[MCQ_SHARD_SETS]; bool tv_ready;
    u64 st_n_s1;
    double st_ms[4]; u64 st_n;
};
static u64 rblk_words(const mcq_shard* c) { return (u64)MCQ_SHARD_HDR + c->capT + c->capF; }
static u64 fblk_words(const mcq_shard* c) { return (u64)MCQ_SHARD_HDR + c->capF; }

// moves send_bytes[p] bytes from send_base + p*send_stride to rank p and receives recv_bytes[p] bytes from it at
// recv_base + p*recv_stride
static int shard_exchange(mcq_shard* c, const void* send_base, u64 send_stride, const u64* send_bytes,
                          void* recv_base, u64 recv_stride, const u64* recv_bytes, hipStream_t st, u64* account = nullptr) {
    const u32 n = c->n;
    if (account) for (u32 p = 0; p < n; ++p) { if (p == c->rank) c->xb_self += send_bytes[p]; else *account += send_bytes[p]; }
    if (c->xfn) {                         // caller's transport (host-staged in the tests): synchronous
        HIPCHK(hipStreamSynchronize(st));
        std::vector<u64> so(n), ro(n);
        for (u32 p = 0; p < n; ++p) { so[p] = p * send_stride; ro[p] = p * recv_stride; }
        if (c->xfn(c->xuser, send_base, so.data(), send_bytes, recv_base, ro.data(), recv_bytes, n, c->rank) != 0)
            return fail(MCQ_E_HIP, "the caller's exchange function failed");
        return MCQ_OK;
    }
    if (n == 1 && !c->have_comm) {
        if (send_bytes[0] != recv_bytes[0]) return fail(MCQ_E_ARG, "self exchange with different sizes");
        if (send_bytes[0]) HIPCHK(hipMemcpyAsync(recv_base, send_base, send_bytes[0], hipMemcpyDeviceToDevice, st));
        return MCQ_OK;
    }
    if (!c->have_comm) return fail(MCQ_E_ARG, "no transport: call mcq_shard_comm_rccl or mcq_shard_set_exchange first");
    NCCLCHK(g_rccl.GroupStart());
    for (u32 p = 0; p < n; ++p) {
        if (send_bytes[p]) NCCLCHK(g_rccl.Send((const char*)send_base + p * send_stride, send_bytes[p], ncclUint8, (int)p, c->comm, st));
        if (recv_bytes[p]) NCCLCHK(g_rccl.Recv((char*)recv_base + p * recv_stride, recv_bytes[p], ncclUint8, (int)p, c->comm, st));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return MCQ_OK;
}

extern "C" int mcq_shard_unique_id(void* out128) {
    if (!out128) return fail(MCQ_E_ARG, "null argument");
    int rc = rccl_load(); if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == MCQ_SHARD_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(out128, &id, sizeof(id));
    return MCQ_OK;
}

extern "C" int mcq_shard_destroy(mcq_shard* c) {
    if (!c) return MCQ_OK;
    (void)hipSetDevice(c->device);
    if (c->have_comm) (void)g_rccl.CommDestroy(c->comm);
    for (auto& b : c->sb) {
        (void)hipFree(b.sendF); (void)hipFree(b.slot_pos); (void)hipFree(b.win_off); (void)hipFree(b.feat_cnt);
        (void)hipFree(b.recvF); (void)hipFree(b.sendR); (void)hipFree(b.recvR); (void)hipFree(b.sendL); (void)hipFree(b.recvL);
    }
    (void)hipFree(c->err); (void)hipHostFree(c->err_host); (void)hipFree(c->cnt_dev); (void)hipHostFree(c->cnt_host);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->xs) (void)hipStreamDestroy(c->xs);
    for (auto e : c->ev_prep) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_done) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_x) if (e) (void)hipEventDestroy(e);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (auto& set : c->tv) for (auto e : set) if (e) (void)hipEventDestroy(e);
    (void)mcq_ws_destroy(c->ws);
    delete c;
    return MCQ_OK;
}

// location blocks of every buffer set for `capL` locations per peer (identical on every rank: the offset of peer p's block is
// p x capL on both sides of the exchange).  The receive side is allocated when a transport needs it: one rank without a
// transport reads its own send buffers in place.  Frees what was there: callers make sure no kernel still reads it.
static bool shard_alias(const mcq_shard* c);
static int shard_alloc_locations(mcq_shard* c, u64 capL) {
    if (capL >= (1ull << 32)) return fail(MCQ_E_UNSUPPORTED, "more than 2^32 locations per peer and batch: use smaller batches");
    for (auto& b : c->sb) {
        if (b.sendL) { (void)hipFree(b.sendL); b.sendL = nullptr; }
        if (b.recvL) { (void)hipFree(b.recvL); b.recvL = nullptr; }
    }
    c->capL = 0;
    for (auto& b : c->sb) {
        HIPCHK(hipMalloc(&b.sendL, std::max<u64>(1, (u64)c->n * capL * c->locb)));
        if (!shard_alias(c)) HIPCHK(hipMalloc(&b.recvL, std::max<u64>(1, (u64)c->n * capL * c->locb)));
    }
    c->capL = capL;
    return MCQ_OK;
}
// (a transport was attached after the blocks were allocated)
static int shard_ensure_recv(mcq_shard* c) {
    if (shard_alias(c) || !c->capL) return MCQ_OK;
    for (auto& b : c->sb) if (!b.recvL) HIPCHK(hipMalloc(&b.recvL, std::max<u64>(1, (u64)c->n * c->capL * c->locb)));
    return MCQ_OK;
}
__global__ void k_shard_clear_bits(u32* err, u32 bits) { atomicAnd(err, ~bits); }

extern "C" int mcq_shard_create(const mcq_db* shard, const mcq_shard_cfg* cfg, mcq_shard** out) {
    if (!shard || !cfg || !out) return fail(MCQ_E_ARG, "null argument");
    if (cfg->n_ranks < 1 || cfg->n_ranks > MCQ_SHARD_MAX_RANKS) return fail(MCQ_E_UNSUPPORTED, "n_ranks must be 1..32");
    if (cfg->rank >= cfg->n_ranks) return fail(MCQ_E_ARG, "rank >= n_ranks");
    if (shard->n_shards != cfg->n_ranks || shard->shard_id != cfg->rank)
        return fail(MCQ_E_ARG, "the handle must be shard `rank` of `n_ranks` (mcq_db_desc.n_shards / shard_id)");
    const u64 n = cfg->n_ranks;
    const u64 max_seqs = cfg->max_seqs ? cfg->max_seqs : cfg->max_queries;
    // a sequence of L bases has at most L / stride + 2 windows
    const u64 max_win = cfg->max_bases / shard->d.winstride + 2 * max_seqs;
    const u64 max_slots = max_win * shard->d.s;
    if (max_slots >= (1ull << 32)) return fail(MCQ_E_UNSUPPORTED, "more than 2^32 feature slots per batch");
    // block capacities: features hash uniformly over the owners, so twice the even share (plus slack for small batches)
    // holds any real batch; adversarial ones (all reads alike) are reported as MCQ_E_CAPACITY, never mis-answered
    u64 capF = cfg->max_features_per_peer ? cfg->max_features_per_peer : std::min<u64>(max_slots, 2 * max_slots / n + 65536);
    capF = (capF + MCQ_SHARD_TILE - 1) / MCQ_SHARD_TILE * MCQ_SHARD_TILE;
    if (capF >= (1ull << MCQ_SHARD_POS_BITS)) return fail(MCQ_E_UNSUPPORTED, "more than 2^27 features per peer and batch");
    // locations: the blocks are sized by the table, not by a constant: the first batch of a context runs in the exact mode, whose
    // owner-side lookup first only counts (k_shard_lookup, count_only), every rank learns the largest count any rank served,
    // and the blocks are allocated for that plus an eighth (shard_grow_locations).  Later exact batches grow them the same way
    // when they overflow; a padded batch that overflows is reported (MCQ_E_CAPACITY) and its repeat with MCQ_SHARD_EXACT grows them.
    // (Until r03 this was 16 locations per feature slot: a RefSeq-scale table delivers 28.)
    const u64 capL = cfg->max_locations_per_peer;
    if (capL >= (1ull << 32)) return fail(MCQ_E_UNSUPPORTED, "more than 2^32 locations per peer and batch");
    HIPCHK(hipSetDevice(shard->device));
    mcq_shard* c = new mcq_shard();
    memset(c, 0, sizeof(*c));
    c->db = shard; c->device = shard->device; c->n = (u32)n; c->rank = cfg->rank;
    c->max_queries = cfg->max_queries; c->max_seqs = max_seqs; c->max_slots = max_slots;
    c->capF = (u32)capF; c->capT = (u32)(capF / MCQ_SHARD_TILE); c->capL = capL;
    c->locb = shard->d.compact ? 4 : 8;
    int rc = mcq_ws_create(shard, cfg->max_queries, 1, cfg->max_locs_per_query, &c->ws);
    if (rc) { delete c; return rc; }
#define SCHK(expr) HIPCHK_OR(expr, (void)mcq_shard_destroy(c))
    for (auto& b : c->sb) {
        SCHK(hipMalloc(&b.sendF, n * fblk_words(c) * 4));
        SCHK(hipMalloc(&b.slot_pos, std::max<u64>(1, max_slots) * 4));
        SCHK(hipMalloc(&b.win_off, (max_seqs + 1) * 8));
        SCHK(hipMalloc(&b.feat_cnt, 8));
        SCHK(hipMalloc(&b.recvF, n * fblk_words(c) * 4));
        SCHK(hipMalloc(&b.sendR, n * rblk_words(c) * 4));
        SCHK(hipMalloc(&b.recvR, n * rblk_words(c) * 4));
    }
    if (capL) { const int rcg = shard_alloc_locations(c, capL); if (rcg) { (void)mcq_shard_destroy(c); return rcg; } }
    SCHK(hipMalloc(&c->err, 4 * (1 + MCQ_SHARD_SETS))); SCHK(hipMemset(c->err, 0, 4 * (1 + MCQ_SHARD_SETS)));
    SCHK(hipHostMalloc(&c->err_host, 4));
    SCHK(hipMalloc(&c->cnt_dev, 4 * n * 8)); SCHK(hipHostMalloc(&c->cnt_host, 4 * n * 8));
    SCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    SCHK(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
    for (auto& e : c->ev_x) SCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    SCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    for (auto& e : c->ev_prep) SCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_done) SCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& set : c->tv) for (auto& e : set) SCHK(hipEventCreate(&e));
    c->tv_ready = true;
#undef SCHK
    *out = c;
    return MCQ_OK;
}

extern "C" int mcq_shard_comm_rccl(mcq_shard* c, const void* unique_id) {
    if (!c || !unique_id) return fail(MCQ_E_ARG, "null argument");
    // one rank: nothing to connect, its own blocks are read in place -- unless the test hook asks for the real thing
    // (a communicator of one rank whose blocks go through ncclSend / ncclRecv to itself: the only way to run the RCCL
    // code path on a box with one GPU)
    if (c->n == 1 && !getenv("MCQ_SHARD_FORCE_RCCL")) return MCQ_OK;
    int rc = rccl_load(); if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId id; memcpy(&id, unique_id, sizeof(id));
    NCCLCHK(g_rccl.CommInitRank(&c->comm, (int)c->n, id, (int)c->rank));
    c->have_comm = true;
    return MCQ_OK;
}

extern "C" int mcq_shard_set_exchange(mcq_shard* c, mcq_exchange_fn fn, void* user) {
    if (!c) return fail(MCQ_E_ARG, "null argument");
    c->xfn = fn; c->xuser = user;
    return MCQ_OK;
}

extern "C" int mcq_shard_set_caps(mcq_shard* c, uint64_t features_per_peer, uint64_t locations_per_peer) {
    if (!c) return fail(MCQ_E_ARG, "null argument");
    if (features_per_peer > c->capF) return fail(MCQ_E_ARG, "beyond the feature blocks' capacity");
    if (locations_per_peer > c->capL) {                   // (collective like every call: all ranks pass the same sizes)
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        int rc = shard_alloc_locations(c, locations_per_peer); if (rc) return rc;
    }
    c->capFx = (u32)features_per_peer; c->capLx = locations_per_peer;
    return MCQ_OK;
}
extern "C" int mcq_shard_get_caps(const mcq_shard* c, uint64_t* features_per_peer, uint64_t* locations_per_peer) {
    if (!c) return fail(MCQ_E_ARG, "null argument");
    if (features_per_peer) *features_per_peer = c->capFx;
    if (locations_per_peer) *locations_per_peer = c->capLx;
    return MCQ_OK;
}

// stage times of the batch that last used buffer set k (its events are about to be recorded again)
static int shard_harvest(mcq_shard* c, int k) {
    if (!c->tv_pending[k]) return MCQ_OK;
    HIPCHK(hipEventSynchronize(c->tv[k][5]));
    const int pair[4][2] = {{0, 1}, {2, 3}, {3, 4}, {4, 5}};        // S1, X1, S2, X2
    for (int i = 0; i < 4; ++i) {
        if (i == 0 && !c->tv_s1[k]) continue;            // (this batch's S1 ran before timing was switched on)
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->tv[k][pair[i][0]], c->tv[k][pair[i][1]]));
        c->st_ms[i] += ms;
    }
    c->st_n += 1; c->st_n_s1 += c->tv_s1[k] ? 1 : 0;
    c->tv_pending[k] = false; c->tv_s1[k] = false;
    return MCQ_OK;
}

// S1 of a batch into buffer slot k, on stream st
static int shard_prepare(mcq_shard* c, int k, const mcq_batch* in, hipStream_t st) {
    ShardBuf& b = c->sb[k];
    if (c->ws->timing) { int rch = shard_harvest(c, k); if (rch) return rch; HIPCHK(hipEventRecord(c->tv[k][0], st)); }
    int rc = mcq_count_windows(c->db, in, b.win_off, st); if (rc) return rc;
    hipLaunchKernelGGL(k_shard_zero_headers, dim3(1), dim3(64), 0, st, b.sendF, fblk_words(c), c->n);
    HIPCHK(hipMemsetAsync(b.feat_cnt, 0, 8, st));
    HIPCHK(hipMemsetAsync(c->err + 1 + k, 0, 4, st));
    BatchDev bd; rc = batch_dev(in, in->bases, in->seq_off, bd); if (rc) return rc;
    if (in->n_seqs) {
        const u32 grid = (u32)std::min<u64>((in->n_seqs + 3) / 4, 256ull * 8);
        // a wave's reservation in an owner's block: an eighth of its expected share (one window's features at least)
        const u64 slots_guess = (in->n_seqs * 2) * c->db->d.s;       // ~2 windows per sequence; only a granularity, any value is correct
        const u32 chunk = (u32)std::min<u64>(4096, std::max<u64>(32, slots_guess / ((u64)c->n * grid * 4 * 8)));
        hipLaunchKernelGGL(k_shard_sketch, dim3(grid), dim3(256), 0, st, c->db->d, bd, (const u64*)b.win_off, c->n, chunk, b.sendF, c->capF,
                           b.slot_pos, c->max_slots, b.feat_cnt, c->err + 1 + k);
    }
    HIPCHK(hipGetLastError());
    if (c->ws->timing) { HIPCHK(hipEventRecord(c->tv[k][1], st)); c->tv_s1[k] = true; }
    return MCQ_OK;
}

// exact mode helper: my[p] (u64, device) -> theirs[p] on every rank, through the host
static int shard_exchange_counts(mcq_shard* c, const u64* mine_host, u64* theirs_host, hipStream_t st) {
    const u32 n = c->n;
    u64* d_send = reinterpret_cast<u64*>(c->cnt_dev);
    u64* d_recv = d_send + n;
    HIPCHK(hipMemcpyAsync(d_send, mine_host, n * 8, hipMemcpyHostToDevice, st));
    std::vector<u64> eight(n, 8);
    int rc = shard_exchange(c, d_send, 8, eight.data(), d_recv, 8, eight.data(), st); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(theirs_host, d_recv, n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MCQ_OK;
}

// X1 + S2 + X2 of the batch whose S1 sits in buffer set k, on stream st.  exact: blocks travel at their exact sizes (two
// count exchanges through the host, and the padded mode's block sizes are learned from them); else at the fixed sizes
// capFx / capLx with the counts inside, nothing on the host.
static bool shard_alias(const mcq_shard* c) { return c->n == 1 && !c->xfn && !c->have_comm; }
static int shard_owner_round(mcq_shard* c, int k, hipStream_t st, bool exact) {
    const u32 n = c->n;
    ShardBuf& b = c->sb[k];
    u32* const err = c->err + 1 + k;
    // a rank's own blocks never travel: with one rank the owner side reads the home side's send buffers in place
    const bool alias = shard_alias(c);
    u32* const recvF = alias ? b.sendF : b.recvF;
    u32* const recvR = alias ? b.sendR : b.recvR;
    std::vector<u64> sbytes(n), rbytes(n), cnt_mine(n), cnt_theirs(n);
    int rc;
    // ---- X1: feature blocks to their owners
    if (c->ws->timing) HIPCHK(hipEventRecord(c->tv[k][2], st));
    c->xb_batches += 1;
    u32 capFx = c->capFx; u64 capLx = c->capLx;
    if (exact && !c->fmt_checked) {
        // before the first location word travels: every rank must read the words the way their owner wrote them (same format, field
        // widths, window offsets of the targets, sketch parameters, rank count) -- shards built from different data or by different
        // routes would otherwise answer with wrong candidates in silence
        std::vector<u64> sig(n, c->db->fmt_sig), theirs(n);
        rc = shard_exchange_counts(c, sig.data(), theirs.data(), st); if (rc) return rc;
        for (u32 p = 0; p < n; ++p)
            if (theirs[p] != c->db->fmt_sig)
                return fail(MCQ_E_ARG, "rank " + std::to_string(p) + " holds its shard with other location words than rank " + std::to_string(c->rank) +
                            " (format, window counts of the targets or sketch parameters differ): build every shard from the same description");
        c->fmt_checked = true;
    }
    if (exact) {
        HIPCHK(hipMemcpy2DAsync(c->cnt_host, 4, b.sendF, fblk_words(c) * 4, 4, n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        u64 mx = 0;
        for (u32 p = 0; p < n; ++p) { cnt_mine[p] = std::min<u64>(c->cnt_host[p], c->capF); mx = std::max(mx, cnt_mine[p]); }
        rc = shard_exchange_counts(c, cnt_mine.data(), cnt_theirs.data(), st); if (rc) return rc;
        for (u32 p = 0; p < n; ++p) { sbytes[p] = (MCQ_SHARD_HDR + cnt_mine[p]) * 4; rbytes[p] = (MCQ_SHARD_HDR + cnt_theirs[p]) * 4; mx = std::max(mx, cnt_theirs[p]); }
        capFx = c->capF;                                  // the lookup accepts whatever the header says
        c->seen_features = mx;
    } else {
        for (u32 p = 0; p < n; ++p) sbytes[p] = rbytes[p] = ((u64)MCQ_SHARD_HDR + capFx) * 4;
    }
    if (alias) c->xb_self += sbytes[0];
    else { rc = shard_exchange(c, b.sendF, fblk_words(c) * 4, sbytes.data(), recvF, fblk_words(c) * 4, rbytes.data(), st, &c->xb_x1); if (rc) return rc; }
    if (c->ws->timing) HIPCHK(hipEventRecord(c->tv[k][3], st));

    // ---- S2: owner side
    if (!exact && (!c->capL || capLx > c->capL)) return fail(MCQ_E_ARG, "padded mode before any exact batch has sized the location blocks");
    { rc = shard_ensure_recv(c); if (rc) return rc; }
    auto lookup = [&](int count_only) {
        hipLaunchKernelGGL(k_shard_zero_headers, dim3(1), dim3(64), 0, st, b.sendR, rblk_words(c), n);
        const dim3 grid(n * c->capT);
        if (c->db->d.compact) hipLaunchKernelGGL(k_shard_lookup<u32>, grid, dim3(256), 0, st, c->db->d, n, (const u32*)recvF, c->capF, capFx, c->capT,
                                                 b.sendR, (u32*)b.sendL, c->capL, err, count_only);
        else                  hipLaunchKernelGGL(k_shard_lookup<u64>, grid, dim3(256), 0, st, c->db->d, n, (const u32*)recvF, c->capF, capFx, c->capT,
                                                 b.sendR, (u64*)b.sendL, c->capL, err, count_only);
    };
    lookup(exact && c->capL == 0);
    HIPCHK(hipGetLastError());
    std::vector<u64> served(n), coming(n);
    if (exact) {
        // locations served to each peer (cursor word of its R block), and the largest such count on ANY rank: when that is more
        // than a block holds -- always on a context's first batch, whose lookup only counted -- every rank allocates the same
        // larger blocks and the lookup runs again
        auto cursors = [&](u64& mx) -> int {
            HIPCHK(hipMemcpy2DAsync(c->cnt_host, 4, b.sendR, rblk_words(c) * 4, 4, n, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            mx = 0;
            for (u32 p = 0; p < n; ++p) { served[p] = c->cnt_host[p]; mx = std::max(mx, served[p]); }
            return MCQ_OK;
        };
        u64 lmax = 0;
        rc = cursors(lmax); if (rc) return rc;
        std::vector<u64> mine1(n, lmax), all1(n);
        rc = shard_exchange_counts(c, mine1.data(), all1.data(), st); if (rc) return rc;
        u64 gmax = 0;
        for (u32 p = 0; p < n; ++p) gmax = std::max(gmax, all1[p]);
        if (gmax > c->capL || c->capL == 0) {
            HIPCHK(hipDeviceSynchronize());                   // (other buffer sets may still be read by an earlier batch's reduce kernels)
            rc = shard_alloc_locations(c, gmax + gmax / 8 + (1u << 17)); if (rc) return rc;
            hipLaunchKernelGGL(k_shard_clear_bits, dim3(1), dim3(1), 0, st, err, 4u);
            lookup(0);
            HIPCHK(hipGetLastError());
            rc = cursors(lmax); if (rc) return rc;
        }
    }
    if (c->ws->timing) HIPCHK(hipEventRecord(c->tv[k][4], st));

    // ---- X2: list ends + tile starts, and the location blocks, back to the requesters
    void* const recvL = alias ? b.sendL : b.recvL;            // (after the lookup: the exact mode may have allocated larger blocks)
    if (exact) {
        // ends of the features each peer sent (counts known from X1), locations served to each peer (cursor word of its R block)
        for (u32 p = 0; p < n; ++p) { sbytes[p] = ((u64)MCQ_SHARD_HDR + c->capT + cnt_theirs[p]) * 4; rbytes[p] = ((u64)MCQ_SHARD_HDR + c->capT + cnt_mine[p]) * 4; }
        if (alias) c->xb_self += sbytes[0];
        else { rc = shard_exchange(c, b.sendR, rblk_words(c) * 4, sbytes.data(), recvR, rblk_words(c) * 4, rbytes.data(), st, &c->xb_x2r); if (rc) return rc; }
        u64 mx = 0;
        for (u32 p = 0; p < n; ++p) { served[p] = std::min<u64>(served[p], c->capL); mx = std::max(mx, served[p]); }
        rc = shard_exchange_counts(c, served.data(), coming.data(), st); if (rc) return rc;
        for (u32 p = 0; p < n; ++p) { sbytes[p] = served[p] * c->locb; rbytes[p] = coming[p] * c->locb; mx = std::max(mx, coming[p]); }
        if (alias) c->xb_self += sbytes[0];
        else { rc = shard_exchange(c, b.sendL, c->capL * c->locb, sbytes.data(), recvL, c->capL * c->locb, rbytes.data(), st, &c->xb_x2l); if (rc) return rc; }
        c->seen_locations = mx;
        // learn the padded mode's block sizes: the largest count any rank saw this batch, plus a sixteenth (counts of millions of
        // uniformly hashed features vary by well under a percent from batch to batch; a block that still overflows is reported,
        // and the exact mode answers it)
        std::vector<u64> mine2(n, (c->seen_features << 32) | std::min<u64>(c->seen_locations, 0xFFFFFFFFull)), all2(n);
        rc = shard_exchange_counts(c, mine2.data(), all2.data(), st); if (rc) return rc;
        u64 gf = 0, gl = 0;
        for (u32 p = 0; p < n; ++p) { gf = std::max<u64>(gf, all2[p] >> 32); gl = std::max<u64>(gl, all2[p] & 0xFFFFFFFFull); }
        c->capFx = std::max<u32>(c->capFx, (u32)std::min<u64>(c->capF, (gf + gf / 16 + 4096 + MCQ_SHARD_TILE - 1) / MCQ_SHARD_TILE * MCQ_SHARD_TILE));
        c->capLx = std::max<u64>(c->capLx, std::min<u64>(c->capL, gl + gl / 16 + 65536));
    } else {
        for (u32 p = 0; p < n; ++p) sbytes[p] = rbytes[p] = ((u64)MCQ_SHARD_HDR + c->capT + capFx) * 4;
        if (alias) c->xb_self += sbytes[0];
        else { rc = shard_exchange(c, b.sendR, rblk_words(c) * 4, sbytes.data(), recvR, rblk_words(c) * 4, rbytes.data(), st, &c->xb_x2r); if (rc) return rc; }
        for (u32 p = 0; p < n; ++p) sbytes[p] = rbytes[p] = capLx * c->locb;
        if (alias) c->xb_self += sbytes[0];
        else { rc = shard_exchange(c, b.sendL, c->capL * c->locb, sbytes.data(), recvL, c->capL * c->locb, rbytes.data(), st, &c->xb_x2l); if (rc) return rc; }
    }
    if (c->ws->timing) { HIPCHK(hipEventRecord(c->tv[k][5], st)); c->tv_pending[k] = true; }
    // what this rank lost, flagged on this rank (exact mode: the buffers' capacities; padded mode: the sizes that travelled)
    hipLaunchKernelGGL(k_shard_check, dim3(1), dim3(64), 0, st, (const u32*)recvR, rblk_words(c), (const u32*)b.sendF, fblk_words(c), n,
                       exact ? c->capL : capLx, exact ? c->capF : capFx, err);
    return MCQ_OK;
}

extern "C" int mcq_shard_query(mcq_shard* c, const mcq_batch* in, const mcq_query_opts* opt, mcq_result* out, void* stream,
                               uint32_t flags, const mcq_batch* next) {
    if (!c || !in || !opt || !out) return fail(MCQ_E_ARG, "null argument");
    if (!(in->flags & MCQ_DEVICE_PTRS) || !(out->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "the sharded path takes device pointers");
    if (next && !(next->flags & MCQ_DEVICE_PTRS)) return fail(MCQ_E_ARG, "the sharded path takes device pointers");
    OptDev od;
    int rc = make_opt(opt, od, c->db); if (rc) return rc;
    const u64 nq = in->paired ? in->n_seqs / 2 : in->n_seqs;
    if (nq > c->max_queries || in->n_seqs > c->max_seqs) return fail(MCQ_E_ARG, "batch larger than the context allows");
    const bool exact = (flags & MCQ_SHARD_EXACT) || c->capFx == 0;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    mcq_ws* ws = c->ws;

    // Three streams, two buffer sets.  Per call:  xs (the context's own): X1, S2, X2 of this batch;  the caller's stream:
    // S3 of this batch once xs is through;  side: S1 of the next batch.  The call only enqueues (padded mode), so the
    // caller's following call puts the next batch's exchanges and lookups on xs while this batch's S3 still runs: S3 and S1
    // (VALU-bound) share the CUs with the lookups (latency-bound) and with the copy engines / xGMI links of the exchanges.
    const int k = c->cur;
    ShardBuf& b = c->sb[k];
    HIPCHK(hipStreamWaitEvent(c->xs, c->ev_done[k], 0));  // this set's exchange buffers were last read by the S3 of MCQ_SHARD_SETS calls ago
    // ---- S1, unless the previous call already ran it for this batch on the side stream (its `next`)
    const bool was_prepared = c->prepared[k] && c->prep_key[k][0] == in->bases && c->prep_key[k][1] == in->seq_off &&
                              c->prep_key[k][2] == (const void*)(uintptr_t)in->n_seqs;
    if (c->prepared[k]) HIPCHK(hipStreamWaitEvent(c->xs, c->ev_prep[k], 0));   // (a prepared batch that is not this one drains, then is overwritten)
    c->prepared[k] = false;
    if (!was_prepared) {                                  // S1 here: behind whatever the caller has enqueued (its inputs)
        HIPCHK(hipEventRecord(c->ev_in, st));
        HIPCHK(hipStreamWaitEvent(c->xs, c->ev_in, 0));
        rc = shard_prepare(c, k, in, c->xs); if (rc) return rc;
    }
    // ---- X1, S2, X2
    rc = shard_owner_round(c, k, c->xs, exact); if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_x[k], c->xs));
    HIPCHK(hipStreamWaitEvent(st, c->ev_x[k], 0));

    // ---- S3: home side, the fused kernels fed from the exchange
    HIPCHK(hipMemsetAsync(ws->ctr, 0, MCQ_CTR_ZEROED, st));
    hipLaunchKernelGGL(k_shard_fold_err, dim3(1), dim3(1), 0, st, c->err, (const u32*)(c->err + 1 + k));
    const bool alias = shard_alias(c);
    u32* const recvR = alias ? b.sendR : b.recvR;
    ShardDev sh;
    sh.slot_pos = b.slot_pos; sh.ends = recvR + MCQ_SHARD_HDR + c->capT; sh.tile_base = recvR + MCQ_SHARD_HDR;
    sh.win_off = b.win_off; sh.ends_stride = (u32)rblk_words(c); sh.tile_stride = (u32)rblk_words(c); sh.capL = c->capL; sh.n_slots = c->max_slots;
    DbDev dbd = c->db->d; dbd.locs = alias ? b.sendL : b.recvL;
    BatchDev bd; rc = batch_dev(in, in->bases, in->seq_off, bd); if (rc) return rc;
    OutDev o; o.cands = (u32*)out->cands; o.ncand = out->n_cand;
    DebugDev dbg; memset(&dbg, 0, sizeof(dbg));
    rc = launch_query(c->db, ws, bd, od, o, st, force_bits(opt->flags), dbg, &sh, &dbd); if (rc) return rc;
    hipLaunchKernelGGL(k_shard_add_count, dim3(1), dim3(1), 0, st, &ws->ctr->n_features, (const unsigned long long*)b.feat_cnt);
    HIPCHK(hipEventRecord(c->ev_done[k], st));            // this buffer set may be overwritten
    c->last_nq = nq;
    const int k2 = (k + 1) % MCQ_SHARD_SETS;
    c->cur = k2;

    // ---- S1 of the next batch on the side stream, into the next buffer set (last read by an earlier S3: ev_done); the
    // caller guarantees that the next batch's inputs are resident and stay unchanged until its call
    if (next && next->n_seqs <= c->max_seqs) {
        HIPCHK(hipStreamWaitEvent(c->side, c->ev_done[k2], 0));
        rc = shard_prepare(c, k2, next, c->side); if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_prep[k2], c->side));
        c->prepared[k2] = true;
        c->prep_key[k2][0] = next->bases; c->prep_key[k2][1] = next->seq_off; c->prep_key[k2][2] = (const void*)(uintptr_t)next->n_seqs;
    }
    return MCQ_OK;
}

extern "C" int mcq_shard_sync(mcq_shard* c, void* stream, mcq_stats* stats) {
    if (!c) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(c->err_host, c->err, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(c->err, 0, 4, st));
    int rc = mcq_ws_sync(c->ws, stream, stats);
    if (stats) stats->n_queries = c->last_nq;
    const u32 e = *c->err_host;
    if (e) return fail(MCQ_E_CAPACITY, std::string("a block of the exchange was too small (") + ((e & 1) ? "features to an owner; " : "") +
                       ((e & 2) ? "features in the padded mode; " : "") + ((e & 4) ? "locations for a requester; " : "") +
                       ((e & 8) ? "locations in the padded mode; " : "") + ((e & 16) ? "more windows than max_bases allows; " : "") + "): results of the batches since the last sync are incomplete -- "
                       "repeat them with MCQ_SHARD_EXACT or larger capacities");
    return rc;
}

extern "C" int mcq_shard_timing(mcq_shard* c, int enable) {
    if (!c) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    for (int k = 0; k < MCQ_SHARD_SETS; ++k) { if (enable) { c->tv_pending[k] = false; c->tv_s1[k] = false; } else { int rc = shard_harvest(c, k); if (rc) return rc; } }
    if (enable) { for (auto& m : c->st_ms) m = 0; c->st_n = 0; c->st_n_s1 = 0; }
    return mcq_ws_timing(c->ws, enable);
}
// milliseconds, summed over the batches since timing was enabled, of S1 (window count + sketch + route, on whichever stream it
// ran), X1 (feature blocks out), S2 (owner-side lookup), X2 (list ends + locations back); n = batches harvested
extern "C" int mcq_shard_stage_times(mcq_shard* c, double* ms, uint64_t* n_batches) {
    if (!c) return fail(MCQ_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    for (int k = 0; k < MCQ_SHARD_SETS; ++k) { int rc = shard_harvest(c, k); if (rc) return rc; }
    if (ms) for (int i = 0; i < 4; ++i) ms[i] = c->st_ms[i];
    if (ms && c->st_n_s1 && c->st_n_s1 != c->st_n) ms[0] *= (double)c->st_n / (double)c->st_n_s1;      // (S1 of the first batches ran untimed)
    if (n_batches) *n_batches = c->st_n;
    return MCQ_OK;
}
// out[0] batches, [1] X1 bytes handed to the transport for other ranks, [2] X2 list ends + tile starts, [3] X2 locations,
// [4] bytes of this rank's own blocks (never travel at n_ranks = 1; over RCCL they are a local copy), [5] ranks of the RCCL
// communicator (0 = no RCCL transport), [6] / [7] the padded mode's block sizes in features / locations per peer
extern "C" int mcq_shard_exchange_bytes(const mcq_shard* c, uint64_t* out) {
    if (!c || !out) return fail(MCQ_E_ARG, "null argument");
    out[0] = c->xb_batches; out[1] = c->xb_x1; out[2] = c->xb_x2r; out[3] = c->xb_x2l; out[4] = c->xb_self;
    out[5] = c->have_comm ? c->n : 0; out[6] = c->capFx; out[7] = c->capLx;
    return MCQ_OK;
}
extern "C" int mcq_shard_kernel_times(mcq_shard* c, double* ms, uint64_t* n_batches) { return c ? mcq_ws_kernel_times(c->ws, ms, n_batches) : fail(MCQ_E_ARG, "null argument"); }
