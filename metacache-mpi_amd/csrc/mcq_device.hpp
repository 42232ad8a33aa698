// mcq_device.hpp -- device-side building blocks of the query path (gfx950, wave64).
//
// Everything here is integer work (hashing, compares, shuffles); there is no MFMA on
// this path.  The unit of parallelism is the wavefront: one wave sketches one window,
// and one wave (or one 1024-thread workgroup for oversized queries) owns one query
// from the table probes to the final candidate list, so the match list never leaves
// LDS on the common path.
//
// Reference behaviour restated (file:line relative to the reference root):
//   tmh / revcomp / canonical        src/hash_int.h:39-45, src/dna_encoding.h:113-121, :187-197
//   window split                     src/dna_encoding.h:259-276
//   k-mer extraction + ambiguity     src/dna_encoding.h:303-348, :457-466
//   unique min-s sketch              src/hash_dna.h:113-152
//   table lookup                     src/hash_multimap.h:1033-1047 (key -> list only)
//   accumulate + sort                src/sketch_database.h:804-823, src/querying.h:88-106
//   contiguous window ranges         src/candidates.h:118-180
//   bounded top list insert          src/candidates.h:236-285
//   P-rank tree fold                 src/querying.h:867-1073
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#define MCQ_EMPTY 0xFFFFFFFFu
#ifndef MCQ_BLOCK_LONG_FIRST
#define MCQ_BLOCK_LONG_FIRST 16384u     // tuning knob: queries this long and longer are the workgroup kernels' first pass; 0 = one pass (measured on ONT-like reads, mean 8 kb:
                                       // one pass 2.04, 8192: 1.92, 16384: 1.80 ms per 16 384 reads)
#endif
#define MCQ_Q_UNPROBED 0x80000000u     // back-queue entry of the direct mode: the second wave stage sketches and probes it itself (queries are < 2^31)
#define MCQ_MAX_FOLD 64
#define MCQ_BIGLIST_MAX 1024u        // 64 virtual ranks x 16 candidates

namespace mcq {

// ------------------------------------------------------------------ structures
struct DbDev {
    const uint4* slots;      // buckets of 64 B (4 uint4: {key, len, 14 words}) or 16 B ({key, len, offset lo, hi}); key == MCQ_EMPTY:
                             // unused.  In a 64-B bucket a list of up to 14 compact (7 wide) locations sits in the bucket itself;
                             // any other list behind the buckets: words 2,3 = its offset there.  The layout is chosen per table
                             // (mcq_db_create): short lists -> 64-B buckets, long lists / large tables -> 16-B slots
    u32 slot_mask;           // nslots - 1 (power of two)
    u32 bsh;                 // log2(uint4 per bucket): 2 (64-B buckets) or 0 (16-B slots)
    const void* locs;        // base of all lists = the bucket array itself (the long lists follow it in the same allocation):
                             // u64 (tgt<<32)|win, or a 32-bit word when `compact`: (tgt<<wb)|win, or (gw) the global window
                             // index gw_off[tgt] + win
    u32 wb;                  // window-id bits inside a location word (32 for u64 locations; unused with gw)
    u32 compact;             // 32-bit location words
    const u32* tgt2tax;
    u32 n_targets;
    u32 k, s, winlen, winstride, tgt_winstride;
    u32 magic_stride, magic_tgt_stride;   // floor(2^32 / stride): udiv_magic
};
// 32-bit location = global window index: for tables whose (target, window) space does not fit 32 bits as two fields (RefSeq
// scale: >= 2^15 sequences, chromosomes of >= 2^17 windows).  A kernel argument of its own, the LAST one, read by the GW
// instantiations only: inside DbDev it moved every later argument of the hot kernel, and the different SGPR allocation
// that followed cost it 2 % (+270 spill reloads).
struct GwDev {
    const u32* off;          // [n_targets + 1] first global window of every target
    const u32* blk;          // [(n_windows >> shift) + 2] pairs: target that holds window b << shift, its first window (LocGW)
    u32 shift;               // blk has one entry per 2^shift windows
    u32 on;
};

// ---- location formats -----------------------------------------------------------------------------------------------
// Everything behind the gather works on location words that sort like (target, window) and needs of a word k only
//   tbeg(k)  the smallest word of k's target (so: same target <=> prev >= tbeg(k) for prev <= k; window = k - tbeg(k))
//   tgt(k)   the target id (for the run heads only: taxon key and virtual rank).
// LocShift: two bit fields, (tgt << wb) | win, 32- or 64-bit words -- both are shifts.
// LocGW:    the global window index of the build (first window of the target + window): one 32-bit word for any table
//           of fewer than 2^32 - 1 windows (485 Gbp at the default stride), whatever the number of targets.  The target
//           comes from two small tables: gw_blk[k >> shift] brackets it (a block of 2^shift windows rarely holds more
//           than one target's start: then a short binary search over gw_off), gw_off[t] is tbeg.  Both tables are sized
//           to stay in L2 (<= 1 MB + 4 B per target); a lookup is two dependent L2 loads per DISTINCT location of a read.
template <class KeyT>
struct LocShift {
    static constexpr bool lookup = false;      // tgt / tbeg are arithmetic on the word
    u32 wb;
    __device__ __forceinline__ KeyT tbeg(KeyT k) const { return k & ~((((KeyT)1) << wb) - 1); }
    __device__ __forceinline__ u32 tgt(KeyT k) const { return (u32)(k >> wb); }
    __device__ __forceinline__ void locate(KeyT k, u32& t, KeyT& tb) const { t = tgt(k); tb = tbeg(k); }
};
struct LocGW {
    static constexpr bool lookup = true;       // tgt / tbeg cost memory accesses: callers keep what they looked up
    const u32* __restrict__ off; const uint2* __restrict__ blk; u32 shift;
    // blk[b] = (last target t with off[t] <= b << shift, off[t]): the target of k lies between blk[k >> shift] and the next entry,
    // and is one of those two unless two or more targets start inside the block -- ONE round trip to memory then (two loads side by
    // side) where a table of target numbers alone took two, three with the search (r04: the lists of the two-class tail wait for this)
    __device__ __forceinline__ void locate(u32 k, u32& t, u32& tb) const {
        const u32 b = k >> shift;
        const uint2 a = blk[b], c = blk[b + 1];
        u32 lo = a.x, hi = c.x, o = a.y;
        if (c.y <= k) { lo = hi; o = c.y; }              // (also when both entries name the same target)
        else {
            --hi;                                        // c.y > k >= a.y: hi > lo
            while (lo < hi) {                            // several targets start inside the block
                const u32 mid = (lo + hi + 1) >> 1;
                const u32 om = off[mid];
                if (om <= k) { lo = mid; o = om; } else hi = mid - 1;
            }
        }
        t = lo; tb = o;
    }
    __device__ __forceinline__ u32 tbeg(u32 k) const { u32 t, tb; locate(k, t, tb); return tb; }
    __device__ __forceinline__ u32 tgt(u32 k) const { u32 t, tb; locate(k, t, tb); return t; }
};
template <class KeyT, bool GW> struct LocOf { typedef LocShift<KeyT> type; };
template <> struct LocOf<u32, true> { typedef LocGW type; };
template <class KeyT, bool GW>
__device__ __forceinline__ typename LocOf<KeyT, GW>::type loc_format(const DbDev& db, const GwDev& g) {
    if constexpr (GW) { LocGW f; f.off = g.off; f.blk = reinterpret_cast<const uint2*>(g.blk); f.shift = g.shift; return f; }
    else { LocShift<KeyT> f; f.wb = db.wb; return f; }
}
// smallest word of a window range of `numWindows` windows that ends at k (never below the target's first word)
template <class KeyT>
__device__ __forceinline__ KeyT range_low(KeyT k, KeyT tb, u32 numWindows) {
    return (k - tb >= (KeyT)numWindows) ? k - (KeyT)(numWindows - 1) : tb;
}

struct BatchDev {
    const char* bases;       // ASCII, or (packed) the u32 words of MCQ_BATCH_PACKED
    const u64* seq_off;      // [n_seq+1] back-to-back, or [2*n_seq] (begin,end) pairs when `ranges`; always in bases
    u64 n_seq;
    u64 nq;
    u32 paired;
    u32 ranges;
    u32 packed;              // MCQ_BATCH_PACKED: 2-bit codes (16 per word, first base in the top bits) + ambiguity bits (32 per word)
    u32 amb_off;             // first word of the ambiguity bits
    u32 last_word;           // index of the (zero) pad word behind the 2-bit codes: loads are clamped to it
    u32 amb_last;            // the same for the ambiguity plane (relative to amb_off)
};
// byte range [beg,end) of sequence a of the batch
__device__ __forceinline__ void seq_bounds(const u64* seq_off, u32 ranges, u64 a, u64& beg, u64& end) {
    if (ranges) { beg = seq_off[2 * a]; end = seq_off[2 * a + 1]; }
    else { beg = seq_off[a]; end = seq_off[a + 1]; }
}

struct OptDev {
    u32 max_cand;            // M
    u32 P;                   // emulate_ranks
    u32 seg;                 // lanes per virtual-rank list = 64 / pow2ceil(P); with `big`: entries per list = M
    u32 big;                 // pow2ceil(P) x M > 64: the P lists do not fit one wave's lanes -- every query takes the
                             // workgroup kernel, which keeps them in LDS (MCQ_BIGLIST_MAX entries)
    u32 quirk_seq_drop;
    u32 hooks;               // 8 = no two-class tail; staged reduce kernel: 1 = no de-duplicating pass, 2 = no second wave stage, 4 = workgroup kernel only
    u64 insert_size_max;
    u32 n_fold;              // fold schedule: (snd -> rcv) in the reference's order
    u32 n_levels;            // rounds of the tree; the edges of one round touch disjoint ranks
    unsigned char level_end[8];          // edges [level_end[l-1], level_end[l]) belong to round l
    unsigned char fold_snd[MCQ_MAX_FOLD];
    unsigned char fold_rcv[MCQ_MAX_FOLD];
    u64 tc_limit;            // workgroup kernels: queries shorter than this belong to the two-class kernel (0 = there is none)
    u32 lin;                 // P > 1 as ONE selection in the order (hits, rank, position) instead of P lists and the tree fold
                             // (topk_lin_write; seg = 64, big = 0, no fold schedule); not with MCQ_QUIRK_SEQ_DROP on a table
                             // that has sequence-level taxa
    u32 keep;                // lin: the ranks the tree routes to rank 0 = [0, keep), keep = 2^floor(log2 P)
};

struct OutDev {
    u32* cands;              // nq * M * 4
    u32* ncand;              // nq
};

struct CountersDev {         // one block of u64/u32 words, zeroed per call
    unsigned long long n_features, n_hit_features, n_locations, n_cands;
    u32 ovf_count;           // queries queued for the block-per-query path
    u32 err_count;           // queries that exceeded the block path's capacity
    u32 ovf_mid_count;       // queued from the back of the same array: <= 64 features, 513..1024 locations (k_query_wave16)
    u32 n_ovf;               // queries queued (the two counts above are reserved slots: a few are left empty)
    unsigned long long n_two_class;   // queries answered by the two-class tail
    unsigned long long n_two_class_retry;   // ... that it gave up and handed to the exact path (lists not provably exact)
    unsigned long long n_narrow;      // front-queue entries with narrow window ranges (query shorter than OptDev::tc_limit): the two-class
                                      // workgroup kernel's share, if it is worth a kernel
    // not zeroed per call (set once per workspace): the first wave stage leaves the probe results of the queries it
    // queues by their length here -- 64 words per back-queue slot: list offset << 16 | list length of the
    // lane's feature -- so the second stage neither sketches nor probes them again.  (In a cache line of its own: the
    // counters above are hammered by atomics, and a load from their line queues behind them.)
    unsigned long long n_short;       // direct mode: queued queries whose list the first wave stage would have kept (<= 512 locations)
    unsigned long long n_geom;        // queries with more than 128 features: the workgroup kernels' by their geometry alone
    u32 blk_cursor[4];                // the workgroup kernels' shared cursors over the front queue ([0,1] plain kernel: long queries first, then the rest; [2,3] two-class kernel)
    u32 w_cursor[2];                  // (-DMCQ_WAVE_DYNQ) the same for the second / third wave stage
    unsigned long long n_long;        // queries of MCQ_BLOCK_LONG_FIRST bases and more (the workgroup kernels take those first: a pass of its own)
    unsigned long long pad_[17];      // (diagnostic builds, -DMCQ_PHASE_CLOCK: phase clocks of the workgroup kernel)
    unsigned long long* probe_buf;
    unsigned long long* probe_front;  // the same for FRONT-queue slots of queries with <= 64 features (third wave stage); a lane without
                                      // a feature holds 0xFFFF
    // not zeroed per call either: how the NEXT batch on this workspace enters (k_next_mode, after the last kernel of a batch).
    // Direct entry: the next batch's first stage only looks at the geometry of its queries -- one LANE per query -- and queues all
    // of them.  bit 1: most of this batch's queries had more than 128 features (long reads: the first stage did nothing for them
    // but push one queue entry per wave).  bit 0: most of them left the first stage one way or the other (a table with long
    // lists: RefSeq scale, 99 %) -- taken by the sharded home side only, whose first stage has no sketch to lose: on the fused
    // path the second stage (4 waves per SIMD) sketches and probes no faster than the first (8), measured +-2 %.  Results are
    // the same either way; only the path differs.
    u32 direct_mode;
};
static_assert(offsetof(CountersDev, probe_buf) == 256, "probe_buf sits 256 bytes into the block");
#define MCQ_CTR_ZEROED offsetof(CountersDev, probe_buf)

// ---- feature-sharded path: what a home rank's reduce kernels read instead of sketching and probing ------------
// The home rank sent every feature slot of the batch (slot = global window index x s + i) to the rank that owns the
// feature; slot_pos remembers where it went (owner << 27 | position in that owner's feature block, MCQ_EMPTY for an
// unused slot).  The owner answered per feature, in block order, with the inclusive end of the feature's location
// list inside its tile of MCQ_SHARD_TILE features, per tile with the start of the tile's lists inside the
// owner's location block, and with the lists themselves.  So a probe result (off, len) is three small loads away,
// and `off` indexes the received location buffer: the reduce kernels gather from it as the fused kernels do from
// the table, the exchanged lists are never copied again.
#define MCQ_SHARD_TILE 1024u
#define MCQ_SHARD_POS_BITS 27
#define MCQ_SHARD_MAX_RANKS 32u
struct ShardDev {
    const u32* slot_pos;     // [n_slots]
    const u32* ends;         // owner o's list ends start at ends + o * ends_stride
    const u32* tile_base;    // owner o's tile starts at tile_base + o * tile_stride
    const u64* win_off;      // [n_seqs + 1]: first global window of every sequence
    u32 ends_stride;         // u32 words between two owners' list ends
    u32 tile_stride;         // u32 words between two owners' tile starts
    u64 capL;                // locations per peer block
    u64 n_slots;             // entries of slot_pos (a batch with more windows than the context was sized for reads nothing beyond)
};
__device__ __forceinline__ void shard_fetch(const ShardDev& sh, u64 slot, u64& off, u32& len) {
    off = 0; len = 0;
    if (slot >= sh.n_slots) return;          // (flagged by k_shard_sketch)
    const u32 sp = sh.slot_pos[slot];
    if (sp == MCQ_EMPTY) return;
    const u32 o = sp >> MCQ_SHARD_POS_BITS, pos = sp & ((1u << MCQ_SHARD_POS_BITS) - 1);
    const u32* tb = sh.tile_base + (u64)o * sh.tile_stride;
    if (pos >= tb[-3]) return;               // header word 1 of the owner's answer: the features it served (a block that
                                             // travelled truncated answers only those; the error is flagged elsewhere)
    const u32* e = sh.ends + (u64)o * sh.ends_stride + pos;
    const u32 e1 = e[0], e0 = (pos & (MCQ_SHARD_TILE - 1)) ? e[-1] : 0u;
    const u64 start = (u64)tb[pos / MCQ_SHARD_TILE] + e0;
    if (start + (e1 - e0) > sh.capL) return; // the owner's location block was full: nothing was copied for this list
    len = e1 - e0;
    off = (u64)o * sh.capL + start;
}

// ---- overflow queues ---------------------------------------------------------------------------------
// One array, two queues: the front one grows from index 0, the back one downwards from the array's last entry.  A wave
// reserves MCQ_OVF_CHUNK slots per global atomic (with most of a batch overflowing, one atomic per query on a
// single address cost 5 ms per 1 M reads; 8 slots: 16 would be 1 % faster there and 3 % slower with 11 % overflowing) and keeps its reservation in five words of LDS (st: next slot and slots
// left of the front and the back queue, queries queued); when it ends it fills what is left with MCQ_EMPTY, which the
// draining kernels skip.
// Capacity: every query can sit in the back queue AND be passed on to the front queue by the second wave stage while
// other waves still drain the back queue, so the array has 2 x nq entries plus the unused reservation tails of three
// sets of at most 32768 waves (first stage front, first stage back, second stage front): the two queues never meet.
#ifndef MCQ_OVF_CHUNK
#define MCQ_OVF_CHUNK 8u            // tuning knob: 1u = one atomic per queued query, no empty slots
#endif
#define MCQ_OVF_TAIL (MCQ_OVF_CHUNK * 32768u)      // unused reservation tails of one set of waves
#define MCQ_OVF_PAD (3u * MCQ_OVF_TAIL)
// (+ nq / 8: the direct mode's first stage reserves whole chunks per 64 queries, ~5 % of its slots stay empty)
__device__ __host__ __forceinline__ u64 ovf_capacity(u64 nq) { return 2 * nq + nq / 8 + MCQ_OVF_PAD; }
__device__ __forceinline__ u64 ovf_slot(u64 nq, int back, u32 i) { return back ? ovf_capacity(nq) - 1 - i : (u64)i; }
// Draining order: slot of the it-th visit.  Reservations are filled from their first slot, so the real entries sit at
// the low positions of every chunk; a drainer striding through the slots by a multiple of its size (5120 waves, 512
// workgroups) would see the same position every time -- some would get all the work and others only empty slots.
// Visiting position-major (all first slots, then all second ones, ...) gives every drainer the same mix.
__device__ __forceinline__ u32 ovf_visit(u32 it, u32 n_reserved) {
    const u32 nch = n_reserved / MCQ_OVF_CHUNK;          // reservations
    const u32 pos = it / nch;
    return (it - pos * nch) * MCQ_OVF_CHUNK + pos;
}
__device__ __forceinline__ void ovf_init(u32* st) { st[0] = 0; st[1] = 0; st[2] = 0; st[3] = 0; st[4] = 0; }
__device__ __forceinline__ void ovf_push(u32* st, int back, CountersDev* ctr, u32* list, u64 nq, u32 q) {   // one lane
    u32 next = st[back], left = st[2 + back];
    if (left == 0) { next = atomicAdd(back ? &ctr->ovf_mid_count : &ctr->ovf_count, MCQ_OVF_CHUNK); left = MCQ_OVF_CHUNK; }
    list[ovf_slot(nq, back, next)] = q;
    st[back] = next + 1; st[2 + back] = left - 1; st[4] += 1;
}
__device__ __forceinline__ void ovf_flush(u32* st, CountersDev* ctr, u32* list, u64 nq) {                    // one lane
    for (int back = 0; back < 2; ++back) {
        u32 next = st[back];
        for (u32 left = st[2 + back]; left; --left, ++next) list[ovf_slot(nq, back, next)] = MCQ_EMPTY;
    }
    if (st[4]) atomicAdd(&ctr->n_ovf, st[4]);
}

// ------------------------------------------------------------------ scalars
__device__ __forceinline__ u32 tmh(u32 x) {
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    return (x >> 16) ^ x;
}

__device__ __forceinline__ u32 revcomp(u32 s, u32 k) {
    s = ((s >> 2) & 0x33333333u) | ((s & 0x33333333u) << 2);
    s = ((s >> 4) & 0x0F0F0F0Fu) | ((s & 0x0F0F0F0Fu) << 4);
    s = __builtin_bswap32(s);                 // the 8- and 16-bit swap steps together
    return (0xFFFFFFFFu - s) >> (32 - 2 * k);
}

__device__ __forceinline__ u32 canonical(u32 s, u32 k) {
    u32 r = revcomp(s, k);
    return s < r ? s : r;
}

// number of windows of a sequence of n bases (src/dna_encoding.h:259-276)
__device__ __host__ __forceinline__ u32 num_windows(u64 n, u32 W, u32 S) {
    if (n <= W) return 1;
    u64 nfull = (n - W) / S + 1;
    return (u32)(nfull + ((nfull * S < n) ? 1 : 0));
}
// window j of a sequence of n bases -> [beg, beg+len)
__device__ __forceinline__ void window_of(u64 n, u32 W, u32 S, u32 j, u64& beg, u32& len) {
    if (n <= W) { beg = 0; len = (u32)n; return; }
    u64 nfull = (n - W) / S + 1;
    beg = (u64)j * S;
    len = (j < nfull) ? W : (u32)(n - beg);
}

// 32-bit variants for sequences shorter than 2^31 bases; magic = floor(2^32 / S)
__device__ __forceinline__ u32 udiv_magic(u32 n, u32 d, u32 magic) {
    u32 q = __umulhi(n, magic);
    if (n - q * d >= d) ++q;
    return q;
}
__device__ __forceinline__ u32 num_windows32(u32 n, u32 W, u32 S, u32 magic) {
    if (n <= W) return 1;
    u32 nfull = udiv_magic(n - W, S, magic) + 1;
    return nfull + ((nfull * S < n) ? 1u : 0u);
}
__device__ __forceinline__ void window_of32(u32 n, u32 W, u32 S, u32 magic, u32 j, u32& beg, u32& len) {
    if (n <= W) { beg = 0; len = n; return; }
    u32 nfull = udiv_magic(n - W, S, magic) + 1;
    beg = j * S;
    len = (j < nfull) ? W : (n - beg);
}
// row 9's range width: 2 + max(len1+len2, insertSizeMax) / target stride (src/classification.cpp:217-219)
__device__ __forceinline__ u32 range_width(u64 qlen, u64 insert_size_max, u32 tgt_stride, u32 magic) {
    u64 m = qlen > insert_size_max ? qlen : insert_size_max;
    if (m < (1ull << 31)) return 2 + udiv_magic((u32)m, tgt_stride, magic);
    return (u32)(2 + m / tgt_stride);
}

// ------------------------------------------------------------------ wave primitives
__device__ __forceinline__ u32 lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of a wave-wide mask below this lane (v_mbcnt_lo/hi: the lane id itself is mbcnt(~0))
__device__ __forceinline__ u32 lane_rank(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// Orders this wave's LDS/global accesses across lanes.  A wave executes its memory
// instructions in order, so only the compiler has to be stopped from reordering.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// full-wave DPP permutation: every lane is written, so no previous value is tied to the result
// (update_dpp(old = v, ...) would cost a v_mov copy per use)
template <int CTRL>
__device__ __forceinline__ u32 dpp_mov(u32 v) {
    return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, false);
}

// minimum over the 64 lanes, returned wave-uniform.  All lanes must be active.
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
    u32 t;
    t = dpp_mov<0xB1>(v);  v = t < v ? t : v;   // quad_perm [1,0,3,2]
    t = dpp_mov<0x4E>(v);  v = t < v ? t : v;   // quad_perm [2,3,0,1]
    t = dpp_mov<0x141>(v); v = t < v ? t : v;   // row_half_mirror
    t = dpp_mov<0x140>(v); v = t < v ? t : v;   // row_mirror: every lane of a row holds the row min
    u32 a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    u32 c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    a = a < b ? a : b; c = c < d ? c : d;
    return a < c ? a : c;
}

__device__ __forceinline__ u32 bcast(u32 v, u32 src_lane) {       // src_lane wave-uniform
    return __builtin_amdgcn_readlane(v, src_lane);
}

// ------------------------------------------------------------------ rows 2-5: one window, one wave
// value of lane (l ^ J), J a power of two < 64: VALU only (DPP / permlane swaps), no LDS pipe
template <int J>
__device__ __forceinline__ u32 xor_lane(u32 v, u32 lane) {
    if constexpr (J == 1) return dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {                                          // banks 0,2 <- lane+4, banks 1,3 <- lane-4
        u32 t = (u32)__builtin_amdgcn_mov_dpp((int)v, 0x104, 0xF, 0x5, false);              // row_shl:4
        return (u32)__builtin_amdgcn_update_dpp((int)t, (int)v, 0x114, 0xF, 0xA, false);    // row_shr:4
    }
    else if constexpr (J == 8) return dpp_mov<0x128>(v);                 // row_ror:8
    else if constexpr (J == 16) { auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? a[0] : a[1]; }
    else { auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? a[0] : a[1]; }
}

// lanes that keep the minimum in a compare-exchange at distance J of a bitonic level K
// (K >= 64: every lane sorts ascending unless `up` says otherwise)
constexpr u64 keepmin_mask(int K, int J, bool up_const) {
    u64 m = 0;
    for (int l = 0; l < 64; ++l) {
        bool up = (K >= 64) ? up_const : ((l & K) == 0);
        bool lower = (l & J) == 0;
        if (lower == up) m |= 1ull << l;
    }
    return m;
}
// compare-exchange result for this lane: keepmin lanes take min(v,o), the others max(v,o).
// One v_cmp, one s_xor with a constant lane mask, one v_cndmask.  All lanes must be active.
__device__ __forceinline__ u32 cmpex(u32 v, u32 o, u64 keepmin) {
    u64 take = __ballot(v < o) ^ keepmin;              // take the partner's value
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(o), "s"(take));
    return r;
}
__device__ __forceinline__ u64 cmpex(u64 v, u64 o, u64 keepmin) {
    u64 take = __ballot(v < o) ^ keepmin;
    u32 lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((u32)v), "v"((u32)o), "s"(take));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((u32)(v >> 32)), "v"((u32)(o >> 32)), "s"(take));
    return ((u64)hi << 32) | lo;
}

// ---- one u32 per lane: sorting network on v_min_u32 / v_max_u32 with DPP operands -------------
// A compare-exchange whose partner is one DPP permutation away needs no compare, no lane mask
// arithmetic and no VCC round trip: min and max of (own, partner) are one VOP2+DPP instruction each.
//   MCQ_CX2: the lanes that keep the minimum are a union of 4-lane banks / 16-lane rows, so the two
//            instructions write disjoint lane sets of the result directly (DPP bank_mask / row_mask).
//   MCQ_CX3: any other lane pattern (period 2 or 4): min and max everywhere, v_cndmask picks per lane.
// Network: "flip" form of the bitonic sorter (first stage of a K-block merge pairs i with i ^ (K-1),
// then half-cleaners i ^ J), every comparator ascending, up to K = 32; for the last level the keys of
// rows 2,3 are kept complemented during levels 2..32 so that lanes 32..63 come out descending and the
// K = 64 merge can start with the cheap lane ^ 32 exchange (v_permlane32_swap).
// The leading s_nop covers the 2 wait states between a VALU write of a VGPR and a DPP read of it (the
// compiler does not look inside asm blocks).
#define MCQ_DPP_ALL " row_mask:0xf bank_mask:0xf"
#define MCQ_CX2(v, DPP_MIN, DPP_MAX) do { u32 r_; \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 " DPP_MIN "\n\tv_max_u32_dpp %0, %1, %1 " DPP_MAX \
        : "=&v"(r_) : "v"(v)); v = r_; } while (0)
#ifdef MCQ_CX3_SGPR_MASK        // tuning knob (A/B; r03: -382 static SALU, +329 s_nop, +129 spill reloads: no gain): the lane mask as an SGPR-pair operand instead of two s_mov into vcc
#define MCQ_CX3(v, DPP, MASK32) do { u32 lo_, hi_; const u64 m_ = ((u64)(MASK32##u) << 32) | (MASK32##u); \
    asm("s_nop 1\n\t" \
        "v_min_u32_dpp %0, %2, %2 " DPP MCQ_DPP_ALL "\n\tv_max_u32_dpp %1, %2, %2 " DPP MCQ_DPP_ALL "\n\t" \
        "v_cndmask_b32_e64 %0, %1, %0, %3" \
        : "=&v"(lo_), "=&v"(hi_) : "v"(v), "s"(m_)); v = lo_; } while (0)
#else
#define MCQ_CX3(v, DPP, MASK32) do { u32 lo_, hi_; \
    asm("s_mov_b32 vcc_lo, " #MASK32 "\n\ts_mov_b32 vcc_hi, " #MASK32 "\n\t"   /* the two wait states before the DPP reads */ \
        "v_min_u32_dpp %0, %2, %2 " DPP MCQ_DPP_ALL "\n\tv_max_u32_dpp %1, %2, %2 " DPP MCQ_DPP_ALL "\n\t" \
        "v_cndmask_b32_e32 %0, %1, %0, vcc" \
        : "=&v"(lo_), "=&v"(hi_) : "v"(v) : "vcc"); v = lo_; } while (0)
#endif

// half-cleaners at lane distance 8, 4, 2, 1 (ascending everywhere)
__device__ __forceinline__ u32 cx_j8(u32 v) { MCQ_CX2(v, "row_ror:8 row_mask:0xf bank_mask:0x3", "row_ror:8 row_mask:0xf bank_mask:0xc"); return v; }
__device__ __forceinline__ u32 cx_j4(u32 v) { MCQ_CX2(v, "row_shl:4 row_mask:0xf bank_mask:0x5", "row_shr:4 row_mask:0xf bank_mask:0xa"); return v; }
__device__ __forceinline__ u32 cx_j2(u32 v) { MCQ_CX3(v, "quad_perm:[2,3,0,1]", 0x33333333); return v; }
__device__ __forceinline__ u32 cx_j1(u32 v) { MCQ_CX3(v, "quad_perm:[1,0,3,2]", 0x55555555); return v; }
// flips of blocks of 4, 8, 16 lanes
__device__ __forceinline__ u32 cx_flip4(u32 v)  { MCQ_CX3(v, "quad_perm:[3,2,1,0]", 0x33333333); return v; }
__device__ __forceinline__ u32 cx_flip8(u32 v)  { MCQ_CX2(v, "row_half_mirror row_mask:0xf bank_mask:0x5", "row_half_mirror row_mask:0xf bank_mask:0xa"); return v; }
__device__ __forceinline__ u32 cx_flip16(u32 v) { MCQ_CX2(v, "row_mirror row_mask:0xf bank_mask:0x3", "row_mirror row_mask:0xf bank_mask:0xc"); return v; }
// flip of blocks of 32 lanes: partner = mirrored lane of the neighbouring row.  v_permlane16_swap of two
// copies leaves (R0,R0,R2,R2) and (R1,R1,R3,R3): every row finds its neighbour row in place.
__device__ __forceinline__ u32 cx_flip32(u32 v) {
    auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    u32 r;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %2, %3 row_mirror row_mask:0x5 bank_mask:0xf\n\t"
        "v_max_u32_dpp %0, %1, %3 row_mirror row_mask:0xa bank_mask:0xf"
        : "=&v"(r) : "v"(sw[0]), "v"(sw[1]), "v"(v));
    return r;
}
// lane ^ 16 and lane ^ 32 exchanges (ascending): both copies after the swap hold the pair in every lane
__device__ __forceinline__ u32 cx_j16(u32 v) {
    auto sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    u32 r;
    asm("s_nop 1\n\tv_min_u32_e32 %0, %1, %2\n\tv_max_u32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:0xa bank_mask:0xf"
        : "=&v"(r) : "v"(sw[0]), "v"(sw[1]));
    return r;
}
__device__ __forceinline__ u32 cx_j32(u32 v) {
    auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    u32 r;
    asm("s_nop 1\n\tv_min_u32_e32 %0, %1, %2\n\tv_max_u32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf"
        : "=&v"(r) : "v"(sw[0]), "v"(sw[1]));
    return r;
}
// complement the keys of rows 2,3 in place
__device__ __forceinline__ u32 not_rows23(u32 v) {
    asm("s_nop 1\n\tv_not_b32_dpp %0, %0 quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}
// every aligned block of 32 lanes sorted ascending
__device__ __forceinline__ u32 wave_sort_blocks32(u32 v) {
    v = cx_j1(v);                                                   // K = 2 (flip of 2 = lane ^ 1)
    v = cx_flip4(v);  v = cx_j1(v);                                 // K = 4
    v = cx_flip8(v);  v = cx_j2(v); v = cx_j1(v);                   // K = 8
    v = cx_flip16(v); v = cx_j4(v); v = cx_j2(v); v = cx_j1(v);     // K = 16
    v = cx_flip32(v); v = cx_j8(v); v = cx_j4(v); v = cx_j2(v); v = cx_j1(v);   // K = 32
    return v;
}
// ascending sort of one u32 per lane across the wave
__device__ __forceinline__ u32 wave_sort64(u32 v, u32 lane) {
    (void)lane;
    v = not_rows23(v);
    v = wave_sort_blocks32(v);
    v = not_rows23(v);                                              // lanes 32..63 now descending
    v = cx_j32(v); v = cx_j16(v); v = cx_j8(v); v = cx_j4(v); v = cx_j2(v); v = cx_j1(v);
    return v;
}
// ascending sort of lanes 0-31 (lanes 32-63 hold padding and sort among themselves)
__device__ __forceinline__ u32 wave_sort32_low(u32 v, u32 lane) {
    (void)lane;
    return wave_sort_blocks32(v);
}

// ---- the same networks as ONE asm block each (r04) ------------------------------------------------------------------------
// Stage by stage (above) every compare-exchange is its own asm statement: it opens with the two wait states a DPP read needs
// behind a VALU write, and the compiler closes it with another `s_nop 0` (it does not look inside and assumes the worst) --
// a third of the instructions of a sort were wait states, and the kernels that sort are short of issue slots.
// Written as one block the chain needs a wait only where a stage reads what the instruction before it wrote; and TWO
// registers sorted side by side (wave_sort64_x2, cx_chain6_x2: stage by stage, a's instructions, then b's) need none at all --
// each register's DPP read sits two or more instructions behind its own last write -- and share the lane-mask moves.
// Registers ping-pong (x -> y -> x ...): min and max write disjoint lanes of the destination while both read the source.
// -DMCQ_SORT_STAGEWISE (A/B knob): the stage-by-stage forms.
#define MQ_ALL " row_mask:0xf bank_mask:0xf"
#define MQ_NOP2 "s_nop 1\n\t"
#define MQ_NOP1 "s_nop 0\n\t"
#define MQ_VCC(M) "s_mov_b32 vcc_lo, " M "\n\ts_mov_b32 vcc_hi, " M "\n\t"
#define MQ_M3 "0x33333333"
#define MQ_M5 "0x55555555"
#define MQ_CX2(D, S, PMIN, PMAX) "v_min_u32_dpp " D ", " S ", " S " " PMIN "\n\tv_max_u32_dpp " D ", " S ", " S " " PMAX "\n\t"
#define MQ_CX3(D, S, T, P) "v_min_u32_dpp " D ", " S ", " S " " P MQ_ALL "\n\tv_max_u32_dpp " T ", " S ", " S " " P MQ_ALL "\n\t" \
                           "v_cndmask_b32_e32 " D ", " T ", " D ", vcc\n\t"
#define MQ_J8(D, S)  MQ_CX2(D, S, "row_ror:8 row_mask:0xf bank_mask:0x3", "row_ror:8 row_mask:0xf bank_mask:0xc")
#define MQ_J4(D, S)  MQ_CX2(D, S, "row_shl:4 row_mask:0xf bank_mask:0x5", "row_shr:4 row_mask:0xf bank_mask:0xa")
#define MQ_F8(D, S)  MQ_CX2(D, S, "row_half_mirror row_mask:0xf bank_mask:0x5", "row_half_mirror row_mask:0xf bank_mask:0xa")
#define MQ_F16(D, S) MQ_CX2(D, S, "row_mirror row_mask:0xf bank_mask:0x3", "row_mirror row_mask:0xf bank_mask:0xc")
#define MQ_J2(D, S, T) MQ_CX3(D, S, T, "quad_perm:[2,3,0,1]")          /* lane mask 0x33333333 */
#define MQ_J1(D, S, T) MQ_CX3(D, S, T, "quad_perm:[1,0,3,2]")          /* lane mask 0x55555555 */
#define MQ_F4(D, S, T) MQ_CX3(D, S, T, "quad_perm:[3,2,1,0]")          /* lane mask 0x33333333 */
#define MQ_NOT23(S) "v_not_b32_dpp " S ", " S " quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf\n\t"
// the tails of the permlane stages (after the swaps): flip of 32-lane blocks; lane ^ 32; lane ^ 16
#define MQ_F32_TAIL(D, S, T0, T1) "v_min_u32_dpp " D ", " T1 ", " S " row_mirror row_mask:0x5 bank_mask:0xf\n\t" \
                                  "v_max_u32_dpp " D ", " T0 ", " S " row_mirror row_mask:0xa bank_mask:0xf\n\t"
#define MQ_J32_TAIL(D, S, T) "v_min_u32_e32 " D ", " S ", " T "\n\tv_max_u32_dpp " D ", " S ", " T " quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf\n\t"
#define MQ_J16_TAIL(D, S, T) "v_min_u32_e32 " D ", " S ", " T "\n\tv_max_u32_dpp " D ", " S ", " T " quad_perm:[0,1,2,3] row_mask:0xa bank_mask:0xf\n\t"
#define MQ_MOV(D, S) "v_mov_b32 " D ", " S "\n\t"
#define MQ_SWAP16(A, B) "v_permlane16_swap_b32 " A ", " B "\n\t"
#define MQ_SWAP32(A, B) "v_permlane32_swap_b32 " A ", " B "\n\t"
// one register: X -> result in Y (15 stages); temporaries Z, W
#define MQ_BLOCKS32_1(X, Y, Z, W) \
    MQ_VCC(MQ_M5) MQ_J1(Y, X, Z) \
    MQ_VCC(MQ_M3) MQ_F4(X, Y, Z)  MQ_VCC(MQ_M5) MQ_J1(Y, X, Z) \
    MQ_NOP2 MQ_F8(X, Y)  MQ_VCC(MQ_M3) MQ_J2(Y, X, Z)  MQ_VCC(MQ_M5) MQ_J1(X, Y, Z) \
    MQ_NOP2 MQ_F16(Y, X) MQ_NOP2 MQ_J4(X, Y)  MQ_VCC(MQ_M3) MQ_J2(Y, X, Z)  MQ_VCC(MQ_M5) MQ_J1(X, Y, Z) \
    MQ_MOV(Z, X) MQ_MOV(W, X) MQ_NOP2 MQ_SWAP16(Z, W) MQ_NOP2 MQ_F32_TAIL(Y, X, Z, W) \
    MQ_NOP2 MQ_J8(X, Y) MQ_NOP2 MQ_J4(Y, X)  MQ_VCC(MQ_M3) MQ_J2(X, Y, Z)  MQ_VCC(MQ_M5) MQ_J1(Y, X, Z)
// one register, the half-cleaners 32 .. 1: X -> result in X; temporaries Y, Z (X, Y are both written on the way)
#define MQ_CHAIN6_1(X, Y, Z) \
    MQ_MOV(Z, X) MQ_NOP2 MQ_SWAP32(X, Z) MQ_NOP2 MQ_J32_TAIL(Y, X, Z) \
    MQ_MOV(Z, Y) MQ_NOP2 MQ_SWAP16(Y, Z) MQ_NOP2 MQ_J16_TAIL(X, Y, Z) \
    MQ_NOP2 MQ_J8(Y, X) MQ_NOP2 MQ_J4(X, Y)  MQ_VCC(MQ_M3) MQ_J2(Y, X, Z)  MQ_VCC(MQ_M5) MQ_J1(X, Y, Z)
// two registers side by side: no wait states between the stages
#define MQ_CX3_2(M, DA, SA, TA, DB, SB, TB, ST) MQ_VCC(M) ST(DA, SA, TA) ST(DB, SB, TB)
#define MQ_CX2_2(DA, SA, DB, SB, ST) ST(DA, SA) ST(DB, SB)
#define MQ_BLOCKS32_2(XA, YA, ZA, WA, XB, YB, ZB, WB) \
    MQ_CX3_2(MQ_M5, YA, XA, ZA, YB, XB, ZB, MQ_J1) \
    MQ_CX3_2(MQ_M3, XA, YA, ZA, XB, YB, ZB, MQ_F4) MQ_CX3_2(MQ_M5, YA, XA, ZA, YB, XB, ZB, MQ_J1) \
    MQ_CX2_2(XA, YA, XB, YB, MQ_F8) MQ_CX3_2(MQ_M3, YA, XA, ZA, YB, XB, ZB, MQ_J2) MQ_CX3_2(MQ_M5, XA, YA, ZA, XB, YB, ZB, MQ_J1) \
    MQ_CX2_2(YA, XA, YB, XB, MQ_F16) MQ_CX2_2(XA, YA, XB, YB, MQ_J4) MQ_CX3_2(MQ_M3, YA, XA, ZA, YB, XB, ZB, MQ_J2) MQ_CX3_2(MQ_M5, XA, YA, ZA, XB, YB, ZB, MQ_J1) \
    MQ_MOV(ZA, XA) MQ_MOV(WA, XA) MQ_MOV(ZB, XB) MQ_MOV(WB, XB) MQ_SWAP16(ZA, WA) MQ_NOP1 MQ_SWAP16(ZB, WB) \
    MQ_F32_TAIL(YA, XA, ZA, WA) MQ_F32_TAIL(YB, XB, ZB, WB) \
    MQ_CX2_2(XA, YA, XB, YB, MQ_J8) MQ_CX2_2(YA, XA, YB, XB, MQ_J4) MQ_CX3_2(MQ_M3, XA, YA, ZA, XB, YB, ZB, MQ_J2) MQ_CX3_2(MQ_M5, YA, XA, ZA, YB, XB, ZB, MQ_J1)
#define MQ_CHAIN6_2(XA, YA, ZA, XB, YB, ZB) \
    MQ_MOV(ZA, XA) MQ_MOV(ZB, XB) MQ_NOP1 MQ_SWAP32(XA, ZA) MQ_SWAP32(XB, ZB) MQ_J32_TAIL(YA, XA, ZA) MQ_J32_TAIL(YB, XB, ZB) \
    MQ_MOV(ZA, YA) MQ_MOV(ZB, YB) MQ_NOP1 MQ_SWAP16(YA, ZA) MQ_SWAP16(YB, ZB) MQ_J16_TAIL(XA, YA, ZA) MQ_J16_TAIL(XB, YB, ZB) \
    MQ_CX2_2(YA, XA, YB, XB, MQ_J8) MQ_CX2_2(XA, YA, XB, YB, MQ_J4) MQ_CX3_2(MQ_M3, YA, XA, ZA, YB, XB, ZB, MQ_J2) MQ_CX3_2(MQ_M5, XA, YA, ZA, XB, YB, ZB, MQ_J1)

#ifndef MCQ_SORT_STAGEWISE
// every aligned block of 32 lanes sorted ascending
__device__ __forceinline__ u32 wave_sort_blocks32_1(u32 v) {
    u32 y, z, w;
    asm(MQ_BLOCKS32_1("%[x]", "%[y]", "%[z]", "%[w]") : [x] "+v"(v), [y] "=&v"(y), [z] "=&v"(z), [w] "=&v"(w) : : "vcc");
    return y;
}
// ascending sort of one u32 per lane across the wave
__device__ __forceinline__ u32 wave_sort64_1(u32 v) {
    u32 y, z, w;
    asm(MQ_NOP2 MQ_NOT23("%[x]") MQ_BLOCKS32_1("%[x]", "%[y]", "%[z]", "%[w]") MQ_NOP2 MQ_NOT23("%[y]")
        // (MQ_CHAIN6_1 with the roles of x and y exchanged: y -> y)
        MQ_CHAIN6_1("%[y]", "%[x]", "%[z]")
        : [x] "+v"(v), [y] "=&v"(y), [z] "=&v"(z), [w] "=&v"(w) : : "vcc");
    return y;
}
// half-cleaners at lane distance 32 .. 1 (ascending)
__device__ __forceinline__ u32 cx_chain6_1(u32 v) {
    u32 y, z;
    asm(MQ_CHAIN6_1("%[x]", "%[y]", "%[z]") : [x] "+v"(v), [y] "=&v"(y), [z] "=&v"(z) : : "vcc");
    return v;
}
// the same for two registers at once
__device__ __forceinline__ void wave_sort64_x2(u32& a, u32& b) {
    u32 ya, za, wa, yb, zb, wb;
    asm(MQ_NOP2 MQ_NOT23("%[xa]") MQ_NOP1 MQ_NOT23("%[xb]")
        MQ_BLOCKS32_2("%[xa]", "%[ya]", "%[za]", "%[wa]", "%[xb]", "%[yb]", "%[zb]", "%[wb]")
        MQ_NOT23("%[ya]") MQ_NOP1 MQ_NOT23("%[yb]")
        MQ_CHAIN6_2("%[ya]", "%[xa]", "%[za]", "%[yb]", "%[xb]", "%[zb]")
        : [xa] "+v"(a), [ya] "=&v"(ya), [za] "=&v"(za), [wa] "=&v"(wa), [xb] "+v"(b), [yb] "=&v"(yb), [zb] "=&v"(zb), [wb] "=&v"(wb) : : "vcc");
    a = ya; b = yb;
}
__device__ __forceinline__ void cx_chain6_x2(u32& a, u32& b) {
    u32 ya, za, yb, zb;
    asm(MQ_CHAIN6_2("%[xa]", "%[ya]", "%[za]", "%[xb]", "%[yb]", "%[zb]")
        : [xa] "+v"(a), [ya] "=&v"(ya), [za] "=&v"(za), [xb] "+v"(b), [yb] "=&v"(yb), [zb] "=&v"(zb) : : "vcc");
}
#else
__device__ __forceinline__ u32 wave_sort_blocks32_1(u32 v) { return wave_sort_blocks32(v); }
__device__ __forceinline__ u32 wave_sort64_1(u32 v) { return wave_sort64(v, 0); }
__device__ __forceinline__ u32 cx_chain6_1(u32 v) { v = cx_j32(v); v = cx_j16(v); v = cx_j8(v); v = cx_j4(v); v = cx_j2(v); return cx_j1(v); }
__device__ __forceinline__ void wave_sort64_x2(u32& a, u32& b) { a = wave_sort64(a, 0); b = wave_sort64(b, 0); }
__device__ __forceinline__ void cx_chain6_x2(u32& a, u32& b) { a = cx_chain6_1(a); b = cx_chain6_1(b); }
#endif

// Sketch of seq[0..n), n <= 128, by one full wave.  Lane l encodes bases 2l and 2l+1;
// 8 lanes form one 16-base word (2 bits per base, first base in the top bits), 16 lanes
// one 32-base ambiguity word.  Lane l then owns the k-mers starting at l and l+64.
// The min(s, n-k+1) smallest DISTINCT hashes (ascending) are written to dst[0..m);
// returns m (wave-uniform).  tmp: 64 words of this wave's LDS scratch (dst != tmp).
//
// Selection: hashes are ~uniform, so the s smallest of c > 64 values are all below
// thr = 40/c of the hash range except with negligible probability.  Values below thr
// (all values if c <= 64) are compacted to one per lane, sorted by a 21-stage in-register
// bitonic network and de-duplicated.  If the filter let through more than 64 values or
// fewer than s distinct ones, the exact repeated-wave-min selection runs instead.
// the two bases lane `lane` encodes for a window seq[0..n): positions 2*lane and 2*lane+1, as one
// 16-bit word (low byte first; 'N' beyond the end).  One (unaligned) 2-byte load per lane.
struct __attribute__((packed)) PackedU16 { unsigned short v; };
__device__ __forceinline__ u32 window_chars2(const char* __restrict__ seq, u32 n, u32 lane) {
    const u32 p = 2 * lane;
    u32 c = (u32)'N' | ((u32)'N' << 8);
    if (p + 1 < n) c = reinterpret_cast<const PackedU16*>(seq + p)->v;
    else if (p < n) c = (u32)(unsigned char)seq[p] | ((u32)'N' << 8);
    return c;
}
__device__ __forceinline__ void window_chars(const char* __restrict__ seq, u32 n, u32 lane, u32& c0, u32& c1) {
    const u32 c = window_chars2(seq, n, lane);
    c0 = c & 0xFFu; c1 = c >> 8;
}

// the bases of a window, two per lane (window_chars), as the words the sketch works on: w = the 16 bases
// [16 * (lane / 8), +16) of the window, 2 bits each, first base in the top bits; am = the ambiguity bits of the 32 bases
// [32 * (lane / 16), +32), first base in the top bit
__device__ __forceinline__ void wave_words_of_chars(u32 c0, u32 c1, u32 lane, u32& w, u32& am) {
    // A/a=0 C/c=1 G/g=2 T/t=3 (src/dna_encoding.h:326-336); anything else ambiguous
    u32 u0 = c0 & 0xDFu, u1 = c1 & 0xDFu;
    u32 x0 = (u0 >> 1) & 3u, x1 = (u1 >> 1) & 3u;
    x0 ^= x0 >> 1; x1 ^= x1 >> 1;
    u32 a0 = !(u0 == 'A' || u0 == 'C' || u0 == 'G' || u0 == 'T');
    u32 a1 = !(u1 == 'A' || u1 == 'C' || u1 == 'G' || u1 == 'T');
    w = ((x0 << 2) | x1) << (28 - 4 * (lane & 7));
    w |= xor_lane<1>(w, lane); w |= xor_lane<2>(w, lane); w |= xor_lane<4>(w, lane);
    am = ((a0 << 1) | a1) << (30 - 2 * (lane & 15));
    am |= xor_lane<1>(am, lane); am |= xor_lane<2>(am, lane); am |= xor_lane<4>(am, lane); am |= xor_lane<8>(am, lane);
}
// the same words out of an MCQ_BATCH_PACKED buffer: the window starts at base `at` of the batch; two (broadcast) loads and
// a funnel shift per plane.  Bits behind the window's end are never looked at (a k-mer must end inside the window).
__device__ __forceinline__ void wave_words_of_packed(const BatchDev& b, u64 at, u32 lane, u32& w, u32& am) {
    const u32* __restrict__ P2 = reinterpret_cast<const u32*>(b.bases);
    const u32 wi = (u32)(at >> 4) + (lane >> 3), sh = (u32)(at & 15) * 2;
    const u32 p0 = P2[wi < b.last_word ? wi : b.last_word], p1 = P2[wi + 1 < b.last_word ? wi + 1 : b.last_word];
    w = (u32)((((u64)p0 << 32) | p1) >> (32 - sh));
    const u32 ai = (u32)(at >> 5) + (lane >> 4), ash = (u32)(at & 31);
    const u32 alast = b.amb_last;
    const u32* __restrict__ AM = P2 + b.amb_off;
    const u32 a0 = AM[ai < alast ? ai : alast], a1 = AM[ai + 1 < alast ? ai + 1 : alast];
    am = (u32)((((u64)a0 << 32) | a1) >> (32 - ash));
}

// sketch of a window given as words (see wave_words_of_chars)
__device__ __forceinline__ u32 wave_sketch_words(u32 w, u32 am, u32 n, u32 k, u32 s, u32 lane, u32* tmp, u32* dst) {
    if (n < k) return 0;
    u32 cap = n - k + 1;
    u32 sl = s < cap ? s : cap;

    // a window with at most 64 k-mer start positions (the tail window of a read) needs only slot 0
    const bool one_slot = cap <= 64;
    u32 h[2];
    h[1] = MCQ_EMPTY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && one_slot) break;
        u32 pos = lane + 64 * i;
        u32 wi = pos >> 4, sh = (pos & 15) * 2;
        u32 w0 = __shfl(w, (int)((wi * 8) & 63), 64);
        u32 w1 = __shfl(w, (int)(((wi + 1) * 8) & 63), 64);
        u32 win32 = (u32)((((u64)w0 << 32) | w1) >> (32 - sh));
        u32 kmer = win32 >> (32 - 2 * k);
        u32 ai = pos >> 5, ash = pos & 31;
        u32 m0 = __shfl(am, (int)((ai * 16) & 63), 64);
        u32 m1 = __shfl(am, (int)(((ai + 1) * 16) & 63), 64);
        u32 amb32 = (u32)((((u64)m0 << 32) | m1) >> (32 - ash));
        bool ok = (pos + k <= n) && ((amb32 >> (32 - k)) == 0);
        h[i] = ok ? tmh(canonical(kmer, k)) : MCQ_EMPTY;
    }

    // ---- filter + compact + sort
    const u32 c = (u32)__builtin_popcountll(__ballot(h[0] != MCQ_EMPTY)) + (u32)__builtin_popcountll(__ballot(h[1] != MCQ_EMPTY));
    if (c == 0) return 0;
    const bool all_in = c <= 64;
    // 40/c of the hash range; any threshold gives the same sketch (the fallback below is exact), so a float
    // reciprocal replaces the integer division (c <= 128: the product stays below 2^32)
#ifndef MCQ_SKETCH_EXPECT
#define MCQ_SKETCH_EXPECT 40          // expected number of hashes below the threshold (tuning knob)
#endif
    const u32 thr = all_in ? MCQ_EMPTY : (u32)(__builtin_amdgcn_rcpf((float)c) * (4294967296.0f * MCQ_SKETCH_EXPECT));
    const bool s0 = h[0] < thr, s1 = h[1] < thr;
    const u64 m0 = __ballot(s0), m1 = __ballot(s1);
    const u32 n0 = (u32)__builtin_popcountll(m0), cnt = n0 + (u32)__builtin_popcountll(m1);
    bool fallback = cnt > 64;
    u32 m = 0;
    if (!fallback) {
        if (s0) tmp[lane_rank(m0)] = h[0];
        if (s1) tmp[n0 + lane_rank(m1)] = h[1];
        wave_sync();
        u32 v = lane < cnt ? tmp[lane] : MCQ_EMPTY;
        v = (cnt <= 32) ? wave_sort_blocks32_1(v) : wave_sort64_1(v);          // 15 stages when the low half suffices
        asm("s_nop 1" : "+v"(v));                 // v was written inside an asm block: 2 wait states before a DPP read
        const u32 prev = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);    // wave_shr:1
        bool keep = (v != MCQ_EMPTY) && (lane == 0 || v != prev);
        u64 km = __ballot(keep);
        u32 D = (u32)__builtin_popcountll(km);
        if (!all_in && D < sl) fallback = true;
        else {
            m = D < sl ? D : sl;
            u32 rank = lane_rank(km);
            if (keep && rank < m) dst[rank] = v;
        }
    }
    if (fallback) {
        // exact selection: repeated wave-min, equal values retire together
        for (m = 0; m < sl; ++m) {
            u32 lo = h[0] < h[1] ? h[0] : h[1];
            u32 mn = wave_min_u32(lo);
            if (mn == MCQ_EMPTY) break;
            if (lane == 0) dst[m] = mn;
            if (h[0] == mn) h[0] = MCQ_EMPTY;
            if (h[1] == mn) h[1] = MCQ_EMPTY;
        }
    }
    wave_sync();
    return m;
}

__device__ __forceinline__ u32 wave_sketch(const char* __restrict__ seq, u32 n, u32 k, u32 s,
                                           u32 lane, u32* tmp, u32* dst) {
    if (n < k) return 0;
    u32 c0, c1, w, am;
    window_chars(seq, n, lane, c0, c1);
    wave_words_of_chars(c0, c1, lane, w, am);
    return wave_sketch_words(w, am, n, k, s, lane, tmp, dst);
}
// window [at, at + n) of the batch, whichever form the batch is in
__device__ __forceinline__ u32 wave_sketch_b(const BatchDev& b, u64 at, u32 n, u32 k, u32 s, u32 lane, u32* tmp, u32* dst) {
    if (n < k) return 0;
    u32 w, am;
    if (b.packed) wave_words_of_packed(b, at, lane, w, am);
    else { u32 c0, c1; window_chars(b.bases + at, n, lane, c0, c1); wave_words_of_chars(c0, c1, lane, w, am); }
    return wave_sketch_words(w, am, n, k, s, lane, tmp, dst);
}

// ------------------------------------------------------------------ row 6: probe
// Linear probing over 64-B buckets; one 16-B load returns key and list length -- and, for the short lists most features
// have, the bucket IS the list: its locations are in the same 64-B sector the probe has just brought in, so the gather
// that follows hits the cache instead of fetching a second random sector (r01: 55 sectors per read, 29 of them list
// heads).  off = index of the list's first location relative to db.locs, in location units, for both kinds of list.
// Two layouts, chosen per table (mcq_db_create): 64-B buckets with inline lists while lists are short (a third fewer HBM
// requests per read on a 2 Gbp table), 16-B slots {key, len, offset} with every list behind the slot array once they are
// not (>= 10 Gbp: most lists do not fit a bucket any more, and the 64-B array costs +27 GB and 1-3 % time).
#define MCQ_BUCKET_BYTES 64u
// (everything about a layout follows from two scalars, bsh and compact)
__device__ __host__ __forceinline__ u32 bucket_inline_max(u32 bsh, u32 compact) { return ((bsh * 7u) >> 1) << compact; }   // 14 / 7 / 0
// BSH = log2(uint4 per bucket) as a compile-time constant: the query kernels are short of SGPRs, and one more live
// scalar in the probe loop (a run-time bucket size) costs the hot kernel 270 spill reloads and 2 % -- so every caller
// branches once per probe on db.bsh (wave-uniform) into the code of its layout
template <u32 BSH>
__device__ __forceinline__ uint4 bucket_head_t(const DbDev& db, u32 idx) { return db.slots[(u64)idx << BSH]; }
template <u32 BSH>
__device__ __forceinline__ void bucket_list_t(const DbDev& db, u32 idx, const uint4& sl, u64& off, u32& len) {
    len = sl.y;
    const u32 sh = BSH + 1 + db.compact;                     // log2(locations per bucket): 16 B = 2 wide / 4 compact locations
    off = len <= bucket_inline_max(BSH, db.compact) ? ((u64)idx << sh) + (1u << db.compact)      // behind the 8-B head
                                                    : (((u64)db.slot_mask + 1) << sh) + (((u64)sl.w << 32) | sl.z);
}
template <u32 BSH>
__device__ __forceinline__ void probe_t(const DbDev& db, u32 f, u64& off, u32& len) {
    u32 idx = tmh(f) & db.slot_mask;
    while (true) {
        const uint4 sl = bucket_head_t<BSH>(db, idx);
        if (sl.x == f) { bucket_list_t<BSH>(db, idx, sl, off, len); return; }
        if (sl.x == MCQ_EMPTY) return;
        idx = (idx + 1) & db.slot_mask;
    }
}
// BSH = 2 / 0: the layout is known at compile time (the wave kernels are instantiated per layout); -1: one wave-uniform
// branch on db.bsh per probe (workgroup kernel, staged kernels, the match-list taps)
template <int BSH = -1>
__device__ __forceinline__ void probe(const DbDev& db, u32 f, u64& off, u32& len) {
    len = 0; off = 0;
    if (f == MCQ_EMPTY) return;
    if constexpr (BSH >= 0) probe_t<(u32)BSH>(db, f, off, len);
    else { if (db.bsh) probe_t<2>(db, f, off, len); else probe_t<0>(db, f, off, len); }
}
// (the owner-side lookup of the sharded path walks four probes at a time: its own loop, on these)
__device__ __forceinline__ uint4 bucket_head(const DbDev& db, u32 idx) { return db.slots[(u64)idx << db.bsh]; }
__device__ __forceinline__ void bucket_list(const DbDev& db, u32 idx, const uint4& sl, u64& off, u32& len) {
    if (db.bsh) bucket_list_t<2>(db, idx, sl, off, len); else bucket_list_t<0>(db, idx, sl, off, len);
}

// ------------------------------------------------------------------ row 8: sort
// Bitonic sort of buf[0..n) (n a power of two) by G cooperating threads.
template <class KeyT, class Sync>
__device__ __forceinline__ void bitonic_sort(KeyT* buf, u32 n, u32 tid, u32 G, Sync sync) {
    for (u32 k = 2; k <= n; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 t = tid; t < (n >> 1); t += G) {
                u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                u32 l = i | j;
                KeyT a = buf[i], b = buf[l];
                bool up = (i & k) == 0;
                if ((a > b) == up) { buf[i] = b; buf[l] = a; }
            }
            sync();
        }
    }
}

// ------------------------------------------------------------------ row 8 in registers
// Bitonic network over n = 64*E keys held as r[e] = element e*64 + lane.  Exchanges at
// distance j < 64 cross lanes (DPP / swizzle / permlane32_swap, no LDS memory); distance
// j >= 64 pairs two registers of the same lane.  Fully unrolled: every index is static.
template <int J>
__device__ __forceinline__ u32 xlane(u32 v, u32 lane) { return xor_lane<J>(v, lane); }
template <int J>
__device__ __forceinline__ u64 xlane(u64 v, u32 lane) {
    return ((u64)xor_lane<J>((u32)(v >> 32), lane) << 32) | xor_lane<J>((u32)v, lane);
}

// cmpex with the lane mask as a compile-time constant: its two halves are 32-bit literals of the s_xor, so no
// SGPR pair per mask stays live (the compiler otherwise materialises all masks of a sort up front and spills them)
template <u64 KM>
__device__ __forceinline__ u64 take_mask(u64 lt) {
    u32 lo, hi;
    asm("s_xor_b32 %0, %2, %4\n\ts_xor_b32 %1, %3, %5"
        : "=&s"(lo), "=s"(hi) : "s"((u32)lt), "s"((u32)(lt >> 32)), "i"((int)(u32)KM), "i"((int)(u32)(KM >> 32)) : "scc");
    return ((u64)hi << 32) | lo;
}
template <u64 KM>
__device__ __forceinline__ u32 cmpex_c(u32 v, u32 o) {
    const u64 take = take_mask<KM>(__ballot(v < o));
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(o), "s"(take));
    return r;
}
template <u64 KM>
__device__ __forceinline__ u64 cmpex_c(u64 v, u64 o) {
    const u64 take = take_mask<KM>(__ballot(v < o));
    u32 lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((u32)v), "v"((u32)o), "s"(take));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((u32)(v >> 32)), "v"((u32)(o >> 32)), "s"(take));
    return ((u64)hi << 32) | lo;
}
// cross-lane stage J < 64 of level K for register e .. E-1 (e is a template parameter: the mask is a constant)
template <class KeyT, int E, int K, int J, int e>
__device__ __forceinline__ void regsort_xstage(KeyT (&r)[E], u32 lane) {
    if constexpr (e < E) {
        // bit 6+ of the element index lives in e, so for K >= 64 the direction is static
        constexpr u64 km = (K >= 64) ? ((((e * 64) & K) == 0) ? keepmin_mask(64, J, true) : keepmin_mask(64, J, false))
                                     : keepmin_mask(K, J, true);
        r[e] = cmpex_c<km>(r[e], xlane<J>(r[e], lane));
        regsort_xstage<KeyT, E, K, J, e + 1>(r, lane);
    }
}

template <class KeyT, int E, int K, int J>
__device__ __forceinline__ void regsort_stage(KeyT (&r)[E], u32 lane) {
    if constexpr (J >= 64) {
        constexpr int D = J / 64;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if ((e & D) == 0) {
                const bool up = ((e * 64) & K) == 0;            // K >= 128 here: lane bits do not matter
                KeyT a = r[e], b = r[e | D];
                const bool sw = (a > b) == up;
                r[e] = sw ? b : a; r[e | D] = sw ? a : b;
            }
        }
    } else {
#ifdef MCQ_CMPEX_GENERIC        // tuning knob: masks as SGPR constants chosen by the compiler
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u64 km = (K >= 64) ? ((((e * 64) & K) == 0) ? keepmin_mask(64, J, true) : keepmin_mask(64, J, false))
                                     : keepmin_mask(K, J, true);
            r[e] = cmpex(r[e], xlane<J>(r[e], lane), km);
        }
#else
        regsort_xstage<KeyT, E, K, J, 0>(r, lane);
#endif
    }
}
template <class KeyT, int E, int K, int J>
__device__ __forceinline__ void regsort_merge(KeyT (&r)[E], u32 lane) {
    regsort_stage<KeyT, E, K, J>(r, lane);
    if constexpr (J > 1) regsort_merge<KeyT, E, K, J / 2>(r, lane);
}
// the register-pairing stages J = J0 .. 64 of level K
template <int E, int K, int J>
__device__ __forceinline__ void regsort_merge_regs(u32 (&r)[E], u32 lane) {
    regsort_stage<u32, E, K, J>(r, lane);
    if constexpr (J > 64) regsort_merge_regs<E, K, J / 2>(r, lane);
}
template <class KeyT, int E, int K>
__device__ __forceinline__ void regsort_levels(KeyT (&r)[E], u32 lane) {
    if constexpr (K > 2) regsort_levels<KeyT, E, K / 2>(r, lane);
    regsort_merge<KeyT, E, K, K / 2>(r, lane);
}
// 32-bit keys: levels 2..64 are wave_sort64's min/max network per register, and the lane stages of every later level
// its ascending half-cleaners (2-3 VALU per stage instead of compare + mask + select); a register that has to come
// out descending is complemented around them.  Stages at distance >= 64 pair registers as in the generic network.
template <int E, int K>
__device__ __forceinline__ void regsort_levels_u32(u32 (&r)[E], u32 lane) {
    if constexpr (K > 128) regsort_levels_u32<E, K / 2>(r, lane);
    if constexpr (K >= 256) regsort_merge_regs<E, K, K / 2>(r, lane);
    else regsort_stage<u32, E, K, 64>(r, lane);
#pragma unroll
    for (int e = 0; e < E; e += 2) {                  // (E is even: two registers side by side, see cx_chain6_x2)
        const bool up0 = ((e * 64) & K) == 0, up1 = (((e + 1) * 64) & K) == 0;
        u32 v0 = up0 ? r[e] : ~r[e], v1 = up1 ? r[e + 1] : ~r[e + 1];
        cx_chain6_x2(v0, v1);
        r[e] = up0 ? v0 : ~v0; r[e + 1] = up1 ? v1 : ~v1;
    }
}
#ifndef MCQ_REGSORT_U32_MIN_E
#define MCQ_REGSORT_U32_MIN_E 2         // tuning knob: fewest registers per lane that take the min/max form
#endif
template <class KeyT, int E>
__device__ __forceinline__ void wave_regsort(KeyT (&r)[E], u32 lane) {
#ifndef MCQ_REGSORT_GENERIC     // tuning knob (A/B)
    if constexpr (sizeof(KeyT) == 4 && E >= MCQ_REGSORT_U32_MIN_E) {
        static_assert(E % 2 == 0, "registers are sorted in pairs");
#pragma unroll
        for (int e = 0; e < E; e += 2) {
            u32 v0 = r[e], v1 = ~r[e + 1];
            wave_sort64_x2(v0, v1);
            r[e] = v0; r[e + 1] = ~v1;
        }
        regsort_levels_u32<E, 64 * E>(r, lane);
    } else
#endif
    regsort_levels<KeyT, E, 64 * E>(r, lane);
}

// ---- workgroup bitonic sort with the wave-local stages in registers --------------------------------
// All stages at distance j <= 64 stay inside a 128-element chunk, so a wave takes them for its chunks in
// registers (2 keys per lane) without workgroup barriers: levels k <= 128 are a full 128-key register sort
// and of every later level only the stages j >= 128 go through LDS with a barrier each.  n = 4096: 21 barriers
// instead of 78.
template <class KeyT, int J>
__device__ __forceinline__ KeyT cx_dir(KeyT v, u32 lane, bool up) {
    const u64 km = up ? keepmin_mask(64, J, true) : keepmin_mask(64, J, false);
    return cmpex(v, xlane<J>(v, lane), km);
}
template <class KeyT>
__device__ __forceinline__ void merge128(KeyT& r0, KeyT& r1, u32 lane, bool up) {
    { const KeyT a = r0, b = r1; const bool sw = (a > b) == up; r0 = sw ? b : a; r1 = sw ? a : b; }
#ifndef MCQ_REGSORT_GENERIC
    if constexpr (sizeof(KeyT) == 4) {          // min/max half-cleaners (ascending; complemented keys for descending)
        u32 a = up ? (u32)r0 : ~(u32)r0, b = up ? (u32)r1 : ~(u32)r1;
        cx_chain6_x2(a, b);
        r0 = (KeyT)(up ? a : ~a); r1 = (KeyT)(up ? b : ~b);
        return;
    }
#endif
    r0 = cx_dir<KeyT, 32>(r0, lane, up); r1 = cx_dir<KeyT, 32>(r1, lane, up);
    r0 = cx_dir<KeyT, 16>(r0, lane, up); r1 = cx_dir<KeyT, 16>(r1, lane, up);
    r0 = cx_dir<KeyT, 8>(r0, lane, up);  r1 = cx_dir<KeyT, 8>(r1, lane, up);
    r0 = cx_dir<KeyT, 4>(r0, lane, up);  r1 = cx_dir<KeyT, 4>(r1, lane, up);
    r0 = cx_dir<KeyT, 2>(r0, lane, up);  r1 = cx_dir<KeyT, 2>(r1, lane, up);
    r0 = cx_dir<KeyT, 1>(r0, lane, up);  r1 = cx_dir<KeyT, 1>(r1, lane, up);
}
// "Flip" form: every comparator is ascending (the first stage of a K-block merge pairs i with its mirror
// image in the block, the later ones i with i + j), so keys that are +infinity never move and every comparator
// whose upper index is at or beyond the padded length can be left out: the list is padded to the next multiple
// of 128 only (npad), not to n2p -- a 2430-entry list does 60 % of the work of the 4096 network.
template <class KeyT, class Sync>
__device__ __forceinline__ void bitonic_sort_block(KeyT* buf, u32 n2p, u32 npad, u32 tid, u32 G, Sync sync) {
    if (n2p < 256) { bitonic_sort(buf, n2p, tid, G, sync); return; }      // (then npad == n2p)
    const u32 lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = G >> 6, nchunks = npad >> 7;
    for (u32 c = wv; c < nchunks; c += nwv) {                       // levels 2..128
        KeyT r[2] = {buf[c * 128 + lane], buf[c * 128 + 64 + lane]};
        wave_regsort<KeyT, 2>(r, lane);
        buf[c * 128 + lane] = r[0]; buf[c * 128 + 64 + lane] = r[1];
    }
    sync();
    for (u32 k = 256; k <= n2p; k <<= 1) {
        {                                                           // mirror stage of the k-blocks
            const u32 h = k >> 1, hs = (u32)__builtin_ctz(h);
            const u32 tmax = ((npad + k - 1) >> (hs + 1)) << hs;    // blocks that hold real keys
            for (u32 t = tid; t < tmax; t += G) {
                const u32 blk = t >> hs, o = t & (h - 1);
                const u32 i = (blk << (hs + 1)) + o, l = (blk << (hs + 1)) + (k - 1) - o;
                if (l >= npad) continue;
                const KeyT a = buf[i], b = buf[l];
                if (a > b) { buf[i] = b; buf[l] = a; }
            }
            sync();
        }
        for (u32 j = k >> 2; j >= 128; j >>= 1) {
            const u32 tmax = ((npad >> 1) + j - 1) & ~(j - 1);      // i >= npad from here on
            for (u32 t = tid; t < tmax; t += G) {
                const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const u32 l = i | j;
                if (l >= npad) continue;
                const KeyT a = buf[i], b = buf[l];
                if (a > b) { buf[i] = b; buf[l] = a; }
            }
            sync();
        }
        for (u32 c = wv; c < nchunks; c += nwv) {                   // stages j = 64..1 of this level
            KeyT r0 = buf[c * 128 + lane], r1 = buf[c * 128 + 64 + lane];
            merge128<KeyT>(r0, r1, lane, true);
            buf[c * 128 + lane] = r0; buf[c * 128 + 64 + lane] = r1;
        }
        sync();
    }
}

// ------------------------------------------------------------------ row 9: per-target best window range
// buf[0..T) sorted by (tgt,win).  The reference's two-pointer sweep keeps, for the run of
// one target, the first range [fst,lst] with the most entries and win[lst]-win[fst] <
// numWindows.  Because the run is sorted, fst(lst) is simply the first entry of the run
// with win >= win[lst]-numWindows+1, so every entry finds its own count with one binary
// search: hits(j) = j - lower_bound(tgt, win_j - numWindows + 1) + 1.  The run's best is
// the maximum of (hits, -j): folded with an LDS atomic max into H[first entry of the run].
//   H[j0] = (hits << JB) | (JMASK - jbest) for run heads, 0 elsewhere.
// PRE (H in global memory): the entries of a run are neighbours, so a wave first takes the maximum over each run's entries among its
// 64 (segmented prefix maximum, six shuffle steps) and only the last lane of a run's segment issues the atomic -- a read's true
// targets have runs of thousands of entries, which would otherwise queue on one address in L2.
template <class KeyT, class HT, int JB, class LF, bool PRE = false, class Sync>
__device__ __forceinline__ void sweep_targets(const KeyT* buf, HT* H, u32 T, u32 numWindows, const LF& lf, u32 tid, u32 G,
                                              u32* s_w /* G / 64 + 1 words of LDS */, Sync sync) {
    // The head of an entry's run: last run start at or before it -- a ballot of run starts inside the wave, the last
    // start of the earlier waves through s_w, of the earlier chunks through s_w[nwv] (all as index + 1, 0 = none).
    // The lower bound is then searched inside [head, j] only: runs are a few entries long.
    const HT JMASK = ((HT)1 << JB) - 1;
    const u32 lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = G >> 6;
    for (u32 j = tid; j < T; j += G) H[j] = 0;
    if (tid == 0) s_w[nwv] = 0;
    sync();
    for (u32 base = 0; base < T; base += G) {
        const u32 j = base + tid;
        const bool valid = j < T;
        const KeyT key = buf[valid ? j : T - 1];
        const KeyT prev = buf[(valid && j > 0) ? j - 1 : 0];
        const KeyT tb = lf.tbeg(key);
        const bool head = valid && (j == 0 || prev < tb);
        const u64 hb = __ballot(head);
        const u32 wbase = base + wv * 64;
        const u32 mylast = hb ? wbase + (63u - (u32)__builtin_clzll(hb)) + 1 : 0u;
        if (lane == 0) s_w[wv] = mylast;
        sync();
        u32 before = s_w[nwv];
        for (u32 w = 0; w < wv; ++w) { const u32 x = s_w[w]; before = x ? x : before; }
        const u64 le = hb & ((2ull << lane) - 1);
        const u32 myhead = le ? wbase + (63u - (u32)__builtin_clzll(le)) : before - 1;
        const KeyT lowkey = range_low<KeyT>(key, tb, numWindows);
        u32 lo = myhead, hi = valid ? j : myhead;
        while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (buf[mid] < lowkey) lo = mid + 1; else hi = mid; }
        if constexpr (PRE) {
            HT v = valid ? (((HT)(j - lo + 1) << JB) | (JMASK - (HT)j)) : (HT)0;
            const u32 hd = valid ? myhead : 0xFFFFFFFFu;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const HT o = (HT)__shfl_up(v, d, 64);
                const u32 oh = (u32)__shfl_up(hd, d, 64);
                if (lane >= (u32)d && oh == hd && o > v) v = o;
            }
            const u32 nh = (u32)__shfl_down(hd, 1, 64);
            if (valid && (lane == 63 || nh != hd)) atomicMax(&H[myhead], v);
        } else
        if (valid) atomicMax(&H[myhead], ((HT)(j - lo + 1) << JB) | (JMASK - (HT)j));
        sync();                                                     // s_w has been read by everyone
        if (tid == G - 1) s_w[nwv] = mylast ? mylast : before;      // last run start so far
    }
    sync();
}

// Wave version (T <= 2^JB, packed into u32): the head of an entry's run comes
// from a ballot of run starts in its 64-chunk (carried across chunks), and the single
// binary search is confined to [head, j].
template <class KeyT, int JB = 9, class LF>
__device__ __forceinline__ void sweep_targets_wave(const KeyT* buf, u32* H, u32 T, u32 numWindows, const LF& lf, u32 lane) {
    for (u32 j = lane; j < T; j += 64) H[j] = 0;
    wave_sync();
    u32 carry_head = 0;
    for (u32 base = 0; base < T; base += 64) {
        const u32 j = base + lane;
        const bool valid = j < T;
        const KeyT key = buf[valid ? j : T - 1];
        const KeyT prev = buf[(valid && j > 0) ? j - 1 : 0];
        const KeyT tb = lf.tbeg(key);
        const bool head = valid && (j == 0 || prev < tb);
        const u64 le = __ballot(head) & ((2ull << lane) - 1);
        const u32 myhead = le ? base + (63u - (u32)__builtin_clzll(le)) : carry_head;
        const KeyT lowkey = range_low<KeyT>(key, tb, numWindows);
        u32 lo = myhead, hi = valid ? j : myhead;
        while (lo < hi) { u32 mid = (lo + hi) >> 1; if (buf[mid] < lowkey) lo = mid + 1; else hi = mid; }
        if (valid) atomicMax(&H[myhead], ((j - lo + 1) << JB) | (((1u << JB) - 1) - j));
        carry_head = bcast(myhead, 63);
    }
    wave_sync();
}

// inclusive prefix sum over the wave on the VALU only: Hillis-Steele inside each row of 16 (row_shr
// 1,2,4,8), then the row totals ripple through row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3)
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

// inclusive prefix MAXIMUM over the wave, same DPP ladder as wave_incl_scan_dpp (values >= 0, identity 0)
__device__ __forceinline__ u32 wave_incl_max_dpp(u32 v) {
    u32 t;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;
    return v;
}

// Weighted form of sweep_targets_wave for a de-duplicated match list: SK[0..D) are the DISTINCT
// sorted keys, WP[j] the inclusive prefix sum of their multiplicities.  A window range holds every
// copy of the keys inside it, so hits(j) = WP[j] - WP[lo-1] with the same lower bound lo, and the
// first range reaching the maximum ends at the same key as in the multiset sweep (the maximum over
// the copies of one key is at its last copy; earlier keys still win ties).  Entry indices are now
// indices of distinct keys: a monotone relabelling, so every later tie-break is unchanged.
template <class LF>
__device__ __forceinline__ void sweep_targets_weighted(const u32* SK, const u32* WP, u32* H, u32 D, u32 numWindows,
                                                       const LF& lf, u32 lane) {
    for (u32 j = lane; j < D; j += 64) H[j] = 0;
    wave_sync();
    u32 carry_head = 0;
    for (u32 base = 0; base < D; base += 64) {
        const u32 j = base + lane;
        const bool valid = j < D;
        const u32 key = SK[valid ? j : D - 1];
        const u32 prev = SK[(valid && j > 0) ? j - 1 : 0];
        const u32 tb = lf.tbeg(key);
        const bool head = valid && (j == 0 || prev < tb);
        const u64 le = __ballot(head) & ((2ull << lane) - 1);
        const u32 myhead = le ? base + (63u - (u32)__builtin_clzll(le)) : carry_head;
        const u32 lowkey = range_low<u32>(key, tb, numWindows);
        u32 lo = myhead, hi = valid ? j : myhead;
        while (lo < hi) { u32 mid = (lo + hi) >> 1; if (SK[mid] < lowkey) lo = mid + 1; else hi = mid; }
        if (valid) {
            const u32 h = WP[j] - (lo ? WP[lo - 1] : 0u);
            atomicMax(&H[myhead], (h << 9) | (511u - j));
        }
        carry_head = bcast(myhead, 63);
    }
    wave_sync();
}

// The same for at most 64 distinct keys held one per lane (k sorted, lanes >= D padded; incl = inclusive
// multiplicity sums) and a narrow window range: distinct keys of one target have distinct windows, so the range
// ending at an entry reaches back over at most numWindows - 1 predecessors -- checked with wave_shr:1 shifts
// instead of a binary search through LDS.  numWindows <= 8.
// tb_in (formats with LF::lookup only): the first word of the lane's target, looked up by the caller -- who keeps the
// target for the top lists; a format whose tbeg is arithmetic computes it here (nothing extra stays live)
template <class LF>
__device__ __forceinline__ void sweep_targets_regs(u32 k, u32 incl, u32 tb_in, u32* H, u32 D, u32 numWindows, const LF& lf, u32 lane) {
    H[lane] = 0;
    wave_sync();
    const bool valid = lane < D;
    asm("s_nop 1" : "+v"(k));                     // k may come straight out of an asm sort block (DPP read hazard)
    const u32 prev = (u32)__builtin_amdgcn_update_dpp(0, (int)k, 0x138, 0xF, 0xF, false);      // wave_shr:1
    u32 tb = tb_in;
    if constexpr (!LF::lookup) tb = lf.tbeg(k);
    const bool head = valid && (lane == 0 || prev < tb);
    const u64 le = __ballot(head) & ((2ull << lane) - 1);
    const u32 myhead = le ? 63u - (u32)__builtin_clzll(le) : 0u;
    const u32 lowkey = range_low<u32>(k, tb, numWindows);
    u32 cnt = 0, kk = k;
    bool ok = valid;
    for (u32 i = 1; i < numWindows; ++i) {
        kk = (u32)__builtin_amdgcn_update_dpp(0, (int)kk, 0x138, 0xF, 0xF, false);
        ok = ok && lane >= i && kk >= lowkey;
        cnt += ok ? 1u : 0u;
    }
    const u32 below = __shfl(incl, (int)((lane - cnt - 1) & 63), 64);                           // WP[lo - 1]
    const u32 hits = incl - (lane > cnt ? below : 0u);
    if (valid) atomicMax(&H[myhead], (hits << 9) | (511u - lane));
    wave_sync();
}

// window range [beg,end] of the best candidate whose packed word is hv (run head j0 irrelevant)
template <class KeyT, class HT, int JB, class LF>
__device__ __forceinline__ void best_range(const KeyT* buf, HT hv, u32 numWindows, const LF& lf, u32& beg, u32& end) {
    const HT JMASK = ((HT)1 << JB) - 1;
    const u32 j = (u32)(JMASK - (hv & JMASK));
    const KeyT key = buf[j];
    const KeyT tb = lf.tbeg(key);
    const KeyT lowkey = range_low<KeyT>(key, tb, numWindows);
    u32 lo = 0, hi = j;
    while (lo < hi) { u32 mid = (lo + hi) >> 1; if (buf[mid] < lowkey) lo = mid + 1; else hi = mid; }
    beg = (u32)(buf[lo] - tb); end = (u32)(key - tb);
}

// ------------------------------------------------------------------ rows 10-11: top lists in lanes
// The reference builds each bounded top list by inserting candidates one at a time
// (src/candidates.h:236-285); the result is order-dependent (max per taxon, evictions,
// stable ties).  It has a closed form (tests/test_toplist_theorem.py checks it against
// the restated insert()):
//     list = first M of { (taxon, max hits, first candidate reaching that max) }
//            ordered by (hits descending, that candidate's position ascending)
// because the minimum of a full list never decreases and equal-hit entries keep the order
// in which they reached their value.  So a list is built by M rounds of "wave-wide max of
// the packed (hits, -position) word, record it, retire every candidate of that taxon" --
// no serial insert loop.  A tree-fold step (src/querying.h:910-971) is the same closed form
// on the concatenation receiver-list ++ sender-list.
//
// The P virtual-rank lists live side by side in one wave: list r occupies lanes
// [r*seg, r*seg + M).  An entry is (tax, hv) with hv the packed sweep word
// (hits << JB | JMASK - jbest); hv == 0 marks an unused lane.
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
    u32 t;
    t = dpp_mov<0xB1>(v);  v = t > v ? t : v;
    t = dpp_mov<0x4E>(v);  v = t > v ? t : v;
    t = dpp_mov<0x141>(v); v = t > v ? t : v;
    t = dpp_mov<0x140>(v); v = t > v ? t : v;
    u32 a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    u32 c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    a = a > b ? a : b; c = c > d ? c : d;
    return a > c ? a : c;
}
__device__ __forceinline__ u32 wave_max(u32 v) { return wave_max_u32(v); }
__device__ __forceinline__ u64 wave_max(u64 v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { u64 o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

template <class KeyT, class HT, int JB, class LF>
__device__ __forceinline__ u32 topk_fold_write(const DbDev& db, const OptDev& opt, const OutDev& out,
                                               const KeyT* buf, HT* H, u32 T, u32 numWindows, const LF& lf,
                                               u64 q, u32 lane) {
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const bool p2 = (P & (P - 1)) == 0;
    const HT JMASK = ((HT)1 << JB) - 1;

    // 1. compact the run heads' packed words to H[0..nheads) (in place: writes trail reads)
    u32 nheads = 0;
    for (u32 base = 0; base < T; base += 64) {
        const u32 j = base + lane;
        const HT hv = (j < T) ? H[j] : 0;
        const u64 hm = __ballot(hv != 0);
        if (hv != 0) H[nheads + lane_rank(hm)] = hv;
        nheads += (u32)__builtin_popcountll(hm);
    }
    wave_sync();

    // 2. stream the candidates through the lists, 64 at a time
    const u32 rl = lane / seg, li = lane - rl * seg;
    const bool lslot = (li < M) && (rl < P);
    u32 Ltax = MCQ_EMPTY; HT Lhv = 0;
    for (u32 base = 0; base < nheads; base += 64) {
        const u32 k = base + lane;
        HT cv = (k < nheads) ? H[k] : 0;
        const u32 jb = (u32)(JMASK - (cv & JMASK));
        const u32 tgt = lf.tgt(buf[cv ? jb : 0]);
        u32 ctax = MCQ_EMPTY;
        if (cv != 0 && tgt < db.n_targets) ctax = db.tgt2tax[tgt];
        if (ctax == MCQ_EMPTY) cv = 0;
        const u32 cr = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
        u32 Ntax = MCQ_EMPTY; HT Nhv = 0;
        bool lalive = Lhv != 0;
        for (u32 r = 0; r < P; ++r) {
            for (u32 i = 0; i < M; ++i) {
                const HT a = (cr == r) ? cv : 0;
                const HT b2 = (lalive && lslot && rl == r) ? Lhv : 0;
                const HT v = a > b2 ? a : b2;
                const HT m = wave_max(v);
                if (m == 0) break;
                const u32 src = (u32)__builtin_ctzll(__ballot(v == m));
                const u32 wtax = bcast((a == m) ? ctax : Ltax, src);
                if (lane == r * seg + i) { Ntax = wtax; Nhv = m; }
                if (cr == r && ctax == wtax) cv = 0;
                if (lslot && rl == r && Ltax == wtax) lalive = false;
            }
        }
        Ltax = Ntax; Lhv = Nhv;
    }

    // 3. tree fold: receiver-list ++ sender-list, positions decide ties
    if (P > 1) {
        for (u32 f = 0; f < opt.n_fold; ++f) {
            const u32 snd = opt.fold_snd[f], rcv = opt.fold_rcv[f];
            const bool mine = lslot && (rl == rcv || rl == snd) && Lhv != 0 &&
                              !(opt.quirk_seq_drop && rl == snd && (Ltax & 0x80000000u));
            u32 fk = mine ? (((u32)(Lhv >> JB)) << 8) | (255u - (rl == snd ? M + li : li)) : 0u;
            u32 Ntax = MCQ_EMPTY; HT Nhv = 0;
            for (u32 i = 0; i < M; ++i) {
                const u32 m = wave_max_u32(fk);
                if (m == 0) break;
                const u32 src = (u32)__builtin_ctzll(__ballot(fk == m));
                const u32 wtax = bcast(Ltax, src);
                if (lane == rcv * seg + i) { Ntax = wtax; Nhv = (HT)(m >> 8) << JB; }
                if (Ltax == wtax) fk = 0;
            }
            if (lslot && (rl == rcv || rl == snd)) { Ltax = Ntax; Lhv = Nhv; }
        }
    }

    // list 0 is the result
    const u32 n = (u32)__builtin_popcountll(__ballot(lane < M && Lhv != 0));
    if (lane < n) {
        u32 beg = 0, end = 0;
        if (P == 1) best_range<KeyT, HT, JB>(buf, Lhv, numWindows, lf, beg, end);
        uint4 v; v.x = Ltax; v.y = (u32)(Lhv >> JB); v.z = beg; v.w = end;
        u32 ln = lane;
        asm volatile("" : "+v"(ln));               // keeps (cands + 16 * lane) from being hoisted out of the query loop and spilled
        reinterpret_cast<uint4*>(out.cands)[q * M + ln] = v;
    }
    if (lane == 0) out.ncand[q] = n;
    return n;
}

// ---- rows 10-11 as ONE selection (OptDev::lin) ----------------------------------------------------------------------
// The P bounded lists and their tree fold (src/querying.h:867-1073) lose nothing the final M entries could contain: an
// entry that misses its rank's list has M better distinct taxa (or a better entry of its own taxon) in that rank alone, a
// fold step is the same bounded insert on receiver-list ++ sender-list, and the tree concatenates the ranks in ascending
// order (the ranks >= 2^floor(log2 P) of a non-power-of-two P never reach rank 0).  So the folded list is the first M
// distinct taxa of ALL candidates of the ranks < keep in the order (hits descending, rank ascending, position ascending):
// tests/test_toplist_theorem.py checks exactly this against the restated insert() + tree.  The rank goes into the packed
// word between hits and position (6 bits), M selection rounds per 64 run heads with DPP maxima, no lists, no fold levels
// -- the same cost for -n 64 as for -n 2.  Not under MCQ_QUIRK_SEQ_DROP on a table with sequence-level taxa (a dropped entry
// has held a slot of an intermediate list: the lists and the levels are carried out then).
#ifdef MCQ_LIN_ONLY                     // tuning knob (A/B): the wave kernels without the code of the lists (wrong results where they are needed)
#define MCQ_OPT_LIN(opt) true
#else
#define MCQ_OPT_LIN(opt) ((opt).lin != 0)
#endif
template <class HT, int JB>
__device__ __forceinline__ HT lin_word(HT cv, u32 cr) {
    return ((cv >> JB) << (JB + 6)) | ((HT)(63u - cr) << JB) | (cv & (((HT)1 << JB) - 1));
}
__device__ __forceinline__ u32 lin_rank(const OptDev& opt, u32 tgt) {
    const u32 P = opt.P;
    return (P & (P - 1)) == 0 ? (tgt & (P - 1)) : (tgt % P);
}
// H[0..nheads): the compacted packed words (hits << JB | JMASK - j) of the run heads, j indexing buf; REGT as below
template <class KeyT, int JB, bool REGT = false, class LF>
__device__ __forceinline__ u32 topk_lin_write(const DbDev& db, const OptDev& opt, const OutDev& out, const KeyT* buf, const u32* H, u32 nheads,
                                              const LF& lf, u64 q, u32 lane, u32 t1 = 0) {
    const u32 M = opt.max_cand;
    const u32 JMASK = (1u << JB) - 1;
    u32 Ltax = MCQ_EMPTY, Lk = 0;                       // lanes < M: the list so far (linear words)
    for (u32 base = 0; base < nheads; base += 64) {
        const u32 k = base + lane;
        const u32 cv = (k < nheads) ? H[k] : 0;
        const u32 jb = JMASK - (cv & JMASK);
        u32 tgt;
        if constexpr (REGT) tgt = __shfl(t1, (int)(cv ? jb : 0), 64);
        else tgt = lf.tgt(buf[cv ? jb : 0]);
        u32 ctax = MCQ_EMPTY;
        if (cv != 0 && tgt < db.n_targets) ctax = db.tgt2tax[tgt];
        const u32 cr = lin_rank(opt, tgt);
        u32 ck = (ctax == MCQ_EMPTY || cr >= opt.keep) ? 0u : lin_word<u32, JB>(cv, cr);
        u32 lk = Lk, Ntax = MCQ_EMPTY, Nk = 0;
        for (u32 i = 0; i < M; ++i) {
            const u32 v = ck > lk ? ck : lk;
            const u32 m = wave_max_u32(v);
            if (m == 0) break;
            const u32 src = (u32)__builtin_ctzll(__ballot(v == m));       // linear words are unique: one winner
            const u32 wtax = bcast(ck == m ? ctax : Ltax, src);
            if (lane == i) { Ntax = wtax; Nk = m; }
            if (ctax == wtax) ck = 0;                                     // every head of the winner's taxon retires
            if (Ltax == wtax) lk = 0;
        }
        Ltax = Ntax; Lk = Nk;
    }
    const u32 n = (u32)__builtin_popcountll(__ballot(lane < M && Lk != 0));
    if (lane < n) {
        u32 ln = lane;
        asm volatile("" : "+v"(ln));               // (see topk_fold_write)
        reinterpret_cast<uint4*>(out.cands)[q * M + ln] = make_uint4(Ltax, Lk >> (JB + 6), 0u, 0u);     // no window ranges after a fold
    }
    if (lane == 0) out.ncand[q] = n;
    return n;
}
// lexicographic wave maximum of (hi, lo) pairs: two DPP reductions
__device__ __forceinline__ unsigned long long wave_max_pair(unsigned long long v) {
    const u32 hi = wave_max_u32((u32)(v >> 32));
    const u32 lo = wave_max_u32((u32)(v >> 32) == hi ? (u32)v : 0u);
    return ((unsigned long long)hi << 32) | lo;
}

// Tree fold of the P virtual-rank lists held one entry per lane (list r in lanes [r*seg, r*seg + M)) and the
// final write of list 0: one round of the tree at a time -- its edges touch disjoint ranks, so every receiver
// selects from receiver-list ++ sender-list in the same M rounds (positions decide ties).  One wave; mx, wt:
// 64 words of LDS each.
template <class KeyT, class HT, int JB, class LF>
__device__ __forceinline__ u32 fold_lists_write(const DbDev& db, const OptDev& opt, const OutDev& out, const KeyT* buf,
                                                u32 Ltax, HT Lhv, u32 numWindows, const LF& lf, u64 q, u32 lane, u32* mx, u32* wt) {
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const u32 rl = lane / seg, li = lane - rl * seg;
    const bool lslot = (li < M) && (rl < P);
    if (P > 1) {
        u32 lb = 0;
        for (u32 L = 0; L < opt.n_levels; ++L) {
            const u32 le = opt.level_end[L];
            u32 my_rcv = 63; bool part = false, is_snd = false;
            for (u32 e = lb; e < le; ++e) {
                const u32 snd = opt.fold_snd[e], rcv = opt.fold_rcv[e];
                if (rl == snd) { my_rcv = rcv; part = true; is_snd = true; }
                if (rl == rcv) { my_rcv = rcv; part = true; }
            }
            lb = le;
            part = part && lslot;
            const bool mine = part && Lhv != 0 && !(opt.quirk_seq_drop && is_snd && (Ltax & 0x80000000u));
            u32 fk = mine ? ((u32)(Lhv >> JB) << 8) | (255u - (is_snd ? M + li : li)) : 0u;
            u32 Ntax = MCQ_EMPTY; HT Nhv = 0;
            for (u32 i = 0; i < M; ++i) {
                mx[lane] = 0;
                wave_sync();
                if (fk != 0) atomicMax(&mx[my_rcv], fk);
                wave_sync();
                const u32 m = mx[my_rcv];
                if (fk != 0 && fk == m) wt[my_rcv] = Ltax;
                wave_sync();
                const u32 wtax = wt[my_rcv];
                if (part && m != 0) {
                    if (!is_snd && li == i) { Ntax = wtax; Nhv = (HT)(m >> 8) << JB; }
                    if (fk != 0 && Ltax == wtax) fk = 0;
                }
                wave_sync();
            }
            if (part) { Ltax = Ntax; Lhv = Nhv; }
        }
    }

    // list 0 is the result
    const u32 n = (u32)__builtin_popcountll(__ballot(lane < M && Lhv != 0));
    if (lane < n) {
        u32 beg = 0, end = 0;
        if (P == 1) best_range<KeyT, HT, JB>(buf, Lhv, numWindows, lf, beg, end);
        uint4 v; v.x = Ltax; v.y = (u32)(Lhv >> JB); v.z = beg; v.w = end;
        u32 ln = lane;
        asm volatile("" : "+v"(ln));               // keeps (cands + 16 * lane) from being hoisted out of the query loop and spilled
        reinterpret_cast<uint4*>(out.cands)[q * M + ln] = v;
    }
    if (lane == 0) out.ncand[q] = n;
    return n;
}

// ---- rows 10-11 with the wave-wide maxima taken in LDS (dedup path: 32-bit keys, packed u32 words) ----
// Same closed form and the same lane layout of the P virtual-rank lists as topk_fold_write, but a selection
// round is one ds_max per candidate into the word of its rank instead of a DPP reduction per rank: all P
// ranks advance in the same round (M rounds per 64 candidates instead of P x M), and a round costs a dozen
// VALU instructions.  scr: 128 words of this wave's LDS segment (scr[0..64) maxima, scr[64..128) winner taxa).
// REGT (T <= 64 only): the target of entry j sits in lane j's t1 (the caller looked it up once, for the sweep) and
// comes by shuffle instead of a second lookup -- for formats whose target is a memory access away (LocGW)
template <int JB = 9, bool REGT = false, class LF>
__device__ __forceinline__ u32 topk_fold_write_lds(const DbDev& db, const OptDev& opt, const OutDev& out,
                                                   const u32* buf, u32* H, u32 T, u32 numWindows, const LF& lf,
                                                   u64 q, u32 lane, u32* scr, u32 t1 = 0) {
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const bool p2 = (P & (P - 1)) == 0;
    const u32 JMASK = (1u << JB) - 1;
    u32* mx = scr; u32* wt = scr + 64;

    // 1. compact the run heads' packed words to H[0..nheads)
    u32 nheads = 0;
    for (u32 base = 0; base < T; base += 64) {
        const u32 j = base + lane;
        const u32 hv = (j < T) ? H[j] : 0;
        const u64 hm = __ballot(hv != 0);
        if (hv != 0) H[nheads + lane_rank(hm)] = hv;
        nheads += (u32)__builtin_popcountll(hm);
    }
    wave_sync();

    // 2. stream the candidates through the lists, 64 at a time; every round serves all ranks
    const u32 rl = lane / seg, li = lane - rl * seg;
    const bool lslot = (li < M) && (rl < P);
    u32 Ltax = MCQ_EMPTY, Lhv = 0;
    for (u32 base = 0; base < nheads; base += 64) {
        const u32 k = base + lane;
        u32 cv = (k < nheads) ? H[k] : 0;
        const u32 jb = JMASK - (cv & JMASK);
        u32 tgt;
        if constexpr (REGT) tgt = __shfl(t1, (int)(cv ? jb : 0), 64);
        else tgt = lf.tgt(buf[cv ? jb : 0]);
        u32 ctax = MCQ_EMPTY;
        if (cv != 0 && tgt < db.n_targets) ctax = db.tgt2tax[tgt];
        if (ctax == MCQ_EMPTY) cv = 0;
        const u32 cr = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
        u32 Ntax = MCQ_EMPTY, Nhv = 0;
        bool lalive = lslot && Lhv != 0;
        for (u32 i = 0; i < M; ++i) {
            mx[lane] = 0;
            wave_sync();
            if (cv != 0) atomicMax(&mx[cr], cv);
            if (lalive) atomicMax(&mx[rl], Lhv);
            wave_sync();
            const u32 mc = mx[cr], ml = mx[rl];
            if (cv != 0 && cv == mc) wt[cr] = ctax;            // packed words are unique: one winner per rank
            if (lalive && Lhv == ml) wt[rl] = Ltax;
            wave_sync();
            const u32 wtc = wt[cr], wtl = wt[rl];
            if (lslot && li == i && ml != 0) { Ntax = wtl; Nhv = ml; }
            if (cv != 0 && ctax == wtc) cv = 0;                // mc != 0 here, so wt[cr] is this round's winner
            if (lalive && Ltax == wtl) lalive = false;
            wave_sync();
        }
        Ltax = Ntax; Lhv = Nhv;
    }

    return fold_lists_write<u32, u32, JB>(db, opt, out, buf, Ltax, Lhv, numWindows, lf, q, lane, mx, wt);
}

// ---- rows 10-11, all run heads at once (raw-sort paths: several hundred heads) -------------------------
// The compacted heads H[0..nheads), nheads <= 64 * NC, sit in registers (packed word with the virtual rank in bits 26+,
// taxon key), so the selection needs M rounds in all instead of M per 64 heads: per round one ds_max per live candidate
// into the word of its rank, the winners publish their taxa, every candidate of a winning taxon retires.  The
// closed form is the same (first M distinct taxa by hits descending, position ascending); the taxon keys of all
// heads are loaded in one go.  scr: 128 words of LDS.
template <int JB, int NC, class LF>
__device__ __forceinline__ u32 topk_all_lds(const DbDev& db, const OptDev& opt, const OutDev& out, const u32* buf,
                                            const u32* H, u32 nheads, u32 numWindows, const LF& lf, u64 q, u32 lane, u32* scr) {
    static_assert(JB <= 10, "hits << JB stays below bit 26");
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const bool p2 = (P & (P - 1)) == 0;
    const u32 JMASK = (1u << JB) - 1, VMASK = (1u << 26) - 1;
    u32* mx = scr; u32* wt = scr + 64;
    u32 cv[NC], ctax[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        cv[c] = 0; ctax[c] = MCQ_EMPTY;
        if ((u32)(c * 64) >= nheads) continue;               // wave-uniform
        const u32 k = c * 64 + lane;
        const u32 v = (k < nheads) ? H[k] : 0;
        const u32 tgt = lf.tgt(buf[v ? JMASK - (v & JMASK) : 0]);
        if (v != 0 && tgt < db.n_targets) ctax[c] = db.tgt2tax[tgt];
        const u32 cr = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
        cv[c] = v | (cr << 26);
        if (v == 0) cv[c] = 0;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) if (ctax[c] == MCQ_EMPTY) cv[c] = 0;
    const u32 rl = lane / seg, li = lane - rl * seg;
    const bool lslot = (li < M) && (rl < P);
    u32 Ltax = MCQ_EMPTY, Lhv = 0;
    for (u32 i = 0; i < M; ++i) {
        mx[lane] = 0;
        wave_sync();
#pragma unroll
        for (int c = 0; c < NC; ++c) if (cv[c] != 0) atomicMax(&mx[cv[c] >> 26], cv[c]);
        wave_sync();
#pragma unroll
        for (int c = 0; c < NC; ++c) if (cv[c] != 0 && cv[c] == mx[cv[c] >> 26]) wt[cv[c] >> 26] = ctax[c];   // packed words are unique
        wave_sync();
        const u32 ml = mx[rl], wtl = wt[rl];
        if (lslot && li == i && ml != 0) { Ltax = wtl; Lhv = ml & VMASK; }
#pragma unroll
        for (int c = 0; c < NC; ++c) if (cv[c] != 0 && ctax[c] == wt[cv[c] >> 26]) cv[c] = 0;
        wave_sync();
    }
    return fold_lists_write<u32, u32, JB>(db, opt, out, buf, Ltax, Lhv, numWindows, lf, q, lane, mx, wt);
}

// ---- rows 10-11 with more lists than lanes (the reference's mpiexec -n 32 / -n 64 with -maxcand 4) ---------------------------
// pow2ceil(P) x seg list slots, seg = pow2ceil(M) lanes per list, wrap over NL registers per lane: slot s = j * 64 + lane of
// register j belongs to list s / seg, entry s % seg.  Same selection rounds and the same fold as the one-register forms
// above (whose code the hot instantiations keep unchanged); NL = 4 covers P x M <= 256.
template <int NL>
struct ListSlots {
    u32 tax[NL]; u32 hv[NL];
};
template <class KeyT, int JB, int NL, class LF>
__device__ __forceinline__ u32 fold_lists_write_n(const DbDev& db, const OptDev& opt, const OutDev& out, const KeyT* buf, ListSlots<NL>& L,
                                                  u32 numWindows, const LF& lf, u64 q, u32 lane, u32* mx, u32* wt) {
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    u32 rl[NL], li[NL]; bool lslot[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) { const u32 s = j * 64 + lane; rl[j] = s / seg; li[j] = s - rl[j] * seg; lslot[j] = li[j] < M && rl[j] < P; }
    if (P > 1) {
        u32 lb = 0;
        for (u32 Lv = 0; Lv < opt.n_levels; ++Lv) {
            const u32 le = opt.level_end[Lv];
            u32 my_rcv[NL], fk[NL], Ntax[NL], Nhv[NL]; bool part[NL], is_snd[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { my_rcv[j] = 63; part[j] = false; is_snd[j] = false; Ntax[j] = MCQ_EMPTY; Nhv[j] = 0; }
            for (u32 e = lb; e < le; ++e) {
                const u32 snd = opt.fold_snd[e], rcv = opt.fold_rcv[e];
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    if (rl[j] == snd) { my_rcv[j] = rcv; part[j] = true; is_snd[j] = true; }
                    if (rl[j] == rcv) { my_rcv[j] = rcv; part[j] = true; }
                }
            }
            lb = le;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                part[j] = part[j] && lslot[j];
                const bool mine = part[j] && L.hv[j] != 0 && !(opt.quirk_seq_drop && is_snd[j] && (L.tax[j] & 0x80000000u));
                fk[j] = mine ? ((L.hv[j] >> JB) << 8) | (255u - (is_snd[j] ? M + li[j] : li[j])) : 0u;
            }
            for (u32 i = 0; i < M; ++i) {
                mx[lane] = 0;
                wave_sync();
#pragma unroll
                for (int j = 0; j < NL; ++j) if (fk[j] != 0) atomicMax(&mx[my_rcv[j]], fk[j]);
                wave_sync();
                u32 m[NL];
#pragma unroll
                for (int j = 0; j < NL; ++j) { m[j] = mx[my_rcv[j]]; if (fk[j] != 0 && fk[j] == m[j]) wt[my_rcv[j]] = L.tax[j]; }
                wave_sync();
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const u32 wtax = wt[my_rcv[j]];
                    if (part[j] && m[j] != 0) {
                        if (!is_snd[j] && li[j] == i) { Ntax[j] = wtax; Nhv[j] = (m[j] >> 8) << JB; }
                        if (fk[j] != 0 && L.tax[j] == wtax) fk[j] = 0;
                    }
                }
                wave_sync();
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) if (part[j]) { L.tax[j] = Ntax[j]; L.hv[j] = Nhv[j]; }
        }
    }
    // list 0 is the result: entries 0..M-1 = lanes 0..M-1 of register 0
    const u32 n = (u32)__builtin_popcountll(__ballot(lane < M && L.hv[0] != 0));
    if (lane < n) {
        u32 beg = 0, end = 0;
        if (P == 1) best_range<KeyT, u32, JB>(buf, L.hv[0], numWindows, lf, beg, end);
        reinterpret_cast<uint4*>(out.cands)[q * M + lane] = make_uint4(L.tax[0], L.hv[0] >> JB, beg, end);
    }
    if (lane == 0) out.ncand[q] = n;
    return n;
}
// all run heads at once (as topk_all_lds): H[0..nheads) compacted packed words, nheads <= 64 * NC
template <int JB, int NC, int NL, class LF>
__device__ __forceinline__ u32 topk_all_lds_n(const DbDev& db, const OptDev& opt, const OutDev& out, const u32* buf, const u32* H, u32 nheads,
                                              u32 numWindows, const LF& lf, u64 q, u32 lane, u32* scr) {
    static_assert(JB <= 10, "hits << JB stays below bit 26");
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const bool p2 = (P & (P - 1)) == 0;
    const u32 JMASK = (1u << JB) - 1, VMASK = (1u << 26) - 1;
    u32* mx = scr; u32* wt = scr + 64;
    u32 cv[NC], ctax[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        cv[c] = 0; ctax[c] = MCQ_EMPTY;
        if ((u32)(c * 64) >= nheads) continue;               // wave-uniform
        const u32 k = c * 64 + lane;
        const u32 v = (k < nheads) ? H[k] : 0;
        const u32 tgt = lf.tgt(buf[v ? JMASK - (v & JMASK) : 0]);
        if (v != 0 && tgt < db.n_targets) ctax[c] = db.tgt2tax[tgt];
        const u32 cr = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
        cv[c] = v | (cr << 26);
        if (v == 0 || ctax[c] == MCQ_EMPTY) cv[c] = 0;
    }
    ListSlots<NL> L;
    u32 rl[NL], li[NL]; bool lslot[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) { const u32 s = j * 64 + lane; rl[j] = s / seg; li[j] = s - rl[j] * seg; lslot[j] = li[j] < M && rl[j] < P; L.tax[j] = MCQ_EMPTY; L.hv[j] = 0; }
    for (u32 i = 0; i < M; ++i) {
        mx[lane] = 0;
        wave_sync();
#pragma unroll
        for (int c = 0; c < NC; ++c) if (cv[c] != 0) atomicMax(&mx[cv[c] >> 26], cv[c]);
        wave_sync();
#pragma unroll
        for (int c = 0; c < NC; ++c) if (cv[c] != 0 && cv[c] == mx[cv[c] >> 26]) wt[cv[c] >> 26] = ctax[c];   // packed words are unique
        wave_sync();
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const u32 r = rl[j] < 64 ? rl[j] : 63;
            const u32 ml = mx[r], wtl = wt[r];
            if (lslot[j] && li[j] == i && ml != 0) { L.tax[j] = wtl; L.hv[j] = ml & VMASK; }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) if (cv[c] != 0 && ctax[c] == wt[cv[c] >> 26]) cv[c] = 0;
        wave_sync();
    }
    return fold_lists_write_n<u32, JB, NL>(db, opt, out, buf, L, numWindows, lf, q, lane, mx, wt);
}

// ---- rows 8-11 in two classes (large tables: most of a read's locations are chance hits on unrelated targets) ------
// On a RefSeq-scale table a 150-base read gathers ~900 locations of which ~700 are single chance hits, each alone on its
// target; sorting and sweeping all of them to find the dozen targets that matter is most of the work.  The window space is
// cut into cells of 2^cs >= numWindows words (cell = word >> cs; bit fields or global-window words alike): a window range
// spans at most two adjacent cells, so a word whose own cell holds nothing else and whose two neighbour cells are empty --
// a LIGHT word -- lies in no range with a second location: every range that contains it has exactly one hit.  All other
// words are HEAVY, and a heavy word's range neighbours are heavy too (they share or neighbour its cell), so the sweep over
// the heavy words alone gives their exact hit counts.  Two bit maps in LDS (cell occupied / cell holds two or more),
// indexed by the cell's low bits: a collision only makes a light word heavy -- more work, same result.
//
// The top lists need no run heads either: list r = the first M distinct taxa among ALL words of virtual rank r taken as
// entries (hits of the range ending at the word, word) in the order (hits descending, word ascending) -- per taxon the
// first such entry is the reference's (max hits, first candidate reaching them), and weaker entries of a target that has a
// better one retire with its taxon.  Light entries all have one hit, so they come last, in word order: only a PREFIX of
// them (the smallest few, words below theta) can enter a list.  The lists are built from the heavy run heads plus that
// prefix; they are exact when every list is full and ends in an entry that precedes every omitted light word (two or more
// hits, or a word below theta) -- else the caller falls back to the exact path over all words.
#define MCQ_CELL_LOG 15u                          // one wave: 32768 cells per map, 1024 words of LDS each (the workgroup kernel: 2^17)
__device__ __forceinline__ u32 cell_shift(u32 numWindows) { return 32u - (u32)__builtin_clz(numWindows - 1); }     // 2^cs >= numWindows (>= 2)
// G threads clear both maps (2^(LOG - 5) words each)
template <u32 LOG = MCQ_CELL_LOG>
__device__ __forceinline__ void cells_clear(u32* occ, u32* multi, u32 tid, u32 G) {
    u32 zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));             // (not hoistable: see dedup_insert)
    const uint4 z = make_uint4(zero, zero, zero, zero);
    for (u32 i = tid; i < (1u << (LOG - 7)); i += G) { reinterpret_cast<uint4*>(occ)[i] = z; reinterpret_cast<uint4*>(multi)[i] = z; }
}
template <u32 LOG = MCQ_CELL_LOG>
__device__ __forceinline__ void cells_insert(u32 key, u32 cs, u32* occ, u32* multi) {
    const u32 bit = (key >> cs) & ((1u << LOG) - 1), m = 1u << (bit & 31);
    if (atomicOr(&occ[bit >> 5], m) & m) atomicOr(&multi[bit >> 5], m);
}
template <u32 LOG = MCQ_CELL_LOG>
__device__ __forceinline__ bool cells_heavy(u32 key, u32 cs, const u32* occ, const u32* multi) {
    const u32 bit = (key >> cs) & ((1u << LOG) - 1);
    const u32 lo = (bit - 1) & ((1u << LOG) - 1), hi = (bit + 1) & ((1u << LOG) - 1);
    return ((multi[bit >> 5] >> (bit & 31)) | (occ[lo >> 5] >> (lo & 31)) | (occ[hi >> 5] >> (hi & 31))) & 1u;
}

// OptDev::lin: the one list of the two-class forms below sits in lanes [0, M) as (hits << 6 | 63 - rank) << 32 | ~word.  Among
// the one-hit entries the order is (rank, word): the list is exact when it is full and its last entry has two or more hits or
// is a rank-0 word below theta (every omitted light word -- one hit, word >= theta -- comes later then).  ~0u: not provable.
__device__ __forceinline__ u32 two_class_lin_write(const OptDev& opt, const OutDev& out, u32 Ltax, unsigned long long Lw, bool light_omitted,
                                                   u32 theta, u64 q, u32 lane) {
    const u32 M = opt.max_cand, hi = (u32)(Lw >> 32);
    if (light_omitted) {
        const bool ok = Lw != 0 && ((hi >> 6) >= 2 || ((hi & 63u) == 63u && (u32)~(u32)Lw < theta));
        if (__ballot(lane == M - 1 && !ok)) return ~0u;
    }
    const u32 n = (u32)__builtin_popcountll(__ballot(lane < M && Lw != 0));
    if (lane < n) reinterpret_cast<uint4*>(out.cands)[q * M + lane] = make_uint4(Ltax, hi >> 6, 0u, 0u);
    if (lane == 0) out.ncand[q] = n;
    return n;
}

// Top lists + fold + write from the heavy run heads (H[0..nheads) packed (hits << JB | JMASK - j), j indexing the sorted
// distinct heavy words SK) and the light prefix (lkey: this lane's light word, MCQ_EMPTY = none; ascending or not does not
// matter; lhits: its hits -- 1, unless the caller has put run heads into the light lanes: NC = 0, everything in one chunk).
// Entries are compared as 64-bit words (hits << 32 | ~word): heavy and light ones in one order.  NC chunks of 64
// heads.  light_omitted: light words >= theta exist that are not among the entries.  Returns the number of candidates
// written, or ~0u when the lists cannot be proven exact (nothing written then).  scr: 192 words of LDS (64 x u64, 64 x u32).
template <int JB, int NC, class LF>
__device__ __forceinline__ u32 topk_two_class(const DbDev& db, const OptDev& opt, const OutDev& out, const u32* SK, u32 D, const u32* H, u32 nheads,
                                              u32 lkey, bool light_omitted, u32 theta, u32 numWindows, const LF& lf, u64 q, u32 lane, u32* scr, u32 lhits = 1) {
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const bool p2 = (P & (P - 1)) == 0;
    const u32 JMASK = (1u << JB) - 1;
    unsigned long long* mx = reinterpret_cast<unsigned long long*>(scr);
    u32* wt = scr + 128;
    unsigned long long w[NC + 1]; u32 tax[NC + 1], rk[NC + 1];
#pragma unroll
    for (int c = 0; c <= NC; ++c) {
        w[c] = 0; tax[c] = MCQ_EMPTY; rk[c] = 0;
        u32 key = MCQ_EMPTY, hits = 0;
        if (c < NC) {
            if ((u32)(c * 64) >= nheads) continue;               // wave-uniform
            const u32 k = c * 64 + lane;
            const u32 v = (k < nheads) ? H[k] : 0u;
            if (v) { key = SK[JMASK - (v & JMASK)]; hits = v >> JB; }
        } else if (lkey != MCQ_EMPTY) { key = lkey; hits = lhits; }
        const u32 tgt = lf.tgt(hits ? key : SK[0]);               // (idle lanes look up a real word)
        if (hits && tgt < db.n_targets) tax[c] = db.tgt2tax[tgt];
        rk[c] = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
        if (hits && tax[c] != MCQ_EMPTY) w[c] = ((unsigned long long)hits << 32) | (u32)~key;
        if (MCQ_OPT_LIN(opt)) {                                            // one list: (hits, rank, word) in one 64-bit order
            if (rk[c] >= opt.keep) w[c] = 0;
            if (w[c] != 0) w[c] = ((unsigned long long)((hits << 6) | (63u - rk[c])) << 32) | (u32)~key;
        }
    }
    if (MCQ_OPT_LIN(opt)) {
        u32 Ltax = MCQ_EMPTY; unsigned long long Lw = 0;
        for (u32 i = 0; i < M; ++i) {
            unsigned long long v = w[0];
#pragma unroll
            for (int c = 1; c <= NC; ++c) v = w[c] > v ? w[c] : v;
            const unsigned long long m = wave_max_pair(v);
            if (m == 0) break;
            u32 mt = MCQ_EMPTY;
#pragma unroll
            for (int c = 0; c <= NC; ++c) if (w[c] == m) mt = tax[c];
            const u32 wtax = bcast(mt, (u32)__builtin_ctzll(__ballot(v == m)));           // entries are distinct: one winner
            if (lane == i) { Ltax = wtax; Lw = m; }
#pragma unroll
            for (int c = 0; c <= NC; ++c) if (tax[c] == wtax) w[c] = 0;
        }
        return two_class_lin_write(opt, out, Ltax, Lw, light_omitted, theta, q, lane);
    }
    const u32 rl = lane / seg, li = lane - rl * seg;
    const bool lslot = (li < M) && (rl < P);
    u32 Ltax = MCQ_EMPTY; unsigned long long Lw = 0;
    for (u32 i = 0; i < M; ++i) {
        mx[lane] = 0;
        wave_sync();
#pragma unroll
        for (int c = 0; c <= NC; ++c) if (w[c] != 0) atomicMax(&mx[rk[c]], w[c]);
        wave_sync();
#pragma unroll
        for (int c = 0; c <= NC; ++c) if (w[c] != 0 && w[c] == mx[rk[c]]) wt[rk[c]] = tax[c];          // words are distinct: one winner per rank
        wave_sync();
        const unsigned long long ml = mx[rl];
        const u32 wtl = wt[rl];
        if (lslot && li == i && ml != 0) { Ltax = wtl; Lw = ml; }
#pragma unroll
        for (int c = 0; c <= NC; ++c) if (w[c] != 0 && tax[c] == wt[rk[c]]) w[c] = 0;                  // (a live entry's rank has a fresh winner)
        wave_sync();
    }
    // exact? every list full and ending before every omitted light word
    if (light_omitted) {
        const bool last = lslot && li == M - 1;
        const bool ok = Lw != 0 && ((u32)(Lw >> 32) >= 2 || (u32)~(u32)Lw < theta);
        if (__ballot(last && !ok)) return ~0u;
    }
    u32* mx32 = scr; u32* wt32 = scr + 64;
    wave_sync();
    if (P > 1) return fold_lists_write<u32, u32, JB>(db, opt, out, SK, Ltax, (u32)(Lw >> 32) << JB, numWindows, lf, q, lane, mx32, wt32);
    // one list, with window ranges: a single hit is its own range; else the range that ends at the word, inside SK
    const u32 n = (u32)__builtin_popcountll(__ballot(lane < M && Lw != 0));
    if (lane < n) {
        const u32 key = ~(u32)Lw, hits = (u32)(Lw >> 32);
        u32 t, tb;
        lf.locate(key, t, tb);
        u32 beg = key - tb;
        if (hits >= 2) {
            const u32 lowkey = range_low<u32>(key, tb, numWindows);
            u32 lo = 0, hi = D;
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (SK[mid] < lowkey) lo = mid + 1; else hi = mid; }
            beg = SK[lo] - tb;
        }
        reinterpret_cast<uint4*>(out.cands)[q * M + lane] = make_uint4(Ltax, hits, beg, key - tb);
    }
    if (lane == 0) out.ncand[q] = n;
    return n;
}

// The same with the entries in LDS instead of registers (the workgroup kernel is compiled for 64 VGPRs; one of its waves
// builds the lists while the others wait): up to NE_MAX - 64 run heads + 64 light words.  ent: 4 * NE_MAX + 192 words of LDS
// (entry words as u64, taxa, ranks, then the lists' scratch).
template <int JB, u32 NE_MAX, class LF>
__device__ __forceinline__ u32 topk_two_class_lds(const DbDev& db, const OptDev& opt, const OutDev& out, const u32* SK, u32 D, const u32* H, u32 nheads,
                                                  u32 lkey, u32 n_light, bool light_omitted, u32 theta, u32 numWindows, const LF& lf, u64 q, u32 lane, u32* ent) {
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const bool p2 = (P & (P - 1)) == 0;
    const u32 JMASK = (1u << JB) - 1;
    unsigned long long* EW = reinterpret_cast<unsigned long long*>(ent);
    u32* ET = ent + 2 * NE_MAX; u32* ER = ent + 3 * NE_MAX;
    u32* scr = ent + 4 * NE_MAX;
    unsigned long long* mx = reinterpret_cast<unsigned long long*>(scr);
    u32* wt = scr + 128;
    const u32 NE = nheads + n_light;
    for (u32 base = 0; base < NE; base += 64) {
        const u32 e = base + lane;
        u32 key = MCQ_EMPTY, hits = 0;
        const u32 lk = __shfl(lkey, (int)((e - nheads) & 63), 64);     // (by every lane: a shuffle inside a branch cannot read the lanes outside it)
        if (e < nheads) { const u32 v = H[e]; key = SK[JMASK - (v & JMASK)]; hits = v >> JB; }
        else if (e < NE) { key = lk; hits = 1; }
        const u32 tgt = lf.tgt(hits ? key : SK[0]);               // (idle lanes look up a real word)
        u32 tax = MCQ_EMPTY;
        if (hits && tgt < db.n_targets) tax = db.tgt2tax[tgt];
        if (e < NE) {
            EW[e] = (hits && tax != MCQ_EMPTY) ? (((unsigned long long)hits << 32) | (u32)~key) : 0ull;
            ET[e] = tax; ER[e] = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
        }
    }
    wave_sync();
    if (MCQ_OPT_LIN(opt)) {
        for (u32 e = lane; e < NE; e += 64) {
            const unsigned long long w = EW[e];
            const u32 r = ER[e];
            EW[e] = (w != 0 && r < opt.keep) ? ((unsigned long long)((((u32)(w >> 32)) << 6) | (63u - r)) << 32) | (u32)w : 0ull;
        }
        u32 Ltax = MCQ_EMPTY; unsigned long long Lw = 0;
        for (u32 i = 0; i < M; ++i) {
            unsigned long long v = 0; u32 ve = 0;
            for (u32 e = lane; e < NE; e += 64) { const unsigned long long w = EW[e]; if (w > v) { v = w; ve = e; } }
            const unsigned long long m = wave_max_pair(v);
            if (m == 0) break;
            const u32 wtax = bcast(ET[ve], (u32)__builtin_ctzll(__ballot(v == m)));      // entries are distinct: one winner
            if (lane == i) { Ltax = wtax; Lw = m; }
            for (u32 e = lane; e < NE; e += 64) if (ET[e] == wtax) EW[e] = 0;
        }
        return two_class_lin_write(opt, out, Ltax, Lw, light_omitted, theta, q, lane);
    }
    const u32 rl = lane / seg, li = lane - rl * seg;
    const bool lslot = (li < M) && (rl < P);
    u32 Ltax = MCQ_EMPTY; unsigned long long Lw = 0;
    for (u32 i = 0; i < M; ++i) {
        mx[lane] = 0;
        wave_sync();
        for (u32 e = lane; e < NE; e += 64) { const unsigned long long w = EW[e]; if (w != 0) atomicMax(&mx[ER[e]], w); }
        wave_sync();
        for (u32 e = lane; e < NE; e += 64) { const unsigned long long w = EW[e]; if (w != 0 && w == mx[ER[e]]) wt[ER[e]] = ET[e]; }
        wave_sync();
        const unsigned long long ml = mx[rl];
        const u32 wtl = wt[rl];
        if (lslot && li == i && ml != 0) { Ltax = wtl; Lw = ml; }
        for (u32 e = lane; e < NE; e += 64) if (EW[e] != 0 && ET[e] == wt[ER[e]]) EW[e] = 0;
        wave_sync();
    }
    if (light_omitted) {
        const bool last = lslot && li == M - 1;
        const bool ok = Lw != 0 && ((u32)(Lw >> 32) >= 2 || (u32)~(u32)Lw < theta);
        if (__ballot(last && !ok)) return ~0u;
    }
    u32* mx32 = scr; u32* wt32 = scr + 64;
    wave_sync();
    if (P > 1) return fold_lists_write<u32, u32, JB>(db, opt, out, SK, Ltax, (u32)(Lw >> 32) << JB, numWindows, lf, q, lane, mx32, wt32);
    const u32 n = (u32)__builtin_popcountll(__ballot(lane < M && Lw != 0));
    if (lane < n) {
        const u32 key = ~(u32)Lw, hits = (u32)(Lw >> 32);
        u32 t, tb;
        lf.locate(key, t, tb);
        u32 beg = key - tb;
        if (hits >= 2) {
            const u32 lowkey = range_low<u32>(key, tb, numWindows);
            u32 lo = 0, hi = D;
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (SK[mid] < lowkey) lo = mid + 1; else hi = mid; }
            beg = SK[lo] - tb;
        }
        reinterpret_cast<uint4*>(out.cands)[q * M + lane] = make_uint4(Ltax, hits, beg, key - tb);
    }
    if (lane == 0) out.ncand[q] = n;
    return n;
}

// ---- rows 10-11 for the workgroup kernels ----------------------------------------------------------
// After the sweep H[j] != 0 marks the head of a target's run and holds its packed best.  All threads offer
// their heads to the list of the head's virtual rank (ds_max per rank; all ranks in the same round, M rounds,
// retired heads are zeroed in H), then wave 0 folds the P lists.  scr: 64 HT maxima, 64 winner taxa, the lists
// (64 taxa + 64 HT words) in lane layout -- LDS of the workgroup.
template <class HT> struct TopkBlockScratch { HT mx[64]; HT lhv[64]; u32 wt[64]; u32 ltax[64]; u32 fmx[64]; u32 fwt[64]; };

// Tree fold of P lists of M entries that live in LDS (bl[0..P*M) hits, bl[MCQ_BIGLIST_MAX ..) taxa; hits == 0 marks an
// unused entry): the emulation of the reference's larger rank counts (mpiexec -n 32 / -n 64 with -maxcand 4,
// script/ft/QueryGeneric_FT.sh:115) where P x M exceeds the 64 lanes of a wave.  One wave per edge of a tree level
// (the edges of a level touch disjoint ranks), lanes [0,M) the receiver's entries, [M,2M) the sender's; the same closed
// form as fold_lists_write: M rounds of "largest (hits, earliest position), record it, retire its taxon".  Then list 0
// is written; window positions are (0,0) after a fold.  Returns the number of candidates (in every thread of wave 0).
template <class Sync>
__device__ __forceinline__ u32 fold_lists_block(const OptDev& opt, const OutDev& out, u32* bl, u64 q, u32 tid, u32 NTB, Sync sync) {
    const u32 M = opt.max_cand, lane = tid & 63, wv = tid >> 6, nwv = NTB >> 6;
    u32* hits = bl; u32* tax = bl + MCQ_BIGLIST_MAX;
    u32 lb = 0;
    for (u32 L = 0; L < opt.n_levels; ++L) {
        const u32 le = opt.level_end[L];
        for (u32 e = lb + wv; e < le; e += nwv) {                 // wave-uniform
            const u32 snd = opt.fold_snd[e], rcv = opt.fold_rcv[e];
            const bool is_snd = lane >= M;
            const u32 src_i = (is_snd ? snd : rcv) * M + (is_snd ? lane - M : lane);
            u32 h = 0, t = MCQ_EMPTY;
            if (lane < 2 * M) { h = hits[src_i]; t = tax[src_i]; }
            if (opt.quirk_seq_drop && is_snd && (t & 0x80000000u)) h = 0;
            u64 key = h ? (((u64)h << 32) | (u64)(0xFFFFFFFFu - lane)) : 0ull;
            u32 Nh = 0, Nt = MCQ_EMPTY;
            for (u32 i = 0; i < M; ++i) {
                const u64 m = wave_max(key);
                if (m == 0) break;
                const u32 src = (u32)__builtin_ctzll(__ballot(key == m));
                const u32 wtax = bcast(t, src);
                if (lane == i) { Nh = (u32)(m >> 32); Nt = wtax; }
                if (key != 0 && t == wtax) key = 0;
            }
            wave_sync();                                          // everyone has read the two lists
            if (lane < M) { hits[rcv * M + lane] = Nh; tax[rcv * M + lane] = Nt; }
        }
        lb = le;
        sync();
    }
    u32 n = 0;
    if (tid < 64) {
        const u32 h = lane < M ? hits[lane] : 0u;
        n = (u32)__builtin_popcountll(__ballot(h != 0));
        if (lane < n) reinterpret_cast<uint4*>(out.cands)[q * M + lane] = make_uint4(tax[lane], h, 0u, 0u);
        if (lane == 0) out.ncand[q] = n;
    }
    return n;
}

// OptDev::lin in the workgroup kernels: one list (see topk_lin_write).  The workgroup kernels of the query path are
// instantiated per form (lists / lists in LDS / one selection: they are compiled for 64 VGPRs and 1024 threads, and either
// form's code inside the other's kernel cost the long reads 8 % in spills whether it ran or not).
template <class KeyT, class HT, int JB, class LF>
__device__ __forceinline__ u32 topk_block_lin(const DbDev& db, const OptDev& opt, const OutDev& out, const KeyT* B, HT* H, u32 T,
                                              const LF& lf, u64 q, u32 tid, u32 NTB, TopkBlockScratch<HT>* scr) {
    const u32 M = opt.max_cand;
    // one list (see topk_lin_write): per round every thread offers its best live head as (hits, 63 - rank, position) in 64
    // bits, one ds_max per wave; the winner publishes its taxon, every head of that taxon retires
    constexpr u64 JM = JB >= 32 ? 0xFFFFFFFFull : ((1ull << (JB & 31)) - 1);
    unsigned long long* gm = reinterpret_cast<unsigned long long*>(scr->fmx);
    u32 n = 0;
    for (u32 i = 0; i < M; ++i) {
        if (tid == 0) gm[0] = 0;
        __syncthreads();
        unsigned long long v = 0; u32 vtax = MCQ_EMPTY;
        for (u32 j = tid; j < T; j += NTB) {
            const HT hv = H[j];
            if (hv == 0) continue;
            const u32 tgt = lf.tgt(B[j]);
            const u32 tax = tgt < db.n_targets ? db.tgt2tax[tgt] : MCQ_EMPTY;
            const u32 r = lin_rank(opt, tgt);
            if (tax == MCQ_EMPTY || r >= opt.keep) { H[j] = 0; continue; }
            const unsigned long long k = ((unsigned long long)(hv >> JB) << 38) | ((unsigned long long)(63u - r) << 32) | ((unsigned long long)hv & JM);
            if (k > v) { v = k; vtax = tax; }
        }
        const unsigned long long wm = wave_max_pair(v);
        if ((tid & 63) == 0 && wm != 0) atomicMax(&gm[0], wm);
        __syncthreads();
        const unsigned long long m = gm[0];
        if (m == 0) break;                                      // (uniform)
        if (v == m) { scr->wt[0] = vtax; scr->ltax[i] = vtax; scr->fwt[i] = (u32)(m >> 38); }       // linear words are unique: one winner
        __syncthreads();
        const u32 wtax = scr->wt[0];
        for (u32 j = tid; j < T; j += NTB) {
            if (H[j] == 0) continue;
            if (db.tgt2tax[lf.tgt(B[j])] == wtax) H[j] = 0;
        }
        ++n;
        __syncthreads();
    }
    if (tid < n) reinterpret_cast<uint4*>(out.cands)[q * M + tid] = make_uint4(scr->ltax[tid], scr->fwt[tid], 0u, 0u);
    if (tid == 0) out.ncand[q] = n;
    return n;
}

template <class KeyT, class HT, int JB, class LF>
__device__ __attribute__((noinline)) u32 topk_block_lin_call(const DbDev& db, const OptDev& opt, const OutDev& out, const KeyT* B, HT* H, u32 T,
                                                             const LF& lf, u64 q, u32 tid, u32 NTB, TopkBlockScratch<HT>* scr) {
    return topk_block_lin<KeyT, HT, JB>(db, opt, out, B, H, T, lf, q, tid, NTB, scr);
}

// FORM (separate instantiations of the workgroup kernels, so that each carries only its own): 0 = the P lists in the lanes
// of a wave, 1 = OptDev::big, 2 = OptDev::lin.  RTLIN (staged reduce kernels, forms 0 / 1): OptDev::lin is looked at at run
// time and served by a call.
template <class KeyT, class HT, int JB, int FORM, bool RTLIN = false, class LF, class Sync>
__device__ __forceinline__ u32 topk_block(const DbDev& db, const OptDev& opt, const OutDev& out, const KeyT* B, HT* H,
                                          u32 T, u32 numWindows, const LF& lf, u64 q, u32 tid, u32 NTB,
                                          TopkBlockScratch<HT>* scr, u32* bl, Sync sync) {
    if constexpr (FORM == 2) return topk_block_lin<KeyT, HT, JB>(db, opt, out, B, H, T, lf, q, tid, NTB, scr);
    if constexpr (RTLIN) { if (MCQ_OPT_LIN(opt)) return topk_block_lin_call<KeyT, HT, JB>(db, opt, out, B, H, T, lf, q, tid, NTB, scr); }
    const u32 M = opt.max_cand, P = opt.P, seg = opt.seg;
    const bool p2 = (P & (P - 1)) == 0;
    constexpr bool big = FORM == 1;
    if (tid < 64) { scr->ltax[tid] = MCQ_EMPTY; scr->lhv[tid] = 0; }
    if constexpr (big) for (u32 i = tid; i < P * M; i += NTB) { bl[i] = 0; bl[MCQ_BIGLIST_MAX + i] = MCQ_EMPTY; }
    for (u32 i = 0; i < M; ++i) {
        if (tid < 64) scr->mx[tid] = 0;
        sync();
        for (u32 j = tid; j < T; j += NTB) {
            const HT v = H[j];
            if (v == 0) continue;
            const u32 tgt = lf.tgt(B[j]);
            const u32 tax = tgt < db.n_targets ? db.tgt2tax[tgt] : MCQ_EMPTY;
            if (tax == MCQ_EMPTY) { H[j] = 0; continue; }
            const u32 r = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
            atomicMax(&scr->mx[r], v);
        }
        sync();
        for (u32 j = tid; j < T; j += NTB) {
            const HT v = H[j];
            if (v == 0) continue;
            const u32 tgt = lf.tgt(B[j]);
            const u32 r = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
            if (v == scr->mx[r]) scr->wt[r] = db.tgt2tax[tgt];       // packed words are unique: one winner per rank
        }
        sync();
        for (u32 j = tid; j < T; j += NTB) {
            const HT v = H[j];
            if (v == 0) continue;
            const u32 tgt = lf.tgt(B[j]);
            const u32 r = (P > 1) ? (p2 ? (tgt & (P - 1)) : (tgt % P)) : 0;
            if (db.tgt2tax[tgt] == scr->wt[r]) H[j] = 0;            // every head of the winner's taxon retires
        }
        if (tid < P && scr->mx[tid] != 0) {
            if constexpr (big) { bl[tid * M + i] = (u32)(scr->mx[tid] >> JB); bl[MCQ_BIGLIST_MAX + tid * M + i] = scr->wt[tid]; }
            else { scr->ltax[tid * seg + i] = scr->wt[tid]; scr->lhv[tid * seg + i] = scr->mx[tid]; }
        }
        sync();
    }
    if constexpr (big) return fold_lists_block(opt, out, bl, q, tid, NTB, sync);
    u32 n = 0;
    if (tid < 64) n = fold_lists_write<KeyT, HT, JB>(db, opt, out, B, scr->ltax[tid], scr->lhv[tid], numWindows, lf, q, tid, scr->fmx, scr->fwt);
    return n;
}

} // namespace mcq
