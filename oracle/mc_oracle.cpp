// =============================================================================
// oracle/mc_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the per-read *query* path of jmabuin/metacache-mpi
// (rows 1-12 of SURVEY.md section 8a).  It is the parity checker for the HIP
// engine and the "port" CPU baseline of bench.py.  Nothing in the product
// package may link, import or call this file; only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() do.
//
// Parity pin: every function below is checked against vectors produced by the
// reference's own headers compiled in the build container (oracle/_ref, recipe
// in oracle/Makefile; vectors under tests/golden/, generator
// tests/golden/make_golden.py).  The reference has no tests of its own.
//
// Each function cites the reference file:line it restates (paths relative to
// the reference root).  The code is written from the behavioural description
// of those lines, not copied from them.
// =============================================================================
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;

namespace {

// ---- row 4: src/hash_int.h:39-45 (thomas_mueller_hash, u32 -> u32) ----------
inline u32 tmh(u32 x) {
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    return (x >> 16) ^ x;
}

// ---- row 3: src/dna_encoding.h:113-121 (reverse complement of a 2-bit k-mer)
inline u32 revcomp(u32 s, u32 k) {
    s = ((s >> 2) & 0x33333333u) | ((s & 0x33333333u) << 2);
    s = ((s >> 4) & 0x0F0F0F0Fu) | ((s & 0x0F0F0F0Fu) << 4);
    s = ((s >> 8) & 0x00FF00FFu) | ((s & 0x00FF00FFu) << 8);
    s = (s >> 16) | (s << 16);
    return (0xFFFFFFFFu - s) >> (32 - 2 * k);
}

// ---- src/dna_encoding.h:187-197 (canonical = min(kmer, revcomp)) -------------
inline u32 canonical(u32 s, u32 k) {
    u32 r = revcomp(s, k);
    return s < r ? s : r;
}

// ---- row 1: src/dna_encoding.h:259-276 (for_each_window) ---------------------
// Emits [beg,end) pairs.  A sequence not longer than the window (including the
// empty one) is ONE window; otherwise full windows every `stride`, then a tail
// starting at the next stride position if that is still inside the sequence.
template <class F>
inline void for_each_window(u64 n, u64 len, u64 stride, F&& f) {
    if (n <= len) { f(u64(0), n); return; }
    u64 first = 0;
    for (u64 wend = len; wend <= n; first += stride, wend += stride) f(first, wend);
    if (first < n) f(first, n);
}

// ---- rows 2+5: src/dna_encoding.h:303-348, :457-466, src/hash_dna.h:113-152 --
// Rolling 2-bit encoder with an ambiguity bit per base; a k-mer is used only if
// none of its k bases is ambiguous.  The sketch is the min(s, n-k+1) smallest
// DISTINCT hashes, ascending; 0xFFFFFFFF never appears (it is the fill value).
inline int sketch(const char* seq, u64 n, u32 k, u32 s, u32* out) {
    if (n < k) return 0;
    u64 cap = n - k + 1;
    u32 sl = (u64)s < cap ? s : (u32)cap;
    if (sl < 1) return 0;
    for (u32 i = 0; i < sl; ++i) out[i] = 0xFFFFFFFFu;

    const u32 kmerMsk = 0xFFFFFFFFu >> (32 - 2 * k);
    const u32 ambMsk = (0xFFFFu >> (16 - k));
    u32 kmer = 0, amb = 0;
    for (u64 i = 0; i < n; ++i) {
        kmer <<= 2; amb <<= 1;
        switch (seq[i]) {
            case 'A': case 'a': break;
            case 'C': case 'c': kmer |= 1; break;
            case 'G': case 'g': kmer |= 2; break;
            case 'T': case 't': kmer |= 3; break;
            default: amb |= 1; break;
        }
        if (i + 1 < k) continue;
        kmer &= kmerMsk; amb &= ambMsk;
        if (amb) continue;
        u32 h = tmh(canonical(kmer, k));
        if (h < out[sl - 1]) {
            u32* pos = std::lower_bound(out, out + sl, h);
            if (*pos != h) {                     // distinct values only
                std::memmove(pos + 1, pos, (size_t)((out + sl - 1) - pos) * sizeof(u32));
                *pos = h;
            }
        }
    }
    int m = 0;
    while (m < (int)sl && out[m] != 0xFFFFFFFFu) ++m;
    return m;
}

// ---- row 6: feature -> location-list map -------------------------------------
// Only the key -> list mapping is contractual.  Layout here:
// open addressing, slot0 = tmh(key) mod nbuckets, hops +1,+2,+3.. with wrap
// (the scheme of src/hash_multimap.h:142-187, :1033-1047), load factor <= 0.8
// (src/sketch_database.h:267); nbuckets is rounded up to a power of two.
struct Db {
    u32 k, s, winlen, winstride;   // query-side sketching (src/sketch_database.h:256-259)
    u32 tgt_winstride;             // target window stride (used by row 9's range width)
    u64 nbuckets;                  // power of two, so triangular probing visits every slot
    std::vector<u32> bkey;      // key per bucket
    std::vector<u64> boff;      // offset into locs
    std::vector<u32> blen;      // 0 = unused bucket
    std::vector<u64> locs;      // (tgt << 32) | win, each list ascending
    std::vector<u32> tgt2tax;   // target -> taxon key used by insert()
    u32 n_targets;

    inline bool find(u32 key, u64& off, u32& len) const {
        u64 pos = tmh(key) & (nbuckets - 1), hop = 1;
        for (u64 tries = 0; tries < nbuckets; ++tries) {
            if (blen[pos] == 0) return false;
            if (bkey[pos] == key) { off = boff[pos]; len = blen[pos]; return true; }
            pos = (pos + hop++) & (nbuckets - 1);
        }
        return false;
    }
};

struct Cand { u32 tax, hits, beg, end; };

// ---- row 7: src/sketch_database.h:804-823 (accumulate_matches) ---------------
// Appends the location list of every sketch feature that has one; `runs`
// receives the end offset of every appended list (run boundaries for row 8).
inline void accumulate(const Db& db, const char* seq, u64 n,
                       std::vector<u64>& res, std::vector<size_t>& runs,
                       u64* n_feat, u64* n_hitfeat) {
    u32 sk[64];
    for_each_window(n, db.winlen, db.winstride, [&](u64 b, u64 e) {
        int m = sketch(seq + b, e - b, db.k, db.s > 64 ? 64 : db.s, sk);
        if (n_feat) *n_feat += (u64)m;
        for (int i = 0; i < m; ++i) {
            u64 off; u32 len;
            if (db.find(sk[i], off, len) && len > 0) {
                res.insert(res.end(), db.locs.begin() + off, db.locs.begin() + off + len);
                runs.push_back(res.size());
                if (n_hitfeat) *n_hitfeat += 1;
            }
        }
    });
}

// ---- row 8: src/querying.h:88-106 (merge_sort of the sorted runs) ------------
// Bottom-up pairwise std::merge; the result is the sorted multiset by
// (tgt,win) with duplicates kept.
inline void merge_runs(std::vector<u64>& a, const std::vector<size_t>& off, std::vector<u64>& b) {
    if (off.size() < 3) return;
    b.resize(a.size());
    int nch = (int)off.size() - 1;
    for (int s = 1; s < nch; s *= 2) {
        for (int i = 0; i < nch; i += 2 * s) {
            size_t beg = off[i];
            size_t mid = off[std::min(i + s, nch)];
            size_t end = off[std::min(i + 2 * s, nch)];
            std::merge(a.begin() + beg, a.begin() + mid, a.begin() + mid, a.begin() + end, b.begin() + beg);
        }
        a.swap(b);
    }
}

// ---- row 9: src/candidates.h:118-180 (for_all_contiguous_window_ranges) ------
// One candidate per target in ascending target order: two-pointer sweep over
// the target's sorted windows, range width < numWindows, first strictly-best.
struct TgtCand { u32 tgt, hits, beg, end; };
inline void target_candidates(const std::vector<u64>& m, u32 numWindows, std::vector<TgtCand>& out) {
    out.clear();
    size_t n = m.size();
    if (n == 0) return;
    size_t fst = 0;
    u32 hits = 1;
    TgtCand best{(u32)(m[0] >> 32), 1, (u32)m[0], (u32)m[0]};
    for (size_t lst = 1; lst < n; ++lst) {
        u32 t = (u32)(m[lst] >> 32), w = (u32)m[lst];
        if (t == best.tgt) {
            ++hits;
            while (fst != lst && (u32)(w - (u32)m[fst]) >= numWindows) { --hits; ++fst; }
            if (hits > best.hits) { best.hits = hits; best.beg = (u32)m[fst]; best.end = w; }
        } else {
            out.push_back(best);
            fst = lst; hits = 1;
            best = TgtCand{t, 1, w, w};
        }
    }
    out.push_back(best);
}

// ---- row 10: src/candidates.h:236-285 (insert into the bounded top list) -----
// `tax` is already the taxon key at mergeBelow (or a per-target unique key when
// the target has no ancestor there: a sequence-level taxon can never collide,
// so the "same taxon" search below simply never matches for it).
// Bounded insert: position = first element with strictly fewer hits
// (upper_bound on "greater"), insert if not at end or list not full, truncate.
// Same-taxon update: overwrite if more hits, then stable re-sort of [0..i]
// (std::sort on <=16 elements is libstdc++ insertion sort => stable).
inline void top_insert(std::vector<Cand>& top, Cand c, size_t maxCand) {
    size_t i = 0;
    for (; i < top.size(); ++i) if (top[i].tax == c.tax) break;
    if (i < top.size()) {
        if (c.hits > top[i].hits) {
            top[i] = c;
            // element i moves left past every element with strictly fewer hits
            size_t j = i;
            while (j > 0 && top[j - 1].hits < c.hits) { top[j] = top[j - 1]; --j; }
            top[j] = c;
        }
        return;
    }
    size_t j = 0;
    while (j < top.size() && top[j].hits >= c.hits) ++j;
    if (j != top.size() || top.size() < maxCand) {
        top.insert(top.begin() + j, c);
        if (top.size() > maxCand) top.resize(maxCand);
    }
}

// ---- row 11: src/querying.h:867-1073 (binary-tree fold of the P rank lists) --
// senders = odd ranks, receivers = even ranks; i-th sender pairs with i-th
// receiver (set order); after a round all used senders retire and every 2nd
// used receiver becomes a sender (src/querying.h:1035-1066).  The receiver
// re-inserts its own list first (first round only, :910-940; a no-op on an
// already-consistent list but restated anyway), then the sender's entries in
// list order.  Window positions do not travel: the folded list has pos (0,0).
// quirk_seq_drop: the reference ships taxon ids as u32, so a sequence-level
// (negative id) taxon sent by any rank is dropped by the receiver
// (:958, src/candidates.h:240).  Keys with bit 31 set are sequence-level here.
inline void tree_fold(std::vector<std::vector<Cand>>& L, size_t maxCand, bool quirk_seq_drop) {
    int P = (int)L.size();
    std::vector<int> senders, receivers;
    for (int i = 0; i < P; ++i) (i % 2 ? senders : receivers).push_back(i);
    std::vector<char> seeded(P, 0);
    for (int k = P; k > 1; k /= 2) {
        size_t np = std::min(senders.size(), receivers.size());
        std::vector<int> used_s(senders.begin(), senders.begin() + np);
        std::vector<int> used_r(receivers.begin(), receivers.begin() + np);
        for (size_t i = 0; i < np; ++i) {
            int snd = used_s[i], rcv = used_r[i];
            if (!seeded[rcv]) {
                std::vector<Cand> own; own.swap(L[rcv]);
                for (auto c : own) { c.beg = c.end = 0; top_insert(L[rcv], c, maxCand); }
                seeded[rcv] = 1;
            }
            for (auto c : L[snd]) {
                if (quirk_seq_drop && (c.tax & 0x80000000u)) continue;
                c.beg = c.end = 0;
                top_insert(L[rcv], c, maxCand);
            }
            L[snd].clear();
        }
        // retire used senders; every 2nd used receiver becomes a sender
        std::vector<int> ns;
        for (int s : senders) if (std::find(used_s.begin(), used_s.end(), s) == used_s.end()) ns.push_back(s);
        std::vector<int> nr;
        for (int r : receivers) {
            auto it = std::find(used_r.begin(), used_r.end(), r);
            bool to_sender = (it != used_r.end()) && (((it - used_r.begin()) % 2) == 1);
            if (to_sender) ns.push_back(r); else nr.push_back(r);
        }
        std::sort(ns.begin(), ns.end());
        senders.swap(ns); receivers.swap(nr);
    }
}

struct QParams {
    u32 max_cand;        // maxNumCandidatesPerQuery (src/query_options.h:134)
    u32 emulate_ranks;   // P of the reference run being matched; 1 = no fold
    u64 insert_size_max; // src/query_options.h:132
    u32 quirk_seq_drop;  // see tree_fold
};

// One query = one read or one pair.  rows 7-11 chained as in
// src/querying.h:792-825 + src/classification.cpp:209-224.
inline u32 query_one(const Db& db, const char* s1, u64 n1, const char* s2, u64 n2,
                     const QParams& qp, Cand* out,
                     std::vector<u64>& m, std::vector<u64>& tmp, std::vector<size_t>& runs,
                     std::vector<TgtCand>& tc, u64* stats) {
    m.clear(); runs.clear(); runs.push_back(0);
    accumulate(db, s1, n1, m, runs, stats ? stats + 0 : nullptr, stats ? stats + 1 : nullptr);
    accumulate(db, s2, n2, m, runs, stats ? stats + 0 : nullptr, stats ? stats + 1 : nullptr);
    if (stats) stats[2] += m.size();
    merge_runs(m, runs, tmp);
    u32 numWindows = (u32)(2 + std::max<u64>(n1 + n2, qp.insert_size_max) / db.tgt_winstride);
    target_candidates(m, numWindows, tc);
    u32 P = qp.emulate_ranks < 1 ? 1 : qp.emulate_ranks;
    std::vector<std::vector<Cand>> L(P);
    for (const auto& c : tc) {
        u32 tax = c.tgt < db.n_targets ? db.tgt2tax[c.tgt] : (0x80000000u | c.tgt);
        top_insert(L[c.tgt % P], Cand{tax, c.hits, c.beg, c.end}, qp.max_cand);
    }
    if (P > 1) tree_fold(L, qp.max_cand, qp.quirk_seq_drop != 0);
    u32 nc = (u32)L[0].size();
    for (u32 i = 0; i < nc; ++i) out[i] = L[0][i];
    if (stats) stats[3] += nc;
    return nc;
}

} // namespace

// =============================================================================
// C interface (ctypes)
// =============================================================================
extern "C" {

u32 orc_tmh(u32 x) { return tmh(x); }
u32 orc_revcomp(u32 x, u32 k) { return revcomp(x, k); }
u32 orc_canonical(u32 x, u32 k) { return canonical(x, k); }

// writes up to cap (beg,end) pairs; returns the number of windows
int orc_windows(u64 n, u64 len, u64 stride, u64* beg, u64* end, int cap) {
    int c = 0;
    for_each_window(n, len, stride, [&](u64 b, u64 e) { if (c < cap) { beg[c] = b; end[c] = e; } ++c; });
    return c;
}

int orc_sketch(const char* seq, u64 n, u32 k, u32 s, u32* out) { return sketch(seq, n, k, s, out); }

// locs: (tgt<<32)|win, list i = [off[i], off[i+1]) ascending
void* orc_db_create(u64 n_keys, const u32* keys, const u64* off, const u64* locs,
                    u32 n_targets, const u32* tgt2tax, u32 k, u32 s, u32 winlen, u32 winstride,
                    u32 tgt_winstride) {
    Db* db = new Db();
    db->k = k; db->s = s; db->winlen = winlen; db->winstride = winstride;
    db->tgt_winstride = tgt_winstride ? tgt_winstride : winstride;
    u64 want = (u64)(1 + (double)n_keys / 0.8);
    db->nbuckets = 1; while (db->nbuckets < want) db->nbuckets <<= 1;
    db->bkey.assign(db->nbuckets, 0); db->boff.assign(db->nbuckets, 0); db->blen.assign(db->nbuckets, 0);
    db->locs.assign(locs, locs + off[n_keys]);
    db->n_targets = n_targets;
    db->tgt2tax.assign(tgt2tax, tgt2tax + n_targets);
    for (u64 i = 0; i < n_keys; ++i) {
        u32 len = (u32)(off[i + 1] - off[i]);
        if (len == 0) continue;
        u64 pos = tmh(keys[i]) & (db->nbuckets - 1), hop = 1;
        while (db->blen[pos] != 0) pos = (pos + hop++) & (db->nbuckets - 1);
        db->bkey[pos] = keys[i]; db->boff[pos] = off[i]; db->blen[pos] = len;
    }
    return db;
}
void orc_db_destroy(void* p) { delete (Db*)p; }

// Sequences: bases[seq_off[i] .. seq_off[i+1]).  paired != 0: sequences 2q and
// 2q+1 are the mates of query q.  out: n_queries x max_cand Cand, n_cand[q].
// stats (optional, 4 x u64): sketch features, hit features, locations, cands.
// threads > 1 splits the queries over std::threads (CPU baseline).
void orc_query(const void* dbp, u64 n_seq, const char* bases, const u64* seq_off, int paired,
               u32 max_cand, u32 emulate_ranks, u64 insert_size_max, u32 quirk_seq_drop,
               u32* out_cand /* n_q*max_cand*4 */, u32* out_ncand, u64* stats, int threads) {
    const Db& db = *(const Db*)dbp;
    QParams qp{max_cand, emulate_ranks, insert_size_max, quirk_seq_drop};
    u64 nq = paired ? n_seq / 2 : n_seq;
    if (threads < 1) threads = 1;
    std::vector<std::vector<u64>> tstats(threads, std::vector<u64>(4, 0));
    auto work = [&](int t) {
        std::vector<u64> m, tmp; std::vector<size_t> runs; std::vector<TgtCand> tc;
        u64 q0 = nq * (u64)t / (u64)threads, q1 = nq * (u64)(t + 1) / (u64)threads;
        for (u64 q = q0; q < q1; ++q) {
            u64 a = paired ? 2 * q : q;
            const char* s1 = bases + seq_off[a]; u64 n1 = seq_off[a + 1] - seq_off[a];
            const char* s2 = s1; u64 n2 = 0;
            if (paired) { s2 = bases + seq_off[a + 1]; n2 = seq_off[a + 2] - seq_off[a + 1]; }
            Cand* o = (Cand*)(out_cand + q * (u64)max_cand * 4);
            out_ncand[q] = query_one(db, s1, n1, s2, n2, qp, o, m, tmp, runs, tc, tstats[t].data());
        }
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    if (stats) for (int i = 0; i < 4; ++i) { stats[i] = 0; for (int t = 0; t < threads; ++t) stats[i] += tstats[t][i]; }
}

// Intermediate dumps for one query (rows 7/8 and 9) -- used by the parity tests
// for the staged HIP kernels.  Returns the number of entries (may exceed cap).
u64 orc_query_matches(const void* dbp, const char* s1, u64 n1, const char* s2, u64 n2, u64* out, u64 cap) {
    const Db& db = *(const Db*)dbp;
    std::vector<u64> m, tmp; std::vector<size_t> runs{0};
    accumulate(db, s1, n1, m, runs, nullptr, nullptr);
    accumulate(db, s2, n2, m, runs, nullptr, nullptr);
    merge_runs(m, runs, tmp);
    for (u64 i = 0; i < m.size() && i < cap; ++i) out[i] = m[i];
    return m.size();
}
u64 orc_query_target_cands(const void* dbp, const char* s1, u64 n1, const char* s2, u64 n2,
                           u64 insert_size_max, u32* out /* x4 */, u64 cap) {
    const Db& db = *(const Db*)dbp;
    std::vector<u64> m, tmp; std::vector<size_t> runs{0};
    accumulate(db, s1, n1, m, runs, nullptr, nullptr);
    accumulate(db, s2, n2, m, runs, nullptr, nullptr);
    merge_runs(m, runs, tmp);
    std::vector<TgtCand> tc;
    target_candidates(m, (u32)(2 + std::max<u64>(n1 + n2, insert_size_max) / db.tgt_winstride), tc);
    for (u64 i = 0; i < tc.size() && i < cap; ++i) {
        out[4 * i] = tc[i].tgt; out[4 * i + 1] = tc[i].hits; out[4 * i + 2] = tc[i].beg; out[4 * i + 3] = tc[i].end;
    }
    return tc.size();
}

// Rows 8-11 from an (unsorted) location segment of one query: what the home GPU of the
// sharded path computes after the lists came back from their owners.
u32 orc_reduce_query(const void* dbp, const u64* locs, u64 n, u64 query_len, u32 max_cand, u32 emulate_ranks,
                     u64 insert_size_max, u32 quirk_seq_drop, u32* out /* max_cand x 4 */) {
    const Db& db = *(const Db*)dbp;
    std::vector<u64> m(locs, locs + n);
    std::sort(m.begin(), m.end());
    std::vector<TgtCand> tc;
    target_candidates(m, (u32)(2 + std::max<u64>(query_len, insert_size_max) / db.tgt_winstride), tc);
    u32 P = emulate_ranks < 1 ? 1 : emulate_ranks;
    std::vector<std::vector<Cand>> L(P);
    for (const auto& c : tc) {
        u32 tax = c.tgt < db.n_targets ? db.tgt2tax[c.tgt] : (0x80000000u | c.tgt);
        top_insert(L[c.tgt % P], Cand{tax, c.hits, c.beg, c.end}, max_cand);
    }
    if (P > 1) tree_fold(L, max_cand, quirk_seq_drop != 0);
    for (size_t i = 0; i < L[0].size(); ++i) { out[4*i] = L[0][i].tax; out[4*i+1] = L[0][i].hits; out[4*i+2] = L[0][i].beg; out[4*i+3] = L[0][i].end; }
    return (u32)L[0].size();
}

// Fold P explicit per-rank candidate lists (row 11 in isolation).
// lists: P x max_cand x (tax,hits); n[r] entries valid.  Result in out.
u32 orc_tree_fold(u32 P, u32 max_cand, const u32* lists, const u32* n, u32 quirk_seq_drop, u32* out) {
    std::vector<std::vector<Cand>> L(P);
    for (u32 r = 0; r < P; ++r)
        for (u32 i = 0; i < n[r]; ++i)
            L[r].push_back(Cand{lists[(r * max_cand + i) * 2], lists[(r * max_cand + i) * 2 + 1], 0, 0});
    if (P > 1) tree_fold(L, max_cand, quirk_seq_drop != 0);
    for (size_t i = 0; i < L[0].size(); ++i) { out[2 * i] = L[0][i].tax; out[2 * i + 1] = L[0][i].hits; }
    return (u32)L[0].size();
}

// Row 10 in isolation: insert a sequence of (tax,hits) candidates into an empty bounded
// list with the reference's insert(); returns the list as (tax, hits, index of the inserted
// candidate) triples.  Used to check the engine's order-free formulation of the same list.
u32 orc_insert_sequence(u32 n, const u32* tax, const u32* hits, u32 max_cand, u32* out) {
    std::vector<Cand> top;
    for (u32 i = 0; i < n; ++i) top_insert(top, Cand{tax[i], hits[i], i, 0}, max_cand);
    for (size_t i = 0; i < top.size(); ++i) { out[3 * i] = top[i].tax; out[3 * i + 1] = top[i].hits; out[3 * i + 2] = top[i].beg; }
    return (u32)top.size();
}

// ---- row 12: src/classification.cpp:235-265 (classify) + ranked_lca ----------
// cands: n x (tax_key,hits); lineage: n_taxa x 21 taxon indices (0xFFFFFFFF = null)
// indexed by (key & 0x7FFFFFFF), rank index as in src/taxonomy.h:62-85; rank_of[idx].
// Returns the taxon key of the classification or 0xFFFFFFFF for "none".
// ranked_lca: src/taxonomy.h:531-537 (first rank <= root where both lineages
// hold the same non-null taxon).
u32 orc_classify(const u32* cands, u32 n, const u32* lineage, const uint8_t* rank_of,
                 u32 hits_min, float hits_diff_fraction, u32 highest_rank) {
    const u32 NONE = 0xFFFFFFFFu;
    if (n == 0 || cands[0] == NONE) return NONE;
    u64 h0 = cands[1];
    if (h0 < hits_min) return NONE;
    u32 lca = cands[0] & 0x7FFFFFFFu;
    const float thr = h0 > hits_min ? (float)(h0 - hits_min) * hits_diff_fraction : 0.0f;
    for (u32 i = 1; i < n; ++i) {
        if ((float)(u64)cands[2 * i + 1] > thr) {
            u32 b = cands[2 * i] == NONE ? NONE : (cands[2 * i] & 0x7FFFFFFFu);
            u32 r = NONE;
            if (b != NONE) {
                for (int j = 0; j <= 20; ++j) {       // rank::root == 20
                    u32 x = lineage[(u64)lca * 21 + j];
                    if (x != NONE && x == lineage[(u64)b * 21 + j]) { r = x; break; }
                }
            }
            lca = r;
            if (lca == NONE || rank_of[lca] > highest_rank) return NONE;
        } else break;
    }
    return rank_of[lca] <= highest_rank ? lca : NONE;
}

} // extern "C"
