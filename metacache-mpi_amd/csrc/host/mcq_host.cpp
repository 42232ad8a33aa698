// mcq_host.cpp -- host companions of the engine: reference shard reader, taxonomy keys,
// classification (include/mcq_host.h).  Plain C++14, no GPU, no MPI.
#include "../../../include/mcq_host.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

const uint64_t kDbVersion = 20181001;      // MC_DB_VERSION
const int kNumRanks = 21;                  // taxonomy::num_ranks; rank::root == 20, rank::none == 21

struct Taxon { int64_t id, parent; uint8_t rank; std::string name; uint64_t windows; };

struct Reader {
    std::vector<unsigned char> buf; size_t pos = 0; bool ok = true;
    template <class T> T get() {
        T v{};
        if (pos + sizeof(T) > buf.size()) { ok = false; return v; }
        std::memcpy(&v, buf.data() + pos, sizeof(T)); pos += sizeof(T);
        return v;
    }
    std::string str() {                     // u64 length + bytes (src/io_serialize.h:48-55)
        uint64_t n = get<uint64_t>();
        if (!ok || pos + n > buf.size()) { ok = false; return ""; }
        std::string s((const char*)buf.data() + pos, n); pos += n;
        return s;
    }
};

}  // namespace

struct mcq_refdb {
    mcq_refdb_info info{};
    std::vector<Taxon> taxa;
    std::unordered_map<int64_t, uint32_t> by_id;
    std::vector<uint32_t> lineage;          // n_taxa x 21, taxon indices or MCQ_NO_TAXON
    std::vector<uint32_t> keys; std::vector<uint64_t> off, locs;
    // mcq_refdb_open_meta: no table in host memory; per shard file where its key records begin, how many there are
    std::string prefix;
    std::vector<uint64_t> table_pos, table_keys, table_locs, file_bytes;
    std::vector<uint32_t> tgt_windows;      // windows of every target, from the rank that owns it
};

namespace {
// sequential reader of one file through a fixed buffer (the streaming route never holds a shard file in memory)
struct FileReader {
    FILE* f = nullptr; std::vector<unsigned char> buf; size_t pos = 0, end = 0; bool ok = true; uint64_t consumed = 0;
    explicit FileReader(size_t cap = 1u << 22) : buf(cap) {}
    ~FileReader() { if (f) std::fclose(f); }
    bool open(const std::string& path) { f = std::fopen(path.c_str(), "rb"); return f != nullptr; }
    bool seek(uint64_t at) { pos = end = 0; consumed = at; return std::fseek(f, (long)at, SEEK_SET) == 0; }
    bool fill(size_t need) {                 // makes `need` bytes available at buf[pos..)
        if (end - pos >= need) return true;
        if (need > buf.size()) buf.resize(need);
        std::memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0;
        while (end < need) {
            const size_t n = std::fread(buf.data() + end, 1, buf.size() - end, f);
            if (n == 0) return false;
            end += n;
        }
        return true;
    }
    template <class T> T get() {
        T v{};
        if (!ok || !fill(sizeof(T))) { ok = false; return v; }
        std::memcpy(&v, buf.data() + pos, sizeof(T)); pos += sizeof(T); consumed += sizeof(T);
        return v;
    }
    const unsigned char* bytes(size_t n) {   // n bytes in place (valid until the next call)
        if (!ok || !fill(n)) { ok = false; return nullptr; }
        const unsigned char* p = buf.data() + pos; pos += n; consumed += n;
        return p;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (!ok || n > (1u << 24)) { ok = false; return ""; }
        const unsigned char* p = bytes((size_t)n);
        return p ? std::string((const char*)p, (size_t)n) : std::string();
    }
};
void build_lineages(mcq_refdb* db) {
    // ranked lineages: walk the parents, record every ranked ancestor (incl. the taxon itself)
    // at lineage[rank] (taxonomy::ranks, src/taxonomy.h:576-597)
    for (uint32_t i = 0; i < db->taxa.size(); ++i) db->by_id[db->taxa[i].id] = i;
    db->lineage.assign((size_t)db->taxa.size() * kNumRanks, MCQ_NO_TAXON);
    for (uint32_t i = 0; i < db->taxa.size(); ++i) {
        int64_t cur = db->taxa[i].id;
        while (cur != 0) {
            auto it = db->by_id.find(cur);
            if (it == db->by_id.end()) break;
            const Taxon& t = db->taxa[it->second];
            if (t.rank < kNumRanks) db->lineage[(size_t)i * kNumRanks + t.rank] = it->second;
            cur = (t.parent != cur) ? t.parent : 0;
        }
    }
}
}  // namespace

static bool read_file(const std::string& path, std::vector<unsigned char>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    bool ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
}

extern "C" const char* mcq_host_last_error(void) { return g_err.c_str(); }

extern "C" int mcq_refdb_open(const char* prefix, uint32_t n_ranks, mcq_refdb** out) {
    if (!prefix || !out || n_ranks < 1) return fail("bad argument");
    mcq_refdb* db = new mcq_refdb();
    // (key, location) pairs of all shards; the union list of a key is the multiset union of
    // the per-rank lists, each target living on exactly one rank (src/sketch_database.h:540)
    std::vector<std::pair<uint32_t, uint64_t>> all;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        Reader rd;
        const std::string path = std::string(prefix) + ".db_" + std::to_string(r);
        if (!read_file(path, rd.buf)) { delete db; return fail("can't open file " + path); }
        if (rd.get<uint64_t>() != kDbVersion) { delete db; return fail("Database " + path + " is incompatible (version)"); }
        uint8_t w[6]; for (auto& x : w) x = rd.get<uint8_t>();
        if (w[0] != 4 || w[1] != 4 || w[2] != 4 || w[3] != 1 || w[4] != 8 || w[5] != kNumRanks) {
            delete db; return fail("Database " + path + " is incompatible due to different data type sizes");
        }
        uint64_t p[9]; for (auto& x : p) x = rd.get<uint64_t>();
        const uint64_t ntaxa = rd.get<uint64_t>();
        std::vector<Taxon> taxa; taxa.reserve(ntaxa);
        for (uint64_t i = 0; i < ntaxa && rd.ok; ++i) {
            Taxon t; t.id = rd.get<int64_t>(); t.parent = rd.get<int64_t>(); t.rank = rd.get<uint8_t>();
            t.name = rd.str(); rd.str(); rd.get<uint64_t>(); t.windows = rd.get<uint64_t>();
            taxa.push_back(std::move(t));
        }
        const uint32_t ntargets = rd.get<uint32_t>();
        if (r == 0) {
            db->info.k = (uint32_t)p[0]; db->info.sketch_size = (uint32_t)p[1]; db->info.winlen = (uint32_t)p[2];
            db->info.winstride = (uint32_t)p[3]; db->info.q_sketch_size = (uint32_t)p[5]; db->info.q_winlen = (uint32_t)p[6];
            db->info.q_winstride = (uint32_t)p[7]; db->info.max_locs_per_feature = (uint32_t)p[8];
            db->info.n_targets = ntargets; db->info.n_taxa = (uint32_t)ntaxa; db->info.n_ranks = n_ranks;
            db->taxa = std::move(taxa);
        } else if (ntargets != db->info.n_targets || ntaxa != db->info.n_taxa) {
            delete db; return fail("shard " + path + " does not belong to the same database");
        }
        if (ntargets >= 1) {
            const uint64_t nkeys = rd.get<uint64_t>(); rd.get<uint64_t>();
            for (uint64_t i = 0; i < nkeys && rd.ok; ++i) {
                const uint32_t key = rd.get<uint32_t>(); const uint8_t n = rd.get<uint8_t>();
                if (n == 0) continue;
                const uint64_t n1 = rd.get<uint64_t>(); const size_t tpos = rd.pos; rd.pos += 4 * n1;
                const uint64_t n2 = rd.get<uint64_t>(); const size_t wpos = rd.pos; rd.pos += 4 * n2;
                if (n1 != n || n2 != n || rd.pos > rd.buf.size()) { rd.ok = false; break; }
                for (uint32_t j = 0; j < n; ++j) {
                    uint32_t t, wi; std::memcpy(&t, rd.buf.data() + tpos + 4 * j, 4); std::memcpy(&wi, rd.buf.data() + wpos + 4 * j, 4);
                    all.emplace_back(key, ((uint64_t)t << 32) | wi);
                }
            }
        }
        if (!rd.ok) { delete db; return fail("Database " + path + " is truncated or corrupt"); }
    }
    std::sort(all.begin(), all.end());
    db->off.push_back(0);
    for (size_t i = 0; i < all.size(); ++i) {
        if (i == 0 || all[i].first != all[i - 1].first) { if (i) db->off.push_back(i); db->keys.push_back(all[i].first); }
        db->locs.push_back(all[i].second);
    }
    if (!all.empty()) db->off.push_back(all.size());
    db->info.n_keys = db->keys.size(); db->info.n_locs = db->locs.size();
    build_lineages(db);
    *out = db;
    return 0;
}

extern "C" int mcq_refdb_close(mcq_refdb* db) { delete db; return 0; }

// ---- the streaming route (r04): shard files of any size without a host-side union ------------------------------------------
// mcq_refdb_open (above) materialises every (key, location) of all P shards in one host vector and sorts it: 16 B per location,
// 240 GB and minutes for a RefSeq-scale database.  The streaming route reads only the head of every file here (parameters,
// taxa, target count: what classify and the taxon keys need), and hands the key records out in chunks of (feature, target,
// window) triples in file order; the consumer (mcq_parts_builder_*, include/mcq.h) turns them into global-window words on the
// GPU, merges the ranks per feature-hash range there and makes the table parts.  Host memory: one read buffer + one chunk.
extern "C" int mcq_refdb_open_meta(const char* prefix, uint32_t n_ranks, mcq_refdb** out) {
    if (!prefix || !out || n_ranks < 1) return fail("bad argument");
    mcq_refdb* db = new mcq_refdb();
    db->prefix = prefix;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        const std::string path = std::string(prefix) + ".db_" + std::to_string(r);
        FileReader rd;
        if (!rd.open(path)) { delete db; return fail("can't open file " + path); }
        if (rd.get<uint64_t>() != kDbVersion) { delete db; return fail("Database " + path + " is incompatible (version)"); }
        uint8_t w[6]; for (auto& x : w) x = rd.get<uint8_t>();
        if (w[0] != 4 || w[1] != 4 || w[2] != 4 || w[3] != 1 || w[4] != 8 || w[5] != kNumRanks) {
            delete db; return fail("Database " + path + " is incompatible due to different data type sizes");
        }
        uint64_t p[9]; for (auto& x : p) x = rd.get<uint64_t>();
        const uint64_t ntaxa = rd.get<uint64_t>();
        std::vector<Taxon> taxa;
        for (uint64_t i = 0; i < ntaxa && rd.ok; ++i) {
            Taxon t; t.id = rd.get<int64_t>(); t.parent = rd.get<int64_t>(); t.rank = rd.get<uint8_t>();
            t.name = rd.str(); rd.str(); rd.get<uint64_t>(); t.windows = rd.get<uint64_t>();
            taxa.push_back(std::move(t));
        }
        const uint32_t ntargets = rd.get<uint32_t>();
        if (!rd.ok) { delete db; return fail("Database " + path + " is truncated or corrupt"); }
        if (r == 0) {
            db->info.k = (uint32_t)p[0]; db->info.sketch_size = (uint32_t)p[1]; db->info.winlen = (uint32_t)p[2];
            db->info.winstride = (uint32_t)p[3]; db->info.q_sketch_size = (uint32_t)p[5]; db->info.q_winlen = (uint32_t)p[6];
            db->info.q_winstride = (uint32_t)p[7]; db->info.max_locs_per_feature = (uint32_t)p[8];
            db->info.n_targets = ntargets; db->info.n_taxa = (uint32_t)ntaxa; db->info.n_ranks = n_ranks;
            db->tgt_windows.assign(ntargets, 0);
        } else if (ntargets != db->info.n_targets || ntaxa != db->info.n_taxa) {
            delete db; return fail("shard " + path + " does not belong to the same database");
        }
        // windows of the targets this rank owns (`windows` is non-zero only there, src/taxonomy.h:326-335)
        for (const Taxon& t : taxa)
            if (t.id < 0 && (uint64_t)(-t.id - 1) < ntargets && t.windows) {
                if (t.windows >= (1ull << 32)) { delete db; return fail("a target with 2^32 windows or more"); }
                db->tgt_windows[(size_t)(-t.id - 1)] = (uint32_t)t.windows;
            }
        if (r == 0) db->taxa = std::move(taxa);
        uint64_t nkeys = 0, nlocs = 0;
        if (ntargets >= 1) { nkeys = rd.get<uint64_t>(); nlocs = rd.get<uint64_t>(); }
        if (!rd.ok) { delete db; return fail("Database " + path + " is truncated or corrupt"); }
        db->table_pos.push_back(rd.consumed); db->table_keys.push_back(nkeys); db->table_locs.push_back(nlocs);
        std::fseek(rd.f, 0, SEEK_END);
        db->file_bytes.push_back((uint64_t)std::ftell(rd.f));
        db->info.n_locs += nlocs;            // (n_keys of the union is not known without reading the tables: left 0)
    }
    build_lineages(db);
    *out = db;
    return 0;
}
extern "C" int mcq_refdb_tgt_windows(const mcq_refdb* db, uint32_t* out) {
    if (!db || !out) return fail("bad argument");
    if (db->tgt_windows.size() != db->info.n_targets) return fail("the handle was not opened with mcq_refdb_open_meta");
    std::memcpy(out, db->tgt_windows.data(), db->tgt_windows.size() * 4);
    return 0;
}
extern "C" int mcq_refdb_file_stats(const mcq_refdb* db, uint32_t rank, uint64_t* bytes, uint64_t* n_keys, uint64_t* n_locs) {
    if (!db || rank >= db->table_pos.size()) return fail("bad argument");
    if (bytes) *bytes = db->file_bytes[rank];
    if (n_keys) *n_keys = db->table_keys[rank];
    if (n_locs) *n_locs = db->table_locs[rank];
    return 0;
}

struct mcq_shard_stream {
    FileReader rd{1u << 24};
    uint64_t keys_left = 0;
};
extern "C" int mcq_shard_stream_open(const mcq_refdb* db, uint32_t rank, mcq_shard_stream** out) {
    if (!db || !out || rank >= db->table_pos.size()) return fail("bad argument (the handle must come from mcq_refdb_open_meta)");
    mcq_shard_stream* s = new mcq_shard_stream();
    const std::string path = db->prefix + ".db_" + std::to_string(rank);
    if (!s->rd.open(path) || !s->rd.seek(db->table_pos[rank])) { delete s; return fail("can't open file " + path); }
    s->keys_left = db->table_keys[rank];
    *out = s;
    return 0;
}
extern "C" int mcq_shard_stream_next(mcq_shard_stream* s, uint32_t* feat, uint32_t* tgt, uint32_t* win, uint64_t cap, uint64_t* n_out) {
    if (!s || !feat || !tgt || !win || !n_out || cap < 255) return fail("bad argument (a chunk holds at least 255 locations)");
    uint64_t n = 0;
    while (s->keys_left && n + 255 <= cap) {                     // (whole key records only: a list has at most 255 entries)
        const uint32_t key = s->rd.get<uint32_t>(); const uint8_t cnt = s->rd.get<uint8_t>();
        --s->keys_left;
        if (!s->rd.ok) return fail("shard file is truncated or corrupt");
        if (cnt == 0) continue;                                  // (an empty bucket is written without its columns)
        const uint64_t n1 = s->rd.get<uint64_t>();
        if (!s->rd.ok || n1 != cnt) return fail("shard file is truncated or corrupt");
        const unsigned char* tp = s->rd.bytes(4 * (size_t)cnt);
        if (!tp) return fail("shard file is truncated or corrupt");
        std::memcpy(tgt + n, tp, 4 * (size_t)cnt);
        const uint64_t n2 = s->rd.get<uint64_t>();
        if (!s->rd.ok || n2 != cnt) return fail("shard file is truncated or corrupt");
        const unsigned char* wp = s->rd.bytes(4 * (size_t)cnt);
        if (!wp) return fail("shard file is truncated or corrupt");
        std::memcpy(win + n, wp, 4 * (size_t)cnt);
        for (uint32_t j = 0; j < cnt; ++j) feat[n + j] = key;
        n += cnt;
    }
    *n_out = n;
    return 0;
}
extern "C" int mcq_shard_stream_close(mcq_shard_stream* s) { delete s; return 0; }

// ---- shard writer: the exact inverse of the reader above
namespace {
struct Writer {
    FILE* f; bool ok = true;
    template <class T> void put(T v) { if (ok && std::fwrite(&v, sizeof(T), 1, f) != 1) ok = false; }
    void str(const char* s) {
        const uint64_t n = s ? std::strlen(s) : 0;
        put<uint64_t>(n);
        if (ok && n && std::fwrite(s, 1, n, f) != n) ok = false;
    }
};
}  // namespace

extern "C" int mcq_refdb_write_shard(const char* path, const mcq_shard_params* p, const mcq_taxon_rec* taxa, uint64_t n_taxa,
                                     uint32_t n_targets, const uint32_t* keys, const uint64_t* list_off, const uint64_t* locs,
                                     uint64_t n_keys) {
    if (!path || !p || (n_taxa && !taxa) || (n_keys && (!keys || !list_off || !locs))) return fail("bad argument");
    for (uint64_t i = 0; i < n_keys; ++i)
        if (list_off[i + 1] < list_off[i] || list_off[i + 1] - list_off[i] > 255)
            return fail("a list has more than 255 locations (bucket size type is uint8, src/config.h:77)");
    Writer w; w.f = std::fopen(path, "wb");
    if (!w.f) return fail(std::string("can't open file ") + path);
    w.put<uint64_t>(kDbVersion);
    const uint8_t widths[6] = {4, 4, 4, 1, 8, (uint8_t)kNumRanks};
    for (uint8_t x : widths) w.put<uint8_t>(x);
    const uint64_t pv[9] = {p->k, p->sketch_size, p->winlen, p->winstride, p->q_k, p->q_sketch_size, p->q_winlen, p->q_winstride,
                            p->max_locs_per_feature};
    for (uint64_t x : pv) w.put<uint64_t>(x);
    w.put<uint64_t>(n_taxa);
    for (uint64_t i = 0; i < n_taxa; ++i) {
        w.put<int64_t>(taxa[i].id); w.put<int64_t>(taxa[i].parent); w.put<uint8_t>(taxa[i].rank);
        w.str(taxa[i].name); w.str(taxa[i].file);
        w.put<uint64_t>(taxa[i].index); w.put<uint64_t>(taxa[i].windows);
    }
    w.put<uint32_t>(n_targets);
    if (n_targets >= 1) {
        uint64_t nonempty = 0;
        for (uint64_t i = 0; i < n_keys; ++i) nonempty += list_off[i + 1] > list_off[i];
        w.put<uint64_t>(nonempty);
        w.put<uint64_t>(n_keys ? list_off[n_keys] - list_off[0] : 0);
        std::vector<uint32_t> col;
        for (uint64_t i = 0; i < n_keys; ++i) {
            const uint64_t b = list_off[i], n = list_off[i + 1] - b;
            if (n == 0) continue;
            w.put<uint32_t>(keys[i]); w.put<uint8_t>((uint8_t)n);
            col.resize(n);
            w.put<uint64_t>(n);
            for (uint64_t j = 0; j < n; ++j) col[j] = (uint32_t)(locs[b + j] >> 32);
            if (w.ok && std::fwrite(col.data(), 4, n, w.f) != n) w.ok = false;
            w.put<uint64_t>(n);
            for (uint64_t j = 0; j < n; ++j) col[j] = (uint32_t)locs[b + j];
            if (w.ok && std::fwrite(col.data(), 4, n, w.f) != n) w.ok = false;
        }
    }
    const bool ok = w.ok;
    if (std::fclose(w.f) != 0 || !ok) return fail(std::string("write error on ") + path);
    return 0;
}
extern "C" int mcq_refdb_get_info(const mcq_refdb* db, mcq_refdb_info* out) { if (!db || !out) return fail("bad argument"); *out = db->info; return 0; }
extern "C" const uint32_t* mcq_refdb_keys(const mcq_refdb* db) { return db->keys.data(); }
extern "C" const uint64_t* mcq_refdb_list_off(const mcq_refdb* db) { return db->off.empty() ? nullptr : db->off.data(); }
extern "C" const uint64_t* mcq_refdb_locs(const mcq_refdb* db) { return db->locs.data(); }

extern "C" int mcq_refdb_tgt2tax(const mcq_refdb* db, uint32_t merge_below_rank, uint32_t* out) {
    if (!db || !out) return fail("bad argument");
    for (uint32_t t = 0; t < db->info.n_targets; ++t) {
        auto it = db->by_id.find(-(int64_t)t - 1);              // taxon_id_of_target (src/sketch_database.h:149-150)
        if (it == db->by_id.end()) return fail("target " + std::to_string(t) + " has no sequence-level taxon");
        uint32_t a = (merge_below_rank > 0 && merge_below_rank < (uint32_t)kNumRanks)
                         ? db->lineage[(size_t)it->second * kNumRanks + merge_below_rank] : MCQ_NO_TAXON;
        out[t] = (a != MCQ_NO_TAXON) ? a : (0x80000000u | it->second);
    }
    return 0;
}

static inline bool valid_key(const mcq_refdb* db, uint32_t key) { return key != MCQ_NO_TAXON && (key & 0x7FFFFFFFu) < db->taxa.size(); }
extern "C" int64_t mcq_refdb_taxon_id(const mcq_refdb* db, uint32_t key) { return valid_key(db, key) ? db->taxa[key & 0x7FFFFFFFu].id : 0; }
extern "C" uint32_t mcq_refdb_taxon_rank(const mcq_refdb* db, uint32_t key) { return valid_key(db, key) ? db->taxa[key & 0x7FFFFFFFu].rank : MCQ_RANK_NONE; }
extern "C" const char* mcq_refdb_taxon_name(const mcq_refdb* db, uint32_t key) { return valid_key(db, key) ? db->taxa[key & 0x7FFFFFFFu].name.c_str() : "--"; }
extern "C" uint32_t mcq_refdb_ancestor(const mcq_refdb* db, uint32_t key, uint32_t rank) {
    if (!valid_key(db, key) || rank >= (uint32_t)kNumRanks) return MCQ_NO_TAXON;
    return db->lineage[(size_t)(key & 0x7FFFFFFFu) * kNumRanks + rank];
}

extern "C" uint32_t mcq_refdb_classify(const mcq_refdb* db, const uint32_t* c, uint32_t n,
                                       uint32_t hits_min, float hits_diff_fraction, uint32_t highest_rank) {
    if (n == 0 || !valid_key(db, c[0])) return MCQ_NO_TAXON;
    const uint64_t h0 = c[1];
    if (h0 < hits_min) return MCQ_NO_TAXON;                     // below threshold: not classifiable
    uint32_t lca = c[0] & 0x7FFFFFFFu;
    const float thr = h0 > hits_min ? (float)(h0 - hits_min) * hits_diff_fraction : 0.0f;
    for (uint32_t i = 1; i < n; ++i) {
        if (!((float)(uint64_t)c[4 * i + 1] > thr)) break;
        uint32_t r = MCQ_NO_TAXON;
        if (valid_key(db, c[4 * i])) {
            const uint32_t b = c[4 * i] & 0x7FFFFFFFu;
            for (int j = 0; j <= 20; ++j) {                      // ranked_lca: first shared non-null rank up to root
                uint32_t x = db->lineage[(size_t)lca * kNumRanks + j];
                if (x != MCQ_NO_TAXON && x == db->lineage[(size_t)b * kNumRanks + j]) { r = x; break; }
            }
        }
        lca = r;
        if (lca == MCQ_NO_TAXON || db->taxa[lca].rank > highest_rank) return MCQ_NO_TAXON;
    }
    return db->taxa[lca].rank <= highest_rank ? lca : MCQ_NO_TAXON;
}

extern "C" uint32_t mcq_default_hits_min(uint32_t s) { return s >= 6 ? (uint32_t)(s / 3.0) : (s >= 4 ? 2u : 1u); }

static const char* kRankNames[] = {"sequence", "form", "variety", "subspecies", "species", "subgenus", "genus", "subtribe",
                                   "tribe", "subfamily", "family", "suborder", "order", "subclass", "class", "subphylum",
                                   "phylum", "subkingdom", "kingdom", "domain", "root", "none"};
extern "C" const char* mcq_rank_name(uint32_t r) { return kRankNames[r < 21 ? r : 21]; }
extern "C" uint32_t mcq_rank_from_name(const char* name) {
    if (!name) return MCQ_RANK_NONE;
    std::string s(name);
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    static const std::map<std::string, uint32_t> m = {
        {"sequence", 0}, {"genome", 0}, {"form", 1}, {"forma", 1}, {"variety", 2}, {"varietas", 2}, {"subspecies", 3},
        {"species", 4}, {"species group", 5}, {"species subgroup", 5}, {"subgenus", 5}, {"genus", 6}, {"subtribe", 7},
        {"tribe", 8}, {"subfamily", 9}, {"family", 10}, {"superfamily", 11}, {"parvorder", 11}, {"infraorder", 11},
        {"suborder", 11}, {"order", 12}, {"superorder", 13}, {"infraclass", 13}, {"subclass", 13}, {"class", 14},
        {"superclass", 15}, {"subphylum", 15}, {"phylum", 16}, {"division", 16}, {"superphylum", 17}, {"subkingdom", 17},
        {"kingdom", 18}, {"subdomain", 18}, {"superkingdom", 19}, {"domain", 19}, {"root", 20}};
    auto it = m.find(s);
    return it == m.end() ? MCQ_RANK_NONE : it->second;
}
