/* mcq_open.hpp -- header-only C++ helper above the two C ABIs (include/mcq.h, include/mcq_host.h): the reference's shard files
 * -> the queryable GPU handle, by whichever route fits the database's size.  Used by the drop-in CLIs (csrc/host/) and by the
 * reference-side binding (integration/mcq_reference_binding.cpp).
 *
 * Replaces sketch_database::read (src/sketch_database.h:858-952) + hash_multimap::deserialize (src/hash_multimap.h:923-964):
 *   small databases: mcq_refdb_open unions the P shard tables on the host (16 B per location) -> mcq_db_create;
 *   from `stream_min_bytes` of shard files on: mcq_refdb_open_meta reads only the heads; the key records are streamed to the GPU in
 *   chunks of 4 M locations (48 MB of host memory each; up to 8 files are parsed at a time by as many host threads), the P ranks
 *   merged there per feature-hash range (mcq_parts_builder_*), the handle made from the parts (32-bit global-window words) -- host
 *   memory stays at a few chunks whatever the database's size.                                                              */
#ifndef MCQ_OPEN_HPP
#define MCQ_OPEN_HPP
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mcq.h"
#include "mcq_host.h"

#define MCQ_STREAM_LOAD_DEFAULT_MIN_BYTES (1ull << 30)

/* MCQ_BUILD_TRACE=1: resident set of the process at the steps of the streaming route, on stderr */
inline void mcq_trace_rss(const char* what) {
    if (!std::getenv("MCQ_BUILD_TRACE")) return;
    long pages = 0, rss = 0;
    if (FILE* f = std::fopen("/proc/self/statm", "r")) { if (std::fscanf(f, "%ld %ld", &pages, &rss) != 2) rss = 0; std::fclose(f); }
    std::fprintf(stderr, "[mcq_open] %-34s rss %8.1f MB\n", what, rss * 4096.0 / 1048576.0);
}

/* threshold in bytes: MCQ_STREAM_LOAD_MIN_MB overrides the default of 1 GB (0 = always stream) */
inline uint64_t mcq_stream_load_min_bytes() {
    if (const char* e = std::getenv("MCQ_STREAM_LOAD_MIN_MB")) return std::strtoull(e, nullptr, 10) << 20;
    return MCQ_STREAM_LOAD_DEFAULT_MIN_BYTES;
}

/* opens <prefix>.db_0 .. db_<P-1>: the full host-side union below the threshold, heads only (mcq_refdb_open_meta) from it on;
 * *streamed says which.  0 on success, else the text in err. */
inline int mcq_open_refdb(const std::string& prefix, uint32_t P, uint64_t stream_min_bytes, mcq_refdb** rdb, bool* streamed, std::string& err) {
    uint64_t total = 0;
    for (uint32_t r = 0; r < P; ++r)
        if (FILE* f = std::fopen((prefix + ".db_" + std::to_string(r)).c_str(), "rb")) { std::fseek(f, 0, SEEK_END); total += (uint64_t)std::ftell(f); std::fclose(f); }
    *streamed = total >= stream_min_bytes;
    if (*streamed ? mcq_refdb_open_meta(prefix.c_str(), P, rdb) : mcq_refdb_open(prefix.c_str(), P, rdb)) { err = mcq_host_last_error(); return -1; }
    return 0;
}

/* the handle of shard `shard_id` of `n_shards` from an opened mcq_refdb; tgt2tax: [n_targets] taxon keys (mcq_refdb_tgt2tax or the
 * caller's own).  0 on success, else the text in err. */
inline int mcq_make_db(mcq_refdb* rdb, bool streamed, const uint32_t* tgt2tax, uint32_t n_shards, uint32_t shard_id, int device,
                       mcq_db** out, std::string& err) {
    mcq_refdb_info info; mcq_refdb_get_info(rdb, &info);
    if (!streamed) {
        mcq_db_desc d; std::memset(&d, 0, sizeof(d));
        d.k = info.k; d.sketch_size = info.q_sketch_size; d.winlen = info.q_winlen; d.winstride = info.q_winstride;
        d.tgt_winstride = info.winstride; d.n_targets = info.n_targets; d.n_keys = info.n_keys; d.n_locs = info.n_locs;
        d.keys = mcq_refdb_keys(rdb); d.list_off = mcq_refdb_list_off(rdb); d.locs = mcq_refdb_locs(rdb); d.tgt2tax = tgt2tax;
        d.n_shards = n_shards; d.shard_id = shard_id; d.flags = 0; d.device = device;
        static const uint64_t zero_off[1] = {0};
        if (!d.list_off) d.list_off = zero_off;
        if (mcq_db_create(&d, out)) { err = mcq_last_error(); return -1; }
        return 0;
    }
    mcq_trace_rss("heads read");
    std::vector<uint32_t> tw(info.n_targets);
    if (mcq_refdb_tgt_windows(rdb, tw.data())) { err = mcq_host_last_error(); return -1; }
    mcq_parts_builder_desc bd; std::memset(&bd, 0, sizeof(bd));
    bd.k = info.k; bd.sketch_size = info.q_sketch_size; bd.winlen = info.q_winlen; bd.winstride = info.q_winstride; bd.tgt_winstride = info.winstride;
    bd.n_targets = info.n_targets; bd.tgt_windows = tw.data(); bd.expected_locations = info.n_locs;
    bd.n_shards = n_shards; bd.shard_id = shard_id; bd.device = device;
    mcq_parts_builder* pb = nullptr;
    if (mcq_parts_builder_create(&bd, &pb)) { err = mcq_build_last_error(); return -1; }
    mcq_trace_rss("builder created (GPU runtime up)");
    // the files are parsed by up to 8 host threads (one open stream and one 48 MB chunk each: record parsing is ~1 GB/s per thread),
    // which take turns handing their chunks to the builder (MCQ_STREAM_LOAD_THREADS overrides; 1 = the calling thread alone)
    const uint64_t chunk = 1u << 22;
    unsigned n_thr = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("MCQ_STREAM_LOAD_THREADS")) n_thr = (unsigned)std::strtoul(e, nullptr, 10);
    n_thr = std::max(1u, std::min(std::min(n_thr, 8u), info.n_ranks));
    std::mutex mu; std::string first_err;
    auto work = [&](unsigned tix) {
        std::vector<uint32_t> cf(chunk), ct(chunk), cw(chunk);
        for (uint32_t r = tix; r < info.n_ranks; r += n_thr) {
            mcq_shard_stream* st = nullptr;
            if (mcq_shard_stream_open(rdb, r, &st)) { std::lock_guard<std::mutex> g(mu); if (first_err.empty()) first_err = mcq_host_last_error(); return; }
            for (;;) {
                uint64_t n = 0;
                if (mcq_shard_stream_next(st, cf.data(), ct.data(), cw.data(), chunk, &n)) {
                    std::lock_guard<std::mutex> g(mu); if (first_err.empty()) first_err = mcq_host_last_error(); n = 0; mcq_shard_stream_close(st); return;
                }
                if (!n) break;
                std::lock_guard<std::mutex> g(mu);
                if (!first_err.empty()) { mcq_shard_stream_close(st); return; }           // (another thread failed: stop)
                if (mcq_parts_builder_add(pb, cf.data(), ct.data(), cw.data(), n, 0)) { first_err = mcq_build_last_error(); mcq_shard_stream_close(st); return; }
            }
            mcq_shard_stream_close(st);
            mcq_trace_rss("a shard file streamed");
        }
    };
    if (n_thr == 1) work(0);
    else {
        std::vector<std::thread> thr;
        for (unsigned t = 0; t < n_thr; ++t) thr.emplace_back(work, t);
        for (auto& t : thr) t.join();
    }
    if (!first_err.empty()) { err = first_err; mcq_parts_builder_free(pb); return -1; }
    mcq_parts* parts = nullptr;
    if (mcq_parts_builder_finish(pb, &parts)) { err = mcq_build_last_error(); mcq_parts_builder_free(pb); return -1; }
    mcq_trace_rss("ranges sorted: parts made");
    const int rc = mcq_db_from_parts(parts, tgt2tax, n_shards, shard_id, 0, out);
    mcq_parts_free(parts);
    if (rc) { err = mcq_build_last_error(); return -1; }
    mcq_trace_rss("table made");
    return 0;
}
#endif
