// mcq_query_cli -- `metacache query <db> r1.fq r2.fq -pairfiles ...` on one GPU: see mcq_cli_common.hpp for what is
// written and which reference code each part stands in for.
#include "mcq_cli_common.hpp"

int main(int argc, char** argv) {
    Options p;
    if (!parse_options(argc, argv, p)) return 2;
    mcq_refdb* rdb = nullptr; std::vector<uint32_t> t2t; uint32_t hitmin = 0;
    mcq_db* edb = nullptr;
    if (!open_database(p, &rdb, t2t, &edb, hitmin, 1, 0, 0)) return 1;

    const auto t_start = std::chrono::steady_clock::now();                  // the reference times map_queries_to_targets, readers included (src/mode_query.cpp:130-132)
    std::vector<Rec> r1, r2;
    if (!read_records(p.f1, r1)) { std::fprintf(stderr, "FAIL: can't open file %s\n", p.f1.c_str()); return 1; }
    const bool paired = p.paired();
    if (paired && !read_records(p.f2, r2)) { std::fprintf(stderr, "FAIL: can't open file %s\n", p.f2.c_str()); return 1; }
    const size_t nq = paired ? std::min(r1.size(), r2.size()) : r1.size();

    // the reads go through in batches of at most -batch queries and -batch-bases bases (mcq_query_pipelined: batch j+1 is
    // copied in and batch j-1 copied out while batch j computes; three host sets, the mapping lines of a batch are
    // written while the next ones run) -- the reference's blocks of numThreads x queryLimit reads, src/querying.h:1371-1377
    std::vector<size_t> cut{0};
    {
        size_t nb_q = 0; uint64_t nb_b = 0;
        for (size_t q = 0; q < nq; ++q) {
            const uint64_t len = r1[q].seq.size() + (paired ? r2[q].seq.size() : 0);
            if (nb_q && (nb_q == p.batch || nb_b + len > p.batch_bases)) { cut.push_back(q); nb_q = 0; nb_b = 0; }
            ++nb_q; nb_b += len;
        }
        cut.push_back(nq);
    }
    const size_t nb = cut.size() - 1;
    uint64_t max_q = 1, max_b = 1;
    for (size_t j = 0; j < nb; ++j) {
        uint64_t bb = 0;
        for (size_t q = cut[j]; q < cut[j + 1]; ++q) bb += r1[q].seq.size() + (paired ? r2[q].seq.size() : 0);
        max_q = std::max<uint64_t>(max_q, cut[j + 1] - cut[j]); max_b = std::max(max_b, bb);
    }
    mcq_ws* ws = nullptr;
    if (mcq_ws_create(edb, max_q, max_b + 1, 0, &ws)) { std::fprintf(stderr, "ABORT: %s\n", mcq_last_error()); return 1; }
    mcq_query_opts qo; qo.max_cand = p.maxcand; qo.emulate_ranks = p.P; qo.insert_size_max = p.insertsize;
    qo.flags = p.quirks ? MCQ_QUIRK_SEQ_DROP : 0;
    constexpr int NS = 3;
    std::string bases[NS]; std::vector<uint64_t> off[NS];
    std::vector<mcq_cand> cands[NS]; std::vector<uint32_t> ncand[NS];
    mcq_batch in[NS]; mcq_result res[NS]; uint64_t ticket[NS] = {0, 0, 0};
    for (int s = 0; s < NS; ++s) { cands[s].resize(max_q * p.maxcand); ncand[s].resize(max_q); }

    std::ofstream fout; if (!p.outfile.empty()) fout.open(p.outfile);
    std::ostream& os = p.outfile.empty() ? std::cout : fout;
    const Out o = make_out(rdb, p);
    write_head(os, o, p, hitmin);
    uint64_t assigned[MCQ_RANK_NONE + 1] = {0};
    auto finish = [&](size_t j) -> bool {
        const int s = (int)(j % NS);
        if (mcq_ws_wait(ws, ticket[s])) { std::fprintf(stderr, "FAIL: %s\n", mcq_last_error()); return false; }
        for (size_t q = cut[j]; q < cut[j + 1]; ++q)
            write_query(os, o, p, hitmin, r1[q].header, &cands[s][(q - cut[j]) * p.maxcand], ncand[s][q - cut[j]], assigned);
        return true;
    };
    for (size_t j = 0; j < nb; ++j) {
        const int s = (int)(j % NS);
        if (j >= 2 && !finish(j - 2)) return 1;                 // two batches in flight; set s was batch j-3's (finished)
        bases[s].clear(); off[s].assign(1, 0);
        for (size_t q = cut[j]; q < cut[j + 1]; ++q) {
            bases[s] += r1[q].seq; off[s].push_back(bases[s].size());
            if (paired) { bases[s] += r2[q].seq; off[s].push_back(bases[s].size()); }
        }
        std::memset(&in[s], 0, sizeof(mcq_batch));
        in[s].n_seqs = off[s].size() - 1; in[s].bases = bases[s].data(); in[s].seq_off = off[s].data(); in[s].paired = paired ? 1 : 0;
        res[s].cands = cands[s].data(); res[s].n_cand = ncand[s].data(); res[s].flags = 0;
        if (mcq_query_pipelined(edb, ws, &in[s], &qo, &res[s], &ticket[s])) { std::fprintf(stderr, "FAIL: %s\n", mcq_last_error()); return 1; }
    }
    if (nb >= 2 && !finish(nb - 2)) return 1;
    if (nb >= 1 && !finish(nb - 1)) return 1;
    write_summary(os, o, p, assigned, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    mcq_ws_destroy(ws); mcq_db_destroy(edb); mcq_refdb_close(rdb);
    return 0;
}
