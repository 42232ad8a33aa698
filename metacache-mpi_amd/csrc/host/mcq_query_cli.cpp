// mcq_query_cli -- `metacache query <db> r1.fq r2.fq -pairfiles ...` on one GPU: see mcq_cli_common.hpp for what is
// written and which reference code each part stands in for.
#include "mcq_cli_common.hpp"

int main(int argc, char** argv) {
    Options p;
    if (!parse_options(argc, argv, p)) return 2;
    mcq_refdb* rdb = nullptr; std::vector<uint32_t> t2t; mcq_db_desc d; uint32_t hitmin = 0;
    if (!open_database(p, &rdb, t2t, d, hitmin, 1, 0, 0)) return 1;
    mcq_db* edb = nullptr;
    if (mcq_db_create(&d, &edb)) { std::fprintf(stderr, "ABORT: %s\n", mcq_last_error()); return 1; }

    const auto t_start = std::chrono::steady_clock::now();                  // the reference times map_queries_to_targets, readers included (src/mode_query.cpp:130-132)
    std::vector<Rec> r1, r2;
    if (!read_records(p.f1, r1)) { std::fprintf(stderr, "FAIL: can't open file %s\n", p.f1.c_str()); return 1; }
    const bool paired = p.paired();
    if (paired && !read_records(p.f2, r2)) { std::fprintf(stderr, "FAIL: can't open file %s\n", p.f2.c_str()); return 1; }
    const size_t nq = paired ? std::min(r1.size(), r2.size()) : r1.size();

    std::string bases; std::vector<uint64_t> off{0};
    for (size_t q = 0; q < nq; ++q) {
        bases += r1[q].seq; off.push_back(bases.size());
        if (paired) { bases += r2[q].seq; off.push_back(bases.size()); }
    }
    mcq_ws* ws = nullptr;
    if (mcq_ws_create(edb, nq, bases.size() + 1, 0, &ws)) { std::fprintf(stderr, "ABORT: %s\n", mcq_last_error()); return 1; }
    mcq_batch in; std::memset(&in, 0, sizeof(in)); in.n_seqs = off.size() - 1; in.bases = bases.data(); in.seq_off = off.data(); in.paired = paired ? 1 : 0;
    mcq_query_opts qo; qo.max_cand = p.maxcand; qo.emulate_ranks = p.P; qo.insert_size_max = p.insertsize;
    qo.flags = p.quirks ? MCQ_QUIRK_SEQ_DROP : 0;
    std::vector<mcq_cand> cands(std::max<size_t>(1, nq) * p.maxcand);
    std::vector<uint32_t> ncand(std::max<size_t>(1, nq));
    mcq_result res; res.cands = cands.data(); res.n_cand = ncand.data(); res.flags = 0;
    if (mcq_query(edb, ws, &in, &qo, &res, nullptr)) { std::fprintf(stderr, "FAIL: %s\n", mcq_last_error()); return 1; }

    std::ofstream fout; if (!p.outfile.empty()) fout.open(p.outfile);
    std::ostream& os = p.outfile.empty() ? std::cout : fout;
    const Out o = make_out(rdb, p);
    write_head(os, o, p, hitmin);
    uint64_t assigned[MCQ_RANK_NONE + 1] = {0};
    for (size_t q = 0; q < nq; ++q) write_query(os, o, p, hitmin, r1[q].header, &cands[q * p.maxcand], ncand[q], assigned);
    write_summary(os, o, p, assigned, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    mcq_ws_destroy(ws); mcq_db_destroy(edb); mcq_refdb_close(rdb);
    return 0;
}
