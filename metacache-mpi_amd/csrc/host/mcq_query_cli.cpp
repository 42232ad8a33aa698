// mcq_query_cli -- minimal stand-in for `metacache query <db> r1.fq r2.fq -pairfiles ...`
// (src/mode_query.cpp:404-458) around the engine: reads the reference's shard files, runs the
// per-read path on the GPU through the C ABI, classifies on the host and prints mapping lines in
// the reference's `-tophits -taxids-only -omit-ranks` layout (src/classification.cpp:583-632,
// src/printing.cpp:333-360):   <header> \t|\t <taxid:hits,...> \t|\t <taxid or 0>
//
// usage: mcq_query_cli <dbprefix> <n_ranks> <r1.fq> <r2.fq|-> [-lowest R] [-highest R] [-maxcand N]
//                      [-hitmin N] [-hitdiff X] [-insertsize N] [-noquirks] [-out FILE]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/mcq.h"
#include "../../../include/mcq_host.h"

struct Rec { std::string header, seq; };

// FASTA ('>') and FASTQ ('@') records; sequence may span lines in FASTA (src/sequence_io.cpp:122-285)
static bool read_records(const std::string& path, std::vector<Rec>& out) {
    std::ifstream is(path);
    if (!is.good()) return false;
    std::string line;
    while (std::getline(is, line)) {
        if (line.empty()) continue;
        if (line[0] == '@') {
            Rec r; r.header = line.substr(1);
            std::getline(is, r.seq);
            std::getline(is, line); std::getline(is, line);          // '+' and qualities
            out.push_back(std::move(r));
        } else if (line[0] == '>') {
            Rec r; r.header = line.substr(1);
            out.push_back(std::move(r));
        } else if (!out.empty()) {
            out.back().seq += line;
        }
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s <dbprefix> <n_ranks> <r1> <r2|-> [options]\n", argv[0]); return 2; }
    const std::string prefix = argv[1];
    const uint32_t P = (uint32_t)std::atoi(argv[2]);
    const std::string f1 = argv[3], f2 = argv[4];
    uint32_t lowest = MCQ_RANK_SEQUENCE, highest = MCQ_RANK_DOMAIN, maxcand = 2, hitmin = 0;
    float hitdiff = 1.0f; uint64_t insertsize = 0; bool quirks = true; std::string outfile;
    for (int i = 5; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "-lowest") { uint32_t r = mcq_rank_from_name(next()); if (r < MCQ_RANK_ROOT) lowest = r; }
        else if (a == "-highest") { uint32_t r = mcq_rank_from_name(next()); if (r <= MCQ_RANK_ROOT) highest = r; }
        else if (a == "-maxcand" || a == "-max-cand") maxcand = (uint32_t)std::atoi(next());
        else if (a == "-hitmin") hitmin = (uint32_t)std::atoi(next());
        else if (a == "-hitdiff") { hitdiff = (float)std::atof(next()); if (hitdiff > 1) hitdiff *= 0.01; }   // src/query_options.cpp:167-169
        else if (a == "-insertsize") insertsize = std::strtoull(next(), nullptr, 10);
        else if (a == "-noquirks") quirks = false;
        else if (a == "-out") outfile = next();
    }
    if (lowest > highest) lowest = highest;

    mcq_refdb* rdb = nullptr;
    if (mcq_refdb_open(prefix.c_str(), P, &rdb)) { std::fprintf(stderr, "ABORT: %s\n", mcq_host_last_error()); return 1; }
    mcq_refdb_info info; mcq_refdb_get_info(rdb, &info);
    if (hitmin < 1) hitmin = mcq_default_hits_min(info.sketch_size);
    std::vector<uint32_t> t2t(info.n_targets);
    if (mcq_refdb_tgt2tax(rdb, lowest, t2t.data())) { std::fprintf(stderr, "ABORT: %s\n", mcq_host_last_error()); return 1; }

    mcq_db_desc d; std::memset(&d, 0, sizeof(d));
    d.k = info.k; d.sketch_size = info.q_sketch_size; d.winlen = info.q_winlen; d.winstride = info.q_winstride;
    d.tgt_winstride = info.winstride; d.n_targets = info.n_targets; d.n_keys = info.n_keys; d.n_locs = info.n_locs;
    d.keys = mcq_refdb_keys(rdb); d.list_off = mcq_refdb_list_off(rdb); d.locs = mcq_refdb_locs(rdb); d.tgt2tax = t2t.data();
    d.n_shards = 1; d.shard_id = 0; d.flags = 0; d.device = 0;
    static const uint64_t zero_off[1] = {0};
    if (!d.list_off) d.list_off = zero_off;
    mcq_db* edb = nullptr;
    if (mcq_db_create(&d, &edb)) { std::fprintf(stderr, "ABORT: %s\n", mcq_last_error()); return 1; }

    std::vector<Rec> r1, r2;
    if (!read_records(f1, r1)) { std::fprintf(stderr, "FAIL: can't open file %s\n", f1.c_str()); return 1; }
    const bool paired = f2 != "-";
    if (paired && !read_records(f2, r2)) { std::fprintf(stderr, "FAIL: can't open file %s\n", f2.c_str()); return 1; }
    const size_t nq = paired ? std::min(r1.size(), r2.size()) : r1.size();

    std::string bases; std::vector<uint64_t> off{0};
    for (size_t q = 0; q < nq; ++q) {
        bases += r1[q].seq; off.push_back(bases.size());
        if (paired) { bases += r2[q].seq; off.push_back(bases.size()); }
    }
    mcq_ws* ws = nullptr;
    if (mcq_ws_create(edb, nq, bases.size() + 1, 0, &ws)) { std::fprintf(stderr, "ABORT: %s\n", mcq_last_error()); return 1; }
    mcq_batch in; in.n_seqs = off.size() - 1; in.bases = bases.data(); in.seq_off = off.data(); in.paired = paired ? 1 : 0; in.flags = 0;
    mcq_query_opts qo; qo.max_cand = maxcand; qo.emulate_ranks = P; qo.insert_size_max = insertsize;
    qo.flags = quirks ? MCQ_QUIRK_SEQ_DROP : 0;
    std::vector<mcq_cand> cands(std::max<size_t>(1, nq) * maxcand);
    std::vector<uint32_t> ncand(std::max<size_t>(1, nq));
    mcq_result res; res.cands = cands.data(); res.n_cand = ncand.data(); res.flags = 0;
    if (mcq_query(edb, ws, &in, &qo, &res, nullptr)) { std::fprintf(stderr, "FAIL: %s\n", mcq_last_error()); return 1; }

    std::ofstream fout; if (!outfile.empty()) fout.open(outfile);
    std::ostream& os = outfile.empty() ? std::cout : fout;
    size_t classified = 0;
    for (size_t q = 0; q < nq; ++q) {
        const std::string& h = r1[q].header;
        os << h.substr(0, h.find(' ')) << "\t|\t";
        for (uint32_t i = 0; i < ncand[q]; ++i) {
            const mcq_cand& c = cands[q * maxcand + i];
            if (i) os << ',';
            // show_matches: taxa below `lowest` are shown by their ancestor there, else by name
            uint32_t key = c.tax;
            if (lowest > MCQ_RANK_SEQUENCE && mcq_refdb_taxon_rank(rdb, key) < lowest) {
                uint32_t a = mcq_refdb_ancestor(rdb, key, lowest);
                if (a != MCQ_NO_TAXON) os << mcq_refdb_taxon_id(rdb, a); else os << mcq_refdb_taxon_name(rdb, key);
            } else os << mcq_refdb_taxon_id(rdb, key);
            os << ':' << c.hits;
        }
        const uint32_t best = mcq_refdb_classify(rdb, reinterpret_cast<const uint32_t*>(&cands[q * maxcand]), ncand[q], hitmin, hitdiff, highest);
        os << "\t|\t" << mcq_refdb_taxon_id(rdb, best) << '\n';
        classified += best != MCQ_NO_TAXON;
    }
    std::fprintf(stderr, "# queries: %zu  classified: %zu\n", paired ? 2 * nq : nq, classified);
    mcq_ws_destroy(ws); mcq_db_destroy(edb); mcq_refdb_close(rdb);
    return 0;
}
