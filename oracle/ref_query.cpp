// oracle/ref_query.cpp -- TEST INFRASTRUCTURE, build container only.
// Header-level driver around the REFERENCE's own query-path code (compiled in
// place from /root/reference/src by oracle/Makefile into oracle/_ref/ref_query;
// no libmpi at link time: nothing MPI is instantiated on the read/query side).
// It dumps the per-rank intermediates the reference CLI never prints, for the
// golden fixtures of rows 7-10 of SURVEY.md section 8a.  It contains no reference
// source; it only calls the reference's public functions:
//   database::read                         src/sketch_database.h:858
//   database::accumulate_matches           src/sketch_database.h:826
//   merge_sort                             src/querying.h:88
//   distinct_matches_in_contiguous_window_ranges / classification_candidates
//                                          src/candidates.h:296 / :189
//
// usage: ref_query <dbprefix> <P> <queries.txt> <maxcand> <lowest-rank> <insert-size-max>
//   queries.txt: one query per line "seq1 seq2" (seq2 "-" = single read)
// output per query q and rank r (db file <dbprefix>.db_<r>):
//   M q r n  tgt:win ...                      sorted match list (row 8)
//   T q r n  tgt:hits:beg:end ...             one candidate per target (row 9)
//   C q r n  taxid:hits:beg:end ...           bounded top list (row 10)
// and once per database:
//   P k s winlen winstride qwinlen qwinstride ntargets
//   L tgt id_rank0 ... id_rank20              ranked lineage of target's taxon (0 = none)
//   N taxid rank name                          every taxon
#include <cstdint>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "timer.h"
#include "config.h"
#include "sketch_database.h"
#include "querying.h"
#include "candidates.h"

using namespace mc;

int main(int argc, char** argv) {
    if (argc < 7) { std::cerr << "usage\n"; return 2; }
    std::string prefix = argv[1];
    int P = std::stoi(argv[2]);
    std::string qfile = argv[3];
    std::size_t maxcand = std::stoul(argv[4]);
    taxon_rank lowest = taxonomy::rank_from_name(argv[5]);
    std::size_t insmax = std::stoul(argv[6]);

    std::vector<std::pair<std::string, std::string>> queries;
    {
        std::ifstream is(qfile);
        std::string a, b;
        while (is >> a >> b) {
            if (a == "-") a.clear();
            if (b == "-") b.clear();
            queries.emplace_back(a, b);
        }
    }

    for (int r = 0; r < P; ++r) {
        database db;
        db.read(prefix + ".db_" + std::to_string(r));
        if (r == 0) {
            std::cout << "P " << int(db.target_sketcher().kmer_size()) << ' '
                      << db.target_sketcher().sketch_size() << ' '
                      << db.target_window_size() << ' ' << db.target_window_stride() << ' '
                      << db.query_window_size() << ' ' << db.query_window_stride() << ' '
                      << db.target_count() << '\n';
            for (std::uint64_t t = 0; t < db.target_count(); ++t) {
                const taxon* tax = db.taxon_of_target(target_id(t));
                std::cout << "L " << t;
                for (const taxon* a : db.ranks(tax)) std::cout << ' ' << (a ? a->id() : 0);
                std::cout << '\n';
            }
            for (const auto& t : db.taxa()) {
                std::cout << "N " << t.id() << ' ' << int(t.rank()) << ' ' << t.name() << '\n';
            }
        }
        for (std::size_t q = 0; q < queries.size(); ++q) {
            database::match_target_locations m, buf;
            std::vector<std::size_t> offsets{0};
            db.accumulate_matches(queries[q].first, m, offsets);
            db.accumulate_matches(queries[q].second, m, offsets);
            merge_sort(m, offsets, buf);

            std::cout << "M " << q << ' ' << r << ' ' << m.size();
            for (const auto& x : m) std::cout << ' ' << x.tgt << ':' << x.win;
            std::cout << '\n';

            match_locations ml;
            for (const auto& x : m) ml.emplace_back(db.taxon_of_target(x.tgt), x.win);

            candidate_generation_rules rules;
            rules.maxWindowsInRange = window_id(2 + (
                std::max(queries[q].first.size() + queries[q].second.size(), insmax) /
                db.target_window_stride()));

            distinct_matches_in_contiguous_window_ranges all{db, ml, rules};
            std::cout << "T " << q << ' ' << r << ' ' << all.size();
            for (const auto& c : all)
                std::cout << ' ' << (-(c.tax->id()) - 1) << ':' << c.hits << ':' << c.pos.beg << ':' << c.pos.end;
            std::cout << '\n';

            rules.mergeBelow = lowest;
            rules.maxCandidates = maxcand;
            classification_candidates top{db, ml, rules};
            std::cout << "C " << q << ' ' << r << ' ' << top.size();
            for (const auto& c : top)
                std::cout << ' ' << c.tax->id() << ':' << c.hits << ':' << c.pos.beg << ':' << c.pos.end;
            std::cout << '\n';
        }
    }
    return 0;
}
