/* binding_selftest.cpp -- drives mcq_reference_binding against the REFERENCE'S OWN CODE on the CPU (build
 * container only; the GPU calls of the binding are compiled and linked but not executed here).
 *
 *   binding_selftest table <db prefix> <n ranks>
 *        the reference reads every shard (database::read) and dumps its table through its public
 *        print_feature_map; the binding's file route (libmcq_host, what gpu_engine::open uses), the text dump
 *        and a live mc::hash_multimap filled from it and flattened by flatten_feature_store must agree.
 *   binding_selftest keys <db prefix> <lowest rank>
 *        prints the taxon keys make_taxon_keys assigns: "<target> <key> <taxon id>" per target.
 *   binding_selftest map <db prefix> <lowest rank> <max cand> <cands.txt>
 *        cands.txt: per query "<header>\t<n>\t<key>:<hits>:<beg>:<end> ..." (engine output with the keys above);
 *        converts every list with to_candidates (the reference's insert()) and hands them to the reference's own
 *        map_candidates_to_targets (src/classification.cpp:889-925: classify + show_query_mapping), with the
 *        options of the fixture runs; the mapping lines go to stdout.
 */
#include <fstream>
#include <iostream>
#include <sstream>

#include "mcq_reference_binding.h"
#include "args_parser.h"
#include "classification.h"
#include "hash_multimap.h"

using namespace mc;

/* The seam itself, type-checked against the reference's own container: query_batched_parallel2 keeps a block's results in
 * `tsl::hopscotch_map<uint_least64_t, classification_candidates> allhits` (src/querying.h:733) and fills it from the tree
 * merge (:1112-1117); gpu_engine::query_block fills the same map instead.  An explicit instantiation: every dependent line
 * of the template is compiled against the reference's headers (the GPU call behind it is linked, not run, in the build
 * container; `binding_selftest gpu ...` runs it where there is a GPU). */
#include <tsl/hopscotch_map.h>
using reference_result_map = tsl::hopscotch_map<std::uint_least64_t, mc::classification_candidates>;
template void mcq_binding::gpu_engine::query_block<reference_result_map>(
    const std::vector<mc::sequence_pair_reader::sequence_pair>&, bool, std::uint32_t, reference_result_map&);

static database read_db(const std::string& file, database::scope what)
{
    database db;
    db.read(file, what);
    return db;
}

struct loc_t { std::uint32_t tgt, win; };

static mcq_binding::flat_table table_from_dump(const database& db)
{
    std::ostringstream os;
    db.print_feature_map(os);                       /* "key -> (tgt,win)(tgt,win)...\n" per non-empty bucket */
    std::istringstream is(os.str());
    mcq_binding::flat_table t;
    std::string line;
    std::map<std::uint32_t, std::vector<std::uint64_t>> sorted;
    while (std::getline(is, line)) {
        const auto arrow = line.find(" -> ");
        if (arrow == std::string::npos) continue;
        const std::uint32_t key = static_cast<std::uint32_t>(std::stoll(line.substr(0, arrow)));
        auto& l = sorted[key];
        std::size_t p = arrow + 4;
        while (p < line.size() && line[p] == '(') {
            const auto comma = line.find(',', p), close = line.find(')', p);
            l.push_back((static_cast<std::uint64_t>(std::stoull(line.substr(p + 1, comma - p - 1))) << 32) |
                        std::stoull(line.substr(comma + 1, close - comma - 1)));
            p = close + 1;
        }
    }
    for (auto& kv : sorted) {
        t.keys.push_back(kv.first);
        t.locs.insert(t.locs.end(), kv.second.begin(), kv.second.end());
        t.list_off.push_back(t.locs.size());
    }
    return t;
}

static bool same(const mcq_binding::flat_table& a, const mcq_binding::flat_table& b)
{
    return a.keys == b.keys && a.list_off == b.list_off && a.locs == b.locs;
}

static int cmd_table(const std::string& prefix, unsigned P)
{
    std::vector<mcq_binding::flat_table> per_rank, per_rank_live;
    for (unsigned r = 0; r < P; ++r) {
        database db = read_db(prefix + ".db_" + std::to_string(r), database::scope::everything);
        per_rank.push_back(table_from_dump(db));
        /* a live hash_multimap with the same content, flattened through its bucket interface */
        hash_multimap<std::uint32_t, loc_t, feature_hash> live;       /* the database's own h2 (src/config.h:104) */
        const auto& t = per_rank.back();
        live.max_load_factor(0.8f);                 /* src/sketch_database.h:267 */
        live.reserve_keys(t.keys.size());           /* as hash_multimap::deserialize does (src/hash_multimap.h:940-941): its probing
                                                       visits few slots, and an insert that finds none free is dropped silently */
        for (std::size_t i = 0; i < t.keys.size(); ++i)
            for (std::uint64_t j = t.list_off[i]; j < t.list_off[i + 1]; ++j)
                live.insert(t.keys[i], loc_t{static_cast<std::uint32_t>(t.locs[j] >> 32), static_cast<std::uint32_t>(t.locs[j])});
        auto fl = mcq_binding::flatten_feature_store(live);
        /* bucket order of a hash table is arbitrary: bring both to key order before comparing */
        per_rank_live.push_back(mcq_binding::union_tables({fl}));
        { const auto a = per_rank_live.back(), b = mcq_binding::union_tables({t});
          if (!same(a, b)) { std::cerr << "live hash_multimap flatten differs on rank " << r << ": keys " << a.keys.size() << " / " << b.keys.size()
                                       << " locs " << a.locs.size() << " / " << b.locs.size() << " live.key_count " << live.key_count() << " values " << live.value_count() << "\n"; return 1; } }
    }
    const auto u = mcq_binding::union_tables(per_rank);
    mcq_refdb* rdb = nullptr;
    if (mcq_refdb_open(prefix.c_str(), P, &rdb) != 0) { std::cerr << mcq_host_last_error() << "\n"; return 1; }
    mcq_refdb_info info; mcq_refdb_get_info(rdb, &info);
    mcq_binding::flat_table f;
    f.keys.assign(mcq_refdb_keys(rdb), mcq_refdb_keys(rdb) + info.n_keys);
    f.list_off.assign(mcq_refdb_list_off(rdb), mcq_refdb_list_off(rdb) + info.n_keys + 1);
    f.locs.assign(mcq_refdb_locs(rdb), mcq_refdb_locs(rdb) + info.n_locs);
    mcq_refdb_close(rdb);
    if (!same(u, f)) { std::cerr << "file route and reference dump differ\n"; return 1; }
    std::uint64_t sum = 1469598103934665603ull;
    for (auto k : u.keys) sum = (sum ^ k) * 1099511628211ull;
    for (auto l : u.locs) sum = (sum ^ l) * 1099511628211ull;
    std::cout << "table ok keys " << u.keys.size() << " locs " << u.locs.size() << " fnv " << sum << "\n";
    return 0;
}

static int cmd_keys(const std::string& prefix, const std::string& lowest)
{
    database db = read_db(prefix + ".db_0", database::scope::metadata_only);
    const auto keys = mcq_binding::make_taxon_keys(db, taxonomy::rank_from_name(lowest));
    for (std::size_t t = 0; t < keys.tgt2tax.size(); ++t) {
        const taxon* tax = keys(keys.tgt2tax[t]);
        std::cout << t << ' ' << keys.tgt2tax[t] << ' ' << (tax ? tax->id() : 0) << '\n';
    }
    return 0;
}

static int cmd_map(const std::string& prefix, const std::string& lowest, const std::string& maxcand, const std::string& file)
{
    database db = read_db(prefix + ".db_0", database::scope::metadata_only);
    /* the options of the fixture runs (tests/golden/make_golden.py: ref_query_cli), parsed by the reference itself */
    args_parser args{std::vector<std::string>{"query", prefix, "-pairfiles", "-lowest", lowest, "-maxcand", maxcand,
                                              "-hitmin", "4", "-hitdiff", "80", "-tophits", "-taxids-only", "-omit-ranks"}};
    query_options opt = get_query_options(args, {});
    const auto keys = mcq_binding::make_taxon_keys(db, opt.classify.lowestRank);
    candidate_generation_rules rules;
    rules.mergeBelow    = opt.classify.lowestRank;
    rules.maxCandidates = opt.classify.maxNumCandidatesPerQuery;

    std::vector<std::string> headers;
    std::vector<classification_candidates> lists;
    std::ifstream is(file);
    std::string line;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        std::string header, tok; unsigned n = 0;
        std::getline(ls, header, '\t');
        ls >> n;
        std::vector<mcq_cand> c;
        while (ls >> tok) {
            mcq_cand x{};
            unsigned long long a, b, cc, d;
            if (std::sscanf(tok.c_str(), "%llu:%llu:%llu:%llu", &a, &b, &cc, &d) != 4) return 2;
            x.tax = static_cast<std::uint32_t>(a); x.hits = static_cast<std::uint32_t>(b);
            x.win_beg = static_cast<std::uint32_t>(cc); x.win_end = static_cast<std::uint32_t>(d);
            c.push_back(x);
        }
        if (c.size() != n) return 2;
        headers.push_back(header);
        lists.push_back(mcq_binding::to_candidates(c.data(), n, keys, db, rules));
        /* the list must come out of insert() exactly as it went in */
        if (lists.back().size() != n) { std::cerr << "insert() changed the size of the list of " << header << "\n"; return 1; }
        for (unsigned i = 0; i < n; ++i)
            if (lists.back()[i].tax != keys(c[i].tax) || lists.back()[i].hits != c[i].hits) { std::cerr << "insert() reordered the list of " << header << "\n"; return 1; }
    }
    std::ostringstream devnull;
    classification_results results{std::cout, devnull, devnull, devnull};
    map_candidates_to_targets(headers, lists, db, opt, results);
    results.flush_all_streams();
    return 0;
}

/* binding_selftest gpu <db prefix> <n ranks> <lowest rank> <max cand> <reads.txt>   (needs a GPU)
 *      reads.txt: "<header>\t<mate 1>\t<mate 2>" per pair; one block through gpu_engine::query_block into the reference's
 *      map type, then the reference's own map_candidates_to_targets: the mapping lines go to stdout. */
static int cmd_gpu(const std::string& prefix, unsigned n_ranks, const std::string& lowest, const std::string& maxcand, const std::string& reads_file)
{
    database db = read_db(prefix + ".db_0", database::scope::metadata_only);
    /* the options of the fixture runs, parsed by the reference itself (as in cmd_map) */
    args_parser args{std::vector<std::string>{"query", prefix, "-pairfiles", "-lowest", lowest, "-maxcand", maxcand,
                                              "-hitmin", "4", "-hitdiff", "80", "-tophits", "-taxids-only", "-omit-ranks"}};
    query_options opt = get_query_options(args, {});
    mcq_binding::gpu_engine eng(db, prefix, n_ranks, opt.classify, 0, 1 << 20, 1ull << 30);
    std::vector<sequence_pair_reader::sequence_pair> block;
    std::vector<std::string> headers;
    std::ifstream is(reads_file);
    std::string line;
    std::uint_least64_t idx = 0;
    while (std::getline(is, line)) {
        const auto t1 = line.find('\t'), t2 = line.find('\t', t1 + 1);
        if (t1 == std::string::npos || t2 == std::string::npos) continue;
        sequence_pair_reader::sequence_pair p;
        p.first.index = p.second.index = idx++;
        p.first.header = line.substr(0, t1); p.first.data = line.substr(t1 + 1, t2 - t1 - 1);
        p.second.header = p.first.header; p.second.data = line.substr(t2 + 1);
        headers.push_back(p.first.header);
        block.push_back(std::move(p));
    }
    reference_result_map allhits;
    eng.query_block(block, true, n_ranks, allhits);
    std::vector<classification_candidates> lists;
    for (std::uint_least64_t q = 0; q < idx; ++q) {
        auto it = allhits.find(q);
        lists.push_back(it == allhits.end() ? classification_candidates{} : it->second);
    }
    std::ostringstream devnull;
    classification_results results{std::cout, devnull, devnull, devnull};
    map_candidates_to_targets(headers, lists, db, opt, results);
    results.flush_all_streams();
    return 0;
}

int main(int argc, char** argv)
{
    try {
        const std::string cmd = argc > 1 ? argv[1] : "";
        if (cmd == "table" && argc == 4) return cmd_table(argv[2], static_cast<unsigned>(std::stoul(argv[3])));
        if (cmd == "keys" && argc == 4) return cmd_keys(argv[2], argv[3]);
        if (cmd == "map" && argc == 6) return cmd_map(argv[2], argv[3], argv[4], argv[5]);
        if (cmd == "gpu" && argc == 7) return cmd_gpu(argv[2], static_cast<unsigned>(std::stoul(argv[3])), argv[4], argv[5], argv[6]);
        std::cerr << "usage: binding_selftest table|keys|map ...\n";
        return 2;
    } catch (std::exception& e) {
        std::cerr << "ERROR: " << e.what() << "\n";
        return 1;
    }
}
