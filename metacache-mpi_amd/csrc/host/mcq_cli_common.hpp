#pragma once
// mcq_cli_common.hpp -- what mcq_query_cli (one GPU) and mcq_query_mpi (one process per GPU under mpiexec) share: the
// readers, the options and the writers of the reference's -out file.
//
// mcq_query_cli -- stand-in for `mpiexec -n P metacache query <db> r1.fq r2.fq -pairfiles ...`
// (src/mode_query.cpp:404-458) around the engine: reads the reference's shard files, runs the per-read path on
// the GPU through the C ABI, classifies on the host and writes what the reference writes to its -out file:
//   * the parameter lines                      show_query_parameters    src/printing.cpp:40-113
//   * "# TABLE_LAYOUT: ..."                    show_query_mapping_header src/classification.cpp:486-512, printing.cpp:243-300
//   * "# <file1> + <file2>"                    src/querying.h:1337
//   * one mapping line per read (pair)         show_query_mapping       src/classification.cpp:583-632
//         taxon formats (rank:name default, -taxids, -taxids-only, -omit-ranks, -lineage)  show_taxon / show_lineage /
//         show_no_taxon src/printing.cpp:117-201, :305-330;  -tophits list  show_matches src/printing.cpp:333-360
//   * the summary                              show_summary             src/printing.cpp:622-641,
//                                              show_taxon_statistics    src/printing.cpp:522-555
// After sorting, the file equals the reference's byte for byte except for the measured values of the "# time:" and
// "# speed:" lines (tests/test_gpu_cli.py).  Not reproduced: the reference prints nothing for a thread's chunk in which
// no read was classified (src/querying.h:1091, :1129).
//
// usage: mcq_query_cli <dbprefix> <n_ranks> <r1.fq> <r2.fq|-> [-lowest R] [-highest R] [-maxcand N] [-hitmin N]
//            [-hitdiff X] [-insertsize N] [-threads N] [-tophits] [-taxids] [-taxids-only] [-omit-ranks] [-lineage]
//            [-mapped-only] [-nomap] [-noquirks] [-out FILE] [-batch N] [-batch-bases N]
// (-batch / -batch-bases: queries / bases per batch; the reads go through in batches, see mcq_query_cli.cpp)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/mcq.h"
#include "../../../include/mcq_host.h"
#include "../../../include/mcq_open.hpp"

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

// How a taxon is written (the reference's taxon_print_mode, src/query_options.h:68-71; output of src/printing.cpp:117-176,
// :243-300) as two independent choices: an optional "<rank>:" prefix, and one of three bodies -- name, id, name(id).
struct Mode {
    bool rank_prefix; int body;                          // body: 0 = name, 1 = id, 2 = name(id)
    static constexpr Mode make(bool show_ranks, bool taxids, bool taxids_only) { return Mode{show_ranks, taxids_only ? 1 : (taxids ? 2 : 0)}; }
    bool ids_only() const { return body == 1; }
};

struct Out {
    mcq_refdb* db;
    Mode mode = Mode{true, 0};
    uint32_t lowest = MCQ_RANK_SEQUENCE, highest = MCQ_RANK_DOMAIN;
    bool lineage = false, tophits = false;
    const char* comment = "# "; const char* none = "--"; const char* col = "\t|\t";

    // one taxon column entry: [prefix ':'] body, the body built from a name text and an id text
    template <class Name, class Id>
    void entry(std::ostream& os, const char* prefix, const Name& name, const Id& id) const {
        if (mode.rank_prefix) os << prefix << ':';
        if (mode.body != 1) os << name;
        if (mode.body == 2) os << '(';
        if (mode.body != 0) os << id;
        if (mode.body == 2) os << ')';
    }
    void taxon(std::ostream& os, uint32_t key) const {
        entry(os, mcq_rank_name(mcq_refdb_taxon_rank(db, key)), mcq_refdb_taxon_name(db, key), mcq_refdb_taxon_id(db, key));
    }
    void no_taxon(std::ostream& os, uint32_t rank) const { entry(os, mcq_rank_name(rank), none, 0); }
    // classification column: show_taxon(os, db, opt, tax), src/printing.cpp:305-330 (collapseUnclassified is on)
    void best(std::ostream& os, uint32_t key) const {
        if (key == MCQ_NO_TAXON || mcq_refdb_taxon_rank(db, key) > highest) {
            if (mode.ids_only() && !mode.rank_prefix) os << 0; else os << none;
            return;
        }
        const uint32_t tr = mcq_refdb_taxon_rank(db, key);
        const uint32_t rmin = lowest < tr ? tr : lowest, rmax = lineage ? highest : rmin;
        for (uint32_t r = rmin; r <= rmax; ++r) {                           // show_lineage, src/printing.cpp:181-201
            const uint32_t a = mcq_refdb_ancestor(db, key, r);
            if (a != MCQ_NO_TAXON) taxon(os, a); else no_taxon(os, r);
            if (r < rmax) os << ',';
        }
    }
    void header_taxon(std::ostream& os) const {          // the TABLE_LAYOUT line's taxon column: one entry per rank shown
        const uint32_t rmax = lineage ? highest : lowest;
        for (uint32_t r = lowest; r <= rmax; ++r) {
            entry(os, lowest == rmax ? "rank" : mcq_rank_name(r), "taxname", "taxid");
            if (r < rmax) os << ',';
        }
    }
};


struct Options {
    std::string prefix, f1, f2, outfile;
    uint32_t P = 1;                      // ranks of the reference build / run whose results are reproduced
    uint32_t lowest = MCQ_RANK_SEQUENCE, highest = MCQ_RANK_DOMAIN, maxcand = 2, hitmin = 0, threads = 1;
    float hitdiff = 1.0f; uint64_t insertsize = 0; bool quirks = true;
    bool show_ranks = true, taxids = false, taxids_only = false, lineage = false, tophits = false, mapped_only = false, nomap = false;
    std::string transport = "rccl";      // mcq_query_mpi: rccl | mpi (blocks through the host and MPI_Alltoallv)
    uint64_t batch = 1u << 19, batch_bases = 256u << 20;   // mcq_query_mpi: queries / bases per rank and batch
    bool paired() const { return f2 != "-"; }
};

static bool parse_options(int argc, char** argv, Options& o) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s <dbprefix> <n_ranks> <r1> <r2|-> [options]\n", argv[0]); return false; }
    o.prefix = argv[1]; o.P = (uint32_t)std::atoi(argv[2]); o.f1 = argv[3]; o.f2 = argv[4];
    for (int i = 5; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "-lowest") { uint32_t r = mcq_rank_from_name(next()); if (r < MCQ_RANK_ROOT) o.lowest = r; }
        else if (a == "-highest") { uint32_t r = mcq_rank_from_name(next()); if (r <= MCQ_RANK_ROOT) o.highest = r; }
        else if (a == "-maxcand" || a == "-max-cand") o.maxcand = (uint32_t)std::atoi(next());
        else if (a == "-hitmin") o.hitmin = (uint32_t)std::atoi(next());
        else if (a == "-hitdiff") { o.hitdiff = (float)std::atof(next()); if (o.hitdiff > 1) o.hitdiff *= 0.01; }   // src/query_options.cpp:167-169
        else if (a == "-insertsize") o.insertsize = std::strtoull(next(), nullptr, 10);
        else if (a == "-threads") o.threads = (uint32_t)std::atoi(next());
        else if (a == "-tophits" || a == "-top-hits") o.tophits = true;
        else if (a == "-taxids" || a == "-taxid") o.taxids = true;
        else if (a == "-taxids-only" || a == "-taxidsonly" || a == "-taxid-only") o.taxids_only = true;
        else if (a == "-omit-ranks" || a == "-omitranks") o.show_ranks = false;
        else if (a == "-lineage") o.lineage = true;
        else if (a == "-mapped-only" || a == "-mappedonly") o.mapped_only = true;
        else if (a == "-nomap" || a == "-no-map") o.nomap = true;
        else if (a == "-noquirks") o.quirks = false;
        else if (a == "-transport") o.transport = next();
        else if (a == "-batch") o.batch = std::max<uint64_t>(1, std::strtoull(next(), nullptr, 10));
        else if (a == "-batch-bases") o.batch_bases = std::max<uint64_t>(1024, std::strtoull(next(), nullptr, 10));
        else if (a == "-out") o.outfile = next();
    }
    if (o.lowest > o.highest) o.lowest = o.highest;
    if (o.nomap && o.tophits) { o.nomap = false; o.mapped_only = true; }   // "showing hits changes the mapping mode", src/query_options.cpp:289-292
    return true;
}

static Out make_out(mcq_refdb* rdb, const Options& p) {
    Out o; o.db = rdb; o.lowest = p.lowest; o.highest = p.highest; o.lineage = p.lineage; o.tophits = p.tophits;
    o.mode = Mode::make(p.show_ranks, p.taxids, p.taxids_only);          // -taxids-only wins over -taxids (src/query_options.cpp:262-274)
    return o;
}

// show_query_parameters (src/printing.cpp:40-113) + show_query_mapping_header (src/classification.cpp:486-512) + the file line
static void write_head(std::ostream& os, const Out& o, const Options& p, uint32_t hitmin) {
    const char* cm = o.comment;
    if (!p.nomap) {
        os << cm << "Reporting per-read mappings (non-mapping lines start with '" << cm << "').\n";
        if (p.lineage) os << cm << "The complete lineage will be reported starting with the lowest match.\n";
        else os << cm << "Only the lowest matching rank will be reported.\n";
    } else os << cm << "Per-Read mappings will not be shown.\n";
    os << cm << "Classification will be constrained to ranks from '" << mcq_rank_name(p.lowest) << "' to '" << mcq_rank_name(p.highest) << "'.\n";
    os << cm << "Classification hit threshold is " << hitmin << " per query\n";
    os << cm << "At maximum " << p.maxcand << " classification candidates will be considered per query.\n";
    if (p.paired()) os << cm << "File based paired-end mode:\n" << cm << "  Reads from two consecutive files will be interleaved.\n"
                       << cm << "  Max insert size considered " << p.insertsize << ".\n";
    os << cm << "Using " << p.threads << " threads\n";
    if (!p.nomap) {
        os << cm << "TABLE_LAYOUT: query_header" << o.col;
        if (p.tophits) os << "top_hits" << o.col;
        o.header_taxon(os);
        os << '\n';
    }
    os << cm << (p.paired() ? p.f1 + " + " + p.f2 : p.f1) << '\n';          // src/querying.h:1337
}

// one query: classification (src/classification.cpp:235-265), statistics (classification_statistics::assign,
// src/classification_statistics.h:69-78) and its mapping line (show_query_mapping, src/classification.cpp:583-632)
static void write_query(std::ostream& os, const Out& o, const Options& p, uint32_t hitmin, const std::string& header,
                        const mcq_cand* cands, uint32_t ncand, uint64_t* assigned /* [MCQ_RANK_NONE + 1] */) {
    mcq_refdb* rdb = o.db;
    const uint32_t best = mcq_refdb_classify(rdb, reinterpret_cast<const uint32_t*>(cands), ncand, hitmin, p.hitdiff, p.highest);
    if (best == MCQ_NO_TAXON) ++assigned[MCQ_RANK_NONE];
    else for (uint32_t r = mcq_refdb_taxon_rank(rdb, best); r <= MCQ_RANK_ROOT; ++r) ++assigned[r];
    if (p.nomap || (p.mapped_only && best == MCQ_NO_TAXON)) return;
    os << header.substr(0, header.find(' ')) << o.col;
    if (p.tophits) {                                                         // show_matches, src/printing.cpp:333-360
        for (uint32_t i = 0; i < ncand && cands[i].hits > 0; ++i) {
            const mcq_cand& c = cands[i];
            if (i) os << ',';
            const uint32_t key = c.tax;
            if (p.lowest == MCQ_RANK_SEQUENCE) os << mcq_refdb_taxon_name(rdb, key);
            else {
                const uint32_t a = mcq_refdb_taxon_rank(rdb, key) < p.lowest ? mcq_refdb_ancestor(rdb, key, p.lowest) : key;
                if (a != MCQ_NO_TAXON) os << mcq_refdb_taxon_id(rdb, a); else os << mcq_refdb_taxon_name(rdb, key);
            }
            os << ':' << c.hits;
        }
        os << o.col;
    }
    o.best(os, best);
    os << '\n';
}

// show_summary (src/printing.cpp:622-641) + show_taxon_statistics (:522-555)
static void write_summary(std::ostream& os, const Out& o, const Options& p, const uint64_t* assigned, double ms) {
    const char* cm = o.comment;
    const uint64_t total = assigned[MCQ_RANK_ROOT] + assigned[MCQ_RANK_NONE];
    const uint64_t num_queries = p.paired() ? 2 * total : total;             // paired reads count twice (:626-627)
    os << cm << "queries: " << num_queries << '\n'
       << cm << "time:    " << (long long)ms << " ms\n"
       << cm << "speed:   " << num_queries / (ms / 60000.0) << " queries/min\n";
    if (total > 0) {
        if (assigned[MCQ_RANK_ROOT] < 1) os << "None of the input sequences could be classified.\n";
        else {
            if (assigned[MCQ_RANK_NONE] > 0)
                os << cm << "unclassified: " << (100 * (assigned[MCQ_RANK_NONE] / double(total))) << "% (" << assigned[MCQ_RANK_NONE] << ")\n";
            os << cm << "classified:\n";
            static const uint32_t ranks[] = {0 /*sequence*/, 3 /*subspecies*/, 4 /*species*/, 6 /*genus*/, 10 /*family*/, 12 /*order*/,
                                             14 /*class*/, 16 /*phylum*/, 18 /*kingdom*/, 19 /*domain*/, 20 /*root*/};
            for (uint32_t r : ranks) {
                if (assigned[r] == 0) continue;
                std::string rn = mcq_rank_name(r);
                rn.resize(11, ' ');
                os << cm << "  " << rn << (100 * (assigned[r] / double(total))) << "% (" << assigned[r] << ")\n";
            }
        }
    } else std::cerr << cm << "No valid query sequences found.\n";
}

// The reference's shard files -> the queryable handle of shard `shard_id` of `n_shards` (include/mcq_open.hpp: the host-side union
// for small databases, the streaming route -- heads on the host, tables merged on the GPU -- from MCQ_STREAM_LOAD_MIN_MB, default
// 1024, MB of shard files on)
static bool open_database(const Options& p, mcq_refdb** rdb, std::vector<uint32_t>& t2t, mcq_db** edb, uint32_t& hitmin,
                          uint32_t n_shards, uint32_t shard_id, int device) {
    std::string err; bool streamed = false;
    if (mcq_open_refdb(p.prefix, p.P, mcq_stream_load_min_bytes(), rdb, &streamed, err)) { std::fprintf(stderr, "ABORT: %s\n", err.c_str()); return false; }
    mcq_refdb_info info; mcq_refdb_get_info(*rdb, &info);
    hitmin = p.hitmin < 1 ? mcq_default_hits_min(info.sketch_size) : p.hitmin;
    t2t.resize(info.n_targets);
    if (mcq_refdb_tgt2tax(*rdb, p.lowest, t2t.data())) { std::fprintf(stderr, "ABORT: %s\n", mcq_host_last_error()); return false; }
    if (mcq_make_db(*rdb, streamed, t2t.data(), n_shards, shard_id, device, edb, err)) { std::fprintf(stderr, "ABORT: %s\n", err.c_str()); return false; }
    return true;
}
