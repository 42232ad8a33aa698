/* mcq_reference_binding.h -- the reference-side binding of the MI355X query engine.
 *
 * This is the file a maintainer of jmabuin/metacache-mpi would add to its src/ to route the
 * per-read query path through libmcq_hip.so (include/mcq.h).  It is written against the
 * reference's real headers (config.h, sketch_database.h, candidates.h, querying.h,
 * hash_multimap.h) and is compiled and linked against them -- in the build container only --
 * by integration/Makefile (`make -C integration check`).  Nothing of the reference is copied
 * here and nothing built from it travels.
 *
 * What it replaces in the reference (paths relative to the reference root):
 *   gpu_engine::open         sketch_database::read -> hash_multimap::deserialize, src/sketch_database.h:858-952,
 *                            src/hash_multimap.h:923-964 (the table goes to HBM instead of the host heap;
 *                            taxonomy and targets are still read by the reference's own read(.., metadata_only))
 *   gpu_engine::query_block  the worker body of query_batched_parallel2 (src/querying.h:792-825) and the
 *                            MPI tree merge (src/querying.h:867-1073): one block of reads in, the per-read
 *                            classification_candidates out, keyed by query id like all_results_map (:733)
 *   to_candidates            mcq_cand[] -> classification_candidates through the reference's own
 *                            insert(cand, db, rules) (src/candidates.h:236-285), the way its receiver side
 *                            rebuilds lists from the wire (src/querying.h:954-971)
 *   flatten_feature_store    a live hash_multimap (src/hash_multimap.h:250-416, bucket iteration) -> the flat
 *                            keys / list_off / locs arrays of mcq_db_desc, for callers that have built or
 *                            modified the table in memory instead of reading shard files
 */
#ifndef MCQ_REFERENCE_BINDING_H
#define MCQ_REFERENCE_BINDING_H

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "timer.h"          /* querying.h uses `timer` without including it (src/querying.h:446) */
#include "config.h"
#include "sketch_database.h"
#include "candidates.h"
#include "query_options.h"
#include "sequence_io.h"

#include "mcq.h"
#include "mcq_host.h"

namespace mcq_binding {

/* feature -> sorted (tgt,win) lists in the layout mcq_db_desc takes */
struct flat_table {
    std::vector<std::uint32_t> keys;
    std::vector<std::uint64_t> list_off{0};
    std::vector<std::uint64_t> locs;            /* (tgt << 32) | win */
};

/* Flattens any feature store with the reference's hash_multimap interface: iteration over buckets
 * (begin()/end(), src/hash_multimap.h:860-880), bucket.empty()/key()/begin()/end() (:250-330), values
 * with .tgt/.win (target_location, src/sketch_database.h:157-189).  sketch_database keeps its store
 * private (features_, src/sketch_database.h:1109); a maintainer exposes it with one line,
 *     const feature_store& features() const noexcept { return features_; }
 * and calls flatten_feature_store(db.features()).  Lists come out in the order the reference keeps
 * them: ascending (tgt,win) (targets are inserted in increasing id, src/sketch_database.h:1079-1097). */
template <class FeatureStore>
flat_table flatten_feature_store(const FeatureStore& store)
{
    flat_table t;
    for (const auto& bucket : store) {
        if (bucket.empty()) continue;
        t.keys.push_back(static_cast<std::uint32_t>(bucket.key()));
        for (const auto& loc : bucket)
            t.locs.push_back((static_cast<std::uint64_t>(loc.tgt) << 32) | static_cast<std::uint64_t>(loc.win));
        t.list_off.push_back(t.locs.size());
    }
    return t;
}

/* Unions the tables of several ranks (each flat, keys unique inside a rank): the list of a key is the
 * concatenation of its per-rank lists re-sorted by (tgt,win) -- what all the reference's ranks hold together. */
flat_table union_tables(const std::vector<flat_table>& ranks);

/* taxon keys the engine carries (opaque to it except bit 31 = sequence-level taxon, mcq.h) */
struct taxon_keys {
    std::vector<std::uint32_t> tgt2tax;          /* [target_count] -> key */
    std::vector<const mc::taxon*> taxon_of_key;  /* key & 0x7FFFFFFF -> taxon */
    const mc::taxon* operator()(std::uint32_t key) const {
        const std::uint32_t i = key & 0x7FFFFFFFu;
        return (key == 0xFFFFFFFFu || i >= taxon_of_key.size()) ? nullptr : taxon_of_key[i];
    }
};
/* tgt2tax[t] = what insert() turns a candidate of target t into (src/candidates.h:242-245):
 * db.ancestor(db.taxon_of_target(t), mergeBelow) if it exists, else the sequence-level taxon itself (bit 31) */
taxon_keys make_taxon_keys(const mc::database& db, mc::taxon_rank mergeBelow);

/* mcq_cand[0..n) (one query's list as mcq_query returns it) -> the reference's candidate list, through
 * the real insert(): entries arrive in list order (hits descending, ties in arrival order), so insert()
 * appends each one behind its equals and the list comes out identical.                                 */
mc::classification_candidates
to_candidates(const mcq_cand* cands, std::uint32_t n, const taxon_keys& keys, const mc::database& db,
              const mc::candidate_generation_rules& rules);

class gpu_engine {
public:
    /* db: the reference's database with (at least) its metadata read; shard_prefix / n_ranks: the
     * <db>.db_<r> files of the build whose results are to be reproduced (their union goes to HBM).    */
    gpu_engine(const mc::database& db, const std::string& shard_prefix, std::uint32_t n_ranks,
               const mc::classification_options& opt, int device, std::uint64_t max_block_reads, std::uint64_t max_block_bases);
    /* the same from tables already in memory (flatten_feature_store / union_tables) */
    gpu_engine(const mc::database& db, const flat_table& table, const mc::classification_options& opt,
               int device, std::uint64_t max_block_reads, std::uint64_t max_block_bases);
    ~gpu_engine();
    gpu_engine(const gpu_engine&) = delete;
    gpu_engine& operator=(const gpu_engine&) = delete;

    /* One block of read pairs (what the threads of query_batched_parallel2 slurp at src/querying.h:784-790):
     * sketches, looks up, sorts, reduces and folds on the GPU, and leaves every read's candidates in `results`
     * under its query id (sequence.index), exactly what rank 0 holds after the tree merge (:1112-1117).
     * Reads with an empty header are skipped like in :793.  emulate_ranks = the mpiexec -n of the run to match.
     * Throws std::runtime_error with mcq_last_error() on failure (worker exceptions are logged as
     * "FAIL: ..." by the caller, src/querying.h:833-847).                                               */
    template <class ResultMap>
    void query_block(const std::vector<mc::sequence_pair_reader::sequence_pair>& reads, bool paired,
                     std::uint32_t emulate_ranks, ResultMap& results);

    const taxon_keys& keys() const noexcept { return keys_; }

private:
    void create(const flat_table& table, int device, std::uint64_t max_block_reads, std::uint64_t max_block_bases);
    void run(const std::string& bases, const std::vector<std::uint64_t>& off, bool paired, std::uint32_t emulate_ranks,
             std::vector<mcq_cand>& cands, std::vector<std::uint32_t>& ncand);

    const mc::database& db_;
    mc::classification_options opt_;
    taxon_keys keys_;
    mcq_db* gdb_ = nullptr;
    mcq_ws* ws_ = nullptr;
};

template <class ResultMap>
void gpu_engine::query_block(const std::vector<mc::sequence_pair_reader::sequence_pair>& reads, bool paired,
                             std::uint32_t emulate_ranks, ResultMap& results)
{
    std::string bases;
    std::vector<std::uint64_t> off{0};
    std::vector<std::uint_least64_t> ids;
    for (const auto& p : reads) {
        if (p.first.header.empty()) continue;                       /* src/querying.h:793 */
        bases += p.first.data;  off.push_back(bases.size());
        if (paired) { bases += p.second.data; off.push_back(bases.size()); }
        ids.push_back(p.first.index);
    }
    std::vector<mcq_cand> cands;
    std::vector<std::uint32_t> ncand;
    run(bases, off, paired, emulate_ranks, cands, ncand);

    mc::candidate_generation_rules rules;                           /* as the merge re-applies them, src/querying.h:881-884 */
    rules.mergeBelow    = opt_.lowestRank;
    rules.maxCandidates = opt_.maxNumCandidatesPerQuery;
    const std::size_t M = opt_.maxNumCandidatesPerQuery;
    for (std::size_t q = 0; q < ids.size(); ++q) {
        if (ncand[q] == 0) continue;                                /* a read without candidates has no map entry (:1112-1117) */
        results.insert(std::make_pair(ids[q], to_candidates(&cands[q * M], ncand[q], keys_, db_, rules)));
    }
}

} // namespace mcq_binding
#endif
