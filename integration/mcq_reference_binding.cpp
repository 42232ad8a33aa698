/* mcq_reference_binding.cpp -- see mcq_reference_binding.h.  Compiled against the reference's headers by
 * integration/Makefile in the build container; not part of libmcq_hip.so. */
#include "mcq_reference_binding.h"
#include "mcq_open.hpp"

#include <algorithm>
#include <map>
#include <stdexcept>

namespace mcq_binding {

flat_table union_tables(const std::vector<flat_table>& ranks)
{
    std::map<std::uint32_t, std::vector<std::uint64_t>> all;
    for (const auto& t : ranks)
        for (std::size_t i = 0; i < t.keys.size(); ++i) {
            auto& l = all[t.keys[i]];
            l.insert(l.end(), t.locs.begin() + t.list_off[i], t.locs.begin() + t.list_off[i + 1]);
        }
    flat_table u;
    for (auto& kv : all) {
        std::sort(kv.second.begin(), kv.second.end());      /* (tgt << 32) | win: the order of target_location::operator< */
        u.keys.push_back(kv.first);
        u.locs.insert(u.locs.end(), kv.second.begin(), kv.second.end());
        u.list_off.push_back(u.locs.size());
    }
    return u;
}

taxon_keys make_taxon_keys(const mc::database& db, mc::taxon_rank mergeBelow)
{
    taxon_keys k;
    std::unordered_map<const mc::taxon*, std::uint32_t> index;
    auto key_of = [&](const mc::taxon* t) {
        auto it = index.find(t);
        if (it != index.end()) return it->second;
        const std::uint32_t i = static_cast<std::uint32_t>(k.taxon_of_key.size());
        k.taxon_of_key.push_back(t);
        index.emplace(t, i);
        return i;
    };
    const std::uint64_t n = db.target_count();
    k.tgt2tax.resize(n);
    for (std::uint64_t t = 0; t < n; ++t) {
        const mc::taxon* tax = db.taxon_of_target(static_cast<mc::target_id>(t));
        const mc::taxon* anc = (tax && mergeBelow > mc::taxon_rank::Sequence) ? db.ancestor(tax, mergeBelow) : nullptr;
        if (anc) tax = anc;                                           /* src/candidates.h:242-245 */
        if (!tax) { k.tgt2tax[t] = 0xFFFFFFFFu; continue; }
        std::uint32_t key = key_of(tax);
        if (tax->rank() == mc::taxon_rank::Sequence) key |= 0x80000000u;   /* bounded insert without the same-taxon search (:251-259) */
        k.tgt2tax[t] = key;
    }
    return k;
}

mc::classification_candidates
to_candidates(const mcq_cand* cands, std::uint32_t n, const taxon_keys& keys, const mc::database& db,
              const mc::candidate_generation_rules& rules)
{
    mc::classification_candidates cls;
    for (std::uint32_t i = 0; i < n; ++i) {
        mc::match_candidate c{keys(cands[i].tax), cands[i].hits};     /* src/candidates.h:66-81 */
        c.pos = mc::window_range{static_cast<mc::window_id>(cands[i].win_beg), static_cast<mc::window_id>(cands[i].win_end)};
        cls.insert(c, db, rules);                                     /* src/candidates.h:236-285 */
    }
    return cls;
}

static void check(int rc, const char* what)
{
    if (rc != MCQ_OK) throw std::runtime_error(std::string(what) + ": " + mcq_last_error());
}

gpu_engine::gpu_engine(const mc::database& db, const std::string& shard_prefix, std::uint32_t n_ranks,
                       const mc::classification_options& opt, int device, std::uint64_t max_block_reads, std::uint64_t max_block_bases)
: db_(db), opt_(opt), keys_(make_taxon_keys(db, opt.lowestRank))
{
    /* the shard files hold the tables in the reference's own serialisation: libmcq_host reads them -- unioned on the host when
     * they are small, streamed to the GPU and merged there from 1 GB on (include/mcq_open.hpp) -- with the taxon keys of the
     * reference's own database object */
    mcq_refdb* rdb = nullptr;
    std::string err; bool streamed = false;
    if (mcq_open_refdb(shard_prefix, n_ranks, mcq_stream_load_min_bytes(), &rdb, &streamed, err) != 0)
        throw std::runtime_error("opening the shard files: " + err);
    const int rc = mcq_make_db(rdb, streamed, keys_.tgt2tax.data(), 1, 0, device, &gdb_, err);
    mcq_refdb_close(rdb);
    if (rc != 0) throw std::runtime_error("making the GPU table: " + err);
    check(mcq_ws_create(gdb_, max_block_reads, max_block_bases, 0, &ws_), "mcq_ws_create");
}

gpu_engine::gpu_engine(const mc::database& db, const flat_table& table, const mc::classification_options& opt,
                       int device, std::uint64_t max_block_reads, std::uint64_t max_block_bases)
: db_(db), opt_(opt), keys_(make_taxon_keys(db, opt.lowestRank))
{
    create(table, device, max_block_reads, max_block_bases);
}

void gpu_engine::create(const flat_table& t, int device, std::uint64_t max_block_reads, std::uint64_t max_block_bases)
{
    mcq_db_desc d{};
    d.k             = db_.query_sketcher().kmer_size();               /* src/hash_dna.h:80 */
    d.sketch_size   = static_cast<std::uint32_t>(db_.query_sketcher().sketch_size());
    d.winlen        = static_cast<std::uint32_t>(db_.query_window_size());      /* src/sketch_database.h:334-352 */
    d.winstride     = static_cast<std::uint32_t>(db_.query_window_stride());
    d.tgt_winstride = static_cast<std::uint32_t>(db_.target_window_stride());   /* src/classification.cpp:217-219 */
    d.n_targets     = static_cast<std::uint32_t>(db_.target_count());
    d.n_keys = t.keys.size(); d.keys = t.keys.data(); d.list_off = t.list_off.data();
    d.n_locs = t.locs.size(); d.locs = t.locs.data(); d.tgt2tax = keys_.tgt2tax.data();
    d.n_shards = 1; d.shard_id = 0; d.flags = 0; d.device = device;
    check(mcq_db_create(&d, &gdb_), "mcq_db_create");
    check(mcq_ws_create(gdb_, max_block_reads, max_block_bases, 0, &ws_), "mcq_ws_create");
}

gpu_engine::~gpu_engine()
{
    mcq_ws_destroy(ws_);
    mcq_db_destroy(gdb_);
}

void gpu_engine::run(const std::string& bases, const std::vector<std::uint64_t>& off, bool paired, std::uint32_t emulate_ranks,
                     std::vector<mcq_cand>& cands, std::vector<std::uint32_t>& ncand)
{
    const std::uint64_t n_seqs = off.size() - 1, nq = paired ? n_seqs / 2 : n_seqs;
    mcq_batch in{};
    in.n_seqs = n_seqs; in.bases = bases.data(); in.seq_off = off.data(); in.paired = paired ? 1u : 0u; in.flags = 0;
    mcq_query_opts qo{};
    qo.max_cand        = static_cast<std::uint32_t>(opt_.maxNumCandidatesPerQuery);   /* src/query_options.h:134 */
    qo.emulate_ranks   = emulate_ranks;                                               /* fold order of src/querying.h:867-1073 */
    qo.insert_size_max = opt_.insertSizeMax;
    qo.flags           = MCQ_QUIRK_SEQ_DROP;             /* the u32 wire of :983-985 drops sequence-level taxa of non-root ranks */
    cands.assign(nq * qo.max_cand, mcq_cand{});
    ncand.assign(nq, 0);
    mcq_result out{};
    out.cands = cands.data(); out.n_cand = ncand.data(); out.flags = 0;
    check(mcq_query(gdb_, ws_, &in, &qo, &out, nullptr), "mcq_query");
}

} // namespace mcq_binding
