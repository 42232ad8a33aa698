/* mcq_host.h -- host-side companions of the query engine (C ABI, no GPU needed).
 *
 * Rows f1 and f3 of SURVEY.md section 8: reading the reference's database shard files
 * and the final per-read classification.  Each entry point names the reference code it
 * stands in for (paths relative to the reference root).
 *
 *   mcq_refdb_open      sketch_database::read            src/sketch_database.h:858-952
 *                       hash_multimap::deserialize        src/hash_multimap.h:923-964
 *                       taxon / taxonomy read_binary       src/taxonomy.h:312-335, :660-676
 *   mcq_refdb_tgt2tax   db.ancestor(taxon_of_target, r)   src/sketch_database.h:146, :717-720
 *                       as used by candidates insert()    src/candidates.h:242-245
 *   mcq_refdb_classify  classify()                        src/classification.cpp:235-265
 *                       ranked_lca()                      src/taxonomy.h:531-537
 *   mcq_rank_from_name  taxonomy::rank_from_name          src/taxonomy.h:173-213
 *   mcq_refdb_write_shard  sketch_database::write         src/sketch_database.h:959-998
 *   mcq_refdb_open_meta + mcq_shard_stream_*   the same reader, streaming (no host-side table)
 *
 * A taxon *key* is the index of the taxon in the database's taxon list; bit 31 marks a
 * sequence-level taxon (rank Sequence), 0xFFFFFFFF is "no taxon".  These are the keys
 * mcq_db_desc.tgt2tax carries and mcq_cand.tax returns.
 */
#ifndef MCQ_HOST_H
#define MCQ_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcq_refdb mcq_refdb;

typedef struct {
    uint32_t k, sketch_size, winlen, winstride;        /* target sketching parameters      */
    uint32_t q_sketch_size, q_winlen, q_winstride;      /* query sketching parameters       */
    uint32_t max_locs_per_feature;
    uint32_t n_ranks;                                    /* shard files read                 */
    uint32_t n_targets, n_taxa;
    uint64_t n_keys, n_locs;                             /* of the union table               */
} mcq_refdb_info;

#define MCQ_NO_TAXON 0xFFFFFFFFu
#define MCQ_RANK_SEQUENCE 0u
#define MCQ_RANK_SPECIES 4u
#define MCQ_RANK_DOMAIN 19u
#define MCQ_RANK_ROOT 20u
#define MCQ_RANK_NONE 21u

/* ---- writing a shard file the reference can read (sketch_database::write src/sketch_database.h:959-998,
 * hash_multimap::serialize src/hash_multimap.h:972-1029, taxon write_binary src/taxonomy.h:326-335) ------
 * One file = one rank: its parameters, the whole taxon list (sequence-level taxa have id -(target+1);
 * `windows` is non-zero only for the targets this rank owns) and this rank's table.  keys / list_off /
 * locs as in mcq_db_desc ((tgt << 32) | win, lists sorted, at most 255 entries each), host memory.
 * Keys are written in the order given (the reference's reader inserts them one by one, any order works). */
typedef struct {
    int64_t id, parent;
    uint8_t rank;               /* MCQ_RANK_* / taxonomy::rank, 21 = none */
    const char* name;
    const char* file;           /* source file name of a sequence-level taxon, "" otherwise */
    uint64_t index;             /* source.index */
    uint64_t windows;           /* source.windows */
} mcq_taxon_rec;
typedef struct {
    uint64_t k, sketch_size, winlen, winstride;         /* target sketcher            */
    uint64_t q_k, q_sketch_size, q_winlen, q_winstride; /* query sketcher             */
    uint64_t max_locs_per_feature;
} mcq_shard_params;
int mcq_refdb_write_shard(const char* path, const mcq_shard_params* params, const mcq_taxon_rec* taxa, uint64_t n_taxa,
                          uint32_t n_targets, const uint32_t* keys, const uint64_t* list_off, const uint64_t* locs,
                          uint64_t n_keys);

/* reads <prefix>.db_0 .. <prefix>.db_<n_ranks-1> and unions their tables */
int mcq_refdb_open(const char* prefix, uint32_t n_ranks, mcq_refdb** out);
int mcq_refdb_close(mcq_refdb* db);
int mcq_refdb_get_info(const mcq_refdb* db, mcq_refdb_info* out);

/* ---- the streaming route: shard files of any size (RefSeq scale: >= 1.5e10 locations) without a host-side union ----------
 * mcq_refdb_open_meta reads only the head of every shard file -- parameters, taxa, target count (sketch_database::read up to
 * the feature store, src/sketch_database.h:858-930) -- so that the taxon functions and classify work; info.n_keys stays 0,
 * the table accessors return nothing.  mcq_shard_stream_* hands the key records of one file out in file order as chunks of
 * (feature, target, window) triples (hash_multimap::deserialize, src/hash_multimap.h:923-964: per key {u32 key, u8 n, u64 n,
 * u32 tgt[n], u64 n, u32 win[n]}), whole key records per chunk; mcq_refdb_tgt_windows gives the windows of every target (the
 * taxon `windows` field of its owning rank) -- what turns (target, window) into a global window index.  The consumer is
 * mcq_parts_builder_* of include/mcq.h: the ranks are merged per feature-hash range ON THE GPU.  Host memory: one 16 MB read
 * buffer per open stream + the caller's chunk.                                                                            */
int mcq_refdb_open_meta(const char* prefix, uint32_t n_ranks, mcq_refdb** out);
int mcq_refdb_tgt_windows(const mcq_refdb* db, uint32_t* out /* [n_targets] */);
int mcq_refdb_file_stats(const mcq_refdb* db, uint32_t rank, uint64_t* bytes, uint64_t* n_keys, uint64_t* n_locs);
typedef struct mcq_shard_stream mcq_shard_stream;
int mcq_shard_stream_open(const mcq_refdb* db /* from mcq_refdb_open_meta */, uint32_t rank, mcq_shard_stream** out);
/* up to `cap` (>= 255) locations into feat / tgt / win; *n = 0 when the file is exhausted */
int mcq_shard_stream_next(mcq_shard_stream* s, uint32_t* feat, uint32_t* tgt, uint32_t* win, uint64_t cap, uint64_t* n);
int mcq_shard_stream_close(mcq_shard_stream* s);

/* union table in the layout mcq_db_desc wants; valid until mcq_refdb_close */
const uint32_t* mcq_refdb_keys(const mcq_refdb* db);
const uint64_t* mcq_refdb_list_off(const mcq_refdb* db);
const uint64_t* mcq_refdb_locs(const mcq_refdb* db);

/* taxon key per target for candidate merging below `merge_below_rank` */
int mcq_refdb_tgt2tax(const mcq_refdb* db, uint32_t merge_below_rank, uint32_t* out /* [n_targets] */);

int64_t mcq_refdb_taxon_id(const mcq_refdb* db, uint32_t key);       /* 0 for MCQ_NO_TAXON        */
uint32_t mcq_refdb_taxon_rank(const mcq_refdb* db, uint32_t key);
const char* mcq_refdb_taxon_name(const mcq_refdb* db, uint32_t key);
/* taxon index at `rank` in the ranked lineage of `key`, MCQ_NO_TAXON if none */
uint32_t mcq_refdb_ancestor(const mcq_refdb* db, uint32_t key, uint32_t rank);

/* cands: n x {tax key, hits, (2 ignored words)} as mcq_query returns them.
 * hits_diff_fraction as the reference stores it (-hitdiff 80 -> 0.8f).
 * Returns the taxon index of the classification or MCQ_NO_TAXON.                      */
uint32_t mcq_refdb_classify(const mcq_refdb* db, const uint32_t* cands, uint32_t n,
                            uint32_t hits_min, float hits_diff_fraction, uint32_t highest_rank);

/* default of -hitmin when unset: src/mode_query.cpp:247-259 */
uint32_t mcq_default_hits_min(uint32_t sketch_size);
uint32_t mcq_rank_from_name(const char* name);
const char* mcq_rank_name(uint32_t rank);
const char* mcq_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
