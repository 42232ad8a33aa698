/* mcq.h -- C ABI of the MI355X query-path engine for MetaCache-MPI.
 *
 * This is the drop-in boundary.  The reference has no
 * plugin/FFI API; the seam is the per-read body + candidate merge inside
 * query_batched_parallel2 (reference src/querying.h:792-825 and :867-1073), i.e.
 *
 *   database::accumulate_matches(seq, res, offsets)     src/sketch_database.h:826-833
 *     -> sketcher::operator()(first,last)               src/hash_dna.h:113-152
 *     -> hash_multimap::find(key)                       src/hash_multimap.h:778-789
 *   merge_sort(a, offsets, b)                           src/querying.h:88-106
 *   classification_candidates{db, matches, rules}       src/candidates.h:207-219
 *   classification_candidates::insert(cand, db, rules)  src/candidates.h:236-285
 *   MPI tree merge of (qid,taxid,hits) triplets         src/querying.h:867-1073
 *
 * Each entry point below says which of these it replaces.  Plain pointers and
 * sizes only; 0 = success, negative = error (mcq_last_error() gives the text);
 * nothing throws across the boundary.  All buffers are caller-owned.
 *
 * Vocabulary: a *sequence* is one read (or mate); a *query* is one read or one
 * read pair; a *location* is (target id, window id) packed as (tgt << 32) | win;
 * a *feature* is one 32-bit min-hash value of a window sketch.
 */
#ifndef MCQ_H
#define MCQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcq_db mcq_db;   /* GPU-resident feature -> locations multimap (one shard) */
typedef struct mcq_ws mcq_ws;   /* per-caller workspace; one per concurrent mcq_query     */

enum {
    MCQ_OK = 0,
    MCQ_E_ARG = -1,        /* bad argument                                            */
    MCQ_E_HIP = -2,        /* HIP runtime error                                       */
    MCQ_E_CAPACITY = -3,   /* a query exceeded the workspace's per-query capacity     */
    MCQ_E_UNSUPPORTED = -4 /* parameter outside what the kernels are built for        */
};

/* flags */
enum {
    MCQ_DEVICE_PTRS = 1u,       /* the pointers in this struct are device pointers     */
    MCQ_QUIRK_SEQ_DROP = 2u,    /* emulate the reference's u32 wire format: a sequence-
                                   level taxon (key bit 31 set) sent by a non-root rank
                                   is dropped (src/querying.h:958, :983-985)            */
    MCQ_BATCH_RANGES = 8u,      /* mcq_batch.flags (device pointers only): seq_off holds 2*n_seqs
                                   (begin,end) byte ranges into `bases` instead of n_seqs+1 offsets:
                                   the sequences may sit anywhere in the buffer, e.g. inside raw
                                   FASTQ text indexed by mcq_fastq_index                          */
    MCQ_BATCH_PACKED = 0x10u,   /* mcq_batch.flags: `bases` is not ASCII but the packed form mcq_pack_bases writes (3 bits per
                                   base instead of 8: what a host that keeps its reads packed sends over PCIe): u32 words of
                                   2-bit codes, 16 bases per word, first base in the top bits (A/a 0, C/c 1, G/g 2, T/t 3), one
                                   zero pad word, then u32 words of ambiguity bits, 32 bases per word, first base in the top bit
                                   (1 = not ACGT: src/dna_encoding.h:326-336), one zero pad word.  seq_off stays in bases;
                                   mcq_batch.n_bases = seq_off[n_seqs].  Not with MCQ_BATCH_RANGES                        */
    MCQ_FORCE_BLOCK_PATH = 0x100u, /* test hook: every query takes the workgroup path    */
    MCQ_FORCE_RAW_SORT = 0x400u,   /* test hook: the wave path sorts the raw match list instead of
                                      de-duplicating it first (the path of > 256 distinct keys / 64-bit keys) */
    MCQ_NO_WAVE16 = 0x800u,        /* test hook: queries of 513..1024 locations take the workgroup path
                                      instead of the second wave stage                     */
    MCQ_NO_TWO_CLASS = 0x4000u,    /* test hook: long match lists are sorted whole instead of taking the two-class tail
                                      (light / heavy locations: DESIGN.md section 4); same results either way            */
    MCQ_FOLD_BY_LISTS = 0x8000u,   /* test hook: emulate_ranks > 1 builds the P bounded lists and folds them level by level (the form
                                      MCQ_QUIRK_SEQ_DROP needs on a table with sequence-level taxa) instead of the one selection
                                      in the order (hits, rank, position) that gives the same list        */
    MCQ_BUILD_REMOVE_OVERPOPULATED = 0x1000u, /* mcq_build_desc.flags: the build option
                                   -remove-overpopulated-features (src/mode_build.cpp:847-1074): a feature whose
                                   per-rank location counts (after the per-rank limit) sum to more than
                                   max_locs - 1 is removed from every rank                                  */
    MCQ_DB_LOCS_64 = 0x200u,    /* mcq_db_desc.flags: keep 64-bit locations in HBM even when
                                   (tgt,win) would fit a 32-bit form                     */
    MCQ_DB_LOCS_GW = 0x2000u,   /* mcq_db_desc.flags: 32-bit locations in the global-window form (first window of the
                                   target + window) even when the two bit fields would fit 32 bits.  Without either
                                   flag the handle picks: bit fields (tgt << wb) | win if they fit 32 bits, else the
                                   global-window form if the table has fewer than 2^32 - 1 windows (any table the
                                   reference can hold below ~485 Gbp, src/config.h:54-67), else 64-bit words  */
    MCQ_DB_SLOTS_16 = 0x4000u,  /* mcq_db_desc.flags: 16-B slots, every list behind the slot array           */
    MCQ_DB_BUCKETS_64 = 0x8000u /* mcq_db_desc.flags: 64-B buckets, lists of up to 14 (7) locations inside them.
                                   Without either flag the layout follows the table: buckets while the mean list has
                                   at most 4 locations, slots beyond                  */
};

/* Database description = the union of the reference's P shard tables
 * (hash_multimap contents, src/hash_multimap.h:923-964) plus the metadata the
 * candidate step needs.  keys are unique; list i = locs[list_off[i] .. list_off[i+1])
 * sorted ascending by (tgt,win).  tgt2tax[t] is the taxon key insert() works on for
 * target t: db.ancestor(taxon_of_target(t), mergeBelow) if it exists, else a key with
 * bit 31 set that is unique to the target (src/candidates.h:242-245).  Keys are opaque
 * to the engine except bit 31.
 * n_shards/shard_id: this handle stores only keys with mcq_owner(key, n_shards) ==
 * shard_id (the caller may pass all keys; foreign ones are skipped).               */
typedef struct {
    /* Limits narrower than the reference's, by construction of the kernels (MCQ_E_UNSUPPORTED beyond them):
     * k <= 16 is the reference's own limit for 32-bit k-mers (src/config.h:47); winlen <= 128 (`-winlen` is free in
     * the reference, src/mode_build.cpp:67; one wave holds a window as 2 bases per lane) and sketch_size <= 32 (the
     * selection network sorts 64 candidates) cover its defaults (128 / 16) and every shipped script.              */
    uint32_t k;               /* k-mer length, 1..16 (src/hash_dna.h:75)              */
    uint32_t sketch_size;     /* s, 1..32 (src/mode_build.cpp:66)                     */
    uint32_t winlen;          /* query window length, k..128 (src/sketch_database.h:256) */
    uint32_t winstride;       /* query window stride                                  */
    uint32_t tgt_winstride;   /* target window stride (range width, classification.cpp:217) */
    uint32_t n_targets;
    uint64_t n_keys;
    uint64_t n_locs;
    const uint32_t* keys;       /* [n_keys]                                           */
    const uint64_t* list_off;   /* [n_keys + 1]                                       */
    const uint64_t* locs;       /* [n_locs]  (tgt << 32) | win                        */
    const uint32_t* tgt2tax;    /* [n_targets]                                        */
    uint32_t n_shards;          /* >= 1                                               */
    uint32_t shard_id;
    uint32_t flags;             /* MCQ_DEVICE_PTRS, MCQ_DB_LOCS_64 / _GW, MCQ_DB_SLOTS_16 / MCQ_DB_BUCKETS_64 */
    int32_t  device;            /* HIP device ordinal                                 */
    uint32_t loc_win_bits;      /* 0 = derive from the data.  Shards built from different
                                   data must agree on the location format: pass the bit width
                                   of the largest window id of the WHOLE database          */
    const uint32_t* tgt_windows; /* [n_targets] windows of every target (the reference's taxon `windows` field of the
                                   owning rank, src/taxonomy.h:326-335), or NULL = 1 + the largest window id found in
                                   `locs`.  Only the global-window form reads it; shards that are created from
                                   different subsets of the locations must all pass it (same location words everywhere) */
} mcq_db_desc;

/* One batch of sequences: bases[seq_off[i] .. seq_off[i+1]) is sequence i (ASCII,
 * A/C/G/T any case, anything else is ambiguous: src/dna_encoding.h:326-336).
 * paired != 0: sequences 2q and 2q+1 are the mates of query q
 * (sequence_pair_reader::sequence_pair, src/sequence_io.h:204).                     */
typedef struct {
    uint64_t n_seqs;
    const char* bases;
    const uint64_t* seq_off;    /* [n_seqs + 1] */
    uint32_t paired;
    uint32_t flags;             /* MCQ_DEVICE_PTRS, MCQ_BATCH_RANGES, MCQ_BATCH_PACKED */
    uint64_t n_bases;           /* MCQ_BATCH_PACKED with device pointers: bases of the whole batch (locates the planes) */
} mcq_batch;

/* bytes of the packed form of n_bases bases, and the packer: ASCII (host, or device with MCQ_DEVICE_PTRS: then a
 * kernel on `stream`) -> packed buffer of mcq_packed_bytes(n_bases) bytes in the same memory space                  */
uint64_t mcq_packed_bytes(uint64_t n_bases);
int mcq_pack_bases(const char* bases, uint64_t n_bases, void* out, uint32_t flags, void* stream);

/* classification_options / candidate_generation_rules subset that the path uses
 * (src/query_options.h:123-135, src/candidates.h:89-101).                            */
typedef struct {
    uint32_t max_cand;          /* maxNumCandidatesPerQuery, 1..16.  (The reference's `-maxcand 0` = unbounded list,
                                   src/query_options.cpp:176-178, is not offered: past 16 entries its std::sort
                                   is an unstable introsort, so results stop being defined by the inputs.)   */
    uint32_t emulate_ranks;     /* P of the reference run to match (fold order of src/querying.h:867-1073),
                                   1..64; 1 = single list, no fold.  The P lists and their fold are computed as ONE
                                   bounded selection in the order (hits, rank, position) -- the same list,
                                   at the same cost for every P and max_cand (the reference's scripted -n 32 / -n 64 with
                                   -maxcand 4, script/ft/QueryGeneric_FT.sh:115, included).  Only MCQ_QUIRK_SEQ_DROP on a
                                   table with sequence-level taxa carries the lists out: in the lanes of a wave while
                                   pow2ceil(P) x max_cand <= 64, in four registers per lane up to 256 list slots, beyond
                                   that in the LDS of the workgroup kernel (every query, several times slower)          */
    uint64_t insert_size_max;   /* insertSizeMax                                         */
    uint32_t flags;             /* MCQ_QUIRK_SEQ_DROP and the MCQ_FORCE_* / MCQ_NO_WAVE16 / MCQ_NO_TWO_CLASS / MCQ_FOLD_BY_LISTS test hooks; any other
                                   bit is rejected with MCQ_E_ARG                         */
} mcq_query_opts;

/* match_candidate (src/candidates.h:66-81) with the taxon as its key.  After a fold
 * (emulate_ranks > 1) window positions are (0,0), as in the reference's wire format. */
typedef struct {
    uint32_t tax;
    uint32_t hits;
    uint32_t win_beg;
    uint32_t win_end;
} mcq_cand;

typedef struct {
    mcq_cand* cands;            /* [n_queries * max_cand]                              */
    uint32_t* n_cand;           /* [n_queries]                                         */
    uint32_t flags;             /* MCQ_DEVICE_PTRS                                     */
} mcq_result;

/* counters of the last mcq_query on a workspace (for the roofline accounting) */
typedef struct {
    uint64_t n_queries;
    uint64_t n_features;        /* sketch features looked up                           */
    uint64_t n_hit_features;    /* features with a non-empty list                      */
    uint64_t n_locations;       /* locations gathered                                  */
    uint64_t n_cands;           /* candidates written                                  */
    uint64_t n_overflow;        /* queries that left the first wave stage (second wave
                                   stage or block-per-query path)                      */
    uint64_t n_two_class;       /* queries answered by the two-class tail (heavy locations sorted, light ones only
                                   as far as they can enter a top list)                */
    uint64_t n_two_class_retry; /* queries that tail could not prove exact and handed on to the exact path */
    uint64_t n_narrow_queued;   /* queries with narrow window ranges (short reads, pairs) counted in the workgroup kernels' queue:
                                   from 4096 on they get the workgroup kernel with the two-class tail (an upper bound: an
                                   entry may be counted twice)                                            */
} mcq_stats;

/* replaces sketch_database::read -> hash_multimap::deserialize (the table build) */
int mcq_db_create(const mcq_db_desc* desc, mcq_db** out);
int mcq_db_destroy(mcq_db* db);
/* The same for a table that is larger than the memory for its one-piece description (RefSeq scale: 1.7e10 locations are
 * 136 GB as 64-bit words): handed over in parts, e.g. one per feature-hash range (mcq_build_parts makes them) or one per
 * shard file of the reference (src/sketch_database.h:858-999, after mapping (tgt, win) to tgt_windows' prefix sums + win).
 * Device memory only (MCQ_DEVICE_PTRS).  desc->keys / list_off / locs / n_keys / n_locs are ignored, desc->tgt_windows is
 * required: part locations are global-window words, and the handle keeps that form.  A key may appear in one part only. */
typedef struct {
    uint64_t n_keys, n_locs;
    const uint32_t* keys;       /* [n_keys]                                                             */
    const uint32_t* list_len;   /* [n_keys] lengths of the lists, which follow one another in `locs`    */
    const uint32_t* locs;       /* [n_locs] global window indices, every list ascending                 */
} mcq_db_part;
int mcq_db_create_parts(const mcq_db_desc* desc, const mcq_db_part* parts, uint32_t n_parts, mcq_db** out);
/* bytes of HBM held by the handle */
uint64_t mcq_db_bytes(const mcq_db* db);

/* max_queries / max_bases bound one batch; max_locs_per_query bounds the match list
 * of a single query on the block-per-query path (0 = default 1<<18).                 */
int mcq_ws_create(const mcq_db* db, uint64_t max_queries, uint64_t max_bases,
                  uint64_t max_locs_per_query, mcq_ws** out);
int mcq_ws_destroy(mcq_ws* ws);

/* The whole per-read path for one batch: rows 1-11 of SURVEY.md 8a.  Replaces the
 * worker body of query_batched_parallel2 (src/querying.h:792-825) and the MPI tree
 * merge (:867-1073).  stream is a hipStream_t (NULL = default stream).  With device
 * pointers the call only enqueues work; with host pointers it copies in, runs,
 * copies out and synchronises the stream.                                           */
int mcq_query(const mcq_db* db, mcq_ws* ws, const mcq_batch* in, const mcq_query_opts* opt,
              mcq_result* out, void* stream);

/* The same for a caller that streams batch after batch from HOST buffers (pinned memory for full speed): the call
 * returns once the work is enqueued; batch i+1 is copied in and batch i-1 copied out on their own streams while
 * batch i computes (two batches in flight, staging inside the workspace).  *ticket names the call; mcq_ws_wait(ticket)
 * returns when its results are in `out` -- until then `in` and `out` must stay untouched.  Calls complete in order.
 * ASCII bases: bound by PCIe (157 MB per 1 M x 150 bp reads); MCQ_BATCH_PACKED: bound by the kernels.               */
int mcq_query_pipelined(const mcq_db* db, mcq_ws* ws, const mcq_batch* in, const mcq_query_opts* opt,
                        mcq_result* out, uint64_t* ticket);
int mcq_ws_wait(mcq_ws* ws, uint64_t ticket);

/* Waits for the stream, returns MCQ_E_CAPACITY if any query of the last call on ws
 * overflowed the workspace, fills stats (may be NULL).                              */
int mcq_ws_sync(mcq_ws* ws, void* stream, mcq_stats* stats);

/* ---- staged entry points (feature-sharded multi-GPU path) ---------
 * mcq_sketch  : rows 1-5.  features[w * sketch_size + i], n_feat[w] for window w of
 *               the batch; win_query[w] = query index of window w; returns the number
 *               of windows in *n_windows (device scalar when MCQ_DEVICE_PTRS).
 * mcq_lookup_*: rows 6-7 on the owning shard (counts, then gather).
 * mcq_assemble / mcq_reduce: rows 8-11 on the home GPU.                              */
int mcq_count_windows(const mcq_db* db, const mcq_batch* in, uint64_t* win_off /* [n_seqs+1] */, void* stream);
int mcq_sketch(const mcq_db* db, const mcq_batch* in, const uint64_t* win_off,
               uint32_t* features, uint32_t* n_feat, void* stream);
/* mcq_lookup_count: list length per feature (0 for absent / foreign / 0xFFFFFFFF); if
 *   list_src is not NULL it also receives where each list starts in the shard, so that
 *   mcq_lookup_gather need not probe again.
 * mcq_lookup_gather: concatenates the lists at out_off[i] in the handle's NATIVE location
 *   width: mcq_db_loc_bytes() = 4 ((tgt << mcq_db_win_bits()) | win, or the global window index: mcq_db_layout_get)
 *   or 8 ((tgt << 32) | win).
 *   The native width is what travels between GPUs.                                     */
int mcq_lookup_count(const mcq_db* db, const uint32_t* features, uint64_t n_features,
                     uint32_t* list_len, uint64_t* list_src, void* stream);
int mcq_lookup_gather(const mcq_db* db, const uint32_t* features, uint64_t n_features,
                      const uint32_t* list_len, const uint64_t* list_src,
                      const uint64_t* out_off /* [n_features+1] */, void* out_locs, void* stream);
uint32_t mcq_db_loc_bytes(const mcq_db* db);
uint32_t mcq_db_win_bits(const mcq_db* db);
/* what a handle was built as (chosen per table by mcq_db_create) */
enum { MCQ_LOC_FIELDS64 = 0, MCQ_LOC_FIELDS32 = 1, MCQ_LOC_GLOBAL_WINDOW = 2 };
typedef struct {
    uint32_t loc_bytes;         /* 4 or 8                                                                */
    uint32_t loc_format;        /* MCQ_LOC_*: (tgt << 32) | win, (tgt << win_bits) | win, or gw_offsets[tgt] + win */
    uint32_t win_bits;
    uint32_t bucket_bytes;      /* 64 (lists of up to 14 / 7 locations inside the bucket) or 16           */
    uint32_t slots_per_key;     /* 4 or 2: load factor <= 0.25 / 0.5                                      */
    uint64_t n_slots, n_keys, n_locs;   /* of this shard                                                  */
    uint64_t n_ext_locs;        /* locations behind the slot array                                        */
    uint64_t n_windows;         /* global-window form: windows of the whole database                      */
    uint64_t bytes;             /* = mcq_db_bytes                                                         */
    const uint32_t* gw_offsets; /* device, [n_targets + 1]; NULL unless MCQ_LOC_GLOBAL_WINDOW             */
} mcq_db_layout;
int mcq_db_layout_get(const mcq_db* db, mcq_db_layout* out);
/* mcq_assemble: home side, after the lists came back.  List i (list_len[i] native-width
 *   locations, consecutive in src_locs) belongs to feature slot src_slot[i] of the batch's
 *   [window][sketch_size] feature array.  Produces the per-query segments mcq_reduce takes:
 *   loc_off[n_queries+1], query_len[n_queries] (sum of the mates' lengths) and dst_locs.   */
int mcq_assemble(const mcq_db* db, uint64_t n_lists, const uint32_t* list_len, const uint32_t* src_slot,
                 uint64_t n_slots, const void* src_locs, const mcq_batch* in, const uint64_t* win_off,
                 uint64_t* loc_off, uint32_t* query_len, void* dst_locs, void* stream);
/* mcq_reduce: rows 8-11 per query from native-width location segments (any order inside). */
int mcq_reduce(const mcq_db* db, mcq_ws* ws, uint64_t n_queries, const uint64_t* loc_off /* [n_queries+1] */,
               const void* locs, const uint32_t* query_len, const mcq_query_opts* opt, mcq_result* out, void* stream);

/* mcq_bucket_features: groups the non-empty features by owning shard.  counts is a device
 *   array of n_shards u64 receiving the per-shard counts; bucketed/src_index receive,
 *   shard after shard, the features and the slot (index into `features`) each came from. */
int mcq_bucket_features(const uint32_t* features, uint64_t n, uint32_t n_shards,
                        uint64_t* counts, uint32_t* bucketed, uint32_t* src_index, void* stream);

/* shard that owns a feature: a range of h2(f) = thomas_mueller_hash(f)
 * (src/hash_int.h:39-45), never of f itself                          */
uint32_t mcq_owner(uint32_t feature, uint32_t n_shards);

/* Per-kernel timing of the path's kernels (HIP events between them on the call's stream), for the
 * roofline line of bench.py.  enable != 0 starts recording.  mcq_ws_kernel_times returns, summed over
 * the batches since enabling, the milliseconds of ms[0] the first wave stage (k_query_wave / k_reduce_wave),
 * ms[1] the second and third wave stages (k_query_wave16 + k_query_wave32 / k_reduce_wave16), ms[2] the workgroup kernels
 * (k_query_block plain + two-class / k_reduce_block), and the number of batches; mcq_ws_kernel_time their total.
 * Both synchronise the recorded events.                                                   */
int mcq_ws_timing(mcq_ws* ws, int enable);
int mcq_ws_kernel_times(mcq_ws* ws, double* ms /* [3] */, uint64_t* n_batches);
int mcq_ws_kernel_time(mcq_ws* ws, double* total_ms, uint64_t* n_batches);

/* ---- feature-sharded multi-GPU path behind one call (csrc/mcq_shard.hpp) ----------------------
 * One process per GPU.  Replaces, for a table that is partitioned over n_ranks GPUs, what mcq_query replaces on one:
 * the worker body of query_batched_parallel2 and the MPI tree merge (src/querying.h:792-825, :867-1073).  The table
 * is partitioned by hash range of h2(feature) (mcq_owner) instead of the reference's tgt % P
 * (src/sketch_database.h:540-542); every rank queries its own reads; features travel to their owners and location
 * lists back (two exchanges per batch), the reduce kernels read the received lists in place.  Results equal
 * mcq_query's on the whole table, bit for bit (emulate_ranks reproduces the reference's fold order as there).
 *
 * Transport of the exchanges: RCCL (ncclSend / ncclRecv groups on the call's stream; librccl is loaded at run time)
 * after mcq_shard_comm_rccl, a device copy when n_ranks == 1, or the caller's function (mcq_shard_set_exchange;
 * synchronous: the engine waits for the stream before calling it).
 *
 * All calls are collective: every rank makes the same calls with the same flags in the same order.               */
typedef struct mcq_shard mcq_shard;
#define MCQ_SHARD_UNIQUE_ID_BYTES 128
enum { MCQ_SHARD_EXACT = 1u };   /* mcq_shard_query flags: exchange exact sizes (count exchanges through the host) */

typedef struct {
    uint32_t n_ranks;                  /* 1..32 */
    uint32_t rank;
    uint64_t max_queries;              /* per batch and rank */
    uint64_t max_seqs;                 /* 0 = max_queries (2 x for pairs) */
    uint64_t max_bases;                /* per batch and rank: bounds the number of windows */
    uint64_t max_locs_per_query;       /* as in mcq_ws_create; 0 = default */
    uint64_t max_features_per_peer;    /* capacity of one peer's feature block; 0 = twice the even share of the batch */
    uint64_t max_locations_per_peer;   /* capacity of one peer's location block to start with; 0 = sized by the table: the first
                                          batch of a context runs in the exact mode, its owner-side lookup first counts, and
                                          every rank allocates blocks for the largest count any rank served + 1/8.  An exact
                                          batch that outgrows the blocks allocates larger ones the same way (collectively);
                                          a padded batch that does is reported by mcq_shard_sync as MCQ_E_CAPACITY (never
                                          answered wrongly in silence) and its repeat with MCQ_SHARD_EXACT grows them.
                                          Identical on every rank.                                                          */
} mcq_shard_cfg;

/* moves send_bytes[p] bytes at send_base + send_off[p] to rank p and receives recv_bytes[p] bytes from rank p at
 * recv_base + recv_off[p], p = 0 .. n_ranks-1 (device pointers; own rank included); returns 0 on success, and only
 * when the received bytes are in place                                                                            */
typedef int (*mcq_exchange_fn)(void* user, const void* send_base, const uint64_t* send_off, const uint64_t* send_bytes,
                               void* recv_base, const uint64_t* recv_off, const uint64_t* recv_bytes,
                               uint32_t n_ranks, uint32_t rank);

int mcq_shard_create(const mcq_db* shard /* n_shards = n_ranks, shard_id = rank */, const mcq_shard_cfg* cfg, mcq_shard** out);
int mcq_shard_destroy(mcq_shard* ctx);
/* rank 0 makes the id (ncclGetUniqueId), the launcher carries it to the other ranks (the reference's world would use
 * MPI_Bcast; bench.py a torch.distributed broadcast), every rank passes it to mcq_shard_comm_rccl (ncclCommInitRank) */
int mcq_shard_unique_id(void* out /* MCQ_SHARD_UNIQUE_ID_BYTES */);
int mcq_shard_comm_rccl(mcq_shard* ctx, const void* unique_id);
int mcq_shard_set_exchange(mcq_shard* ctx, mcq_exchange_fn fn, void* user);
/* One batch (device pointers).  Default (padded mode): every block travels at a fixed size learned from the last exact
 * batch (largest count any rank saw, plus a sixteenth), its count inside it: the call only enqueues work, no host
 * round trip.  The first batch of a context, and any batch with MCQ_SHARD_EXACT, exchanges exact sizes after two
 * count exchanges through the host.  `next` (may be NULL): the batch of the following call, resident in device
 * memory and unchanged until that call -- its sketching is enqueued on a second stream now.  The exchanges and the
 * owner-side lookups run on a third stream of the context, the reduce kernels on `stream` behind them: a caller that
 * keeps calling without waiting has the next batch's exchanges and lookups under this batch's reduce kernels.
 * `out` and the statistics are complete when `stream` is (mcq_shard_sync).                                           */
int mcq_shard_query(mcq_shard* ctx, const mcq_batch* in, const mcq_query_opts* opt, mcq_result* out, void* stream,
                    uint32_t flags, const mcq_batch* next);
/* as mcq_ws_sync; MCQ_E_CAPACITY also when a block of the exchange was too small since the last sync (repeat those
 * batches with MCQ_SHARD_EXACT), or when a batch had more windows than cfg.max_bases was given for (no repeat helps:
 * create the context for larger batches).  The flags of a `next` batch's S1 are reported with THAT batch.            */
int mcq_shard_sync(mcq_shard* ctx, void* stream, mcq_stats* stats);
/* block sizes of the padded mode (features / locations per peer); setting them skips the learning batch */
int mcq_shard_set_caps(mcq_shard* ctx, uint64_t features_per_peer, uint64_t locations_per_peer);
int mcq_shard_get_caps(const mcq_shard* ctx, uint64_t* features_per_peer, uint64_t* locations_per_peer);
/* timing of the home side's reduce kernels, as mcq_ws_timing / mcq_ws_kernel_times */
int mcq_shard_timing(mcq_shard* ctx, int enable);
int mcq_shard_kernel_times(mcq_shard* ctx, double* ms /* [3] */, uint64_t* n_batches);
/* ... and of the other stages of a batch (events on the streams they run on; enabled by mcq_shard_timing): ms[0] S1 (window
 * count + sketch + route), ms[1] X1 (feature blocks to their owners), ms[2] S2 (owner-side lookup), ms[3] X2 (list ends and
 * locations back), summed over the n batches harvested so far.  The stages of consecutive batches overlap: the sum of a
 * batch's stages is more than the step time.                                                                          */
int mcq_shard_stage_times(mcq_shard* ctx, double* ms /* [4] */, uint64_t* n_batches);
/* accounting of the exchanges since the context was created: out[0] batches; bytes handed to the transport for OTHER ranks in
 * out[1] X1, out[2] X2 list ends + tile starts, out[3] X2 locations; out[4] bytes of this rank's own blocks; out[5] ranks of
 * the RCCL communicator (0 = another transport); out[6], out[7] block sizes of the padded mode (features / locations per peer) */
int mcq_shard_exchange_bytes(const mcq_shard* ctx, uint64_t* out /* [8] */);

/* ---- row f4: FASTQ ingest on the GPU ------------------------------------------------
 * text: raw FASTQ bytes in DEVICE memory (4 lines per record, as fastq_reader::read_next reads
 * them: src/sequence_io.cpp:251-285).  Writes the (begin,end) byte range of every record's
 * sequence line to seq_ranges[2*r], [2*r+1] (device, capacity 2*max_seqs) and the record count
 * to *n_seqs_out (device).  Feed mcq_query with bases = text, seq_off = seq_ranges,
 * flags = MCQ_DEVICE_PTRS | MCQ_BATCH_RANGES: the bases are read in place, never copied.  */
int mcq_fastq_index(const char* text, uint64_t n_bytes, uint64_t* seq_ranges, uint64_t max_seqs,
                    uint64_t* n_seqs_out, void* stream);
/* the same for FASTA text with every sequence on ONE line (2 lines per record: what read files in FASTA look like; the
 * reference's fasta_reader also joins sequences that span lines, src/sequence_io.cpp:122-200: such files -- genomes --
 * go through a host reader, their bases are not contiguous in the text)                                             */
int mcq_fasta_index(const char* text, uint64_t n_bytes, uint64_t* seq_ranges, uint64_t max_seqs,
                    uint64_t* n_seqs_out, void* stream);

/* ---- row f2: building the table from reference sequences on the GPU -----------------
 * Replaces the build-side loop add_all_window_sketches (src/sketch_database.h:1079-1097) with
 * target t inserted on rank t % P (src/sketch_database.h:540-542): every window of every target is
 * sketched, and per (feature, rank) the first max_locs locations in (target, window) order are
 * kept (src/sketch_database.h:1090-1092, bucket limit 254).  The result is the union of the P rank
 * tables in the layout mcq_db_desc takes.  Target sketching uses (k, sketch_size, winlen,
 * winstride); target t = bases[seq_off[t] .. seq_off[t+1]).                                   */
typedef struct {
    uint32_t k, sketch_size, winlen, winstride;
    uint32_t n_targets;
    const char* bases;            /* host, or device with MCQ_DEVICE_PTRS                     */
    const uint64_t* seq_off;      /* [n_targets+1]                                            */
    const uint32_t* tgt2tax;      /* [n_targets]; only mcq_db_build reads it                  */
    uint32_t emulate_ranks;       /* P of the build being reproduced (0 = 1)                  */
    uint32_t max_locs;            /* per (feature, rank); 0 = 254                             */
    uint32_t n_shards, shard_id;  /* mcq_db_build: as in mcq_db_desc                          */
    uint32_t flags;               /* MCQ_DEVICE_PTRS, MCQ_BUILD_REMOVE_OVERPOPULATED; mcq_db_build: MCQ_DB_LOCS_* and the layout flags too */
    int32_t device;
} mcq_build_desc;

typedef struct mcq_table mcq_table;   /* keys / list_off / locs in device memory */

int mcq_build_table(const mcq_build_desc* desc, mcq_table** out);
/* device pointers into the table (valid until mcq_table_free); any out pointer may be NULL.
 * win_off[n_targets+1] = first global window of every target.                               */
int mcq_table_info(const mcq_table* t, uint64_t* n_keys, uint64_t* n_locs, const uint32_t** keys,
                   const uint64_t** list_off, const uint64_t** locs, const uint64_t** win_off);
int mcq_table_free(mcq_table* t);
/* mcq_build_table + mcq_db_create in one call; the queryable handle is the only thing left in HBM */
int mcq_db_build(const mcq_build_desc* desc, mcq_db** out);
/* The build in parts, for tables whose one-piece temporaries do not fit the GPU (RefSeq scale: ~60 B per feature slot of
 * the sequences; mcq_db_build takes this way by itself then).  The features are cut into ranges of h2(feature) -- sub-ranges
 * of shard `shard_id` of `n_shards` when the table is sharded: only that shard's features are built -- and every range is
 * built on its own from the sequences (device memory only): what stays is the table in the form the query side stores, keys
 * + list lengths + 32-bit global-window words.  Same lists as mcq_build_table (per (feature, rank) limit, rank merge,
 * -remove-overpopulated-features).  The caller may release the sequences before mcq_db_from_parts makes the handle.       */
typedef struct mcq_parts mcq_parts;
int mcq_build_parts(const mcq_build_desc* desc, mcq_parts** out);
int mcq_parts_info(const mcq_parts* parts, uint64_t* n_keys, uint64_t* n_locs, uint64_t* n_windows, uint32_t* n_parts, uint64_t* bytes);
/* tgt2tax: [n_targets] host, or device with MCQ_DEVICE_PTRS in flags; flags also takes MCQ_DB_SLOTS_16 / MCQ_DB_BUCKETS_64 */
int mcq_db_from_parts(const mcq_parts* parts, const uint32_t* tgt2tax, uint32_t n_shards, uint32_t shard_id, uint32_t flags, mcq_db** out);
int mcq_parts_free(mcq_parts* parts);

/* ---- parts from streamed (feature, target, window) triples: a table of any size from the REFERENCE'S OWN shard files --------
 * (row f1 at scale: sketch_database::read + hash_multimap::deserialize, src/sketch_database.h:858-952, src/hash_multimap.h:923-964,
 * without the host-side union of mcq_refdb_open -- 16 B per location and a sort on the host: 240 GB for a RefSeq build.)
 * The host library streams every shard file as chunks of triples (include/mcq_host.h: mcq_refdb_open_meta, mcq_shard_stream_*);
 * mcq_parts_builder_add takes a chunk (host pointers, or device with MCQ_DEVICE_PTRS), turns (target, window) into the global
 * window index of tgt_windows' prefix sums on the GPU and appends the (feature, word) pairs of this shard's features to the buffer
 * of their feature-hash range; mcq_parts_builder_finish sorts every range -- the merge of the reference's P per-rank lists of a
 * feature into (target, window) order -- and leaves the parts mcq_db_from_parts takes (the builder is released).  Chunks may
 * come in any order, from any number of files.  Device memory at the peak: 8 B per location of this shard + ~30 B per location
 * of one range.                                                                                                          */
typedef struct {
    uint32_t k, sketch_size, winlen, winstride;   /* what QUERIES are sketched with (the q_* fields of mcq_refdb_info)      */
    uint32_t tgt_winstride;                       /* window stride of the targets (0 = winstride)                           */
    uint32_t n_targets;
    const uint32_t* tgt_windows;                  /* host [n_targets]: mcq_refdb_tgt_windows                                */
    uint64_t expected_locations;                  /* of the WHOLE table (sum of the shard files' location counts): sizes the ranges */
    uint32_t n_ranges;                            /* 0 = from expected_locations and the free memory                        */
    uint32_t n_shards, shard_id;                  /* as in mcq_db_desc: only this shard's features are kept                 */
    int32_t device;
} mcq_parts_builder_desc;
typedef struct mcq_parts_builder mcq_parts_builder;
int mcq_parts_builder_create(const mcq_parts_builder_desc* desc, mcq_parts_builder** out);
int mcq_parts_builder_add(mcq_parts_builder* b, const uint32_t* feat, const uint32_t* tgt, const uint32_t* win, uint64_t n, uint32_t flags);
int mcq_parts_builder_finish(mcq_parts_builder* b, mcq_parts** out);
int mcq_parts_builder_free(mcq_parts_builder* b);
const char* mcq_build_last_error(void);

/* ---- debug / parity taps ------------------------------------------------------------
 * Row 5 in isolation is mcq_count_windows + mcq_sketch above (the sketches of every window).
 * Rows 7-8 in isolation: the sorted match list of every query (what merge_sort returns,
 * src/querying.h:88-106): match_off[q..q+1) into matches (capacity cap; pass matches = NULL first to
 * learn the sizes).  path_flags = 0 taps every query on the path it takes in mcq_query (first or second
 * wave stage, workgroup kernel); MCQ_FORCE_BLOCK_PATH / MCQ_FORCE_RAW_SORT / MCQ_NO_WAVE16 tap the path
 * those hooks select.  The taps live in separate instantiations of the kernels.                     */
int mcq_debug_matches(const mcq_db* db, mcq_ws* ws, const mcq_batch* in, uint32_t path_flags,
                      uint64_t* match_off /* host [n_queries+1] */, uint64_t* matches /* host */, uint64_t cap);

/* diagnostic builds only (-DMCQ_PHASE_CLOCK): shader clocks per phase of the workgroup kernel of the last synchronised call */
int mcq_debug_phase_clocks(mcq_ws* ws, uint64_t* out /* [22] */);

const char* mcq_last_error(void);
const char* mcq_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MCQ_H */
