// mcq_query_mpi -- `mpiexec -n N mcq_query_mpi <db> P r1.fq r2.fq ...`: the multi-GPU form of the reference's
// `mpiexec -n P metacache query` (src/main.cpp:41-104, src/mode_query.cpp:404-458), host code in C++ around the C ABI.
//
// One process per GPU (rank r uses device r mod device count).  Where the reference gives every rank the targets
// tgt % P and lets every rank sketch every read (src/sketch_database.h:540-542, src/querying.h:792-825), here every
// rank holds the hash range of the feature table it owns (mcq_db_desc.n_shards = N, .shard_id = rank: the union of the
// reference's P shard files, filtered by mcq_owner) and queries ITS slice of the reads through mcq_shard_query:
// features travel to their owners and location lists back -- ncclSend / ncclRecv groups over RCCL, the communicator's
// id made on rank 0 and carried by MPI_Bcast -- instead of the reference's tree of blocking MPI_Send / MPI_Recv of
// (query, taxon, hits) triplets (src/querying.h:867-1073).  emulate_ranks = P reproduces that tree's fold order, so the
// output is the reference's for `mpiexec -n P`, whatever N is.  Rank 0 gathers the mapping lines (MPI_Gatherv) and the
// statistics (MPI_Reduce) and writes the -out file (mcq_cli_common.hpp).
//
// -transport mpi moves the blocks through the host and MPI_Alltoallv instead (mcq_shard_set_exchange): for boxes where
// several ranks share one GPU, which RCCL refuses -- and the way this program is tested on a one-GPU box.
//
// The reads go through in batches (-batch queries, -batch-bases bases per rank and batch; the context and the communicator
// are set up for that shape before the clock starts): batch j+1 is staged and announced while batch j is enqueued and
// batch j-1 may still run, results come back behind the kernels and are written out while the next batches run.
//
// usage: mpiexec -n N mcq_query_mpi <dbprefix> <P> <r1.fq> <r2.fq|-> [options of mcq_query_cli] [-transport rccl|mpi]
//                                   [-batch N] [-batch-bases N]
#include <mpi.h>
#include <hip/hip_runtime_api.h>

#include <sstream>

#include "mcq_cli_common.hpp"

#define HIP_OR_DIE(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    std::fprintf(stderr, "ABORT: %s: %s\n", #expr, hipGetErrorString(e_)); MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)
#define MCQ_OR_DIE(expr) do { if ((expr) != MCQ_OK) { \
    std::fprintf(stderr, "ABORT: %s: %s\n", #expr, mcq_last_error()); MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

// mcq_exchange_fn over MPI: device blocks -> host, MPI_Alltoallv, host -> device
static int exchange_over_mpi(void*, const void* send_base, const uint64_t* send_off, const uint64_t* send_bytes,
                             void* recv_base, const uint64_t* recv_off, const uint64_t* recv_bytes, uint32_t n, uint32_t) {
    std::vector<int> sc(n), sd(n), rc(n), rd(n);
    uint64_t st = 0, rt = 0;
    for (uint32_t p = 0; p < n; ++p) {
        if (send_bytes[p] > 0x7FFFFFFFull || recv_bytes[p] > 0x7FFFFFFFull || st > 0x7FFFFFFFull || rt > 0x7FFFFFFFull) return 1;   // int counts of MPI
        sc[p] = (int)send_bytes[p]; sd[p] = (int)st; st += send_bytes[p];
        rc[p] = (int)recv_bytes[p]; rd[p] = (int)rt; rt += recv_bytes[p];
    }
    std::vector<char> hs(st ? st : 1), hr(rt ? rt : 1);
    for (uint32_t p = 0; p < n; ++p)
        if (sc[p] && hipMemcpy(hs.data() + sd[p], (const char*)send_base + send_off[p], send_bytes[p], hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (MPI_Alltoallv(hs.data(), sc.data(), sd.data(), MPI_BYTE, hr.data(), rc.data(), rd.data(), MPI_BYTE, MPI_COMM_WORLD) != MPI_SUCCESS) return 1;
    for (uint32_t p = 0; p < n; ++p)
        if (rc[p] && hipMemcpy((char*)recv_base + recv_off[p], hr.data() + rd[p], recv_bytes[p], hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    int rank = 0, N = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &N);
    Options p;
    if (!parse_options(argc, argv, p)) { MPI_Finalize(); return 2; }
    int n_dev = 0;
    HIP_OR_DIE(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { std::fprintf(stderr, "ABORT: no GPU\n"); MPI_Abort(MPI_COMM_WORLD, 1); }
    const int device = rank % n_dev;
    HIP_OR_DIE(hipSetDevice(device));

    // this rank's hash range of the table
    mcq_refdb* rdb = nullptr; std::vector<uint32_t> t2t; uint32_t hitmin = 0;
    mcq_db* shard = nullptr;
    if (!open_database(p, &rdb, t2t, &shard, hitmin, (uint32_t)N, (uint32_t)rank, device)) MPI_Abort(MPI_COMM_WORLD, 1);

    // the context for a fixed batch shape and the communicator: set up before the clock starts, like the reference's
    // database load and MPI_Init (identical capacities on every rank)
    const uint64_t B = p.batch, MB = p.batch_bases;
    mcq_shard_cfg cfg; std::memset(&cfg, 0, sizeof(cfg));
    cfg.n_ranks = (uint32_t)N; cfg.rank = (uint32_t)rank;
    cfg.max_queries = B; cfg.max_seqs = 2 * B; cfg.max_bases = MB;
    mcq_shard* ctx = nullptr;
    MCQ_OR_DIE(mcq_shard_create(shard, &cfg, &ctx));
    if (p.transport == "mpi") MCQ_OR_DIE(mcq_shard_set_exchange(ctx, exchange_over_mpi, nullptr));
    else if (N > 1 || std::getenv("MCQ_SHARD_FORCE_RCCL")) {
        char id[MCQ_SHARD_UNIQUE_ID_BYTES];
        if (rank == 0) MCQ_OR_DIE(mcq_shard_unique_id(id));                // ncclGetUniqueId
        MPI_Bcast(id, sizeof id, MPI_BYTE, 0, MPI_COMM_WORLD);
        MCQ_OR_DIE(mcq_shard_comm_rccl(ctx, id));                          // ncclCommInitRank
    }
    // three sets of device inputs / outputs: batch j+1 is staged while batch j-1 may still run (its set is the one batch
    // j+2 will take), results come back on the stream behind the kernels
    constexpr int NS = 3;
    hipStream_t st = nullptr;
    HIP_OR_DIE(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char* d_bases[NS]; uint64_t* d_off[NS]; mcq_cand* d_cands[NS]; uint32_t* d_ncand[NS];
    mcq_cand* h_cands[NS]; uint32_t* h_ncand[NS]; hipEvent_t ev_out[NS];
    for (int s = 0; s < NS; ++s) {
        HIP_OR_DIE(hipMalloc(&d_bases[s], MB + 64));
        HIP_OR_DIE(hipMalloc(&d_off[s], (2 * B + 1) * 8));
        HIP_OR_DIE(hipMalloc(&d_cands[s], B * p.maxcand * sizeof(mcq_cand)));
        HIP_OR_DIE(hipMalloc(&d_ncand[s], B * 4));
        HIP_OR_DIE(hipHostMalloc(&h_cands[s], B * p.maxcand * sizeof(mcq_cand)));
        HIP_OR_DIE(hipHostMalloc(&h_ncand[s], B * 4));
        HIP_OR_DIE(hipEventCreateWithFlags(&ev_out[s], hipEventDisableTiming));
    }

    MPI_Barrier(MPI_COMM_WORLD);                                            // src/mode_query.cpp:129
    const auto t_start = std::chrono::steady_clock::now();
    // every rank reads the files and keeps its contiguous slice of the queries
    std::vector<Rec> r1, r2;
    if (!read_records(p.f1, r1)) { std::fprintf(stderr, "FAIL: can't open file %s\n", p.f1.c_str()); MPI_Abort(MPI_COMM_WORLD, 1); }
    const bool paired = p.paired();
    if (paired && !read_records(p.f2, r2)) { std::fprintf(stderr, "FAIL: can't open file %s\n", p.f2.c_str()); MPI_Abort(MPI_COMM_WORLD, 1); }
    const size_t nq_all = paired ? std::min(r1.size(), r2.size()) : r1.size();
    const size_t q0 = nq_all * (size_t)rank / (size_t)N, q1 = nq_all * (size_t)(rank + 1) / (size_t)N;
    // the slice in batches of at most B queries and MB bases; every rank makes the same number of (collective) calls
    std::vector<size_t> cut{q0};
    {
        size_t nb_q = 0; uint64_t nb_b = 0;
        for (size_t q = q0; q < q1; ++q) {
            const uint64_t len = r1[q].seq.size() + (paired ? r2[q].seq.size() : 0);
            if (len > MB) { std::fprintf(stderr, "ABORT: query %zu is longer than -batch-bases\n", q); MPI_Abort(MPI_COMM_WORLD, 1); }
            if (nb_q == B || nb_b + len > MB) { cut.push_back(q); nb_q = 0; nb_b = 0; }
            ++nb_q; nb_b += len;
        }
        cut.push_back(q1);
    }
    unsigned long long nb_mine = cut.size() - 1, nb = 0;
    MPI_Allreduce(&nb_mine, &nb, 1, MPI_UNSIGNED_LONG_LONG, MPI_MAX, MPI_COMM_WORLD);
    auto lo = [&](size_t j) { return j < cut.size() - 1 ? cut[j] : q1; };
    auto hi = [&](size_t j) { return j < cut.size() - 1 ? cut[j + 1] : q1; };

    mcq_query_opts qo; qo.max_cand = p.maxcand; qo.emulate_ranks = p.P; qo.insert_size_max = p.insertsize;
    qo.flags = p.quirks ? MCQ_QUIRK_SEQ_DROP : 0;
    const Out o = make_out(rdb, p);
    std::ostringstream lines;
    uint64_t assigned[MCQ_RANK_NONE + 1] = {0};
    std::vector<mcq_batch> in(nb ? nb : 1);
    std::string bases; std::vector<uint64_t> off;
    auto stage = [&](size_t j) {                                           // batch j of this rank to its device set
        const int s = (int)(j % NS);
        bases.clear(); off.assign(1, 0);
        for (size_t q = lo(j); q < hi(j); ++q) {
            bases += r1[q].seq; off.push_back(bases.size());
            if (paired) { bases += r2[q].seq; off.push_back(bases.size()); }
        }
        if (!bases.empty()) HIP_OR_DIE(hipMemcpy(d_bases[s], bases.data(), bases.size(), hipMemcpyHostToDevice));
        HIP_OR_DIE(hipMemcpy(d_off[s], off.data(), off.size() * 8, hipMemcpyHostToDevice));
        std::memset(&in[j], 0, sizeof(mcq_batch));
        in[j].n_seqs = off.size() - 1; in[j].bases = d_bases[s]; in[j].seq_off = d_off[s]; in[j].paired = paired ? 1 : 0; in[j].flags = MCQ_DEVICE_PTRS;
    };
    auto finish = [&](size_t j) {                                          // results of batch j: wait, write its mapping lines
        const int s = (int)(j % NS);
        HIP_OR_DIE(hipEventSynchronize(ev_out[s]));
        for (size_t q = lo(j); q < hi(j); ++q)
            write_query(lines, o, p, hitmin, r1[q].header, &h_cands[s][(q - lo(j)) * p.maxcand], h_ncand[s][q - lo(j)], assigned);
    };
    // the first batch of a context exchanges exact sizes and learns the block sizes the others travel at; should a later
    // batch not fit them (MCQ_E_CAPACITY at the end), everything is repeated with exact sizes
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint32_t flags = attempt ? MCQ_SHARD_EXACT : 0;
        lines.str(""); std::memset(assigned, 0, sizeof(assigned));
        if (nb) stage(0);
        for (size_t j = 0; j < nb; ++j) {
            const int s = (int)(j % NS);
            if (j >= 2) finish(j - 2);                                      // (its device set is the one batch j+1 takes)
            if (j + 1 < nb) stage(j + 1);
            mcq_result res; res.cands = d_cands[s]; res.n_cand = d_ncand[s]; res.flags = MCQ_DEVICE_PTRS;
            MCQ_OR_DIE(mcq_shard_query(ctx, &in[j], &qo, &res, st, flags, j + 1 < nb ? &in[j + 1] : nullptr));
            const size_t nqj = hi(j) - lo(j);
            if (nqj) {
                HIP_OR_DIE(hipMemcpyAsync(h_cands[s], d_cands[s], nqj * p.maxcand * sizeof(mcq_cand), hipMemcpyDeviceToHost, st));
                HIP_OR_DIE(hipMemcpyAsync(h_ncand[s], d_ncand[s], nqj * 4, hipMemcpyDeviceToHost, st));
            }
            HIP_OR_DIE(hipEventRecord(ev_out[s], st));
        }
        if (nb >= 2) finish(nb - 2);
        if (nb >= 1) finish(nb - 1);
        const int rc = mcq_shard_sync(ctx, st, nullptr);
        int bad = rc == MCQ_E_CAPACITY ? 1 : 0, any = 0;
        if (rc != MCQ_OK && rc != MCQ_E_CAPACITY) { std::fprintf(stderr, "ABORT: %s\n", mcq_last_error()); MPI_Abort(MPI_COMM_WORLD, 1); }
        MPI_Allreduce(&bad, &any, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
        if (!any) break;
        if (attempt) { std::fprintf(stderr, "ABORT: %s\n", mcq_last_error()); MPI_Abort(MPI_COMM_WORLD, 1); }
    }

    // rank 0 collects the mapping lines and the statistics and writes
    const std::string mine_s = lines.str();
    if (mine_s.size() > 0x7FFFFFFFull) { std::fprintf(stderr, "ABORT: more than 2 GB of mapping lines on one rank\n"); MPI_Abort(MPI_COMM_WORLD, 1); }
    int len = (int)mine_s.size();
    std::vector<int> lens(N), disp(N);
    MPI_Gather(&len, 1, MPI_INT, lens.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
    std::string all;
    if (rank == 0) { int t = 0; for (int r = 0; r < N; ++r) { disp[r] = t; t += lens[r]; } all.resize((size_t)t); }
    MPI_Gatherv(mine_s.data(), len, MPI_CHAR, rank == 0 ? &all[0] : nullptr, lens.data(), disp.data(), MPI_CHAR, 0, MPI_COMM_WORLD);
    unsigned long long a_loc[MCQ_RANK_NONE + 1], a_all[MCQ_RANK_NONE + 1];
    for (int i = 0; i <= (int)MCQ_RANK_NONE; ++i) a_loc[i] = assigned[i];
    MPI_Reduce(a_loc, a_all, MCQ_RANK_NONE + 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
    if (rank == 0) {
        std::ofstream fout; if (!p.outfile.empty()) fout.open(p.outfile);
        std::ostream& os = p.outfile.empty() ? std::cout : fout;
        write_head(os, o, p, hitmin);
        os << all;
        for (int i = 0; i <= (int)MCQ_RANK_NONE; ++i) assigned[i] = a_all[i];
        write_summary(os, o, p, assigned, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    }
    for (int s = 0; s < NS; ++s) {
        (void)hipFree(d_bases[s]); (void)hipFree(d_off[s]); (void)hipFree(d_cands[s]); (void)hipFree(d_ncand[s]);
        (void)hipHostFree(h_cands[s]); (void)hipHostFree(h_ncand[s]); (void)hipEventDestroy(ev_out[s]);
    }
    (void)hipStreamDestroy(st);
    mcq_shard_destroy(ctx); mcq_db_destroy(shard); mcq_refdb_close(rdb);
    MPI_Finalize();
    return 0;
}
