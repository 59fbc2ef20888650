// comm.hip -- the multi-GPU exchange (replaces SpGEMM_mpi, final/SpGEMM_mpi_omp.c:155-225): communicator
// over RCCL or host callbacks, the row_ptr stitch, the root gather of col_idx, SpGEMM_hip_multi.
#include "internal.hpp"

#include <rccl/rccl.h>
#include <unistd.h>

using namespace bsp;

#define NCCLCHK(call)                                                                      \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) {                                                           \
            snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", #call, ncclGetErrorString(r_), \
                     __FILE__, __LINE__);                                                  \
            return BSPGEMM_ERR_COMM;                                                       \
        }                                                                                  \
    } while (0)

// ------------------------------------------------------------------ multi-GPU stitch -----
// One protocol, two transports.  Every rank contributes the int32 LENGTHS of its shard's rows,
// padded to the longest shard; one all-gather; every rank scans the gathered lengths into the
// global int64 C.row_ptr on its device (bspgemm_lengths_to_row_ptr).  The all-gather is RCCL over
// xGMI (bspgemm_comm_create) or a host callback (bspgemm_comm_create_host: MPI_Allgather in the C
// drivers when ranks share a GPU, a fake in the tests) -- everything around it is the same code.
struct bspgemm_comm {
    bspgemm_context *ctx;
    ncclComm_t comm;                    // RCCL transport (NULL with the host transport)
    bspgemm_host_transport host;        // host transport callbacks (allgather NULL with RCCL)
    int rank, nranks;
    int *d_send = nullptr;              // width ints
    int *d_recv = nullptr;              // nranks * width ints
    size_t width_cap = 0;
    long long *d_global = nullptr;      // stitched row_ptr, grown on demand
    size_t global_cap = 0;
    long long *d_edges = nullptr;       // global row_ptr at the shard bounds (nranks + 1)
    int *d_bounds = nullptr;            // nranks + 1
    int *d_status = nullptr;            // 1 + nranks ints: this rank's status, everybody's (bspgemm_comm_agree)
    double timeout_s = 120.0;           // BSPGEMM_COMM_TIMEOUT_S: a collective that has not completed by then is aborted
    int inject = 0;                     // bspgemm_comm_inject_failure (tests): 1 root allocation fails, 2 the next RCCL wait "times out"
    bool rccl = false;                  // the transport this communicator was created with (never changes)
    bool dead = false;                  // a collective failed or timed out: the RCCL communicator has been aborted (or the host
                                        // transport reported an error); every later collective returns BSPGEMM_ERR_COMM
};

// A communicator is usable while it is alive and has the transport it was created with.  After an abort (comm_kill)
// every collective entry point returns BSPGEMM_ERR_COMM at once -- round 3 nulled c->comm instead and the next call
// took the host-transport branch through NULL callbacks.
static bspgemm_status comm_usable(const bspgemm_comm *c, const char *what)
{
    if (c->dead) {
        snprintf(g_err, sizeof g_err, "%s: the communicator is dead (an earlier collective failed or timed out and it was aborted)", what);
        return BSPGEMM_ERR_COMM;
    }
    if (c->rccl ? c->comm == nullptr : c->host.allgather == nullptr) {
        snprintf(g_err, sizeof g_err, "%s: the communicator has no transport", what);
        return BSPGEMM_ERR_COMM;
    }
    return BSPGEMM_OK;
}
static void comm_kill(bspgemm_comm *c)
{
    if (c->rccl && c->comm && !c->dead) ncclCommAbort(c->comm);   // frees the communicator: the handle must not be used again
    c->comm = nullptr;
    c->dead = true;
}

// Waits for the stream behind an RCCL call without trusting it to finish: a peer that died or left the
// protocol leaves the others inside the collective for ever (the reference's MPI calls have the same
// property: final/SpGEMM_mpi_omp.c:178-204 checks nothing).  Polls the stream and RCCL's asynchronous
// error state; on an error or after timeout_s the communicator is aborted and the call FAILS.
static bspgemm_status comm_wait(bspgemm_comm *c, hipStream_t s, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (c->inject == 2) {                                    // test hook: behave as if this wait had run out
        c->inject = 0;
        (void)hipStreamSynchronize(s);                       // (the collective itself did complete: drain it before the abort)
        snprintf(g_err, sizeof g_err, "%s: no completion after %.0f s (injected); communicator aborted", what, c->timeout_s);
        comm_kill(c);
        return BSPGEMM_ERR_COMM;
    }
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return BSPGEMM_OK;
        if (q != hipErrorNotReady) {
            snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(q));
            comm_kill(c);
            return BSPGEMM_ERR_HIP;
        }
        ncclResult_t async = ncclSuccess;
        if (c->comm && ncclCommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
            snprintf(g_err, sizeof g_err, "%s: RCCL reported %s; communicator aborted", what, ncclGetErrorString(async));
            comm_kill(c);
            return BSPGEMM_ERR_COMM;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
            snprintf(g_err, sizeof g_err, "%s: no completion after %.0f s (a peer is missing?); communicator aborted", what, c->timeout_s);
            comm_kill(c);
            return BSPGEMM_ERR_COMM;
        }
        usleep(200);
    }
}

extern "C" bspgemm_status bspgemm_comm_unique_id(unsigned char id[BSPGEMM_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == BSPGEMM_UNIQUE_ID_BYTES, "RCCL unique id size");
    if (!id) return FAIL(BSPGEMM_ERR_INVALID, "id is NULL");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return BSPGEMM_OK;
}

static bspgemm_status comm_new(bspgemm_context *ctx, int rank, int nranks, bspgemm_comm **out)
{
    bspgemm_comm *c = new (std::nothrow) bspgemm_comm();
    if (!c) return FAIL(BSPGEMM_ERR_ALLOC, "comm");
    c->ctx = ctx;
    c->comm = nullptr;
    c->host = bspgemm_host_transport{nullptr, nullptr, nullptr};
    c->rank = rank;
    c->nranks = nranks;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_edges), ((size_t)nranks + 1) * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_bounds), ((size_t)nranks + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_status), ((size_t)nranks + 1) * sizeof(int));
    if (const char *t = getenv("BSPGEMM_COMM_TIMEOUT_S")) { const double v = atof(t); if (v > 0) c->timeout_s = v; }
    if (e != hipSuccess) {
        bspgemm_comm_destroy(c);
        snprintf(g_err, sizeof g_err, "comm buffers: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? BSPGEMM_ERR_ALLOC : BSPGEMM_ERR_HIP;
    }
    *out = c;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_comm_create(bspgemm_context *ctx, const unsigned char id[BSPGEMM_UNIQUE_ID_BYTES],
                                              int rank, int nranks, bspgemm_comm **out)
{
    if (!ctx || !id || !out || nranks <= 0 || rank < 0 || rank >= nranks) return FAIL(BSPGEMM_ERR_INVALID, "comm_create");
    *out = nullptr;
    if (bspgemm_status st = use_device(ctx)) return st;
    bspgemm_comm *c = nullptr;
    if (bspgemm_status st = comm_new(ctx, rank, nranks, &c)) return st;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    // RCCL prints a version banner on STDOUT when the first communicator is created; the drop-in
    // binaries' stdout is the reference's CSV line / parity message and nothing else, so the
    // banner is sent to stderr
    fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) dup2(2, 1);
    c->rccl = true;
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (saved_out >= 0) {
        fflush(stdout);
        dup2(saved_out, 1);
        close(saved_out);
    }
    if (r != ncclSuccess) {
        c->comm = nullptr;
        bspgemm_comm_destroy(c);
        snprintf(g_err, sizeof g_err, "ncclCommInitRank: %s", ncclGetErrorString(r));
        return BSPGEMM_ERR_COMM;
    }
    *out = c;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_comm_create_host(bspgemm_context *ctx, const bspgemm_host_transport *t,
                                                   int rank, int nranks, bspgemm_comm **out)
{
    if (!ctx || !t || !t->allgather || !out || nranks <= 0 || rank < 0 || rank >= nranks)
        return FAIL(BSPGEMM_ERR_INVALID, "comm_create_host");
    *out = nullptr;
    if (bspgemm_status st = use_device(ctx)) return st;
    bspgemm_comm *c = nullptr;
    if (bspgemm_status st = comm_new(ctx, rank, nranks, &c)) return st;
    c->host = *t;
    *out = c;
    return BSPGEMM_OK;
}

extern "C" void bspgemm_comm_destroy(bspgemm_comm *c)
{
    if (!c) return;
    hipSetDevice(c->ctx->device);
    if (c->comm) ncclCommDestroy(c->comm);
    hipFree(c->d_send);
    hipFree(c->d_recv);
    hipFree(c->d_global);
    hipFree(c->d_edges);
    hipFree(c->d_bounds);
    hipFree(c->d_status);
    delete c;
}

extern "C" void bspgemm_comm_inject_failure(bspgemm_comm *c, int what) { if (c) c->inject = what; }

// Every rank contributes its status; everybody gets the worst.  Called before a collective that a failed
// rank would not enter: either all ranks go in, or none does (a rank that left the protocol alone leaves
// the others blocked in ncclSend / MPI_Gatherv for ever).  Collective.  On a dead communicator (an earlier
// collective failed or timed out) it returns BSPGEMM_ERR_COMM without touching the transport.
extern "C" bspgemm_status bspgemm_comm_agree(bspgemm_comm *c, bspgemm_status mine)
{
    if (!c) return FAIL(BSPGEMM_ERR_INVALID, "comm is NULL");
    if (bspgemm_status st = comm_usable(c, "bspgemm_comm_agree")) return st;
    bspgemm_context *ctx = c->ctx;
    const int n = c->nranks;
    if (n > 1024) return FAIL(BSPGEMM_ERR_INVALID, "more than 1024 ranks");      // (the same on every rank: nobody enters)
    int worst = (int)mine;
    int v = (int)mine;
    if (c->rccl) {
        // a rank-local HIP failure in front of the collective must not keep this rank out of it: the status
        // goes up by a synchronous copy, and if even that fails the rank still enters, reporting BSPGEMM_ERR_HIP
        hipStream_t s = ctx->stream;
        bool staged = hipSetDevice(ctx->device) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        if (!staged) v = v > (int)BSPGEMM_ERR_HIP ? v : (int)BSPGEMM_ERR_HIP;
        if (hipMemcpy(c->d_status, &v, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) (void)hipGetLastError();
        const ncclResult_t nr = ncclAllGather(c->d_status, c->d_status + 1, 1, ncclInt32, c->comm, s);
        if (nr != ncclSuccess) {
            snprintf(g_err, sizeof g_err, "status all-gather: %s; communicator aborted", ncclGetErrorString(nr));
            comm_kill(c);
            return BSPGEMM_ERR_COMM;
        }
        if (bspgemm_status st = comm_wait(c, s, "status all-gather")) return st;
        int all[1024];
        if (hipMemcpy(all, c->d_status + 1, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return FAIL(BSPGEMM_ERR_HIP, "status read-back");
        for (int r = 0; r < n; r++) if (all[r] > worst) worst = all[r];
        if (v > worst) worst = v;
    } else {
        int all[1024];                                       // (n <= 1024, checked above: no allocation that could fail here)
        const int rc = c->host.allgather(c->host.user, &v, all, sizeof(int));
        if (rc != 0) {
            c->dead = true;
            return FAIL(BSPGEMM_ERR_COMM, "host all-gather failed (communicator marked dead)");
        }
        for (int r = 0; r < n; r++) if (all[r] > worst) worst = all[r];
    }
    if (worst != BSPGEMM_OK && mine == BSPGEMM_OK)
        snprintf(g_err, sizeof g_err, "another rank failed with status %d (%s)", worst, bspgemm_status_string((bspgemm_status)worst));
    return (bspgemm_status)worst;
}

extern "C" int bspgemm_comm_rank(const bspgemm_comm *c) { return c ? c->rank : -1; }
extern "C" int bspgemm_comm_size(const bspgemm_comm *c) { return c ? c->nranks : 0; }

// lengths[i] = row_ptr[i+1] - row_ptr[i]  (|C_i| < 2^31 always; the slice-local offsets are not)
__global__ void k_row_lengths(const long long *__restrict__ row_ptr, int n, int *__restrict__ len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len[i] = (int)(row_ptr[i + 1] - row_ptr[i]);
}
// edges[r] = global[bounds[r]]
__global__ void k_pick_edges(const long long *__restrict__ global, const int *__restrict__ bounds, int n,
                             long long *__restrict__ edges)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) edges[r] = global[bounds[r]];
}

static bspgemm_status comm_bounds_ok(const bspgemm_comm *c, const int *bounds)
{
    if (bounds[0] != 0) return FAIL(BSPGEMM_ERR_INVALID, "bounds[0] != 0");
    for (int r = 0; r < c->nranks; r++)
        if (bounds[r + 1] < bounds[r]) return FAIL(BSPGEMM_ERR_INVALID, "bounds not ascending");
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_comm_stitch_row_ptr(bspgemm_comm *c, const bspgemm_result *local,
                                                      const int *bounds, const int64_t **d_row_ptr_global,
                                                      int64_t *shard_nnz)
{
    if (!c || !local || !bounds || !d_row_ptr_global) return FAIL(BSPGEMM_ERR_INVALID, "stitch_row_ptr");
    if (bspgemm_status st = comm_usable(c, "bspgemm_comm_stitch_row_ptr")) return st;
    if (bspgemm_status st = comm_bounds_ok(c, bounds)) return st;
    if (c->nranks > 1024) return FAIL(BSPGEMM_ERR_INVALID, "more than 1024 ranks");
    const int my_rows = bounds[c->rank + 1] - bounds[c->rank];
    if (my_rows != local->rows) return FAIL(BSPGEMM_ERR_INVALID, "local result does not match bounds[rank]");
    bspgemm_context *ctx = c->ctx;
    if (bspgemm_status st = use_device(ctx)) return st;
    hipStream_t s = ctx->stream;
    int width = 1;
    for (int r = 0; r < c->nranks; r++)
        if (bounds[r + 1] - bounds[r] > width) width = bounds[r + 1] - bounds[r];
    const size_t need = (size_t)bounds[c->nranks] + 1;
    // Staging buffers grow here.  Whether they must grow follows from `bounds` alone, which every rank passes
    // identically (the protocol's precondition), so all ranks take this branch together -- and because an
    // allocation can fail on ONE rank, they agree on the outcome before anybody enters the all-gather (round 3
    // returned from here alone and left the peers to the collective's timeout).
    if ((size_t)width > c->width_cap || need > c->global_cap) {
        auto grow = [&]() -> bspgemm_status {
            HIPCHK(hipStreamSynchronize(s));
            if ((size_t)width > c->width_cap) {
                hipFree(c->d_send); hipFree(c->d_recv);
                c->d_send = c->d_recv = nullptr;
                c->width_cap = 0;
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_send), (size_t)width * sizeof(int)));
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_recv), (size_t)width * c->nranks * sizeof(int)));
                c->width_cap = (size_t)width;
            }
            if (need > c->global_cap) {
                hipFree(c->d_global);
                c->d_global = nullptr;
                c->global_cap = 0;
                HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_global), need * sizeof(long long)));
                c->global_cap = need;
            }
            return BSPGEMM_OK;
        };
        bspgemm_status st = c->inject == 3 ? FAIL(BSPGEMM_ERR_ALLOC, "staging buffers (injected)") : grow();
        if (c->inject == 3) c->inject = 0;
        st = bspgemm_comm_agree(c, st);
        if (st) {
            // EVERY rank forgets its staging buffers, also those whose allocation went through: the next stitch must find
            // all ranks in this branch again (a rank that kept its buffers would skip the agree the others enter)
            hipFree(c->d_send); hipFree(c->d_recv); hipFree(c->d_global);
            c->d_send = c->d_recv = nullptr;
            c->d_global = nullptr;
            c->width_cap = c->global_cap = 0;
            return st;
        }
    }
    // 1. this shard's row lengths (pad slots are never read by the scan)
    if (my_rows > 0)
        hipLaunchKernelGGL(k_row_lengths, dim3((my_rows + 255) / 256), dim3(256), 0, s, local->d_row_ptr, my_rows, c->d_send);
    // 2. the one collective
    if (c->rccl) {
        const ncclResult_t nr = ncclAllGather(c->d_send, c->d_recv, (size_t)width, ncclInt32, c->comm, s);
        if (nr != ncclSuccess) {
            snprintf(g_err, sizeof g_err, "row-length all-gather: %s; communicator aborted", ncclGetErrorString(nr));
            comm_kill(c);
            return BSPGEMM_ERR_COMM;
        }
        if (bspgemm_status st = comm_wait(c, s, "row-length all-gather")) return st;
    } else {
        // host staging: a rank that cannot stage still enters the all-gather (with whatever it has) and fails afterwards,
        // so that nobody is left inside it
        const size_t bytes = (size_t)width * sizeof(int);
        int *hs = static_cast<int *>(calloc(1, bytes)), *hr = static_cast<int *>(malloc(bytes * c->nranks));
        bspgemm_status st = (hs && hr) ? BSPGEMM_OK : FAIL(BSPGEMM_ERR_ALLOC, "host staging");
        if (!st && hipMemcpyAsync(hs, c->d_send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "lengths to host");
        if (!st && hipStreamSynchronize(s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "sync");
        if (hs && hr) {
            if (c->host.allgather(c->host.user, hs, hr, bytes) != 0) {
                c->dead = true;
                if (!st) st = FAIL(BSPGEMM_ERR_COMM, "host all-gather failed (communicator marked dead)");
            }
        } else {
            c->dead = true;                                  // this rank could not take part: the others will see the transport fail
        }
        if (!st && hipMemcpyAsync(c->d_recv, hr, bytes * c->nranks, hipMemcpyHostToDevice, s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "lengths to device");
        if (!st && hipStreamSynchronize(s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "sync");
        free(hs); free(hr);
        if (st) return st;
    }
    // 3. every rank scans the gathered lengths: shard r continues where shard r-1 ended
    if (bspgemm_status st = bspgemm_lengths_to_row_ptr(ctx, c->d_recv, c->nranks, width, bounds,
                                                       reinterpret_cast<int64_t *>(c->d_global), s))
        return st;
    *d_row_ptr_global = reinterpret_cast<const int64_t *>(c->d_global);
    if (shard_nnz) {
        HIPCHK(hipMemcpyAsync(c->d_bounds, bounds, ((size_t)c->nranks + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pick_edges, dim3(1), dim3(c->nranks + 1 <= 1024 ? c->nranks + 1 : 1024), 0, s,
                           c->d_global, c->d_bounds, c->nranks + 1, c->d_edges);
        long long edges[1025];
        HIPCHK(hipMemcpyAsync(edges, c->d_edges, ((size_t)c->nranks + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int r = 0; r < c->nranks; r++) shard_nnz[r] = edges[r + 1] - edges[r];
    } else {
        HIPCHK(hipStreamSynchronize(s));
    }
    return BSPGEMM_OK;
}

// Root gather of the sharded col_idx (what MPI_Gatherv does at final/SpGEMM_mpi_omp.c:203): with
// RCCL every rank sends its shard and the root receives them at their global offsets -- one
// grouped send/recv, the root's own shard included, so that one rank runs the same calls as N.
// Everything that can fail on one rank only (the root's device buffer, a rank's host staging) is done FIRST and
// agreed on (bspgemm_comm_agree): either every rank enters the exchange or none does.
extern "C" bspgemm_status bspgemm_comm_gather_col_idx(bspgemm_comm *c, const bspgemm_result *local,
                                                      const int64_t *shard_nnz, int root, int *col_idx_host)
{
    if (!c || !local || !shard_nnz || root < 0 || root >= c->nranks) return FAIL(BSPGEMM_ERR_INVALID, "gather_col_idx");
    if (bspgemm_status st = comm_usable(c, "bspgemm_comm_gather_col_idx")) return st;
    bspgemm_context *ctx = c->ctx;
    hipStream_t s = ctx->stream;
    long long total = 0;
    for (int r = 0; r < c->nranks; r++) total += shard_nnz[r];
    // A root WITHOUT a destination (its malloc failed) still runs the whole collective into scratch and reports
    // afterwards (callers that did not agree on their allocation beforehand: the second line of defence).
    const bool root_blind = c->rank == root && total > 0 && !col_idx_host;
    // ---- rank-local preparation ---------------------------------------------------------------------------
    bspgemm_status prep = BSPGEMM_OK;
    if (shard_nnz[c->rank] != local->nnz) prep = FAIL(BSPGEMM_ERR_INVALID, "shard_nnz[rank] != nnz of the local result");
    if (!prep) prep = use_device(ctx);
    int *d_all = nullptr;                                    // RCCL: the root's device-side destination
    int *mine = nullptr, *scratch = nullptr;                 // host transport: this rank's shard on the host, a blind root's sink
    size_t *bytes = nullptr;
    if (!prep && c->rccl && c->rank == root) {
        const hipError_t e = c->inject == 4 ? hipErrorOutOfMemory : result_alloc(ctx, reinterpret_cast<void **>(&d_all), result_bytes_colidx(total));
        if (c->inject == 4) c->inject = 0;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            prep = FAIL(e == hipErrorOutOfMemory ? BSPGEMM_ERR_ALLOC : BSPGEMM_ERR_HIP, "device buffer of the gathered col_idx");
        }
    }
    if (!prep && !c->rccl) {
        if (!c->host.gatherv) prep = FAIL(BSPGEMM_ERR_INVALID, "host transport has no gatherv");
        if (!prep) {
            mine = static_cast<int *>(malloc((size_t)(local->nnz > 0 ? local->nnz : 1) * sizeof(int)));
            bytes = static_cast<size_t *>(malloc((size_t)c->nranks * sizeof(size_t)));
            if (root_blind) scratch = static_cast<int *>(malloc((size_t)total * sizeof(int)));
            if (!mine || !bytes || (root_blind && !scratch)) prep = FAIL(BSPGEMM_ERR_ALLOC, "host staging of the col_idx gather");
        }
        if (!prep) prep = bspgemm_result_download(ctx, local, nullptr, mine);
    }
    auto release = [&] {
        if (d_all) result_release(ctx, d_all, result_bytes_colidx(total));
        free(mine); free(bytes); free(scratch);
    };
    // ---- everybody or nobody -------------------------------------------------------------------------------
    if (bspgemm_status st = bspgemm_comm_agree(c, prep)) { release(); return st; }
    bspgemm_status st = BSPGEMM_OK;
    if (c->rccl) {
        ncclResult_t nr = ncclGroupStart();
        if (nr == ncclSuccess && local->nnz > 0)
            nr = ncclSend(local->d_col_idx, (size_t)local->nnz, ncclInt32, root, c->comm, s);
        if (c->rank == root) {
            long long off = 0;
            for (int r = 0; r < c->nranks && nr == ncclSuccess; r++) {
                if (shard_nnz[r] > 0) nr = ncclRecv(d_all + off, (size_t)shard_nnz[r], ncclInt32, r, c->comm, s);
                off += shard_nnz[r];
            }
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) {
            snprintf(g_err, sizeof g_err, "col_idx gather: %s; communicator aborted", ncclGetErrorString(nr));
            (void)hipStreamSynchronize(s);
            comm_kill(c);
            st = BSPGEMM_ERR_COMM;
        }
        if (!st) st = comm_wait(c, s, "col_idx gather");
        if (!st && c->rank == root && total > 0 && !root_blind &&
            hipMemcpyAsync(col_idx_host, d_all, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess)
            st = FAIL(BSPGEMM_ERR_HIP, "gathered col_idx to host");
        if (hipStreamSynchronize(s) != hipSuccess && !st) st = FAIL(BSPGEMM_ERR_HIP, "sync");
    } else {
        for (int r = 0; r < c->nranks; r++) bytes[r] = (size_t)shard_nnz[r] * sizeof(int);
        if (c->host.gatherv(c->host.user, mine, (size_t)local->nnz * sizeof(int), root_blind ? scratch : col_idx_host, bytes, root) != 0) {
            c->dead = true;
            st = FAIL(BSPGEMM_ERR_COMM, "host gatherv failed (communicator marked dead)");
        }
    }
    release();
    if (!st && root_blind) st = FAIL(BSPGEMM_ERR_INVALID, "root has no destination (the gather was run and discarded)");
    return st;
}

// replaces SpGEMM_mpi (final/SpGEMM_mpi_omp.c:155-225): same arguments plus the communicator the
// reference takes implicitly (MPI_COMM_WORLD).  Every rank passes the whole A and B (every rank of
// the reference reads the whole file, :309); rows are cut at equal work instead of An/numtasks
// (:165); the result -- malloc'ed *Ccol, caller's Crow[An+1] -- is valid on rank 0 only, like :200-223.
extern "C" int SpGEMM_hip_multi(bspgemm_comm *c, int *Acol, int *Arow, int An, int *Bcol, int *Brow, int Bm,
                                int **Ccol, int *Crow, int tBlock)
{
    (void)tBlock;
    if (Ccol) *Ccol = nullptr;
    auto run = [&]() -> bspgemm_status {
        if (!c || !Acol || !Arow || !Bcol || !Brow || !Ccol || !Crow || An < 0 || Bm < 0)
            return FAIL(BSPGEMM_ERR_INVALID, "SpGEMM_hip_multi arguments");
        bspgemm_context *ctx = c->ctx;
        const int brows = bspgemm_par_max_plus_one(Acol + Arow[0], (long long)Arow[An] - Arow[0]);
        bspgemm_matrix *A = nullptr, *B = nullptr;
        bspgemm_result *C = nullptr;
        int *bounds = static_cast<int *>(malloc(((size_t)c->nranks + 1) * sizeof(int)));
        int64_t *shard = static_cast<int64_t *>(malloc((size_t)c->nranks * sizeof(int64_t)));
        int64_t *rp64 = nullptr;
        int *dst = nullptr;
        bspgemm_status st = (bounds && shard) ? BSPGEMM_OK : FAIL(BSPGEMM_ERR_ALLOC, "bounds");
        if (!st) st = bspgemm_matrix_upload(ctx, An, brows, Arow, Acol, &A);
        if (!st) st = bspgemm_matrix_upload(ctx, brows, Bm, Brow, Bcol, &B);
        if (!st) st = bspgemm_partition_rows(ctx, A, B, c->nranks, bounds);
        if (!st) st = bspgemm_multiply(ctx, A, B, bounds[c->rank], bounds[c->rank + 1], &C);
        // from here on the ranks act together: nobody enters a collective unless everybody does
        st = bspgemm_comm_agree(c, st);
        const int64_t *d_global = nullptr;
        if (!st) st = bspgemm_comm_stitch_row_ptr(c, C, bounds, &d_global, shard);
        st = bspgemm_comm_agree(c, st);
        long long total = 0;
        if (!st) {
            for (int r = 0; r < c->nranks; r++) total += shard[r];
            if (total > INT_MAX) st = FAIL(BSPGEMM_ERR_OVERFLOW, "nnz(C) > INT_MAX: use the int64 handle API");
        }
        if (!st && c->rank == 0) {
            if (c->inject != 1) dst = static_cast<int *>(malloc((size_t)(total > 0 ? total : 1) * sizeof(int)));
            else c->inject = 0;                              // (one-shot test hook)
            rp64 = static_cast<int64_t *>(malloc(((size_t)An + 1) * sizeof(int64_t)));
            if (!dst || !rp64) st = FAIL(BSPGEMM_ERR_ALLOC, "host result");
        }
        // the root's host allocation can fail on one rank only: agreed on before the gather (which agrees once more on
        // its own rank-local buffers), so that either every rank enters it or none does.  A communicator that died on the
        // way (timeout, RCCL error) makes every further agree return BSPGEMM_ERR_COMM at once.
        st = bspgemm_comm_agree(c, st);
        if (!st) st = bspgemm_comm_gather_col_idx(c, C, shard, 0, dst);
        if (!st && c->rank == 0) {
            if (hipMemcpy(rp64, d_global, ((size_t)An + 1) * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)
                st = FAIL(BSPGEMM_ERR_HIP, "row_ptr to host");
            else {
                for (int i = 0; i <= An; i++) Crow[i] = (int)rp64[i];
                *Ccol = dst;
                dst = nullptr;
            }
        }
        free(dst); free(rp64); free(bounds); free(shard);
        bspgemm_result_free(C);
        bspgemm_matrix_free(A);
        bspgemm_matrix_free(B);
        return st;
    };
    const bspgemm_status st = run();
    return st ? dropin_fail("SpGEMM_hip_multi", st) : 0;
}
