/*
 * mpi_transport.h -- how the C drivers obtain a bspgemm_comm under mpirun (host C, MPI only here).
 *
 * The reference's ranks talk MPI (final/SpGEMM_mpi_omp.c:178-204).  Here MPI carries two things
 * only: the RCCL unique id (one MPI_Bcast), when every rank has a GPU of its own, or -- when
 * ranks share a GPU, e.g. `mpirun -n 4` on a one-GPU box like the reference's `make test`
 * (final/Makefile:11-12); RCCL refuses two ranks on one device -- the all-gather / gatherv of the
 * library's host transport (MPI_Allgather / MPI_Gatherv on host buffers).  The stitch protocol is
 * the library's in both cases (bspgemm_comm_stitch_row_ptr).  BSPGEMM_MPI_HOST_TRANSPORT=1 forces
 * the host transport.
 */
#ifndef BSPGEMM_MPI_TRANSPORT_H
#define BSPGEMM_MPI_TRANSPORT_H
#include "../../include/bspgemm.h"

#include <limits.h>
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>

static int mpi_allgather_cb(void *user, const void *send, void *recv, size_t bytes)
{
    (void)user;
    if (bytes > INT_MAX) return 1;
    return MPI_Allgather(send, (int)bytes, MPI_BYTE, recv, (int)bytes, MPI_BYTE, MPI_COMM_WORLD) != MPI_SUCCESS;
}

static int mpi_gatherv_cb(void *user, const void *send, size_t send_bytes, void *recv, const size_t *recv_bytes, int root)
{
    (void)user;
    int size, rank;
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    /* counts in ints like the reference's (:184-203); a shard above 2^31-1 bytes goes as 4-byte words */
    int *counts = malloc((size_t)size * sizeof(int)), *displs = malloc((size_t)size * sizeof(int));
    if (!counts || !displs) { free(counts); free(displs); return 1; }
    size_t off = 0;
    int bad = send_bytes % 4 != 0;
    for (int r = 0; r < size; r++) {
        if (recv_bytes[r] % 4 != 0 || recv_bytes[r] / 4 > INT_MAX || off / 4 > INT_MAX) bad = 1;
        counts[r] = (int)(recv_bytes[r] / 4);
        displs[r] = (int)(off / 4);
        off += recv_bytes[r];
    }
    int rc = bad ? 1 : MPI_Gatherv(send, (int)(send_bytes / 4), MPI_INT, recv, counts, displs, MPI_INT, root, MPI_COMM_WORLD) != MPI_SUCCESS;
    free(counts); free(displs);
    return rc;
}

/* device for this rank (rank mod #devices unless BSPGEMM_DEVICE is set) + the communicator:
 * RCCL when the ranks fit the visible devices one to one, else the MPI host transport */
static bspgemm_status mpi_make_comm(bspgemm_context *ctx, int rank, int numtasks, int ndev, bspgemm_comm **comm, int *used_rccl)
{
    const char *force = getenv("BSPGEMM_MPI_HOST_TRANSPORT");
    const int host = (force && force[0] == '1') || numtasks > ndev;
    *used_rccl = !host;
    if (host) {
        static bspgemm_host_transport t = {NULL, mpi_allgather_cb, mpi_gatherv_cb};
        return bspgemm_comm_create_host(ctx, &t, rank, numtasks, comm);
    }
    unsigned char id[BSPGEMM_UNIQUE_ID_BYTES];
    bspgemm_status st = BSPGEMM_OK;
    if (rank == 0) st = bspgemm_comm_unique_id(id);
    int ok = st == BSPGEMM_OK;
    MPI_Bcast(&ok, 1, MPI_INT, 0, MPI_COMM_WORLD);
    if (!ok) return rank == 0 ? st : BSPGEMM_ERR_COMM;
    MPI_Bcast(id, BSPGEMM_UNIQUE_ID_BYTES, MPI_BYTE, 0, MPI_COMM_WORLD);
    return bspgemm_comm_create(ctx, id, rank, numtasks, comm);
}
#endif
