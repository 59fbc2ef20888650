"""Single-GPU checks of the multi-GPU plumbing (the driver runs N=2,4,8; the builder has one GPU):
the RCCL paths are exercised with a world of ONE rank -- communicator creation, the zero-copy
torch view of the result's device row_ptr, the all-gather + rebase -- and must reproduce the
local row_ptr exactly."""
import os
import socket

import numpy as np
import pytest

import bspgemm

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_native_rccl_stitch_one_rank():
    import ctypes as C
    L = bspgemm.lib()
    ctx = bspgemm.Context(0)
    rp, ci, n = bspgemm.gen_rmat(13, 8, (0.30, 0.25, 0.25), seed=5)
    A = ctx.upload(rp, ci, n)
    Cres = ctx.multiply(A, A)
    crp, _ = Cres.download(col_idx=False)
    uid = C.create_string_buffer(128)
    assert L.bspgemm_comm_unique_id(uid) == 0
    comm = C.c_void_p()
    assert L.bspgemm_comm_create(ctx._h, uid, 0, 1, C.byref(comm)) == 0, L.bspgemm_last_error()
    bounds = np.array([0, n], dtype=np.int32)
    dptr = C.c_void_p()
    shard = np.zeros(1, dtype=np.int64)
    st = L.bspgemm_comm_stitch_row_ptr(comm, Cres._h, bounds, C.byref(dptr), C.c_void_p(shard.ctypes.data))
    assert st == 0, L.bspgemm_last_error()
    assert shard[0] == crp[-1]
    import torch
    from bspgemm import dist as bdist
    g = bdist.device_tensor(dptr.value, n + 1, torch.int64, torch.device("cuda", 0)).cpu().numpy()
    assert np.array_equal(g, crp)
    L.bspgemm_comm_destroy(comm)
    ctx.close()


def test_torch_rccl_stitch_one_rank():
    import torch
    import torch.distributed as dist
    from bspgemm import dist as bdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ctx = bspgemm.Context(0)
        rp, ci, n = bspgemm.gen_uniform(1 << 14, 8, seed=3)
        A = ctx.upload(rp, ci, n)
        prefix = ctx.row_work_prefix(A, A)
        bounds = bdist.shard_bounds(prefix, 1)
        assert bounds.tolist() == [0, n]
        Cres = ctx.multiply(A, A)
        crp, _ = Cres.download(col_idx=False)
        local = bdist.device_tensor(Cres.row_ptr_device, Cres.rows + 1, torch.int64, dev)   # zero-copy view
        assert np.array_equal(local.cpu().numpy(), crp)
        g, shard = bdist.stitch_row_ptr(local, bounds)
        assert np.array_equal(g.cpu().numpy(), crp) and int(shard[0]) == crp[-1]
        g2, shard2 = bdist.stitch_row_ptr(local, bounds, ctx=ctx)        # lengths scanned by the library
        assert np.array_equal(g2.cpu().numpy(), crp) and int(shard2[0]) == crp[-1]
        # the library half alone, on a made-up 3-rank gather with ragged shards (one of them empty)
        lens = np.diff(crp).astype(np.int32)
        fake_bounds = np.array([0, 5000, 5000, n], dtype=np.int32)
        width = int(np.diff(fake_bounds).max())
        padded = np.zeros((3, width), dtype=np.int32)
        for r in range(3):
            padded[r, : fake_bounds[r + 1] - fake_bounds[r]] = lens[fake_bounds[r]: fake_bounds[r + 1]]
        padded[1, :] = 7                                                 # pad slots must be ignored
        d_len = torch.from_numpy(padded).to(dev)
        out = torch.empty(n + 1, dtype=torch.int64, device=dev)
        ctx.lengths_to_row_ptr(d_len.data_ptr(), 3, width, fake_bounds, out.data_ptr(),
                               torch.cuda.current_stream(dev).cuda_stream)
        assert np.array_equal(out.cpu().numpy(), crp)
        ctx.close()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# N > 1 on ONE GPU: the library's host transport (bspgemm_comm_create_host) lets several "ranks" --
# here threads of this process, each with its own context and streams; under mpirun, processes that
# share the GPU -- run the very code an N-GPU job runs around its one all-gather: ragged equal-work
# shards, padded int32 row lengths, bspgemm_lengths_to_row_ptr, the root gather of col_idx and the
# SpGEMM_mpi-shaped drop-in.  Only the transport differs from the RCCL job (which needs N devices).
def _thread_transport(world):
    import ctypes as C
    import threading
    barrier = threading.Barrier(world, timeout=180)
    slots = [b""] * world

    def make(rank):
        def allgather(user, send, recv, nbytes):
            try:
                slots[rank] = C.string_at(send, nbytes)
                barrier.wait()
                data = b"".join(slots)
                C.memmove(recv, data, len(data))
                barrier.wait()
                return 0
            except Exception:
                return 1

        def gatherv(user, send, send_bytes, recv, recv_bytes, root):
            try:
                slots[rank] = C.string_at(send, send_bytes) if send_bytes else b""
                barrier.wait()
                if rank == root:
                    assert [len(x) for x in slots] == [recv_bytes[r] for r in range(world)]
                    data = b"".join(slots)
                    if data:
                        C.memmove(recv, data, len(data))
                barrier.wait()
                return 0
            except Exception:
                return 1

        t = bspgemm.HostTransport(None, bspgemm.ALLGATHER_FN(allgather), bspgemm.GATHERV_FN(gatherv))
        return t
    return make


def _run_ranks(world, body):
    import threading
    out, errs = [None] * world, []

    def run(rank):
        try:
            out[rank] = body(rank)
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errs.append((rank, repr(e)))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs
    return out


@pytest.mark.parametrize("world", [2, 3, 4])
def test_spgemm_hip_multi_host_transport(world):
    """SpGEMM_hip_multi (the SpGEMM_mpi drop-in, final/SpGEMM_mpi_omp.c:155-225) with `world` ranks on
    one GPU: rank 0's Crow/Ccol must be the whole product, bit for bit"""
    import ctypes as C
    from oracle import oracle as O
    import gen
    L = bspgemm.lib()
    rp, ci, n = gen.rmat(13, 8, (0.57, 0.19, 0.19, 0.05), seed=41)      # skewed: ragged shards, heavy rows
    erp, eci = O.spgemm(rp, ci, rp, ci, n)
    rp, ci = bspgemm._i32(rp), bspgemm._i32(ci)
    make = _thread_transport(world)

    def body(rank):
        ctx = bspgemm.Context(0)
        t = make(rank)
        comm = C.c_void_p()
        assert L.bspgemm_comm_create_host(ctx._h, C.byref(t), rank, world, C.byref(comm)) == 0, L.bspgemm_last_error()
        assert L.bspgemm_comm_rank(comm) == rank and L.bspgemm_comm_size(comm) == world
        crow = np.zeros(n + 1, dtype=np.int32)
        cc = C.POINTER(C.c_int)()
        st = L.SpGEMM_hip_multi(comm, ci, rp, n, ci, rp, n, C.byref(cc), crow, 64)
        assert st == 0, L.bspgemm_last_error()
        res = None
        if rank == 0:
            res = (crow, bspgemm._take_i32(cc, crow[-1]))
        else:
            assert not cc, "non-root ranks get *Ccol = NULL (final/SpGEMM_mpi_omp.c:200 allocates on root only)"
        L.bspgemm_comm_destroy(comm)
        ctx.close()
        return res
    out = _run_ranks(world, body)
    crow, ccol = out[0]
    assert np.array_equal(crow, erp) and np.array_equal(ccol, eci)


def test_stitch_with_an_empty_shard_host_transport():
    """bspgemm_comm_stitch_row_ptr on hand-made bounds: a shard of zero rows between two ragged ones;
    every rank ends with the same global row_ptr, shard_nnz adds up"""
    import ctypes as C
    import torch
    from bspgemm import dist as bdist
    from oracle import oracle as O
    L = bspgemm.lib()
    rp, ci, n = bspgemm.gen_uniform(6000, 6, seed=9)
    erp, _ = O.spgemm(rp, ci, rp, ci, n)
    bounds = np.array([0, 4100, 4100, n], dtype=np.int32)
    world = 3
    make = _thread_transport(world)

    def body(rank):
        ctx = bspgemm.Context(0)
        t = make(rank)
        comm = C.c_void_p()
        assert L.bspgemm_comm_create_host(ctx._h, C.byref(t), rank, world, C.byref(comm)) == 0
        A = ctx.upload(rp, ci, n)
        Cres = ctx.multiply(A, A, int(bounds[rank]), int(bounds[rank + 1]))
        dptr = C.c_void_p()
        shard = np.zeros(world, dtype=np.int64)
        st = L.bspgemm_comm_stitch_row_ptr(comm, Cres._h, bounds, C.byref(dptr), C.c_void_p(shard.ctypes.data))
        assert st == 0, L.bspgemm_last_error()
        g = bdist.device_tensor(dptr.value, n + 1, torch.int64, torch.device("cuda", 0)).cpu().numpy()
        L.bspgemm_comm_destroy(comm)
        ctx.close()
        return g, shard
    for g, shard in _run_ranks(world, body):
        assert np.array_equal(g, erp)
        assert shard.tolist() == [int(erp[4100]), 0, int(erp[n] - erp[4100])]


@pytest.mark.parametrize("world", [2, 3])
def test_spgemm_hip_multi_root_allocation_fails(world):
    """A failure on ONE rank must not leave the others inside a collective (VERDICT r2 #9 / ADVICE: a root
    whose malloc failed returned before the gather while the peers blocked in ncclSend / MPI_Gatherv).
    Rank 0's host allocation is forced to fail: every rank must come back with a non-zero status, within
    the timeout, and *Ccol must stay NULL everywhere."""
    import ctypes as C
    import time
    L = bspgemm.lib()
    L.bspgemm_comm_inject_failure.argtypes = [C.c_void_p, C.c_int]
    L.bspgemm_comm_inject_failure.restype = None
    rp, ci, n = bspgemm.gen_uniform(4096, 8, seed=17)
    rp, ci = bspgemm._i32(rp), bspgemm._i32(ci)
    make = _thread_transport(world)

    def body(rank):
        ctx = bspgemm.Context(0)
        t = make(rank)
        comm = C.c_void_p()
        assert L.bspgemm_comm_create_host(ctx._h, C.byref(t), rank, world, C.byref(comm)) == 0, L.bspgemm_last_error()
        if rank == 0:
            L.bspgemm_comm_inject_failure(comm, 1)
        crow = np.zeros(n + 1, dtype=np.int32)
        cc = C.POINTER(C.c_int)()
        t0 = time.time()
        st = L.SpGEMM_hip_multi(comm, ci, rp, n, ci, rp, n, C.byref(cc), crow, 64)
        dt = time.time() - t0
        L.bspgemm_comm_destroy(comm)
        ctx.close()
        return st, bool(cc), dt
    out = _run_ranks(world, body)
    for rank, (st, has_ccol, dt) in enumerate(out):
        assert st != 0, "rank %d returned success although rank 0 could not allocate the result" % rank
        assert not has_ccol
        assert dt < 60.0, "rank %d took %.1f s: it waited for a collective nobody else entered" % (rank, dt)
    assert out[0][0] == 2, out      # BSPGEMM_ERR_ALLOC on the rank that failed; the others learn a peer failed


def test_comm_agree_host_transport():
    """bspgemm_comm_agree: every rank gets the worst status of all ranks"""
    import ctypes as C
    L = bspgemm.lib()
    L.bspgemm_comm_agree.argtypes = [C.c_void_p, C.c_int]
    world = 3
    make = _thread_transport(world)

    def body(rank):
        ctx = bspgemm.Context(0)
        t = make(rank)
        comm = C.c_void_p()
        assert L.bspgemm_comm_create_host(ctx._h, C.byref(t), rank, world, C.byref(comm)) == 0
        a = L.bspgemm_comm_agree(comm, 0)
        b = L.bspgemm_comm_agree(comm, 5 if rank == 1 else 0)
        L.bspgemm_comm_destroy(comm)
        ctx.close()
        return a, b
    for a, b in _run_ranks(world, body):
        assert a == 0 and b == 5


# ------------------------------------------------------------------------------------------------
# The RCCL failure path, on the one GPU there is (VERDICT r3 weak #9 / ADVICE r3 #1): a communicator whose
# collective timed out is ABORTED and marked dead; round 3 nulled its handle instead, and the next call took the
# host-transport branch through NULL callbacks (a crash).  bspgemm_comm_inject_failure(c, 2) = "the next bounded
# RCCL wait behaves as if it had run out".
def _rccl_comm_one_rank(ctx):
    import ctypes as C
    L = bspgemm.lib()
    L.bspgemm_comm_inject_failure.argtypes = [C.c_void_p, C.c_int]
    L.bspgemm_comm_inject_failure.restype = None
    L.bspgemm_comm_agree.argtypes = [C.c_void_p, C.c_int]
    uid = C.create_string_buffer(128)
    assert L.bspgemm_comm_unique_id(uid) == 0
    comm = C.c_void_p()
    assert L.bspgemm_comm_create(ctx._h, uid, 0, 1, C.byref(comm)) == 0, L.bspgemm_last_error()
    return comm


def test_rccl_timeout_kills_the_communicator_not_the_process():
    import ctypes as C
    from oracle import oracle as O
    L = bspgemm.lib()
    ctx = bspgemm.Context(0)
    rp, ci, n = bspgemm.gen_uniform(3000, 6, seed=23)
    erp, eci = O.spgemm(rp, ci, rp, ci, n)
    rp, ci = bspgemm._i32(rp), bspgemm._i32(ci)
    comm = _rccl_comm_one_rank(ctx)

    def multi():
        crow = np.zeros(n + 1, dtype=np.int32)
        cc = C.POINTER(C.c_int)()
        st = L.SpGEMM_hip_multi(comm, ci, rp, n, ci, rp, n, C.byref(cc), crow, 64)
        return st, crow, cc
    # healthy first: one rank over RCCL == the whole product
    st, crow, cc = multi()
    assert st == 0, L.bspgemm_last_error()
    assert np.array_equal(crow, erp) and np.array_equal(bspgemm._take_i32(cc, crow[-1]), eci)
    # the next bounded wait "times out": ERR_COMM, no result, no crash
    L.bspgemm_comm_inject_failure(comm, 2)
    st, crow, cc = multi()
    assert st == 8 and not cc, (st, L.bspgemm_last_error())               # BSPGEMM_ERR_COMM
    assert b"aborted" in L.bspgemm_last_error() or b"dead" in L.bspgemm_last_error()
    # ... and the communicator stays dead: every collective entry point refuses, nothing dereferences a NULL transport
    st, crow, cc = multi()
    assert st == 8 and not cc
    assert L.bspgemm_comm_agree(comm, 0) == 8
    A = ctx.upload(rp, ci, n)
    Cres = ctx.multiply(A, A)
    bounds = np.array([0, n], dtype=np.int32)
    dptr = C.c_void_p()
    shard = np.array([Cres.nnz], dtype=np.int64)
    assert L.bspgemm_comm_stitch_row_ptr(comm, Cres._h, bounds, C.byref(dptr), C.c_void_p(shard.ctypes.data)) == 8
    host = np.zeros(max(int(Cres.nnz), 1), dtype=np.int32)
    assert L.bspgemm_comm_gather_col_idx(comm, Cres._h, shard, 0, C.c_void_p(host.ctypes.data)) == 8
    L.bspgemm_comm_destroy(comm)                                           # destroying a dead communicator is fine
    # the context is unharmed: a fresh communicator on it works
    comm2 = _rccl_comm_one_rank(ctx)
    assert L.bspgemm_comm_agree(comm2, 0) == 0
    assert L.bspgemm_comm_agree(comm2, 5) == 5
    L.bspgemm_comm_destroy(comm2)
    ctx.close()


@pytest.mark.parametrize("what,transport", [(3, "rccl"), (4, "rccl"), (3, "host"), (4, "host-root-device")],
                         ids=["stitch_staging_rccl", "gather_buffer_rccl", "stitch_staging_host", "host_has_no_device_buffer"])
def test_rank_local_allocation_failures_are_agreed_on(what, transport):
    """Rank-local allocations in front of a collective (the stitch's staging buffers: inject 3; the root's device buffer
    of the gather: inject 4) are followed by bspgemm_comm_agree: the failing rank reports BSPGEMM_ERR_ALLOC, every other
    rank learns that a peer failed, nobody waits in a collective, and the communicator stays usable afterwards."""
    import ctypes as C
    import time
    from oracle import oracle as O
    L = bspgemm.lib()
    L.bspgemm_comm_inject_failure.argtypes = [C.c_void_p, C.c_int]
    L.bspgemm_comm_inject_failure.restype = None
    rp, ci, n = bspgemm.gen_uniform(4096, 8, seed=29)
    erp, eci = O.spgemm(rp, ci, rp, ci, n)
    rp, ci = bspgemm._i32(rp), bspgemm._i32(ci)
    if transport == "rccl":
        ctx = bspgemm.Context(0)
        comm = _rccl_comm_one_rank(ctx)
        L.bspgemm_comm_inject_failure(comm, what)
        crow = np.zeros(n + 1, dtype=np.int32)
        cc = C.POINTER(C.c_int)()
        st = L.SpGEMM_hip_multi(comm, ci, rp, n, ci, rp, n, C.byref(cc), crow, 64)
        assert st == 2 and not cc, (st, L.bspgemm_last_error())            # BSPGEMM_ERR_ALLOC
        st = L.SpGEMM_hip_multi(comm, ci, rp, n, ci, rp, n, C.byref(cc), crow, 64)   # the communicator is still alive
        assert st == 0, L.bspgemm_last_error()
        assert np.array_equal(crow, erp) and np.array_equal(bspgemm._take_i32(cc, crow[-1]), eci)
        L.bspgemm_comm_destroy(comm)
        ctx.close()
        return
    world = 3
    make = _thread_transport(world)

    def body(rank):
        ctx = bspgemm.Context(0)
        t = make(rank)
        comm = C.c_void_p()
        assert L.bspgemm_comm_create_host(ctx._h, C.byref(t), rank, world, C.byref(comm)) == 0
        if rank == 1 and what == 3:
            L.bspgemm_comm_inject_failure(comm, 3)
        crow = np.zeros(n + 1, dtype=np.int32)
        cc = C.POINTER(C.c_int)()
        t0 = time.time()
        st1 = L.SpGEMM_hip_multi(comm, ci, rp, n, ci, rp, n, C.byref(cc), crow, 64)
        dt = time.time() - t0
        got1 = bool(cc)
        if got1:
            bspgemm._libc.free(C.cast(cc, C.c_void_p))
        cc = C.POINTER(C.c_int)()
        st2 = L.SpGEMM_hip_multi(comm, ci, rp, n, ci, rp, n, C.byref(cc), crow, 64)
        ok2 = st2 == 0 and (rank != 0 or (np.array_equal(crow, erp) and np.array_equal(bspgemm._take_i32(cc, crow[-1]), eci)))
        L.bspgemm_comm_destroy(comm)
        ctx.close()
        return st1, got1, dt, ok2
    out = _run_ranks(world, body)
    for rank, (st1, got1, dt, ok2) in enumerate(out):
        assert dt < 60.0 and ok2, (rank, st1, dt, ok2)
        if what == 3:
            assert st1 != 0 and not got1, "rank %d succeeded although rank 1 could not stage the stitch" % rank
        else:                        # inject 4 concerns the RCCL root's device buffer only: the host transport has none
            assert st1 == 0
    if what == 3:
        assert out[1][0] == 2, out   # BSPGEMM_ERR_ALLOC on the rank that failed
