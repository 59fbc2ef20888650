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
