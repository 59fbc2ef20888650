"""world_size-2 test of the multi-GPU layer on CPU (gloo): equal-work shard cut + the row_ptr
all-gather/rebase of bspgemm/dist.py -- the code bench.py runs over RCCL for N > 1.
The per-shard product is stood in by the CPU oracle here (tests may use it; the shard multiply
itself is covered on the GPU by tests/test_gpu_parity.py::test_shards_concatenate_to_whole)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "binary-spgemm_amd")):
        sys.path.insert(0, p)
    import gen
    from oracle import oracle as O
    from bspgemm import dist as bdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rp, ci, n = gen.rmat(11, 8, (0.57, 0.19, 0.19, 0.05), seed=31)      # skewed on purpose
        # per-row work prefix (what bspgemm_row_work_prefix returns on the GPU)
        blen = np.diff(rp).astype(np.int64)
        F = np.add.reduceat(np.concatenate([blen[ci], [0]]), np.minimum(rp[:-1], ci.size))
        F[np.diff(rp) == 0] = 0
        prefix = np.concatenate([[0], np.cumsum(F)])
        bounds = bdist.shard_bounds(prefix, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        lrp, lci = O.spgemm_rows(rp, ci, rp, ci, n, r0, r1)                 # this rank's shard
        grp, shard_nnz = bdist.stitch_row_ptr(torch.from_numpy(lrp), bounds)
        wrp, wci = O.spgemm(rp, ci, rp, ci, n)                              # whole product
        ok = np.array_equal(grp.numpy(), wrp) and int(shard_nnz.sum()) == int(wrp[-1])
        ok = ok and np.array_equal(lci, wci[wrp[r0]:wrp[r1]])
        work = np.diff(prefix[bounds])
        ok = ok and work.max() <= 1.3 * work.mean()
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stitch_row_ptr_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert [out.get(r) for r in range(world)] == [True] * world


def test_shard_bounds_properties():
    sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
    from bspgemm import dist as bdist
    rng = np.random.default_rng(5)
    F = rng.pareto(1.5, size=10000).astype(np.int64) * 10
    prefix = np.concatenate([[0], np.cumsum(F)])
    for parts in (1, 2, 4, 8):
        b = bdist.shard_bounds(prefix, parts)
        assert b[0] == 0 and b[-1] == 10000 and np.all(np.diff(b) >= 0) and b.size == parts + 1
    z = bdist.shard_bounds(np.zeros(11, np.int64), 4)          # all-empty rows still split
    assert z[0] == 0 and z[-1] == 10 and np.all(np.diff(z) >= 0)
