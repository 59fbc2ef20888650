"""Multi-GPU layer: one process per GPU, B replicated, contiguous A-row shards cut at equal work,
C.row_ptr stitched with an all-gather -- the job of SpGEMM_mpi (reference
final/SpGEMM_mpi_omp.c:155-225), whose MPI_Reduce / MPI_Gather / MPI_Gatherv + serial rebase
(:178-223) become one collective on the shards' int32 row lengths plus a device-side scan.  col_idx
stays sharded on the GPUs (the reference ships it to rank 0 with MPI_Gatherv, :203).

The collective runs through torch.distributed: backend "nccl" is RCCL over xGMI on the GPU
box; backend "gloo" runs the same code on CPU tensors (tests/test_dist_gloo.py, world_size 2).
Nothing here computes a product: the shard multiply is bspgemm.Context.multiply (HIP).
"""
import numpy as np
import torch
import torch.distributed as dist

PER_ROW_COST = 32   # products-equivalent charged per row when cutting shards (matches api.hip)


def shard_bounds(work_prefix, parts, per_row=PER_ROW_COST):
    """Cut rows [0,R) into `parts` contiguous shards of equal cost = products + per_row*rows.

    Same rule as bspgemm_partition_rows (csrc/api.hip).  The reference cuts An/numtasks equal
    row counts (final/SpGEMM_mpi_omp.c:165), which is up to 3x unbalanced on skewed inputs
    (SURVEY.md 8e).
    """
    work_prefix = np.asarray(work_prefix, dtype=np.int64)
    R = work_prefix.size - 1
    cost = work_prefix + per_row * np.arange(R + 1, dtype=np.int64)
    total = int(cost[R])
    bounds = np.zeros(parts + 1, dtype=np.int32)
    for p in range(1, parts):
        target = total // parts * p
        bounds[p] = int(np.searchsorted(cost, target, side="left"))
    bounds[parts] = R
    return np.maximum.accumulate(bounds).astype(np.int32)


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_tensor(ptr, n, dtype, device):
    """torch tensor aliasing `n` elements at device pointer `ptr` (no copy)."""
    typestr = {torch.int64: "<i8", torch.int32: "<i4"}[dtype]
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def stitch_row_ptr(local_row_ptr, bounds, group=None, detach=True, ctx=None):
    """All-gather the shards' row lengths and rebuild the global C.row_ptr on every rank.

    local_row_ptr : int64 tensor [rows_r + 1], slice-local (starts at 0) -- on the GPU for
                    backend nccl, on the CPU for gloo.
    bounds        : int array [world+1], the shard row bounds every rank used.
    ctx           : bspgemm.Context of this GPU: the gathered lengths are then scanned into the
                    row_ptr by the library's own kernels (bspgemm_lengths_to_row_ptr) on torch's
                    stream; without it (CPU tensors, gloo) the same arithmetic runs as torch ops.
    detach        : wait (host side) until the send buffer has been built from `local_row_ptr`, so
                    that the caller may release or overwrite it while the collective is in flight
                    (the collective itself stays asynchronous on torch's stream).
    Returns (global_row_ptr int64 [R+1] on the same device, shard_nnz int64 [world] on it too).
    One collective.  What travels is the row LENGTH as int32 (|C_i| < 2^31 always; slice-local
    offsets are not bounded like that): 4 B per row instead of the 8 B of an int64 row_ptr --
    at 8 GPUs and 33.5 M rows that is 134 MB instead of 268 MB into every rank.  Shards are padded
    to the longest (equal-work cuts make them near-equal); every rank then scans the lengths.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = np.asarray(bounds, dtype=np.int64)
    rows = bounds[1:] - bounds[:-1]
    R = int(bounds[-1])
    assert local_row_ptr.numel() == rows[rank] + 1, "local row_ptr does not match bounds[rank]"
    width = max(int(rows.max()), 1)
    dev = local_row_ptr.device
    send = torch.empty(width, dtype=torch.int32, device=dev)    # the pad slots are never read
    send[: rows[rank]] = (local_row_ptr[1:] - local_row_ptr[:-1]).to(torch.int32)
    if detach and local_row_ptr.is_cuda:
        torch.cuda.current_stream(dev).synchronize()
    recv = torch.empty(world * width, dtype=torch.int32, device=dev)
    if local_row_ptr.is_cuda and dist.get_backend(group) == "nccl" and hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(recv, send, group=group)           # RCCL: one flat collective
    else:
        _all_gather_list(recv, send, world, group)                     # gloo (CPU tensors, or a shared-GPU rehearsal)
    if ctx is not None and local_row_ptr.is_cuda:
        out = torch.empty(R + 1, dtype=torch.int64, device=dev)
        ctx.lengths_to_row_ptr(recv.data_ptr(), world, width, bounds, out.data_ptr(),
                               torch.cuda.current_stream(dev).cuda_stream)
    else:
        recv = recv.view(world, width)
        out = torch.zeros(R + 1, dtype=torch.int64, device=dev)
        for r in range(world):
            out[1 + int(bounds[r]): 1 + int(bounds[r + 1])] = recv[r, : int(rows[r])]
        out = torch.cumsum(out, 0)
    edges = torch.as_tensor(bounds, dtype=torch.int64, device=dev)
    shard_nnz = out[edges[1:]] - out[edges[:-1]]
    return out, shard_nnz


def _all_gather_list(recv, send, world, group):
    if send.is_cuda:                      # gloo has no all_gather on GPU tensors: stage through the host
        host = send.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        recv.copy_(torch.cat(parts).to(recv.device))
        return
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    recv.copy_(torch.cat(parts))
