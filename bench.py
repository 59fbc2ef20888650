#!/usr/bin/env python3
"""bench.py -- headline benchmark: boolean SpGEMM C = A*A, output nonzeros per second.

A "step" is one full pass of the hot path over one synthetic matrix that is already resident
in HBM: bspgemm_multiply (row work -> scan/bin -> accumulate+emit -> scan -> compaction) on this
rank's A-row shard, plus, for N > 1, the all-gather that stitches C.row_ptr (the job of
SpGEMM_mpi, reference final/SpGEMM_mpi_omp.c:155-225).  The timed region mirrors the
reference's (:320-324): inputs resident, result allocation included, no file I/O.

Workload at N = 1: BASELINE.json configs[2], the config the north-star target is quoted on --
R-MAT scale 22, edge factor 16, (a,b,c,d) = (0.30,0.25,0.25,0.20), A*A.
For N GPUs the scale is 22 + log2(N) (rows, nonzeros and products all double per step, so the
work per GPU is fixed: weak scaling); rows are cut into N contiguous shards of equal work, B = A
is replicated on every GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "binary-spgemm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
RMAT_MILD = (0.30, 0.25, 0.25)  # d = 0.20


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def make_matrix(args, world):
    import bspgemm
    if args.workload == "rmat":
        scale = args.scale if args.scale else 22 + int(round(math.log2(world)))
        rp, ci, n = bspgemm.gen_rmat(scale, 16, RMAT_MILD, seed=1)
        name = "R-MAT scale %d, edge factor 16, (a,b,c,d)=(0.30,0.25,0.25,0.20), seed 1, A*A" % scale
    elif args.workload == "rmat-g500":
        scale = args.scale or 18
        rp, ci, n = bspgemm.gen_rmat(scale, 16, (0.57, 0.19, 0.19), seed=1)
        name = "R-MAT scale %d, edge factor 16, Graph500 skew (0.57,0.19,0.19,0.05), seed 1, A*A" % scale
    elif args.workload == "uniform":
        scale = args.scale or 18
        rp, ci, n = bspgemm.gen_uniform(1 << scale, 16, seed=1)
        name = "uniform n=2^%d, 16 nnz/row, seed 1, A*A" % scale
    elif args.workload == "powerlaw":
        scale = args.scale or 20
        rp, ci, n = bspgemm.gen_powerlaw(1 << scale, 64, seed=1)
        name = "power-law n=2^%d, mean degree 64 (Pareto 2.1, clip n/16), seed 1, A*A" % scale
    else:
        raise SystemExit("unknown workload " + args.workload)
    return rp, ci, n, name


def bin_of(F, caps):
    """capacity class of a row with F products -- same rule as csrc/prepass.hip bin_of(); `caps` is
    bspgemm_stats.bin_cap: caps[0] = 0 (empty rows), caps[-1] = INT32_MAX (heavy rows)"""
    b = np.zeros(F.shape, dtype=np.int64)
    b[F > 0] = 1
    for k in range(1, len(caps) - 1):
        b[F > caps[k]] = k + 1
    return b


def cpu_baseline(rp, ci, n, budget_s=12.0):
    """Reference CPU path (oracle/_ref SpGEMM_omp, else the in-repo port) on a bounded row sample."""
    from oracle import oracle as O
    cores = min(16, len(os.sched_getaffinity(0)))     # one GPU's CPU share on the box
    R = O.reference()
    kind = "reference" if R is not None else "port"

    def run(row0, rows, tblock):
        t = time.perf_counter()
        if R is not None:
            crow, ccol = R.omp(rp, ci, rp, ci, n, tblock, row0=row0, rows=rows)
        else:
            crow, ccol = O.spgemm_omp(rp, ci, rp, ci, n, tblock, cores, row0=row0, rows=rows)
        return time.perf_counter() - t, int(crow[-1])

    row0 = n // 2
    unit = cores * 8                       # keep the decomposition divisible (reference README.md:14)
    rows = min(unit * 64, n - row0)
    rows -= rows % unit
    if rows <= 0:
        return None
    dt, nnz = run(row0, rows, max(rows // unit, 1))
    for _ in range(3):                     # grow the sample until it is ~budget_s of CPU work
        if dt >= 0.6 * budget_s or row0 + rows >= n:
            break
        per_row = max(nnz / rows, 1.0)
        grow = min(budget_s / max(dt, 1e-3), 16.0)
        new_rows = int(min(n - row0, rows * grow, 1.5e9 / per_row))   # int32 nnz of the reference
        new_rows -= new_rows % unit
        if new_rows <= rows:
            break
        rows = new_rows
        dt, nnz = run(row0, rows, rows // unit)
    tblock = rows // unit
    return {"value": round(nnz / dt / 1e9, 5), "unit": "GNZ/s", "cores": cores, "kind": kind,
            "sample": "rows [%d,%d) of the same matrix (%d output nonzeros), SpGEMM_omp with %d OpenMP "
                      "threads, tBlock=%d, %.2f s" % (row0, row0 + rows, nnz, cores, tblock, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="rmat", choices=["rmat", "rmat-g500", "uniform", "powerlaw"])
    ap.add_argument("--scale", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # Only the JSON line may reach stdout: RCCL prints a version banner there when the first
    # communicator is created.  Everything else is sent to stderr for the whole run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))

    # torch.distributed.run exports OMP_NUM_THREADS=1 for every rank; the host-side generators
    # (OpenMP, csr_gen.c) would then build the scale-25 matrix on one core.  Give each rank its share.
    if world > 1 and os.environ.get("OMP_NUM_THREADS", "1") == "1":
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, (os.cpu_count() or 16) // world)))

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see bspgemm/__init__.py: RCCL's streams must not crowd ours
    import torch
    import torch.distributed as dist
    import bspgemm
    from bspgemm import dist as bdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BSPGEMM_BENCH_FORCE_DIST") == "1"   # 1-rank rehearsal of the N>1 path
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    t0 = time.perf_counter()
    rp, ci, n, wname = make_matrix(args, world)
    log(rank, "generated %s: n=%d nnz(A)=%d in %.1f s" % (wname, n, rp[-1], time.perf_counter() - t0))

    ctx = bspgemm.Context(local_rank)
    A = ctx.upload(rp, ci, n)                       # B = A, replicated on every GPU
    prefix = ctx.row_work_prefix(A, A)
    bounds = bdist.shard_bounds(prefix, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    log(rank, "F=%d products; shard bounds %s" % (prefix[-1], bounds.tolist()))

    def step():
        C = ctx.multiply(A, A, r0, r1)
        if use_dist and os.environ.get("BSPGEMM_BENCH_NO_STITCH") != "1":   # (switch for overhead hunting)
            local_rp = bdist.device_tensor(C.row_ptr_device, C.rows + 1, torch.int64, dev)
            bdist.stitch_row_ptr(local_rp, bounds, detach=False, ctx=ctx)   # C stays alive for two steps
        return C

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step().free()
    bin_ms = None
    phase_ms = np.zeros(4)
    fence()
    t_start = time.perf_counter()
    last = older = None
    for _ in range(args.steps):
        # two results stay alive: the stitch of step k-1 (asynchronous, torch's stream) may still be
        # reading C(k-1).row_ptr when step k starts; it has long finished when step k+1 reuses the buffer
        if older is not None:
            older.free()
        older = last
        last = step()
        st = ctx.stats()
        bin_ms = np.array(st["ms_bin"]) if bin_ms is None else bin_ms + np.array(st["ms_bin"])
        phase_ms += np.array([st["ms_total"], st["ms_symbolic"], st["ms_numeric"], st["ms_stitch"]])
    fence()
    elapsed = time.perf_counter() - t_start
    if older is not None:
        older.free()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([last.nnz, st["products"], st["bytes_alg"]], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        nnz_total, products_total, bytes_total = (int(x) for x in tot.tolist())
    else:
        nnz_total, products_total, bytes_total = int(last.nnz), int(st["products"]), int(st["bytes_alg"])
    bin_ms /= args.steps
    phase_ms /= args.steps

    try:
        # ---- roofline of the dominant kernel (this rank's launches) -------------------------
        # The dominant kernel is k_wave_rows (csrc/wave_rows.inc): ONE kernel source, launched as one
        # template instance per capacity class (16 of them), back to back on two streams so that one
        # instance's tail overlaps the next one's ramp-up.  Its "launch" is therefore the whole family:
        # algorithmic bytes of all one-wave rows (4 B/product + 4 B/output nonzero +
        # 12 B/A-nonzero + 12 B/row) over the HIP-event time from the first instance's start to the
        # last one's end on the multiply's stream (bspgemm_stats.ms_numeric, which also holds the few
        # heavy rows of k_dense_rows and the count scan: conservative).  Per-instance event brackets are
        # listed next to it; they overlap pairwise, so their sum exceeds the family's time.
        crp, _ = last.download(col_idx=False)
        F_row = np.diff(prefix)[r0:r1]
        a_row = np.diff(rp.astype(np.int64))[r0:r1]
        c_row = np.diff(crp)
        caps = st["bin_cap"]
        DENSE_BIN = len(caps) - 1
        bins = bin_of(F_row, caps)
        tiles = max(int(st.get("tiles", 1)), 1)            # each class is launched once per row super-tile
        levels = next((L for L in range(1, 5) if n <= (256 << (5 * L))), 5)      # csrc/kernels.hpp levels_for_cols
        if levels == 4 and n <= (512 << 15):
            levels = 3                                                           # ... wave_levels_for_cols
        wave = (bins >= 1) & (bins < DENSE_BIN)
        bytes_wave = int(4 * F_row[wave].sum() + 4 * c_row[wave].sum() + 12 * a_row[wave].sum() + 12 * wave.sum())
        ms_wave = float(phase_ms[2])
        achieved = bytes_wave / (ms_wave * 1e-3) / 1e9 if ms_wave > 0 else 0.0
        instances = []
        for b in range(1, DENSE_BIN):
            selb = bins == b
            if selb.any():
                by = int(4 * F_row[selb].sum() + 4 * c_row[selb].sum() + 12 * a_row[selb].sum() + 12 * selb.sum())
                ms_b = float(bin_ms[b]) / tiles
                instances.append({"chunks": caps[b] // 64, "rows": int(selb.sum()), "products": int(F_row[selb].sum()),
                                  "bytes": by, "ms": round(ms_b, 4),          # ms: event bracket = rocprofv3 average
                                  "GBps_while_sharing_the_gpu": round(by / (ms_b * 1e-3) / 1e9, 1) if ms_b > 0 else 0.0})
        # HBM traffic: PMC counters cannot be read from inside the process, so the committed
        # rocprofv3 --pmc result of this very command is quoted when the workload is the profiled one
        # (tools/pmc_run.sh -> profiles/*_pmc_traffic.json: FETCH_SIZE + WRITE_SIZE summed over the
        # family's instances); otherwise null.
        traffic, traffic_src = None, None
        try:
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
                prof = json.load(open(path))
                fam = [v for k, v in prof.get("kernels", {}).items() if k.startswith("bsp::k_wave_rows<%d," % levels)]
                if prof.get("workload") == wname and world == 1 and tiles == 1 and len(fam) == len(instances):
                    traffic = int(sum(k["fetch_bytes"] + k["write_bytes"] for k in fam))
                    traffic_src = os.path.relpath(path, ROOT) + " (FETCH_SIZE+WRITE_SIZE of the %d instances, uncorrected)" % len(fam)
                    break
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "k_wave_rows<LEVELS=%d,*> (%d capacity-class instances, two streams)" % (levels, len(instances)),
                    "bytes_per_launch": bytes_wave, "ms_per_launch": round(ms_wave, 4),
                    "launch_rows": int(wave.sum()), "launch_products": int(F_row[wave].sum()),
                    "instances": instances}

    except Exception as e:   # the roofline is a reported extra: never lose the metric line to it
        roofline = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
                    "traffic": None, "error": "%s: %s" % (type(e).__name__, e)}
        caps = list(st.get("bin_cap", []))
    ms_per_step = elapsed / args.steps * 1e3
    value = nnz_total * args.steps / elapsed / 1e9
    out = {
        "metric": "output nnz/sec (GNZ/s)", "value": round(value, 4), "unit": "GNZ/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": wname, "n": int(n), "nnz_a": int(rp[-1]), "products": products_total,
                   "nnz_c": nnz_total, "parallelism": "row-shards x%d cut at equal work, B replicated" % world,
                   "shard_rows": [int(b) for b in bounds.tolist()]},
        "roofline": roofline,
        "whole_job": {"bytes_alg": bytes_total, "alg_GBps": round(bytes_total / (ms_per_step * 1e-3) / 1e9, 1),
                      "alg_frac_of_hbm_peak": round(bytes_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS / world, 4),
                      "rank0_ms": {"total": round(float(phase_ms[0]), 4), "symbolic": round(float(phase_ms[1]), 4),
                                   "numeric": round(float(phase_ms[2]), 4), "stitch": round(float(phase_ms[3]), 4)},
                      "rank0_ms_per_bin": [round(float(x), 4) for x in bin_ms],
                      "rank0_rows_per_bin": [int(x) for x in st["rows_per_bin"]],
                      "bin_cap": [int(x) for x in st["bin_cap"]]},
    }
    last.free()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(rp, ci, n)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line to it
            out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
