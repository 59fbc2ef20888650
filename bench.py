#!/usr/bin/env python3
"""bench.py -- headline benchmark: boolean SpGEMM C = A*A, output nonzeros per second.

A "step" is one full pass of the hot path over one synthetic matrix that is already resident
in HBM: bspgemm_multiply on this rank's A-row shard, plus, for N > 1, the all-gather that stitches
C.row_ptr (the job of SpGEMM_mpi, reference final/SpGEMM_mpi_omp.c:155-225).  The library's DEFAULT
flow is timed (`roofline.flow` names it): products per row -> capacity classes -> rows accumulated and
placed by their product count (an upper bound) -> counts scanned into C.row_ptr -> rows squeezed into
C.col_idx (k_compact).  The north star's symbolic -> scan -> numeric order is the "exact" flow
(BSPGEMM_FLOW=exact: 6.9 ms per step on this workload against 6.3; round 3 also had a single-pass row-order
"fused" flow, removed in round 4 -- DESIGN.md section 2 has all three side by side).  The timed region
mirrors the reference's (:320-324): inputs resident, result allocation included, no file I/O --
"allocation" here is a hit in the context's cache of freed results (config.allocation).

Workloads (BASELINE.json configs, R-MAT edge factor 16, (a,b,c,d) = (0.30,0.25,0.25,0.20), seed 1,
SURVEY.md 8d/9.2):
    --gpus 1          configs[2]: scale 22 -- the config the north-star target is quoted on
    --gpus N (N > 1)  configs[3]: scale 24 row-sharded over the N GPUs, B replicated; the total
                      work is fixed, so "scaling": "strong"
    --gpus N --weak   scale 22 + log2 N: rows, nonzeros and products double with N, the work per
                      GPU is fixed ("scaling": "weak")
Rows are cut into N contiguous shards of equal work; B = A is replicated on every GPU.

    python bench.py --gpus N --steps K --warmup W          (spawns its N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "binary-spgemm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
RMAT_MILD = (0.30, 0.25, 0.25)  # d = 0.20
MAX_STAT_STEPS = 16             # event sets the library keeps (bspgemm_stats_at)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="rmat", choices=["rmat", "rmat-g500", "uniform", "powerlaw"])
    ap.add_argument("--scale", type=int, default=0)
    ap.add_argument("--weak", action="store_true", help="N > 1: scale 22 + log2 N instead of BASELINE cfg4 (scale 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes (this
    process has not touched the GPU and never will), relay rank 0's JSON line, return the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 16) // args.gpus))))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll ALL children: a rank that dies leaves the others in an RCCL init or barrier for ever, and a
    # launcher blocked on rank 0's pipe would hang with them until the driver's limit
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("BSPGEMM_BENCH_DEADLINE_S", "3000"))
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.time() > deadline:
            failed = "rank %s exited with %s" % (bad[0], codes[bad[0]]) if bad else "deadline passed"
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    reader.join(10)
    codes = [p.wait() for p in procs]
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if failed:
        print("[bench] %s: the other ranks were stopped" % failed, file=sys.stderr, flush=True)
        return 1
    return max(abs(c) for c in codes)


def pick_scale(args, world):
    """(scale, scaling) of the R-MAT workload -- see the module docstring"""
    if args.scale:
        return args.scale, ("weak" if args.weak else "strong")
    if world == 1:
        return 22, "weak"                       # one GPU: both readings coincide
    if args.weak:
        return 22 + int(round(math.log2(world))), "weak"
    return 24, "strong"


def make_matrix(args, world):
    import bspgemm
    scaling = "strong"                           # a fixed matrix whatever N is ...
    if args.workload == "rmat":
        scale, scaling = pick_scale(args, world)  # ... except the weak-scaling R-MAT series
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
    return rp, ci, n, name, scaling


def bin_of(F, caps):
    """capacity class of a row with F products -- same rule as csrc/prepass.hip bin_of(); `caps` is
    bspgemm_stats.bin_cap: caps[0] = 0 (empty rows), caps[-1] = INT32_MAX (heavy rows)"""
    import numpy as np
    b = np.zeros(F.shape, dtype=np.int64)
    b[F > 0] = 1
    for k in range(1, len(caps) - 1):
        b[F > caps[k]] = k + 1
    return b


def cpu_baseline(rp, ci, n, budget_s=6.0, gpu_ctx=None):
    """Reference CPU path -- the reference's own SpGEMM_omp compiled into oracle/_ref (kind
    "reference"; the in-repo port when that is absent) -- on bounded row samples of the same matrix,
    with the OpenMP team set explicitly (omp_set_num_threads) to 1, C/2 and C threads."""
    import ctypes
    from oracle import oracle as O
    R = O.reference()
    kind = "reference" if R is not None else "port"
    gomp = ctypes.CDLL("libgomp.so.1", mode=ctypes.RTLD_GLOBAL)     # the runtime oracle/_ref links: one per process
    host_cpus = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0))
    C = min(int(os.environ.get("BSPGEMM_BENCH_CPU_CORES", "16")), usable)   # one GPU's CPU share on the box

    def run(threads, row0, rows):
        unit = threads * 8                  # decomposition stays divisible (reference README.md:14)
        rows -= rows % unit
        if rows <= 0:
            return None
        tblock = rows // unit
        gomp.omp_set_num_threads(int(threads))
        team = int(gomp.omp_get_max_threads())
        t = time.perf_counter()
        if R is not None:
            crow, _ = R.omp(rp, ci, rp, ci, n, tblock, row0=row0, rows=rows)
        else:
            crow, _ = O.spgemm_omp(rp, ci, rp, ci, n, tblock, threads, row0=row0, rows=rows)
        dt = time.perf_counter() - t
        return {"threads": int(threads), "omp_get_max_threads": team, "rows": int(rows), "row0": int(row0),
                "tBlock": int(tblock), "nnz": int(crow[-1]), "seconds": round(dt, 3),
                "GNZ/s": round(int(crow[-1]) / dt / 1e9, 5)}

    # Sample sizing: the reference pays per SLICE for a Bm-int result buffer and a Bm-byte flag array
    # (final/SpGEMM_mpi_omp.c:21,88-92), so tiny tBlock would time its allocator, not its kernel: every
    # sample keeps tBlock >= 4096 rows (8 slices per thread) and is grown once towards
    # `budget_s` seconds, inside the int32 nnz limit of the reference.
    row0 = n // 2
    sweep = []
    for t in sorted({1, max(C // 2, 1), C}):
        rows = min(t * 8 * 4096, n - row0)
        r = run(t, row0, rows)
        if r is None:
            continue
        grow = min(budget_s / max(r["seconds"], 1e-3), 16.0)
        per_row = max(r["nnz"] / r["rows"], 1.0)
        more = int(min(n - row0, r["rows"] * grow, 1.5e9 / per_row))
        if more >= 1.5 * r["rows"]:
            r2 = run(t, row0, more)
            if r2 is not None:
                r = r2
        sweep.append(r)
    if not sweep:
        return None
    best = max(sweep, key=lambda r: r["GNZ/s"])
    out = {"value": best["GNZ/s"], "unit": "GNZ/s", "cores": best["omp_get_max_threads"], "kind": kind,
           "layout": "OpenMP only (SpGEMM_omp through ctypes), %d threads" % best["threads"],
           "sample": "rows [%d,%d) of the same matrix (%d output nonzeros), SpGEMM_omp, omp_set_num_threads(%d), "
                     "tBlock=%d, %.2f s" % (best["row0"], best["row0"] + best["rows"], best["nnz"], best["threads"],
                                            best["tBlock"], best["seconds"]),
           "thread_sweep": sweep, "host_cpus": host_cpus, "usable_cpus": usable,
           "cpu_share": "%d cores = one GPU's share of this host; BSPGEMM_BENCH_CPU_CORES raises it" % C}
    # ... and the reference's MPI + OpenMP program itself, the layout its report found fastest on a node (pure MPI,
    # SURVEY.md 6 Fig. 8).  It runs on an R-MAT scale-20 FILE (the reference's int32 counters and its loader bound the
    # size), i.e. on ANOTHER matrix than `value` above and than the GPU headline: GNZ/s depends on the matrix, so the
    # figure is kept in a field of its own and never folded into `value`; the GPU is timed on that same
    # scale-20 matrix beside it (`gpu_same_matrix`) so that the ratio there is like for like.
    try:
        mpi = cpu_baseline_mpirun(C, gpu_ctx) if kind == "reference" else None
    except Exception as e:
        mpi = {"error": repr(e)}
    out["mpirun_other_workload"] = mpi
    return out


def cpu_baseline_mpirun(budget_cores, gpu_ctx=None):
    """The reference BINARY as its README runs it -- `mpirun -n P SpGEMM_mpi_omp file tBlock T times`
    (final/SpGEMM_mpi_omp.c:294-366, README.md:12-21) -- built untouched into oracle/_ref, on an R-MAT scale-20
    file (the bench matrix's generator, a quarter of its rows: the reference's int32 counters and its loader,
    which every rank runs on the whole file, bound the size) written by bspgemm_write_mtx.  Layouts P x T with
    P * T = the CPU share of one GPU; median of 3 runs each, from the program's own CSV line."""
    import shutil
    import bspgemm
    exe = os.path.join(ROOT, "oracle", "_ref", "SpGEMM_mpi_omp")
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    if not (os.path.exists(exe) and os.path.exists(mpirun)):
        return None
    scale = 20
    n = 1 << scale
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "bspgemm_bench_rmat%d.mtx" % scale)
    t0 = time.perf_counter()
    rp, ci, _ = bspgemm.gen_rmat(scale, 16, RMAT_MILD, seed=1)
    bspgemm.write_mtx(path, rp, ci, n)
    t_write = time.perf_counter() - t0
    layouts = [(p, budget_cores // p) for p in (1, 4, 16, budget_cores) if p <= budget_cores and budget_cores % p == 0]
    layouts = sorted(set(layouts))
    runs = []
    for P, T in layouts:
        tblock = n // (P * T * 8)                      # 8 slices per thread, divisible (README.md:14-17)
        if tblock < 1:
            continue
        env = dict(os.environ, OMP_NUM_THREADS=str(T), PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
        cmd = [mpirun, "-n", str(P), exe, path, str(tblock), str(T), "3"]
        t = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            runs.append({"P": P, "T": T, "tBlock": tblock, "error": "timeout"})
            continue
        wall = time.perf_counter() - t
        line = [x for x in r.stdout.strip().splitlines() if x.count(",") == 10]
        if r.returncode != 0 or not line:
            runs.append({"P": P, "T": T, "tBlock": tblock, "error": (r.stderr or r.stdout)[-200:]})
            continue
        f = line[-1].split(",")
        cnnz, median, fastest = int(f[7]), float(f[9]), float(f[10])
        runs.append({"P": P, "T": T, "cores": P * T, "tBlock": tblock, "Cnnz": cnnz, "median_s": median, "fastest_s": fastest,
                     "GNZ/s": round(cnnz / median / 1e9, 5), "wall_s_with_loading": round(wall, 1)})
    try:
        os.remove(path)
    except OSError:
        pass
    ok = [r for r in runs if "GNZ/s" in r]
    if not ok:
        return {"runs": runs}
    best = max(ok, key=lambda r: r["GNZ/s"])
    res = {"best": best, "runs": runs, "workload": "R-MAT scale %d, edge factor 16, (0.30,0.25,0.25,0.20), seed 1, A*A -- NOT the headline's "
                                                   "scale-22 matrix" % scale,
           "matrix": "R-MAT scale %d (same generator and parameters), %d entries, file written in %.1f s" % (scale, int(rp[-1]), t_write),
           "command": "mpirun -n P oracle/_ref/SpGEMM_mpi_omp <file> tBlock T 3"}
    if gpu_ctx is not None:
        # the GPU on the SAME scale-20 matrix, operands resident, 10 multiplies after 3 warm-ups
        try:
            A = gpu_ctx.upload(rp, ci, n)
            for _ in range(3):
                gpu_ctx.multiply(A, A).free()
            t = time.perf_counter()
            nnz = 0
            for _ in range(10):
                Cg = gpu_ctx.multiply(A, A)
                nnz = Cg.nnz
                Cg.free()
            dt = (time.perf_counter() - t) / 10
            A.free()
            res["gpu_same_matrix"] = {"ms_per_multiply": round(dt * 1e3, 4), "GNZ/s": round(nnz / dt / 1e9, 2), "nnz_c": int(nnz),
                                      "ratio_to_best_mpirun_layout": round(nnz / dt / 1e9 / best["GNZ/s"], 1)}
        except Exception as e:
            res["gpu_same_matrix"] = {"error": repr(e)}
    return res


def pmc_profile(wname, world):
    """the committed rocprofv3 --pmc result of this very command (tools/pmc_run.sh -> profiles/*_pmc_traffic.json)"""
    import glob
    if world != 1:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            prof = json.load(open(path))
        except Exception:
            continue
        if prof.get("workload") == wname:
            return prof, os.path.relpath(path, ROOT)
    return None, None


def main():
    args = parse_args()
    if (args.gpus > 1 or os.environ.get("BSPGEMM_BENCH_FORCE_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))            # before anything in this process touches the GPU

    # Only the JSON line may reach stdout: RCCL prints a version banner there when the first
    # communicator is created.  Everything else is sent to stderr for the whole run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # torch.distributed.run exports OMP_NUM_THREADS=1 for every rank; the host-side generators
    # (OpenMP, csr_gen.c) would then build the matrix on one core.  Give each rank its share.
    if world > 1 and os.environ.get("OMP_NUM_THREADS", "1") == "1":
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, (os.cpu_count() or 16) // world)))

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see bspgemm/__init__.py: RCCL's streams must not crowd ours
    import numpy as np
    import torch
    import torch.distributed as dist
    import bspgemm
    from bspgemm import dist as bdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # BSPGEMM_BENCH_SHARED_GPU=1: a FUNCTIONAL rehearsal of the N-rank path on a one-GPU box -- every rank
    # uses GPU 0 and the collective runs over gloo (RCCL refuses two ranks on one device).  Its numbers
    # say nothing about scaling; the JSON line is labelled.
    shared_gpu = os.environ.get("BSPGEMM_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BSPGEMM_BENCH_FORCE_DIST") == "1"   # 1-rank rehearsal of the N>1 path
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    t0 = time.perf_counter()
    rp, ci, n, wname, scaling = make_matrix(args, world)
    log(rank, "generated %s: n=%d nnz(A)=%d in %.1f s; scaling=%s" % (wname, n, rp[-1], time.perf_counter() - t0, scaling))

    ctx = bspgemm.Context(local_rank)
    A = ctx.upload(rp, ci, n)                       # B = A, replicated on every GPU
    prefix = ctx.row_work_prefix(A, A)
    bounds = bdist.shard_bounds(prefix, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    log(rank, "F=%d products; shard bounds %s" % (prefix[-1], bounds.tolist()))

    stitch_done = []                                # events: stitch k has finished reading C(k).row_ptr

    def step():
        C = ctx.multiply(A, A, r0, r1)
        if use_dist and os.environ.get("BSPGEMM_BENCH_NO_STITCH") != "1":   # (switch for overhead hunting)
            local_rp = bdist.device_tensor(C.row_ptr_device, C.rows + 1, torch.int64, dev)
            bdist.stitch_row_ptr(local_rp, bounds, detach=False, ctx=ctx)   # asynchronous on torch's stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            stitch_done.append(ev)
        return C

    def retire(C):
        """free a result once the stitch that reads its row_ptr has finished (an event, not timing luck)"""
        if stitch_done:
            stitch_done.pop(0).synchronize()
        C.free()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up and timed steps run the SAME rotation (two results alive: the stitch of step k-1 runs under
    # step k), so that every buffer the steady state uses exists before the clock starts: a warm-up that
    # freed each result at once left the second C.col_idx to be hipMalloc'ed inside the timed region
    # (0.5 s for the 21.8 GB of scale 24 on one GPU).  With --warmup 1 one more untimed step is run for the
    # same reason and reported as `priming_steps`; --warmup 0 times the cold start as asked.
    last = older = None

    def rotate():
        nonlocal last, older
        if older is not None:
            retire(older)
        older = last
        last = step()

    priming = 1 if args.warmup == 1 else 0
    for _ in range(args.warmup + priming):
        rotate()
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        rotate()
    fence()
    elapsed = time.perf_counter() - t_start
    # per-step HIP-event brackets recorded on the library's streams DURING the timed region
    hist = [ctx.stats(age) for age in range(min(args.steps, MAX_STAT_STEPS))]
    st = hist[0]
    # the per-class brackets are OFF in the timed steps (an event pair around each of the ~17 class launches costs
    # about 1 %): two extra, untimed steps with them on itemise the classes (roofline.instances)
    ctx.set_class_timing(True)
    for _ in range(2):
        rotate()
    fence()
    itemised = [ctx.stats(age) for age in range(2)]
    ctx.set_class_timing(False)
    if older is not None:
        retire(older)
    keys = ("ms_total", "ms_symbolic", "ms_prepass", "ms_count", "ms_numeric", "ms_stitch")
    phase_ms = {k: float(np.mean([h[k] for h in hist])) for k in keys}
    bin_ms = np.mean([h["ms_bin"] for h in itemised], axis=0)
    bin_count_ms = np.mean([h["ms_bin_count"] for h in itemised], axis=0)
    if use_dist:
        rdev = torch.device("cpu") if shared_gpu else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([last.nnz, st["products"], st["bytes_alg"], st["bytes_read_alg"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        nnz_total, products_total, bytes_total, read_total = (int(x) for x in tot.tolist())
    else:
        nnz_total, products_total, bytes_total, read_total = (int(last.nnz), int(st["products"]), int(st["bytes_alg"]),
                                                              int(st["bytes_read_alg"]))
    ms_per_step = elapsed / args.steps * 1e3

    prof, prof_path = pmc_profile(wname, world)
    try:
        # ---- roofline of the dominant kernel (this rank's launches) -------------------------
        # The dominant kernel is k_wave_rows (csrc/wave_rows.inc), the numeric pass: ONE kernel source,
        # launched as one template instance per capacity class (16 of them), back to back on two
        # streams so that one instance's tail overlaps the next one's ramp-up.  Its "launch" is the
        # whole family: algorithmic bytes of all one-wave rows (4 B/product + 4 B/output
        # nonzero + 12 B/A-nonzero + 12 B/row) over the HIP-event time from the first instance's start
        # to the last one's end on the multiply's stream (bspgemm_stats.ms_numeric).  Per-instance
        # event brackets are listed next to it; they overlap pairwise, so their sum exceeds the
        # family's time.  `symbolic` prices the count kernels (the COUNT instances of the same source) the
        # same way with what they move: 4 B/product + 8 B/A-nonzero + 4 B/row.
        crp, _ = last.download(col_idx=False)
        F_row = np.diff(prefix)[r0:r1]
        a_row = np.diff(rp.astype(np.int64))[r0:r1]
        c_row = np.diff(crp)
        caps = st["bin_cap"]
        DENSE_BIN = 1 + sum(1 for c in caps[1:] if c <= 2048)    # first class that is not one-wave-per-row
        bins = bin_of(F_row, caps)
        levels = next((L for L in range(1, 5) if n <= (256 << (5 * L))), 5)      # csrc/kernels.hpp levels_for_cols
        if levels == 4 and n <= (512 << 15):
            levels = 3                                                           # ... wave_levels_for_cols
        wave = (bins >= 1) & (bins < DENSE_BIN)
        read_wave = int(4 * F_row[wave].sum() + 12 * a_row[wave].sum() + 4 * wave.sum())
        bytes_wave = read_wave + int(4 * c_row[wave].sum() + 8 * wave.sum())
        ms_wave = phase_ms["ms_numeric"]
        achieved = bytes_wave / (ms_wave * 1e-3) / 1e9 if ms_wave > 0 else 0.0
        instances = []
        for b in range(1, DENSE_BIN):
            selb = bins == b
            if selb.any():
                by = int(4 * F_row[selb].sum() + 4 * c_row[selb].sum() + 12 * a_row[selb].sum() + 12 * selb.sum())
                ms_b = float(bin_ms[b])
                instances.append({"chunks": caps[b] // 64, "rows": int(selb.sum()), "products": int(F_row[selb].sum()),
                                  "bytes": by, "ms": round(ms_b, 4),          # ms: event bracket = rocprofv3 average
                                  "count_ms": round(float(bin_count_ms[b]), 4),
                                  "GBps_while_sharing_the_gpu": round(by / (ms_b * 1e-3) / 1e9, 1) if ms_b > 0 else 0.0})
        traffic, traffic_src = None, None
        if prof:
            fam = [v for k, v in prof.get("kernels", {}).items() if k.startswith("bsp::k_wave_rows<%d," % levels) and not k.rstrip().endswith("true>")]
            if len(fam) == len(instances):
                # FETCH_SIZE under-counts wide coalesced reads 2x on gfx950 (MI355X_MICROARCH.md, HBM): the
                # gathers of this kernel are 4-byte accesses, so the raw figure is quoted, uncorrected
                if all("fetch_bytes_calibrated" in k for k in fam):
                    # round 4: FETCH_SIZE calibrated on this access (profiles/r04_fetch_calibration_unaligned.txt) and applied to the
                    # matrix's own row_ptr (tools/gather_traffic_model.py): a request is a 128-byte line's wanted 64-byte sectors,
                    # tallied as 64 bytes -- the factor says what the gather of THIS matrix moves per byte FETCH_SIZE shows
                    traffic = int(sum(k["fetch_bytes_calibrated"] + k["write_bytes"] for k in fam))
                    traffic_src = prof_path + (" (WRITE_SIZE + FETCH_SIZE x %.3f of the %d instances: the counter calibrated on "
                                               "64-80-byte rows at dword alignment and applied to this matrix's row_ptr; uncorrected "
                                               "FETCH_SIZE + WRITE_SIZE = %d)" % (prof.get("gather_calibration_factor") or 0.0, len(fam),
                                                                                  int(sum(k["fetch_bytes"] + k["write_bytes"] for k in fam))))
                else:
                    traffic = int(sum(k.get("fetch_bytes_corrected_high", 2 * k["fetch_bytes"]) + k["write_bytes"] for k in fam))
                    traffic_src = prof_path + (" (WRITE_SIZE + 2 x FETCH_SIZE of the %d instances: gfx950 tallies 128-byte read "
                                               "requests at 64 bytes; the gathers' 64-byte requests make this the HIGH bound, "
                                               "low bound %d)" % (len(fam), int(sum(k.get("fetch_bytes_corrected_low", k["fetch_bytes"])
                                                                                  + k["write_bytes"] for k in fam))))
        sym_bytes = int(4 * F_row[wave].sum() + 8 * a_row[wave].sum() + 4 * wave.sum())
        ms_cnt = phase_ms["ms_count"]
        flow = "exact" if float(np.sum(bin_count_ms)) > 0 else "upper-bound"
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "k_wave_rows<LEVELS=%d,*> (numeric pass: %d capacity-class instances, two streams)" % (levels, len(instances)),
                    "bytes_per_launch": bytes_wave, "ms_per_launch": round(ms_wave, 4),
                    "read_bytes": read_wave,
                    "read_frac": round(read_wave / (ms_wave * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ms_wave > 0 else None,
                    "launch_rows": int(wave.sum()), "launch_products": int(F_row[wave].sum()),
                    "steps_averaged": len(hist), "instances": instances, "flow": flow}
        if flow == "exact":
            roofline["symbolic"] = {"kernel": "k_wave_rows<LEVELS=%d,*,COUNT> (symbolic pass: exact row sizes, the rank bitmap without its emit half)" % levels,
                                    "bytes_per_launch": sym_bytes, "ms_per_launch": round(ms_cnt, 4),
                                    "achieved": round(sym_bytes / (ms_cnt * 1e-3) / 1e9, 1) if ms_cnt > 0 else None,
                                    "frac": round(sym_bytes / (ms_cnt * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ms_cnt > 0 else None}
        else:
            by = int(8 * c_row.sum())                    # the compaction: 4 B read + 4 B written per output nonzero
            ms_st = phase_ms["ms_stitch"]
            roofline["compaction"] = {"kernel": "k_compact (upper-bound placed rows -> C.col_idx) + count scan",
                                      "bytes_per_launch": by, "ms_per_launch": round(ms_st, 4),
                                      "achieved": round(by / (ms_st * 1e-3) / 1e9, 1) if ms_st > 0 else None,
                                      "frac": round(by / (ms_st * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ms_st > 0 else None}
    except Exception as e:   # the roofline is a reported extra: never lose the metric line to it
        roofline = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
                    "traffic": None, "error": "%s: %s" % (type(e).__name__, e)}
    value = nnz_total * args.steps / elapsed / 1e9
    step_traffic = (prof.get("step_total_bytes_calibrated") or prof.get("step_total_bytes")) if prof else None
    out = {
        "metric": "output nnz/sec (GNZ/s)", "value": round(value, 4), "unit": "GNZ/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "int32", "data": "synthetic" if not shared_gpu else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
        "config": {"workload": wname, "n": int(n), "nnz_a": int(rp[-1]), "products": products_total,
                   "nnz_c": nnz_total, "parallelism": "row-shards x%d cut at equal work, B replicated" % world,
                   "shard_rows": [int(b) for b in bounds.tolist()], "priming_steps": priming,
                   "flow": os.environ.get("BSPGEMM_FLOW", "auto"),
                   "allocation": "cached: C.row_ptr/C.col_idx come from the context's cache of freed results "
                                 "(two results rotate), workspaces persist; a cold first step allocates (hipMalloc)"},
        "roofline": roofline,
        "whole_job": {"bytes_alg": bytes_total, "bytes_read_alg": read_total,
                      "alg_GBps": round(bytes_total / (ms_per_step * 1e-3) / 1e9, 1),
                      "alg_frac_of_hbm_peak": round(bytes_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS / world, 4),
                      "read_frac_of_hbm_peak": round(read_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS / world, 4),
                      "pmc_step_traffic_bytes": step_traffic,
                      "traffic_ratio": round(step_traffic / bytes_total, 3) if step_traffic else None,
                      "traffic_source": prof_path if step_traffic else None,
                      "rank0_ms": {"total": round(phase_ms["ms_total"], 4), "symbolic": round(phase_ms["ms_symbolic"], 4),
                                   "prepass": round(phase_ms["ms_prepass"], 4), "count": round(phase_ms["ms_count"], 4),
                                   "numeric": round(phase_ms["ms_numeric"], 4), "stitch": round(phase_ms["ms_stitch"], 4)},
                      "rank0_ms_per_bin": [round(float(x), 4) for x in bin_ms],
                      "rank0_count_ms_per_bin": [round(float(x), 4) for x in bin_count_ms],
                      "rank0_rows_per_bin": [int(x) for x in st["rows_per_bin"]],
                      "bin_cap": [int(x) for x in st["bin_cap"]]},
    }
    if use_dist:
        # what the collective layer itself saw: the first 8-GPU run shows "RCCL saw N ranks" from its own output
        seen = torch.ones(1, dtype=torch.int64, device=torch.device("cpu") if shared_gpu else dev)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        out["comm"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": int(seen.item()),
                       "stitch": "all_gather_into_tensor of int32 row lengths (bspgemm/dist.py) + bspgemm_lengths_to_row_ptr"}
    last.free()
    if rank == 0 and world == 1 and nnz_total <= 2**31 - 1 and os.environ.get("BSPGEMM_BENCH_NO_DROPIN") != "1":
        # The call the reference's driver makes (final/SpGEMM_mpi_omp.c:322-327): host int32 arrays in, a malloc'ed
        # Ccol out, PCIe both ways.  Never the headline: reported beside it.
        try:
            import ctypes
            libc = ctypes.CDLL(None)
            libc.free.argtypes = [ctypes.c_void_p]
            L = bspgemm.lib()
            rp32, ci32 = bspgemm._i32(rp), bspgemm._i32(ci)
            crow = np.zeros(n + 1, dtype=np.int32)
            times = []
            for _ in range(3):
                cc = ctypes.POINTER(ctypes.c_int)()
                t = time.perf_counter()
                rc = L.SpGEMM_hip(ci32, ctypes.c_void_p(rp32.ctypes.data), n, ci32, rp32, n, ctypes.byref(cc), crow, 0)
                times.append(time.perf_counter() - t)
                libc.free(ctypes.cast(cc, ctypes.c_void_p))
                if rc != 0:
                    raise RuntimeError("SpGEMM_hip returned %d" % rc)
            best = min(times[1:])
            out["dropin_e2e"] = {"ms": round(best * 1e3, 1), "GNZ/s": round(int(crow[-1]) / best / 1e9, 3), "nnz_c": int(crow[-1]),
                                 "what": "SpGEMM_hip(Acol,Arow,An,Acol,Arow,An,&Ccol,Crow,tBlock): upload A (B = A is a view), "
                                         "multiply, download into the malloc'ed result pinned piecewise under the DMA; "
                                         "best of 2 after one warm-up call"}
        except Exception as e:
            out["dropin_e2e"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(rp, ci, n, gpu_ctx=ctx)
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
