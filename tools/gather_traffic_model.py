#!/usr/bin/env python3
"""What the B.col_idx gather of a product really moves, and what rocprofv3's FETCH_SIZE says about it (gfx950).

Calibration (profiles/r04_fetch_calibration_unaligned.txt): the L2 asks the fabric for 64-byte sectors; two sectors of one
128-byte line that are wanted together are ONE request, tallied as 64 bytes although 128 move.  So for a gather of whole B rows
    bytes really moved  = 64 * (distinct 64-byte sectors the row touches)
    bytes FETCH_SIZE shows = 64 * (distinct 128-byte lines the row touches)
both summed over every read of a B row (one per A-nonzero), before any L2 hit.  The ratio is the correction factor of the gather
part of a kernel's FETCH_SIZE for THIS matrix; it follows from row_ptr alone.
usage: python tools/gather_traffic_model.py [rmat SCALE | uniform LOG2N | powerlaw LOG2N | g500 SCALE]   (prints one JSON object)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
import bspgemm  # noqa: E402


def model(rp, ci):
    rp = np.asarray(rp, dtype=np.int64)
    length = np.diff(rp)
    reads = np.bincount(np.asarray(ci), minlength=length.size).astype(np.int64)      # how often row j is gathered
    live = length > 0
    first, last = rp[:-1][live], rp[1:][live] - 1
    sectors = last // 16 - first // 16 + 1
    lines = last // 32 - first // 32 + 1
    r = reads[live]
    alg = 4 * int((r * length[live]).sum())
    true = 64 * int((r * sectors).sum())
    tallied = 64 * int((r * lines).sum())
    aligned = 64 * int((r * ((length[live] + 15) // 16)).sum())                      # every row padded to a 64-byte boundary
    return {"products": alg // 4, "gather_bytes_algorithmic": alg, "gather_bytes_moved": true, "gather_bytes_fetch_size_shows": tallied,
            "moved_over_algorithmic": round(true / alg, 3), "correction_factor_moved_over_shown": round(true / tallied, 3),
            "gather_bytes_moved_if_rows_were_64B_aligned": aligned, "aligned_over_algorithmic": round(aligned / alg, 3)}


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "rmat"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 22
    if kind == "rmat":
        rp, ci, n = bspgemm.gen_rmat(k, 16, (0.30, 0.25, 0.25), seed=1)
    elif kind == "g500":
        rp, ci, n = bspgemm.gen_rmat(k, 16, (0.57, 0.19, 0.19), seed=1)
    elif kind == "powerlaw":
        rp, ci, n = bspgemm.gen_powerlaw(1 << k, 64, seed=1)
    else:
        rp, ci, n = bspgemm.gen_uniform(1 << k, 16, seed=1)
    out = model(rp, ci)
    out["workload"] = "%s %d" % (kind, k)
    print(json.dumps(out))
