"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/bspgemm.h declares, and the product path FAILS LOUDLY without a GPU (no CPU fallback)."""
import os
import re

import pytest

import bspgemm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "bspgemm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(bspgemm_[a-zA-Z0-9_]+|SpGEMM_hip[a-zA-Z0-9_]*)\s*\(", text)
    return sorted(set(names))


def test_library_loads_and_exports_every_declared_symbol():
    L = bspgemm.lib()
    declared = _declared_functions()
    assert len(declared) >= 35
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, "declared in include/bspgemm.h but not exported: %s" % missing
    assert sorted(bspgemm.EXPORTS) == declared, "python binding list out of sync with the header"


def test_header_cites_the_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "bspgemm.h")).read()
    for cite in ("final/SpGEMM_mpi_omp.c:71-74", "final/SpGEMM_mpi_omp.c:15-18", "final/utils.c:47-81",
                 "final/SpGEMM_mpi_omp_validity.c:290-302", "final/SpGEMM_mpi_omp.c:155-225",
                 "Matlab/inc/BSpGEMM.h:2-4", "final/SpGEMM_mpi_omp.c:232-235"):
        assert cite in text, cite


def test_header_compiles_as_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "bspgemm.h"\nint main(void){bspgemm_stats s; (void)s; return BSPGEMM_OK;}\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src,
                        "-o", os.path.join(d, "t.o")], check=True)


def test_product_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(bspgemm.BspgemmError) as e:
        bspgemm.Context(0)
    assert e.value.status == 4          # BSPGEMM_ERR_NO_DEVICE
    import numpy as np
    rp = np.array([0, 1, 2], np.int32)
    ci = np.array([0, 1], np.int32)
    with pytest.raises(bspgemm.BspgemmError):
        bspgemm.SpGEMM_hip(ci, rp, 2, ci, rp, 2)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "binary-spgemm_amd")
    for base, _dirs, files in os.walk(pkg):
        if "build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hpp", ".hip", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % os.path.join(base, f)
    assert "oracle" not in open(os.path.join(ROOT, "include", "bspgemm.h")).read().lower()


def test_one_hip_runtime_per_process():
    """bspgemm first, torch second, in a fresh process: exactly one libamdhip64 may be mapped
    (round 1's segfault in the RCCL stitch test was two of them: torch's bundled runtime and
    /opt/rocm's behind libbspgemm.so's rpath -- INTEGRATION.md section 2); and the guard refuses a
    process that already holds two."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import bspgemm; bspgemm.lib()\n"
            "import torch\n"
            "libs = bspgemm.check_single_hip_runtime()\n"
            "assert len(libs) == 1, libs\n"
            "print('one runtime:', libs[0])\n") % os.path.join(ROOT, "binary-spgemm_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "one runtime:" in r.stdout
    bad = ("import ctypes, sys; sys.path.insert(0, %r)\n"
           "ctypes.CDLL(%r)\n"                      # the library before torch: /opt/rocm's runtime comes with it
           "import torch\n"
           "import bspgemm\n"
           "try:\n"
           "    bspgemm.check_single_hip_runtime()\n"
           "except RuntimeError as e:\n"
           "    print('refused:', e)\n") % (os.path.join(ROOT, "binary-spgemm_amd"),
                                            os.path.join(ROOT, "binary-spgemm_amd", "libbspgemm.so"))
    r = subprocess.run([sys.executable, "-c", bad], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "refused: two HIP runtimes" in r.stdout, r.stdout + r.stderr


def test_shipped_library_has_no_ablation_paths():
    """the shipped .so was one -DBSP_ABLATE away from a silent wrong answer.  The timing-only
    ablation branches are gone from the kernel sources, the tuning constants are not overridable, and the
    library says so (bspgemm_build_info)."""
    import glob
    info = bspgemm.lib().bspgemm_build_info().decode()
    assert "BSP_ABLATE=0" in info and "gfx950" in info, info
    src = glob.glob(os.path.join(ROOT, "binary-spgemm_amd", "csrc", "*"))
    assert len(src) > 10
    for path in src:
        text = open(path).read()
        for word in ("BSP_ABLATE", "BSP_DENSE_ABLATE", "BSP_DENSE_NOEMIT", "BSP_COMPACT_NOSLOW"):
            if os.path.basename(path) == "context.hip" and word == "BSP_ABLATE":
                continue                                   # the text of bspgemm_build_info
            assert word not in text, "%s still mentions %s" % (path, word)
        assert "#ifndef BSP_" not in text, "%s has a -D overridable kernel switch" % path
    mk = open(os.path.join(ROOT, "binary-spgemm_amd", "Makefile")).read()
    assert "ABLATE" not in mk and "XDEF" not in mk
