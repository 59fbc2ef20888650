"""The C command-line drivers on a real GPU: same positionals, messages and CSV fields as the
reference's binaries (final/SpGEMM_mpi_omp.c:294-366, final/SpGEMM_mpi_omp_validity.c:308-375)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "binary-spgemm_amd")
MTX = os.path.join(ROOT, "tests", "golden", "validity_test.mtx")


def _run(args, **kw):
    return subprocess.run(args, capture_output=True, text=True, timeout=300, **kw)


def test_validity_driver_prints_the_reference_message():
    """`make test` of the reference: mpirun -n 4 SpGEMM_mpi_omp_validity validity_test.mtx 6250 2"""
    r = _run([os.path.join(PKG, "SpGEMM_hip_validity"), MTX, "6250", "2"])
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "Results of serial and multricore are the same!"


def test_bench_driver_csv_line():
    r = _run([os.path.join(PKG, "SpGEMM_hip"), MTX, "6250", "2", "5"])
    assert r.returncode == 0, r.stderr
    f = r.stdout.strip().split(",")
    # tasks,threads,tasks*threads,tBlock,path,n,nnz(A),nnz(C),mean,median,fastest + 5 extra columns
    assert len(f) == 16
    assert f[:4] == ["1", "2", "2", "6250"] and f[4] == MTX
    assert f[5:8] == ["50000", "25000", "12502"]
    assert all(float(x) > 0 for x in f[8:11]) and float(f[10]) <= float(f[9])


def test_bench_driver_usage_and_bad_file(tmp_path):
    r = _run([os.path.join(PKG, "SpGEMM_hip")])
    assert r.returncode == 1 and r.stdout.startswith("usage: mpirun  -n  numtasks  SpGEMM_mpi_omp")
    bad = tmp_path / "bad.mtx"
    bad.write_text("%MatrixMarket matrix coordinate pattern general\n1 1 1\n1 1\n")
    r = _run([os.path.join(PKG, "SpGEMM_hip"), str(bad), "1", "1", "1"])
    assert r.returncode == 1 and "Could not process Matrix Market banner." in r.stdout
    r = _run([os.path.join(PKG, "SpGEMM_hip"), str(tmp_path / "missing.mtx"), "1", "1", "1"])
    assert r.returncode == 1 and r.stdout == ""


def test_mpi_driver_single_rank():
    """the MPI+RCCL build, one rank (singleton init, no mpirun needed)"""
    exe = os.path.join(PKG, "SpGEMM_hip_mpi")
    if not os.path.exists(exe):
        pytest.skip("MPI not available at build time")
    r = _run([exe, MTX, "6250", "2", "3"])
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().split(",")[7] == "12502"
