import os
import sys

import pytest
import torch  # noqa: F401  -- first: torch bundles its own HIP runtime; loading it before libbspgemm.so
#                              keeps ONE runtime in the process (whichever libamdhip64 is loaded first wins)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "binary-spgemm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
