import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """Tucker cases whose modes exceed the in-LDS eigensolver use rocSOLVER: its libraries must
    enter the process before the HIP runtime is initialised (milliseconds then, minutes later —
    include/ppals.h, ppals_preload_eigensolver), i.e. before the first test creates a context."""
    if "not gpu" in (session.config.getoption("-m") or ""):
        return
    try:
        import ppals
        ppals.preload_eigensolver()
    except Exception as e:  # no product library / no ROCm: the gpu tests will say so themselves
        print(f"[conftest] eigensolver preload skipped: {e}")


def bar_log(test, **kw):
    """PPALS_BAR_LOG=file: one JSON line per asserted fp32 bar with what was measured (how DESIGN.md §5's
    table of measured bars per input class is made: tools/runs/r05_c.sh)"""
    path = os.environ.get("PPALS_BAR_LOG")
    if not path:
        return
    import json
    with open(path, "a") as f:
        f.write(json.dumps(dict(test=test, **{k: (float(v) if hasattr(v, "__float__") else v)
                                              for k, v in kw.items()})) + "\n")
