"""GPU run of tests/padded_cases.py: the scans' compacting epilogues (k_scan_suffix_buf / _fast /
generic, k_slab_reduce), Ops::pad_layout (k_transpose_pad, pitched copy) against the oracle."""
import pytest

from padded_cases import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    import ppals
    return ppals


@pytest.fixture(scope="module")
def ctx(pp):
    c = pp.Context(0)
    yield c
    c.close()
