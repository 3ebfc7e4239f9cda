"""CPU run of tests/padded_cases.py over the host stand-in ops (engine host logic only)."""
import pytest

import hostsim_util
from padded_cases import *  # noqa: F401,F403


@pytest.fixture(scope="module")
def pp():
    return hostsim_util.load()


@pytest.fixture(scope="module")
def ctx(pp):
    c = pp.Context(0)
    yield c
    c.close()
