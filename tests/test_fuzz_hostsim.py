"""The seeded campaign of tests/test_gpu_fuzz_campaign.py on the host stand-in (engine host logic:
schedules, root sets, padded layouts, PP restart logic, Tucker thin route)."""
import pytest

import hostsim_util
import test_gpu_fuzz_campaign as F

test_cp_sweeps = F.test_cp_sweeps
test_cp_pp_driver = F.test_cp_pp_driver
test_tucker_sweeps = F.test_tucker_sweeps


@pytest.fixture(scope="module")
def pp():
    return hostsim_util.load()


@pytest.fixture(scope="module")
def ctx(pp):
    c = pp.Context(0)
    yield c
    c.close()
