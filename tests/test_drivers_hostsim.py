"""CPU runs of the DRIVER tests (tests/test_gpu_driver.py): the product's bin/test_ALS, bin/pp_bench
and bin/run sources linked against the host stand-in (tests/hostsim) — flag parsing, echo block,
CSV, the V/W file exchange and the `-tensor o*` path, pp_bench's line format, against the oracle.
They say nothing about the HIP kernels; `-m gpu` runs the same tests on the real binaries."""
import os
import subprocess

import pytest

import test_gpu_driver as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def BIN():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
    return os.path.join(ROOT, "tests", "hostsim", "build")


@pytest.fixture(scope="module")
def full_size():
    return False


test_test_ALS_matches_oracle = D.test_test_ALS_matches_oracle
test_cli_defaults_and_silent_resets = D.test_cli_defaults_and_silent_resets
test_pp_bench_lines = D.test_pp_bench_lines
test_every_tensor_source_and_pp_mode_matches_oracle = D.test_every_tensor_source_and_pp_mode_matches_oracle
test_run_driver_class_api = D.test_run_driver_class_api
test_run_driver_low_rank_optimizers = D.test_run_driver_low_rank_optimizers
test_cfg1_cli_matches_oracle = D.test_cfg1_cli_matches_oracle
test_file_exchange_and_o_path = D.test_file_exchange_and_o_path
test_o_path_rejects_short_file = D.test_o_path_rejects_short_file
test_pp_bench_tucker_lines = D.test_pp_bench_tucker_lines

# the plug-in point for reference-made fixtures (tests/test_ctf_fixtures.py), on the host stand-in
import test_ctf_fixtures as CF  # noqa: E402


@pytest.mark.parametrize("d", CF._dirs("ctf_selftest"), ids=os.path.basename)
def test_selftest_fixture(BIN, d, tmp_path):
    CF.check_fixture(BIN, d, tmp_path)


@pytest.mark.parametrize("d", CF._dirs("ctf") or [None], ids=lambda d: os.path.basename(d) if d else "none")
def test_ctf_made_fixture(BIN, d, tmp_path):
    if d is None:
        pytest.skip("no reference-made fixture under tests/golden/ctf/ (tools/make_ctf_fixture.md)")
    CF.check_fixture(BIN, d, tmp_path)
