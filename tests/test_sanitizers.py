"""The engine's host logic (engine.cpp, tucker.cpp, ppals_api.cpp over the host stand-in ops) under
AddressSanitizer + UndefinedBehaviorSanitizer: `make -C tests/hostsim asan`, then a cross-section
of the hostsim suites in a child interpreter with the sanitizer runtime preloaded. GPU sanitizers
are not available on the pool: this is the CPU build only."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SELECT = ("dt_sweeps_match or driver_pp_matches or normalize_and_owed or pp_operator_after or "
          "tucker_pp_driver or hosvd_and_dt or schedule_switch or tensor_refill or "
          "context_destroyed or driver_pp_partupdate or placement_measurement")


def test_hostsim_suites_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostsim"), "asan"])
    lib = os.path.join(HERE, "hostsim", "build_asan", "libppals_hostsim.so")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, PPALS_HOSTSIM_LIB=lib,
               LD_PRELOAD=":".join(p for p in (asan, ubsan) if os.path.isabs(p)),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98",
               PPALS_ORACLE_THREADS="2", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(HERE, "test_engine_hostsim.py"),
                        os.path.join(HERE, "test_forced_comm_hostsim.py"), "-k",
                        SELECT + " or forced"],
                       env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout
