"""CPU-side cover of the P > 1-on-one-GPU harness (tests/hipsim, tests/hipsim_rank.py,
tests/hipsim_util.py): the library builds and exports the C ABI, and the rank script — thread
world, process world, every case function — runs end to end over the host stand-in at small sizes,
so that what tests/test_gpu_multirank.py launches on the GPU box has itself been executed."""
import ctypes as C
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = os.path.join(HERE, "hipsim_rank.py")


def test_hipsim_library_builds_and_exports_the_abi():
    import hipsim_util
    path = hipsim_util.build()
    lib = C.CDLL(path)      # loading is not computing: no HIP call happens here
    lib.ppals_version.restype = C.c_char_p
    v = lib.ppals_version()
    assert b"hipsim" in v and b"TEST INFRASTRUCTURE" in v
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "pairwise-perturbation_amd"))
    import ppals
    assert not [s for s in ppals.EXPORTS if not hasattr(lib, s)]
    # without a device the context creation fails with an error code, like the product
    lib.ppals_last_error.restype = C.c_char_p
    h = C.c_void_p()
    if not os.path.exists("/dev/kfd"):   # (not through torch: its HIP runtime must not be mixed in)
        assert lib.ppals_ctx_create(C.byref(h), 0) < 0


def test_product_binding_refuses_test_libraries(monkeypatch):
    """ppals/__init__.py never drives a TEST INFRASTRUCTURE library under the product's name"""
    import hipsim_util
    code = ("import sys; sys.path.insert(0, %r); import ppals\n"
            "try:\n    ppals.lib(%r)\n    print('LOADED')\n"
            "except ppals.PpalsError as e:\n    print('REFUSED', e)\n"
            % (os.path.join(os.path.dirname(HERE), "pairwise-perturbation_amd"), hipsim_util.build()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "REFUSED" in out.stdout and "LOADED" not in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("case,world", [("cp_small", 3), ("tucker_small", 2), ("cfg4_small", 8)])
def test_thread_world_script_on_host_stand_in(case, world):
    env = dict(os.environ, OMP_NUM_THREADS="2", PPALS_ORACLE_THREADS="2")
    out = subprocess.run([sys.executable, SCRIPT, case, "--threads", str(world), "--backend", "hostsim"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert f"all {world} ranks: OK" in out.stdout


def test_thread_world_reports_a_failing_rank():
    """a rank that raises aborts the barrier: the others return instead of waiting forever"""
    import hipsim_util
    import threading
    w = hipsim_util.ThreadWorld(3, timeout=30)
    seen = []

    def body(r):
        try:
            if r == 1:
                raise ValueError("boom")
            w.barrier()
        except ValueError:
            w.abort()
            seen.append("raised")
        except threading.BrokenBarrierError:
            seen.append("broken")

    ths = [threading.Thread(target=body, args=(r,)) for r in range(3)]
    [t.start() for t in ths]
    [t.join(60) for t in ths]
    assert sorted(seen) == ["broken", "broken", "raised"]


def test_process_world_script_on_host_stand_in():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", PPALS_ORACLE_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, SCRIPT, "cp_small", "--backend", "hostsim"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and f"rank {r}: OK" in out, out[-3000:]
