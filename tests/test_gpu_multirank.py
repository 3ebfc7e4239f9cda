"""The HIP kernels at P > 1 on ONE GPU (script/script_strongscaling.py:10,45-46).

No multi-GPU node exists on this pool, so the sharded code paths of engine.cpp / tucker.cpp (row
offsets of real shards, partial sums from different ranks, the blocked update over P gathered
blocks, the Tucker `dist_` branches) meet the real kernels here: P ranks — each a fresh child
process with its own HIP context, or, for world 8 (a box allows 6 processes on its card), 8 threads
of one fresh child — share the one device through tests/hipsim (the product's engine + C ABI + HIP
kernels, a staged callback communicator instead of RCCL). Every result is compared with the
UNSHARDED oracle. tests/hipsim_rank.py holds the cases; timing is ignored throughout."""
import os
import signal
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = os.path.join(HERE, "hipsim_rank.py")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module", autouse=True)
def hipsim_lib():
    import hipsim_util
    hipsim_util.build()   # once, before the ranks race for it (a no-op when the tree shipped it)


def _keep(name, text):
    """PPALS_MULTIRANK_LOG=dir: the ranks' own output is kept there (evidence for profiles/)"""
    d = os.environ.get("PPALS_MULTIRANK_LOG")
    if d:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".log"), "w") as f:
            f.write(text)


def _reap(procs, timeout):
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:   # exactly the process groups this test started
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
            raise
        outs.append(out)
    return outs


def run_processes(case, world, timeout=900, extra_env=None):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", PPALS_ORACLE_THREADS="2")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, SCRIPT, case], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True, start_new_session=True))
    outs = _reap(procs, timeout)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {world} ({case}) failed:\n{out[-4000:]}"
        assert f"rank {r}: OK" in out
    _keep(f"{case}_procs{world}", "\n".join(f"--- rank {r}\n{o}" for r, o in enumerate(outs)))
    return outs


def run_threads(case, world, timeout=1100, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="4", PPALS_ORACLE_THREADS="4")
    env.update(extra_env or {})
    p = subprocess.Popen([sys.executable, SCRIPT, case, "--threads", str(world)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         start_new_session=True)
    out = _reap([p], timeout)[0]
    assert p.returncode == 0, f"{case} with {world} thread ranks failed:\n{out[-6000:]}"
    assert f"all {world} ranks: OK" in out
    _keep(f"{case}_threads{world}", out)
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_hostsim_cases_on_hip_kernels(world):
    """the CPU rehearsal's own cases (tests/hostsim_rank.py: both plans alternating, schedule traces,
    PP, -magni, -pp 2, Tucker hosvd / DT / PP) — now over the HIP kernels, separate processes"""
    run_processes("tiny", world)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_cp_dt_pp_sharded_processes(world):
    """CP DT + PP (+ -pp 2) on mid-size shapes, both collective plans, both schedules; world 3
    and 4 leave the last rank a shorter shard on most shapes (unequal shards)"""
    run_processes("cp_mid", world)


@pytest.mark.parametrize("world", [2, 4])
def test_tucker_sharded_processes(world):
    """Tucker HOSVD + alsTucker_DT + TTMc + alsTucker_PP with modes above 64 rows"""
    run_processes("tucker_mid", world)


def test_rs_plan_unequal_shards_processes():
    """reduce-scatter plan for every mode with a short last rank (hostsim_rank.rs_unequal_cases)"""
    run_processes("tiny_rs_unequal", 3)


def test_padded_layouts_sharded_processes():
    """the padded resident layouts forced on: their leading block holds the LOCAL rows"""
    run_processes("cp_plans", 2, extra_env={"PPALS_PAD_LAYOUT": "1"})


@pytest.mark.parametrize("case", ["cp_plans", "cp_mid", "tucker_mid"])
def test_world8_thread_ranks(case):
    """world 8 — BASELINE configs[3]'s world size — as 8 threads of one child"""
    run_threads(case, 8)


def test_cfg4_full_size_world8():
    """BASELINE configs[3] at FULL size with P = 8 on one MI355X (8 x (12.8 GB shard + its second
    layout)): MTTKRPs and exact sweeps on both collective plans against the closed form"""
    out = run_threads("cfg4", 8)
    assert "cfg4 plan small_bytes=0" in out


@pytest.mark.parametrize("world", [2, 8])
def test_cfg5_full_size_sharded(world):
    """BASELINE configs[4] (Tucker s = 400, core 20^3) sharded over P ranks on one device"""
    run_threads("cfg5", world)
