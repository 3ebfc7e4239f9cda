"""bench.py's own N > 1 code path on the CPU: the process group, the communicator-id hand-over,
init_comm, measure() with its barrier / max-over-ranks, and the reduce-scatter sub-record — two ranks
over gloo and the engine's host stand-in (PPALS_BENCH_BACKEND=hostsim). What the driver's 8-GPU run
executes for the first time on hardware is executed here first (TEST INFRASTRUCTURE: the numbers
mean nothing)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_multi_rank_path_on_the_host_stand_in(world):
    import hostsim_util
    hostsim_util.load()  # build once, before the ranks race for it
    port = free_port()
    procs = []
    workload = "cp4_s12_r3" if world < 8 else "cp4_s16_r3"   # (8 ranks: 2 rows of the leading mode each)
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2" if world < 8 else "1",
                   PPALS_BENCH_BACKEND="hostsim")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3",
             "--warmup", "1", "--workload", workload, "--dtype", "f64", "--no-cpu-baseline"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-2000:]}\n{err[-3000:]}"
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    assert not any(ln.startswith("{") for o, _ in outs[1:] for ln in o.splitlines())  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["rccl_ranks"] == world and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "strong" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "REHEARSAL" in d["data"]
    assert f"x{world}" in d["config"]["sharding"]
    rs = d["sub_records"]["reduce_scatter_plan"]
    assert rs["value"] > 0 and "reduce-scatter" in rs["comm_plan"]
    # the N > 1 line carries BASELINE configs[3] and the N-GPU leg of configs[4] (shrunken here)
    c4 = d["sub_records"]["cfg4_sharded"]
    assert "error" not in c4, c4
    assert c4["rccl_ranks"] == world
    for plan in ("allreduce_plan", "reduce_scatter_plan"):
        assert c4[plan]["steps"] == 5 and c4[plan]["value"] > 0 and c4[plan]["ms_per_step"] > 0
    g = [c4[p_]["final_gradnorm"] for p_ in ("allreduce_plan", "reduce_scatter_plan")]
    assert abs(g[0] - g[1]) < 1e-8 * abs(g[0])       # same iterates on both collective plans
    c5 = d["sub_records"]["cfg5_tucker_sharded"]
    assert "error" not in c5, c5
    assert c5["rccl_ranks"] == world and c5["hosvd_ms"] > 0 and c5["ms_per_hooi_sweep"] > 0
    # the same ALS iterates on every shard plan: the run ends at the same gradient norm as one rank
    env = dict(os.environ, PPALS_BENCH_BACKEND="hostsim", OMP_NUM_THREADS="2")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                          "--warmup", "1", "--workload", workload, "--dtype", "f64",
                          "--no-cpu-baseline", "--no-config-records"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert abs(d1["final_gradnorm"] - d["final_gradnorm"]) < 1e-8 * d1["final_gradnorm"]
    assert abs(d1["final_rel_residual"] - d["final_rel_residual"]) < 1e-9
