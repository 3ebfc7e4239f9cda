"""RCCL plumbing on the GPU box (one rank): see tests/rccl_single_rank.py. The two-and-more-rank
logic is covered on the CPU by tests/test_multirank_gloo.py; real multi-GPU runs are the driver's."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["PPALS_FORCE_COMM"] = "1"
    return env


def test_sharded_paths_through_one_rank_rccl():
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_single_rank.py")], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "RCCL single-rank rehearsal: OK" in r.stdout


def test_bench_under_torchrun_one_rank():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, nccl backend,
    unique-id broadcast, ppals_ctx_init_comm), with one rank and the sharded paths forced."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--workload", "cp4_s64_r10", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import json
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["launches"] > 0
    assert out["config"].get("comm", "").startswith("rccl")
