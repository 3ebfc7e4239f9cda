"""world_size-2 (and 3, 4: a last rank with fewer rows) rehearsal of the sharded ALS engine on CPU: torch.distributed/gloo behind
the communicator callbacks of the host-stand-in build (tests/hostsim). Covers the N>1 shard plan:
leading-mode block partition, reduce-scatter of partial MTTKRP rows, all-gather of updated rows."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,padded", [(2, False), (3, False), (4, False), (2, True)])
def test_sharded_engine_matches_oracle(world, padded):
    """padded: the same run with the padded resident layouts forced on (their leading block holds
    the LOCAL rows of the partitioned mode)"""
    import hostsim_util
    hostsim_util.load()  # build once, before the ranks race for it
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        if padded:
            env.update(PPALS_PAD_LAYOUT="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "hostsim_rank.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "OK" in out


@pytest.mark.parametrize("world", [4, 8])
def test_reduce_scatter_plan_every_mode(world):
    """the reduce-scatter / all-gather plan forced for every mode (PPALS_COMM_SMALL_BYTES=0) on
    cfg-shaped problems — equal shards, s divisible by the world size — at world 8 (what the
    driver's 8-GPU run of configs[3] uses when the messages are not small), see
    hostsim_rank.rs_plan_cases"""
    import hostsim_util
    hostsim_util.load()
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", PPALS_ORACLE_THREADS="1",
                   PPALS_RANK_MODE="rs_plan")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "hostsim_rank.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "OK" in out


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_plan_with_unequal_shards(world):
    """s not divisible by the world size, every mode through reduce-scatter + all-gather
    (hostsim_rank.rs_unequal_cases): the padding of the row blocks and the short last shard"""
    import hostsim_util
    hostsim_util.load()
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", PPALS_RANK_MODE="rs_unequal")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "hostsim_rank.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "OK" in out
