"""The drivers' RCCL unique-id rendezvous (host/driver_common.h: exchange_uid): every rank of a
launch gets rank 0's id of THIS launch — never a file left by an earlier launch from the same
parent process (stale /tmp/ppals_uid_* files) — and nothing is left behind."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "hostsim", "build", "uid_exchange_main")


@pytest.fixture(scope="module")
def harness():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(HERE, "uid_exchange_main.cpp")
    hdr = os.path.join(HERE, "..", "pairwise-perturbation_amd", "host", "driver_common.h")
    if (not os.path.exists(BIN)
            or os.path.getmtime(BIN) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wno-unused-function", "-o", BIN, src])
    return BIN


def launch(harness, world, uid_dir, skew_ms):
    env = dict(os.environ, PPALS_UID_DIR=str(uid_dir), MASTER_PORT="29999", PPALS_UID_TIMEOUT_S="20")
    env.pop("PPALS_UID_FILE", None)
    procs = [subprocess.Popen([harness, str(r), str(world), str(skew_ms[r])], env=env,
                              stdout=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=60)[0].strip() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    return outs


@pytest.mark.parametrize("skew", [(0, 0, 0), (300, 0, 0), (0, 300, 150)])
def test_two_launches_from_one_parent(harness, tmp_path, skew):
    world = 3
    # leftovers of a crashed earlier launch from this very parent: a full-size id file and a hello
    stale = tmp_path / f"ppals_uid_29999_none_{os.getpid()}"
    stale.write_bytes(b"\x11" * (128 + 16 * world))
    (tmp_path / (stale.name + ".r1")).write_bytes(b"\x22" * 16)
    first = launch(harness, world, tmp_path, skew)
    assert len(set(first)) == 1 and len(first[0]) == 256 and first[0] != "11" * 128
    second = launch(harness, world, tmp_path, skew)
    assert len(set(second)) == 1 and second[0] != first[0]
    assert [f for f in os.listdir(tmp_path) if f.startswith("ppals_uid_")] == []
