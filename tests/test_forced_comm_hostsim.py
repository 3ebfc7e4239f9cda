"""CPU run of the single-rank sharded-path rehearsal (tests/rccl_single_rank.py) on the host
stand-in with loopback collectives, so the script the GPU box runs against RCCL is itself tested."""
import ctypes as C
import os

import numpy as np

import hostsim_util
import rccl_single_rank


def test_forced_comm_paths_with_loopback_collectives(monkeypatch):
    monkeypatch.setenv("PPALS_FORCE_COMM", "1")
    pp = hostsim_util.load()
    ctx = pp.Context(0)
    AR = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64)
    RS = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)
    AG = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)
    calls = {"ar": 0, "rs": 0, "ag": 0}

    def allreduce(buf, n):
        calls["ar"] += 1

    def reduce_scatter(send, recv, n):
        calls["rs"] += 1
        np.ctypeslib.as_array(recv, (n,))[:] = np.ctypeslib.as_array(send, (n,)).copy()

    def allgather(send, recv, n):
        calls["ag"] += 1
        np.ctypeslib.as_array(recv, (n,))[:] = np.ctypeslib.as_array(send, (n,)).copy()

    cbs = (AR(allreduce), RS(reduce_scatter), AG(allgather))
    uid = C.create_string_buffer(128)
    for i, cb in enumerate(cbs):
        C.memmove(C.byref(uid, 8 * i), C.byref(C.cast(cb, C.c_void_p)), 8)
    ctx.init_comm(0, 1, uid)
    rccl_single_rank.run(pp, ctx)
    assert calls["ar"] > 0 and calls["rs"] > 0 and calls["ag"] > 0
    ctx.close()
