"""Loads the ppals ctypes binding against tests/hostsim/build/libppals_hostsim.so (TEST
INFRASTRUCTURE: the product's engine + C-ABI sources over a host stand-in for the device ops)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_mod = None


def load():
    global _mod
    if _mod is None:
        libpath = os.environ.get("PPALS_HOSTSIM_LIB")  # tests/test_sanitizers.py: the ASan build
        if not libpath:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
            libpath = os.path.join(ROOT, "tests", "hostsim", "build", "libppals_hostsim.so")
        path = os.path.join(ROOT, "pairwise-perturbation_amd", "ppals", "__init__.py")
        spec = importlib.util.spec_from_file_location("ppals_hostsim", path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["ppals_hostsim"] = mod  # so objects can find the binding they came from
        spec.loader.exec_module(mod)
        mod._LIBPATH = libpath
        assert b"hostsim" in mod.lib().ppals_version()
        _mod = mod
    return _mod


def gloo_comm_uid(rank, world):
    """the 128-byte 'unique id' of the stand-in communicator: three callback pointers (all-reduce,
    reduce-scatter, all-gather of fp64 buffers) served by torch.distributed on the default (gloo)
    process group. Returns (uid, keepalive, calls): keep `keepalive` referenced while the context lives."""
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    AR = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64)
    RS = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)
    AG = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)
    calls = {"ar": 0, "rs": 0, "ag": 0}

    def allreduce(buf, n):
        calls["ar"] += 1
        t = torch.from_numpy(np.ctypeslib.as_array(buf, (n,)))
        dist.all_reduce(t)

    def reduce_scatter(send, recv, n):  # gloo has no reduce_scatter: all_reduce + slice
        calls["rs"] += 1
        t = torch.from_numpy(np.ctypeslib.as_array(send, (n * world,)).copy())
        dist.all_reduce(t)
        np.ctypeslib.as_array(recv, (n,))[:] = t.numpy()[rank * n:(rank + 1) * n]

    def allgather(send, recv, n):
        calls["ag"] += 1
        mine = torch.from_numpy(np.ctypeslib.as_array(send, (n,)).copy())
        outs = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(outs, mine)
        np.ctypeslib.as_array(recv, (n * world,))[:] = torch.cat(outs).numpy()

    cbs = (AR(allreduce), RS(reduce_scatter), AG(allgather))
    uid = C.create_string_buffer(128)
    for i, cb in enumerate(cbs):
        C.memmove(C.byref(uid, 8 * i), C.byref(C.cast(cb, C.c_void_p)), 8)
    return uid, cbs, calls
