"""Loads the ppals ctypes binding against tests/hipsim/build/libppals_hipsim.so (TEST
INFRASTRUCTURE: the product's engine + C ABI + HIP kernels with a staged callback communicator in
place of RCCL), and provides the two communicators its callbacks can sit on:

* between PROCESSES sharing the one GPU: torch.distributed / gloo (hostsim_util.gloo_comm_uid);
* between THREADS of one process (a GPU box allows at most 6 processes on its card, so world 8 —
  BASELINE configs[3] — runs as 8 threads, each with its own ppals_ctx, stream and shard):
  ThreadWorld below, a barrier + fixed-order sum on host buffers.
"""
import ctypes as C
import importlib.util
import os
import subprocess
import sys
import threading

import numpy as np

from hostsim_util import gloo_comm_uid  # noqa: F401  (same three callbacks, host buffers)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBPATH = os.path.join(ROOT, "tests", "hipsim", "build", "libppals_hipsim.so")
_mod = None


def build():
    subprocess.check_call(["make", "-s", "-j", "4", "-C", os.path.join(ROOT, "tests", "hipsim")])
    return LIBPATH


def load(make=True):
    global _mod
    if _mod is None:
        if make or not os.path.exists(LIBPATH):
            build()
        path = os.path.join(ROOT, "pairwise-perturbation_amd", "ppals", "__init__.py")
        spec = importlib.util.spec_from_file_location("ppals_hipsim", path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["ppals_hipsim"] = mod
        spec.loader.exec_module(mod)
        mod._LIBPATH = LIBPATH
        assert b"hipsim" in mod.lib().ppals_version()
        _mod = mod
    return _mod


class ThreadWorld:
    """P ranks = P threads of this process. A collective: every rank deposits a copy of its send
    buffer, all meet at a barrier, every rank forms the result from the deposits IN RANK ORDER (so
    all ranks hold bit-identical sums, as after an RCCL all-reduce), all meet again before a slot is
    reused. A rank that fails aborts the barrier, so the others raise instead of waiting forever."""

    def __init__(self, size, timeout=600.0):
        self.size = size
        self.slots = [None] * size
        self._barrier = threading.Barrier(size, timeout=timeout)
        self.calls = [{"ar": 0, "rs": 0, "ag": 0} for _ in range(size)]
        self.failed = []

    def barrier(self):
        self._barrier.wait()

    def abort(self):
        self._barrier.abort()

    def setenv(self, rank, key, val):
        """an environment switch the engine reads at session creation: set between two barriers so
        that no rank creates a session while another one changes it"""
        self.barrier()
        if rank == 0:
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = str(val)
        self.barrier()

    def _exchange(self, rank, arr):
        self.slots[rank] = arr
        self.barrier()
        got = list(self.slots)
        self.barrier()
        return got

    def comm_uid(self, rank):
        """(uid, keepalive) for ppals_ctx_init_comm of `rank`"""
        P, calls = self.size, self.calls[rank]
        AR = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64)
        RS = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)
        AG = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)

        def total(parts):
            acc = parts[0].copy()
            for p in parts[1:]:
                acc += p
            return acc

        def allreduce(buf, n):
            calls["ar"] += 1
            mine = np.ctypeslib.as_array(buf, (n,))
            parts = self._exchange(rank, mine.copy())
            mine[:] = total(parts)

        def reduce_scatter(send, recv, n):
            calls["rs"] += 1
            parts = self._exchange(rank, np.ctypeslib.as_array(send, (n * P,)).copy())
            np.ctypeslib.as_array(recv, (n,))[:] = total([p[rank * n:(rank + 1) * n] for p in parts])

        def allgather(send, recv, n):
            calls["ag"] += 1
            parts = self._exchange(rank, np.ctypeslib.as_array(send, (n,)).copy())
            np.ctypeslib.as_array(recv, (n * P,))[:] = np.concatenate(parts)

        def guard(f):
            def g(*a):
                try:
                    f(*a)
                except BaseException as e:  # a broken barrier: another rank failed
                    self.failed.append((rank, repr(e)))
                    self.abort()
            return g

        cbs = (AR(guard(allreduce)), RS(guard(reduce_scatter)), AG(guard(allgather)))
        uid = C.create_string_buffer(128)
        for i, cb in enumerate(cbs):
            C.memmove(C.byref(uid, 8 * i), C.byref(C.cast(cb, C.c_void_p)), 8)
        return uid, cbs
