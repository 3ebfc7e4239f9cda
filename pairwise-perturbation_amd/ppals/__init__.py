"""ppals — ctypes binding of libppals.so (include/ppals.h), the MI355X-native ALS sweep engine.

This is plumbing for tests/ and bench.py; the product is the C-ABI library and the C++ `test_ALS`
driver. There is no CPU fallback: creating a Context without a HIP device raises PpalsError.

Names follow the reference (LinjianMa/pairwise-perturbation): `alsCP_DT`, `alsCP_PP`, `hosvd`,
`alsTucker_DT` take the same arguments in the same order as als_CP.h:30-32,105-108 and
als_Tucker.h:14-15,46-48, minus the CTF `World` (the Context) and the always-zero `F`.

Array conventions: tensors are Fortran-ordered numpy fp64 arrays (first index fastest, like CTF);
factor matrices are (s_i, R) Fortran-ordered.
"""
import ctypes as C
import os
import weakref

import numpy as np

F32, F64 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("PPALS_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libppals.so"))
_lib = None

c_dp = C.POINTER(C.c_double)


class PpalsError(RuntimeError):
    pass


class _Opts(C.Structure):
    _fields_ = [("tol", C.c_double), ("timelimit", C.c_double), ("maxiter", C.c_int),
                ("lambda_", C.c_double), ("resprint", C.c_int), ("bench", C.c_int),
                ("tol_init", C.c_double), ("ratio_step", C.c_double), ("csv_path", C.c_char_p),
                ("csv_append", C.c_int), ("verbose", C.c_int),
                ("update_percentage", C.c_double)]


EXPORTS = [
    "ppals_last_error", "ppals_version", "ppals_preload_eigensolver", "ppals_ctx_create",
    "ppals_ctx_destroy",
    "ppals_get_unique_id", "ppals_ctx_init_comm", "ppals_ctx_rank", "ppals_ctx_nranks",
    "ppals_ctx_sync", "ppals_profile_enable", "ppals_profile_read", "ppals_profile_reset",
    "ppals_tensor_create", "ppals_tensor_destroy", "ppals_tensor_local_rows",
    "ppals_tensor_fill_cp", "ppals_tensor_fill_uniform", "ppals_tensor_upload",
    "ppals_tensor_download",
    "ppals_tensor_fill_laplacian", "ppals_tensor_fill_collinear", "ppals_collinear_factors",
    "ppals_tensor_norm", "ppals_fill_uniform_host", "ppals_tree_node", "ppals_mttkrp",
    "ppals_pp_operator", "ppals_cp_residual", "ppals_cp_gram_system", "ppals_cp_create",
    "ppals_cp_destroy", "ppals_cp_set_factors", "ppals_cp_get_factors", "ppals_cp_sweeps_dt",
    "ppals_cp_gradnorm", "ppals_cp_dt", "ppals_cp_pp", "ppals_cp_pp_partupdate",
    "ppals_cpd_als", "ppals_cpd_als_lr", "ppals_cp_set_schedule", "ppals_cp_get_schedule", "ppals_cp_placement_report", "ppals_cp_pp_build_stats",
    "ppals_tucker_create",
    "ppals_tucker_destroy", "ppals_tucker_set_factors", "ppals_tucker_get_factors",
    "ppals_tucker_set_core",
    "ppals_tucker_hosvd", "ppals_tucker_ttmc", "ppals_tucker_sweeps_dt", "ppals_tucker_dt",
    "ppals_tucker_pp",
]


def lib(path=None):
    """load libppals.so; raises (loudly) when it has not been built"""
    global _lib
    if _lib is None:
        p = path or _LIBPATH
        if not os.path.exists(p):
            raise PpalsError(f"{p} not found: build it with `make -C pairwise-perturbation_amd` "
                             "(there is no pure-Python or CPU fallback)")
        _lib = C.CDLL(p)
        _lib.ppals_last_error.restype = C.c_char_p
        _lib.ppals_version.restype = C.c_char_p
        # the product binding only ever drives the HIP build; the host stand-in of tests/hostsim is
        # reachable solely through tests/hostsim_util.py, which loads this file under another name
        if __name__ == "ppals" and b"TEST INFRASTRUCTURE" in _lib.ppals_version():
            _lib = None
            raise PpalsError(f"{p} is the test stand-in, not the HIP engine (no CPU fallback)")
    return _lib


def _check(rc, allow_bool=False):
    if rc < 0:
        raise PpalsError(f"ppals error {rc}: {lib().ppals_last_error().decode()}")
    return rc


def _dp(a):
    return a.ctypes.data_as(c_dp)


def flat(Ws):
    return np.concatenate([np.asfortranarray(W, dtype=np.float64).ravel(order="F") for W in Ws])


def unflat(wflat, lens, ranks):
    out, p = [], 0
    for s, r in zip(lens, ranks):
        out.append(wflat[p:p + s * r].reshape((s, r), order="F").copy(order="F"))
        p += s * r
    return out


def fill_uniform_host(n, seed, offset=0, lo=0.0, hi=1.0):
    out = np.empty(int(n), dtype=np.float64)
    lib().ppals_fill_uniform_host(_dp(out), C.c_int64(int(n)), C.c_uint64(seed),
                                  C.c_uint64(offset), C.c_double(lo), C.c_double(hi))
    return out


def init_factors(lens, R, seed):
    """W_i[e] = u01(seed + i, e): the deterministic stand-in for CTF's W[i].fill_random(0,1)"""
    return [fill_uniform_host(s * R, seed + i).reshape((s, R), order="F")
            for i, s in enumerate(lens)]


def collinear_factors(lens, R, col_min=0.5, col_max=0.9, seed=0):
    """the factor vectors of `-tensor c` (Gen_collinearity, common.cxx:361-423), lambda in mode 0"""
    wf = np.empty(sum(int(s) * R for s in lens))
    arr = (C.c_int64 * len(lens))(*[int(x) for x in lens])
    _check(lib().ppals_collinear_factors(len(lens), arr, R, C.c_double(col_min),
                                         C.c_double(col_max), C.c_uint64(seed), _dp(wf)))
    return unflat(wf, lens, [R] * len(lens))


def preload_eigensolver():
    """load rocBLAS / rocSOLVER now — call before anything initialises the HIP runtime in this
    process when a Tucker session with a mode extent > 64 will follow (include/ppals.h)"""
    _check(lib().ppals_preload_eigensolver())


class Context:
    """replaces CTF::World (test_ALS.cxx:200): one per process, one GPU"""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._children = weakref.WeakSet()  # tensors / sessions that must die before the context
        _check(lib().ppals_ctx_create(C.byref(self._h), device))

    def init_comm(self, rank, nranks, unique_id):
        _check(lib().ppals_ctx_init_comm(self._h, rank, nranks, unique_id))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().ppals_get_unique_id(buf))
        return buf.raw

    @property
    def rank(self):
        return lib().ppals_ctx_rank(self._h)

    @property
    def nranks(self):
        return lib().ppals_ctx_nranks(self._h)

    def sync(self):
        _check(lib().ppals_ctx_sync(self._h))

    def profile_enable(self, level=1):
        """0 off, 1 tensor scans only, 2 scans + the other bracketed kernels"""
        _check(lib().ppals_profile_enable(self._h, int(level)))

    def profile_reset(self):
        _check(lib().ppals_profile_reset(self._h))

    def profile_read(self, which=0):
        n, ms, by = C.c_int64(0), C.c_double(0), C.c_double(0)
        _check(lib().ppals_profile_read(self._h, which, C.byref(n), C.byref(ms), C.byref(by)))
        return n.value, ms.value, by.value

    def close(self):
        if self._h:
            for ch in sorted(list(self._children), key=lambda c: isinstance(c, Tensor)):
                ch.close()  # sessions first, then tensors
            lib().ppals_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tensor:
    """replaces CTF::Tensor<> V: dense, block-partitioned along the leading mode"""

    def __init__(self, ctx, lens, dtype=F32):
        self.ctx = ctx
        self.lens = [int(x) for x in lens]
        self.dtype = dtype
        self._h = C.c_void_p()
        arr = (C.c_int64 * len(lens))(*self.lens)
        _check(lib().ppals_tensor_create(ctx._h, len(lens), arr, dtype, C.byref(self._h)))
        ctx._children.add(self)

    def local_rows(self):
        lo, n = C.c_int64(0), C.c_int64(0)
        _check(lib().ppals_tensor_local_rows(self._h, C.byref(lo), C.byref(n)))
        return lo.value, n.value

    def fill_cp(self, Wtrue):
        wf = flat(Wtrue)
        _check(lib().ppals_tensor_fill_cp(self._h, Wtrue[0].shape[1], _dp(wf)))
        return self

    def fill_uniform(self, seed, lo=0.5, hi=1.0):
        _check(lib().ppals_tensor_fill_uniform(self._h, C.c_uint64(seed), C.c_double(lo),
                                               C.c_double(hi)))
        return self

    def fill_laplacian(self, ndigits, s):
        _check(lib().ppals_tensor_fill_laplacian(self._h, ndigits, s))
        return self

    def fill_collinear(self, R, col_min=0.5, col_max=0.9, ratio_noise=0.01, seed=0):
        _check(lib().ppals_tensor_fill_collinear(self._h, R, C.c_double(col_min),
                                                 C.c_double(col_max), C.c_double(ratio_noise),
                                                 C.c_uint64(seed)))
        return self

    def upload(self, V):
        Vf = np.asfortranarray(V, dtype=np.float64)
        assert list(Vf.shape) == self.lens
        _check(lib().ppals_tensor_upload(self._h, _dp(Vf)))
        return self

    def download(self):
        """the tensor as fp64, first index fastest (this rank's rows; zeros elsewhere)"""
        out = np.zeros(self.lens, dtype=np.float64, order="F")
        _check(lib().ppals_tensor_download(self._h, _dp(out)))
        return out

    def norm(self):
        out = C.c_double(0)
        _check(lib().ppals_tensor_norm(self._h, C.byref(out)))
        return out.value

    def close(self):
        if self._h:
            lib().ppals_tensor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _opts(tol=0.0, timelimit=5e3, maxiter=0, lam=0.0, resprint=10, bench=0, tol_init=1e-2,
          ratio_step=1.0, csv=None, csv_append=0, verbose=0, update_percentage=1.0):
    return _Opts(tol, timelimit, maxiter, lam, resprint, bench, tol_init, ratio_step,
                 csv.encode() if csv else None, csv_append, verbose, update_percentage)


class CP:
    """a CP-ALS session: factors, Grams and dimension-tree caches resident in HBM"""

    def __init__(self, ctx, V, R):
        self.ctx, self.V, self.R = ctx, V, R
        self.lens = V.lens
        self._h = C.c_void_p()
        _check(lib().ppals_cp_create(ctx._h, V._h, R, C.byref(self._h)))
        ctx._children.add(self)

    def set_factors(self, Ws, gradWs=None):
        wf = flat(Ws)
        gf = flat(gradWs) if gradWs is not None else None
        _check(lib().ppals_cp_set_factors(self._h, _dp(wf), _dp(gf) if gf is not None else None))

    def get_factors(self, with_grad=False):
        n = sum(s * self.R for s in self.lens)
        wf = np.empty(n)
        gf = np.empty(n) if with_grad else None
        _check(lib().ppals_cp_get_factors(self._h, _dp(wf), _dp(gf) if with_grad else None))
        W = unflat(wf, self.lens, [self.R] * len(self.lens))
        if with_grad:
            return W, unflat(gf, self.lens, [self.R] * len(self.lens))
        return W

    def set_schedule(self, schedule):
        """"dt" (two first-level nodes, alsCP_DT) or "msdt" (multi-sweep tree, the default)"""
        code = {"dt": 0, "msdt": 1}[schedule] if isinstance(schedule, str) else int(schedule)
        _check(lib().ppals_cp_set_schedule(self._h, code))

    @property
    def schedule(self):
        return {0: "dt", 1: "msdt"}[_check(lib().ppals_cp_get_schedule(self._h))]

    def placement_report(self):
        """where the online placement choice put each root's first-level intermediate (dict)"""
        import json
        buf = C.create_string_buffer(8192)
        _check(lib().ppals_cp_placement_report(self._h, buf, 8192))
        return json.loads(buf.value.decode())

    def pp_build_stats(self, mode=0):
        """(number, seconds) of the PP operator builds since the last reset; mode +1 / -1: reset and
        turn their timing on / off"""
        n, sec = C.c_int64(0), C.c_double(0)
        _check(lib().ppals_cp_pp_build_stats(self._h, mode, C.byref(n), C.byref(sec)))
        return n.value, sec.value

    def sweeps_dt(self, n, lam=0.0):
        _check(lib().ppals_cp_sweeps_dt(self._h, n, C.c_double(lam)))

    def gradnorm(self):
        out = C.c_double(0)
        _check(lib().ppals_cp_gradnorm(self._h, C.byref(out)))
        return out.value

    def residual(self):
        out = C.c_double(0)
        _check(lib().ppals_cp_residual(self._h, C.byref(out)))
        return out.value

    def tree_node(self, key, shape=None):
        n = C.c_int64(0)
        _check(lib().ppals_tree_node(self._h, key.encode(), None, C.byref(n)))
        out = np.empty(n.value)
        _check(lib().ppals_tree_node(self._h, key.encode(), _dp(out), C.byref(n)))
        return out.reshape(shape, order="F") if shape else out

    def mttkrp(self, mode):
        M = np.empty(self.lens[mode] * self.R)
        _check(lib().ppals_mttkrp(self._h, mode, _dp(M)))
        return M.reshape((self.lens[mode], self.R), order="F")

    def pp_operator(self, contracted, shape=None):
        n = C.c_int64(0)
        _check(lib().ppals_pp_operator(self._h, contracted.encode(), None, C.byref(n)))
        out = np.empty(n.value)
        _check(lib().ppals_pp_operator(self._h, contracted.encode(), _dp(out), C.byref(n)))
        return out.reshape(shape, order="F") if shape else out

    def gram_system(self, mode, lam=0.0):
        S = np.empty(self.R * self.R)
        Si = np.empty(self.R * self.R)
        _check(lib().ppals_cp_gram_system(self._h, mode, C.c_double(lam), _dp(S), _dp(Si)))
        return S.reshape((self.R, self.R), order="F"), Si.reshape((self.R, self.R), order="F")

    def run_dt(self, **kw):
        o = _opts(**kw)
        it = C.c_int(0)
        rc = _check(lib().ppals_cp_dt(self._h, C.byref(o), C.byref(it)))
        return rc, it.value

    def run_pp(self, **kw):
        o = _opts(**kw)
        it = C.c_int(0)
        rc = _check(lib().ppals_cp_pp(self._h, C.byref(o), C.byref(it)))
        return rc, it.value

    def cpd_als(self, optimizer, **kw):
        """CPD<dtype, Optimizer>::als (src/CP.cxx:100-186); maxiter= is maxsweep.
        Returns (rc, sweeps, iters)."""
        o = _opts(**kw)
        it = C.c_int(0)
        sw = C.c_double(0)
        rc = _check(lib().ppals_cpd_als(self._h, int(optimizer), C.byref(o), C.byref(sw),
                                        C.byref(it)))
        return rc, sw.value, it.value

    def cpd_als_lr(self, optimizer, update_rank, randomsvd=0, **kw):
        """CPD<dtype, CPDTLROptimizer / CPMSDTLROptimizer>::als (optimizer 3 / 4); randomsvd as
        run.cxx's flag. Returns (rc, sweeps, iters)."""
        o = _opts(**kw)
        it = C.c_int(0)
        sw = C.c_double(0)
        rc = _check(lib().ppals_cpd_als_lr(self._h, int(optimizer), int(update_rank), int(randomsvd),
                                           C.byref(o), C.byref(sw), C.byref(it)))
        return rc, sw.value, it.value

    def run_pp_partupdate(self, **kw):
        o = _opts(**kw)
        it = C.c_int(0)
        rc = _check(lib().ppals_cp_pp_partupdate(self._h, C.byref(o), C.byref(it)))
        return rc, it.value

    def close(self):
        if self._h:
            lib().ppals_cp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tucker:
    def __init__(self, ctx, V, ranks):
        self.ctx, self.V = ctx, V
        self.lens, self.ranks = V.lens, [int(r) for r in ranks]
        self._h = C.c_void_p()
        arr = (C.c_int * len(ranks))(*self.ranks)
        _check(lib().ppals_tucker_create(ctx._h, V._h, arr, C.byref(self._h)))
        ctx._children.add(self)

    def set_factors(self, Ws):
        wf = flat(Ws)
        _check(lib().ppals_tucker_set_factors(self._h, _dp(wf)))

    def set_core(self, core=None):
        """the `core` argument of alsTucker_DT/_PP; None: recompute it from V and the factors"""
        if core is None:
            _check(lib().ppals_tucker_set_core(self._h, None))
        else:
            cf = np.asfortranarray(core, dtype=np.float64).ravel(order="F").copy()
            _check(lib().ppals_tucker_set_core(self._h, _dp(cf)))

    def get_factors(self):
        wf = np.empty(sum(s * r for s, r in zip(self.lens, self.ranks)))
        core = np.empty(int(np.prod(self.ranks)))
        _check(lib().ppals_tucker_get_factors(self._h, _dp(wf), _dp(core)))
        return unflat(wf, self.lens, self.ranks), core.reshape(self.ranks, order="F")

    def hosvd(self):
        _check(lib().ppals_tucker_hosvd(self._h))

    def ttmc(self, skip):
        n = C.c_int64(0)
        shape = [self.lens[i] if i == skip else self.ranks[i] for i in range(len(self.lens))]
        Y = np.empty(int(np.prod(shape)))
        _check(lib().ppals_tucker_ttmc(self._h, skip, _dp(Y), C.byref(n)))
        return Y.reshape(shape, order="F")

    def sweeps_dt(self, n):
        _check(lib().ppals_tucker_sweeps_dt(self._h, n))

    def run_dt(self, **kw):
        o = _opts(**kw)
        it = C.c_int(0)
        rc = _check(lib().ppals_tucker_dt(self._h, C.byref(o), C.byref(it)))
        return rc, it.value

    def run_pp(self, **kw):
        o = _opts(**kw)
        it = C.c_int(0)
        rc = _check(lib().ppals_tucker_pp(self._h, C.byref(o), C.byref(it)))
        return rc, it.value

    def close(self):
        if self._h:
            lib().ppals_tucker_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- reference-named entry points (als_CP.h / als_Tucker.h), argument order preserved ----
def alsCP_DT(V, W, grad_W, tol, timelimit, maxiter, lambda_, Plot_File, resprint, bench, dw):
    """alsCP_DT(V, W, grad_W, F, tol, timelimit, maxiter, lambda, Plot_File, resprint, bench, dw)
    (als_CP.h:30-32) — F is always zero in the reference and is dropped. W/grad_W are updated in
    place (lists of numpy arrays). Returns the reference's bool."""
    s = CP(dw, V, W[0].shape[1])
    s.set_factors(W, grad_W)
    rc, _ = s.run_dt(tol=tol, timelimit=timelimit, maxiter=maxiter, lam=lambda_, csv=Plot_File,
                     resprint=resprint, bench=int(bench))
    Wn, Gn = s.get_factors(with_grad=True)
    for a, b in zip(W, Wn):
        a[...] = b
    for a, b in zip(grad_W, Gn):
        a[...] = b
    s.close()
    return bool(rc)


def alsCP_PP(V, W, grad_W, tol, tol_init, timelimit, maxiter, lambda_, ratio_step, Plot_File,
             resprint, bench, dw):
    """alsCP_PP (als_CP.h:105-108), F dropped"""
    s = CP(dw, V, W[0].shape[1])
    s.set_factors(W, grad_W)
    rc, _ = s.run_pp(tol=tol, tol_init=tol_init, timelimit=timelimit, maxiter=maxiter,
                     lam=lambda_, ratio_step=ratio_step, csv=Plot_File, resprint=resprint,
                     bench=int(bench))
    Wn, Gn = s.get_factors(with_grad=True)
    for a, b in zip(W, Wn):
        a[...] = b
    for a, b in zip(grad_W, Gn):
        a[...] = b
    s.close()
    return bool(rc)
