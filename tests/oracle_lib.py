"""ctypes binding of oracle/liboracle.so — the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Conventions: tensors are numpy fp64 arrays in Fortran order (first index fastest, like the
reference's CTF tensors); factor matrices are (s_i, R) Fortran-ordered.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_ip = C.POINTER(C.c_int)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _LIB = C.CDLL(path)
        _LIB.ppo_residual.restype = C.c_double
        _LIB.ppo_tree_node.restype = C.c_int64
        _LIB.ppo_pp_operator.restype = C.c_int64
        # the parity problems are tiny: on a many-core host (the GPU box exposes 128 threads) the
        # OpenMP fork/join of hundreds of small regions dominates, so cap the team
        _LIB.ppo_set_num_threads(int(os.environ.get("PPALS_ORACLE_THREADS", "4")))
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _lens(lens):
    return (C.c_int64 * len(lens))(*[int(x) for x in lens])


def _ranks(r):
    return (C.c_int * len(r))(*[int(x) for x in r])


def fill_uniform(n, seed, offset=0, lo=0.0, hi=1.0):
    out = np.empty(int(n), dtype=np.float64)
    lib().ppo_fill_uniform(_dp(out), C.c_int64(int(n)), C.c_uint64(seed), C.c_uint64(offset),
                           C.c_double(lo), C.c_double(hi))
    return out


def flat(Ws):
    """concatenate factor matrices (each (s_i, r_i)) in column-major order"""
    return np.concatenate([np.asfortranarray(W, dtype=np.float64).ravel(order="F") for W in Ws])


def unflat(wflat, lens, ranks):
    out, p = [], 0
    for s, r in zip(lens, ranks):
        out.append(wflat[p:p + s * r].reshape((s, r), order="F").copy(order="F"))
        p += s * r
    return out


def init_factors(lens, R, seed):
    """deterministic factor init used by tests, bench and the product's own drivers:
    W_i[e] = u01(seed + i, e), e = row + s_i*col"""
    return [fill_uniform(s * R, seed + i).reshape((s, R), order="F") for i, s in enumerate(lens)]


def dimension_tree(N):
    buf = C.create_string_buffer(4096)
    n = lib().ppo_dimension_tree(N, buf, 4096)
    assert n >= 0
    nodes = {}
    for rec in buf.value.decode().strip(";").split(";"):
        key, parent, sibling = rec.split(":")
        nodes[key] = {"parent": parent, "sibling": sibling}
    return nodes


def build_V(Ws):
    lens = [W.shape[0] for W in Ws]
    R = Ws[0].shape[1]
    V = np.empty(int(np.prod(lens)), dtype=np.float64)
    wf = flat(Ws)
    lib().ppo_build_V(len(lens), _lens(lens), R, _dp(wf), _dp(V))
    return V.reshape(lens, order="F")


def residual(V, Ws):
    lens = V.shape
    wf = flat(Ws)
    Vf = np.asfortranarray(V)
    return lib().ppo_residual(len(lens), _lens(lens), Ws[0].shape[1], _dp(Vf), _dp(wf))


def mttkrp(V, Ws, mode, route):
    lens = V.shape
    R = Ws[0].shape[1]
    wf = flat(Ws)
    Vf = np.asfortranarray(V)
    M = np.empty(lens[mode] * R, dtype=np.float64)
    lib().ppo_mttkrp(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), mode, route, _dp(M))
    return M.reshape((lens[mode], R), order="F")


def tree_node(V, Ws, key):
    lens = V.shape
    R = Ws[0].shape[1]
    wf = flat(Ws)
    Vf = np.asfortranarray(V)
    shape = [lens[ord(c) - 97] for c in key] + [R]
    out = np.empty(int(np.prod(shape)), dtype=np.float64)
    n = lib().ppo_tree_node(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), key.encode(), _dp(out))
    assert n == out.size, (n, out.size)
    return out.reshape(shape, order="F")


def pp_operator(V, Ws, contracted):
    """V contracted with the modes in `contracted` (e.g. "bd"); result over the remaining modes + r"""
    lens = V.shape
    N = len(lens)
    R = Ws[0].shape[1]
    wf = flat(Ws)
    Vf = np.asfortranarray(V)
    keep = [m for m in range(N) if chr(97 + m) not in contracted]
    shape = [lens[m] for m in keep] + [R]
    out = np.empty(int(np.prod(shape)), dtype=np.float64)
    n = lib().ppo_pp_operator(N, _lens(lens), R, _dp(Vf), _dp(wf), contracted.encode(), _dp(out))
    assert n == out.size
    return out.reshape(shape, order="F")


def gram_hadamard(Ws, mode, lam=0.0):
    lens = [W.shape[0] for W in Ws]
    R = Ws[0].shape[1]
    wf = flat(Ws)
    S = np.empty(R * R, dtype=np.float64)
    lib().ppo_gram_hadamard(len(lens), _lens(lens), R, _dp(wf), mode, C.c_double(lam), _dp(S))
    return S.reshape((R, R), order="F")


def svd_solve(M, S):
    rows, R = M.shape
    Mf, Sf = np.asfortranarray(M), np.asfortranarray(S)
    W = np.empty(rows * R, dtype=np.float64)
    lib().ppo_svd_solve(rows, R, _dp(Mf), _dp(Sf), _dp(W))
    return W.reshape((rows, R), order="F")


def normalize(Ws):
    lens = [W.shape[0] for W in Ws]
    R = Ws[0].shape[1]
    wf = flat(Ws)
    lib().ppo_normalize(len(lens), _lens(lens), R, _dp(wf))
    return unflat(wf, lens, [R] * len(lens))


def svd(A):
    m, n = A.shape
    Af = np.asfortranarray(A, dtype=np.float64)
    U = np.empty(m * n)
    s = np.empty(n)
    Vm = np.empty(n * n)
    lib().ppo_svd(m, n, _dp(Af), _dp(U), _dp(s), _dp(Vm))
    return U.reshape((m, n), order="F"), s, Vm.reshape((n, n), order="F")


def _run_cp(fn, V, Ws, gradWs, args, csv, verbose):
    lens = V.shape
    R = Ws[0].shape[1]
    wf, gf = flat(Ws), flat(gradWs)
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = fn(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), _dp(gf), *args,
            (csv.encode() if csv else None), *verbose, C.byref(iters))
    return rc, iters.value, unflat(wf, lens, [R] * len(lens)), unflat(gf, lens, [R] * len(lens))


def als_cp(V, Ws, gradWs, tol, maxiter, timelimit=5e3):
    lens = V.shape
    R = Ws[0].shape[1]
    wf, gf = flat(Ws), flat(gradWs)
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = lib().ppo_als_cp(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), _dp(gf), C.c_double(tol),
                          C.c_double(timelimit), maxiter, 0, C.byref(iters))
    return rc, iters.value, unflat(wf, lens, [R] * len(lens)), unflat(gf, lens, [R] * len(lens))


def als_cp_dt(V, Ws, gradWs, tol, maxiter, lam=0.0, csv=None, resprint=10, timelimit=5e3,
              verbose=0, bench=0):
    lens = V.shape
    R = Ws[0].shape[1]
    wf, gf = flat(Ws), flat(gradWs)
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = lib().ppo_als_cp_dt_ex(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), _dp(gf),
                                C.c_double(tol), C.c_double(timelimit), maxiter, C.c_double(lam),
                                (csv.encode() if csv else None), resprint, verbose, bench,
                                C.byref(iters))
    return rc, iters.value, unflat(wf, lens, [R] * len(lens)), unflat(gf, lens, [R] * len(lens))


def als_cp_pp(V, Ws, gradWs, tol, tol_init, maxiter, lam=0.0, ratio_step=1.0, csv=None,
              resprint=10, timelimit=5e3, verbose=0, bench=0):
    lens = V.shape
    R = Ws[0].shape[1]
    wf, gf = flat(Ws), flat(gradWs)
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = lib().ppo_als_cp_pp_ex(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), _dp(gf),
                                C.c_double(tol), C.c_double(tol_init), C.c_double(timelimit),
                                maxiter, C.c_double(lam), C.c_double(ratio_step),
                                (csv.encode() if csv else None), resprint, verbose, bench,
                                C.byref(iters))
    return rc, iters.value, unflat(wf, lens, [R] * len(lens)), unflat(gf, lens, [R] * len(lens))


def als_cp_pp_partupdate(V, Ws, gradWs, tol, tol_init, maxiter, update_percentage, lam=0.0,
                         ratio_step=1.0, csv=None, resprint=10, timelimit=5e3, verbose=0):
    lens = V.shape
    R = Ws[0].shape[1]
    wf, gf = flat(Ws), flat(gradWs)
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = lib().ppo_als_cp_pp_partupdate(
        len(lens), _lens(lens), R, _dp(Vf), _dp(wf), _dp(gf), C.c_double(tol),
        C.c_double(tol_init), C.c_double(timelimit), maxiter, C.c_double(lam),
        C.c_double(ratio_step), C.c_double(update_percentage), (csv.encode() if csv else None),
        resprint, verbose, C.byref(iters))
    return rc, iters.value, unflat(wf, lens, [R] * len(lens)), unflat(gf, lens, [R] * len(lens))


def cpd_als(V, Ws, gradWs, kind, tol, maxsweep, lam=0.0, csv=None, resprint=10, timelimit=5e3,
            verbose=0):
    """class API CPD<double, Optimizer>::als (src/CP.cxx:100-186); kind 0 simple, 1 DT, 2 MSDT.
    Returns (rc, sweeps, iters, W, gradW)."""
    lens = V.shape
    R = Ws[0].shape[1]
    wf, gf = flat(Ws), flat(gradWs)
    Vf = np.asfortranarray(V)
    sweeps = C.c_double(0)
    iters = C.c_int(0)
    f = lib().ppo_cpd_als
    f.restype = C.c_int
    rc = f(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), _dp(gf), kind, C.c_double(lam),
           C.c_double(tol), C.c_double(timelimit), maxsweep, resprint,
           (csv.encode() if csv else None), verbose, C.byref(sweeps), C.byref(iters))
    return (rc, sweeps.value, iters.value, unflat(wf, lens, [R] * len(lens)),
            unflat(gf, lens, [R] * len(lens)))


def cpd_als_lr(V, Ws, gradWs, kind, update_rank, tol, maxsweep, lam=0.0, csv=None, resprint=10,
               timelimit=5e3, randomsvd=0):
    """class API with the low-rank-update optimizers: kind 3 CPDTLROptimizer, 4 CPMSDTLROptimizer;
    update_rank / randomsvd = run.cxx's -updaterank / -randomsvd. Returns (rc, sweeps, iters, W, gradW)."""
    lens = V.shape
    R = Ws[0].shape[1]
    wf, gf = flat(Ws), flat(gradWs)
    Vf = np.asfortranarray(V)
    sweeps = C.c_double(0)
    iters = C.c_int(0)
    f = lib().ppo_cpd_als_lr
    f.restype = C.c_int
    rc = f(len(lens), _lens(lens), R, _dp(Vf), _dp(wf), _dp(gf), kind, update_rank, int(randomsvd),
           C.c_double(lam), C.c_double(tol), C.c_double(timelimit), maxsweep, resprint,
           (csv.encode() if csv else None), 0, C.byref(sweeps), C.byref(iters))
    assert rc >= 0, "bad argument"
    return (rc, sweeps.value, iters.value, unflat(wf, lens, [R] * len(lens)),
            unflat(gf, lens, [R] * len(lens)))


def sort_indexes(v):
    """the oracle's restatement of sort_indexes (als_CP.cxx:835-843)"""
    v = np.ascontiguousarray(v, dtype=np.float64)
    idx = (C.c_int * len(v))()
    lib().ppo_sort_indexes(len(v), _dp(v), idx)
    return list(idx)


def ttmc(V, Ws, skip):
    lens = V.shape
    ranks = [W.shape[1] for W in Ws]
    shape = [lens[i] if i == skip else ranks[i] for i in range(len(lens))]
    wf = flat(Ws)
    Vf = np.asfortranarray(V)
    Y = np.empty(int(np.prod(shape)))
    lib().ppo_ttmc(len(lens), _lens(lens), _ranks(ranks), _dp(Vf), _dp(wf), skip, _dp(Y))
    return Y.reshape(shape, order="F")


def hosvd(V, ranks):
    lens = V.shape
    wf = np.zeros(sum(s * r for s, r in zip(lens, ranks)))
    core = np.empty(int(np.prod(ranks)))
    Vf = np.asfortranarray(V)
    lib().ppo_hosvd(len(lens), _lens(lens), _ranks(ranks), _dp(Vf), _dp(wf), _dp(core))
    return unflat(wf, lens, ranks), core.reshape(ranks, order="F")


def als_tucker(V, Ws, core, tol, maxiter, timelimit=5e3):
    lens = V.shape
    ranks = [W.shape[1] for W in Ws]
    wf = flat(Ws)
    cf = np.asfortranarray(core).ravel(order="F").copy()
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = lib().ppo_als_tucker(len(lens), _lens(lens), _ranks(ranks), _dp(Vf), _dp(wf), _dp(cf),
                              C.c_double(tol), C.c_double(timelimit), maxiter, 0, C.byref(iters))
    return rc, iters.value, unflat(wf, lens, ranks), cf.reshape(ranks, order="F")


def als_tucker_dt(V, Ws, core, tol, maxiter, csv=None, resprint=10, timelimit=5e3, verbose=0,
                  bench=0):
    lens = V.shape
    ranks = [W.shape[1] for W in Ws]
    wf = flat(Ws)
    cf = np.asfortranarray(core).ravel(order="F").copy()
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = lib().ppo_als_tucker_dt_ex(len(lens), _lens(lens), _ranks(ranks), _dp(Vf), _dp(wf),
                                    _dp(cf), C.c_double(tol), C.c_double(timelimit), maxiter,
                                    (csv.encode() if csv else None), resprint, verbose, bench,
                                    C.byref(iters))
    return rc, iters.value, unflat(wf, lens, ranks), cf.reshape(ranks, order="F")


def als_tucker_pp(V, Ws, core, tol, tol_init, maxiter, csv=None, resprint=10, timelimit=5e3,
                  verbose=0, bench=0):
    lens = V.shape
    ranks = [W.shape[1] for W in Ws]
    wf = flat(Ws)
    cf = np.asfortranarray(core).ravel(order="F").copy()
    Vf = np.asfortranarray(V)
    iters = C.c_int(0)
    rc = lib().ppo_als_tucker_pp_ex(len(lens), _lens(lens), _ranks(ranks), _dp(Vf), _dp(wf),
                                    _dp(cf), C.c_double(tol), C.c_double(tol_init),
                                    C.c_double(timelimit), maxiter,
                                    (csv.encode() if csv else None), resprint, verbose, bench,
                                    C.byref(iters))
    return rc, iters.value, unflat(wf, lens, ranks), cf.reshape(ranks, order="F")


def read_csv(path):
    """parse the reference-format CSV into (header, rows of floats); blank flush lines skipped"""
    with open(path) as f:
        lines = [ln.strip() for ln in f.read().splitlines()]
    header = lines[0].split(",")
    rows = [[float(x) for x in ln.split(",")] for ln in lines[1:] if ln]
    return header, rows
