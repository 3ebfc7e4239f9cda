"""Second, independent restatement of the reference's CP drivers (TEST INFRASTRUCTURE): every
tensor expression evaluated by numpy.einsum with the reference's own index roles, the R x R
inverse through LAPACK's SVD (numpy.linalg.svd — the routine family behind CTF's Matrix::svd),
no shared code with oracle/ppals_oracle.cpp. It records the (iter, pp_update) print rows so that
the phase switching of alsCP_PP can be compared, not only the final factors.

  als_cp_dt : alsCP_DT       als_CP.cxx:127-320
  als_cp_pp : alsCP_PP       als_CP.cxx:1082-1137 with alsCP_DT_sub :418-612, alsCP_PP_sub :621-833
"""
import string

import numpy as np

LET = string.ascii_lowercase


def _mttkrp(V, W, i):
    """M["dk"] = V["abcd"] W1["ak"] W2["bk"] W3["ck"]  (als_CP.cxx:84-86), mode i kept"""
    N = V.ndim
    subs = [LET[:N]] + [LET[j] + "z" for j in range(N) if j != i]
    return np.einsum(",".join(subs) + "->" + LET[i] + "z", V, *[W[j] for j in range(N) if j != i],
                     optimize=True)


def _S(W, i, lam):
    """S["ij"] = prod_{j != i} W_j["ki"] W_j["kj"]  (+ regul)  (als_CP.cxx:288-292)"""
    R = W[0].shape[1]
    S = np.ones((R, R))
    for j in range(len(W)):
        if j != i:
            S = S * (W[j].T @ W[j])
    return S + lam * np.eye(R)


def _svd_inverse(S):
    U, s, Vt = np.linalg.svd(S)  # S_reverse["ij"] = VT["ki"] s["k"]^-1 U["jk"]  (common.cxx:717-722)
    return Vt.T @ np.diag(1.0 / s) @ U.T


def _normalize(W):
    N = len(W)
    norm = 1.0
    for w in W:
        norm = norm * np.linalg.norm(w)
    norm = norm ** (1.0 / N)
    return [w * (norm / np.linalg.norm(w)) for w in W]


def _residual(V, W):
    N = V.ndim
    model = np.einsum(",".join(LET[j] + "z" for j in range(N)) + "->" + LET[:N], *W, optimize=True)
    return np.linalg.norm(V - model)


def _gradnorm(G):
    return np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G))


def _exact_sweep(V, W, G, lam):
    for i in range(V.ndim):
        M = _mttkrp(V, W, i)
        S = _S(W, i, lam)
        G[i] = -M + W[i] @ S
        W[i] = M @ _svd_inverse(S)
    return _normalize(W)


def als_cp_dt(V, W, G, tol, maxiter, lam=0.0, resprint=10):
    W, G = [w.copy() for w in W], [g.copy() for g in G]
    rows = []
    it = 0
    for it in range(maxiter + 2):
        if it > maxiter:
            break
        if it % resprint == 0 or it == maxiter:
            gn = _gradnorm(G)
            rows.append((it, 0, gn, _residual(V, W)))
            if gn < tol:
                break
        W = _exact_sweep(V, W, G, lam)
    return it, W, G, rows


def als_cp_pp(V, W, G, tol, tol_init, maxiter, lam=0.0, ratio_step=1.0, resprint=10):
    N = V.ndim
    W, G = [w.copy() for w in W], [g.copy() for g in G]
    dW = [np.zeros_like(w) for w in W]
    rows = []
    it = 0
    gradnorm = 10.0

    def print_block(it, flag):
        gn = _gradnorm(G)
        rows.append((it, flag, gn, _residual(V, W)))
        return gn

    while gradnorm > tol and it <= maxiter:
        # ---- alsCP_DT_sub
        W_prev = [np.zeros_like(w) for w in W]
        while it <= maxiter:
            if it % resprint == 0 or it == maxiter:
                gradnorm = print_block(it, 0)
                if gradnorm < tol:
                    break
            W = _exact_sweep(V, W, G, lam)
            nbreak = 0
            for i in range(N):
                dW[i] = W[i] - W_prev[i]
                W_prev[i] = W[i].copy()
                if abs(np.linalg.norm(dW[i]) / np.linalg.norm(W[i])) < tol_init:
                    nbreak += 1
            if nbreak == N:
                break          # `return` BEFORE the loop increment: iter is not advanced
            it += 1
        # ---- alsCP_PP_sub
        init_iter = it
        ops, M0, W_init = {}, [None] * N, None
        while it <= maxiter:
            nbreak = sum(1 for i in range(N)
                         if abs(np.linalg.norm(dW[i]) / np.linalg.norm(W[i])) > tol_init)
            if (it - init_iter) % 15 == 0 or nbreak > 0:
                if nbreak > 0 or it != init_iter:
                    break
                W_init = [w.copy() for w in W]
                dW = [np.zeros_like(w) for w in W]
                for a in range(N):       # all pair operators and full MTTKRPs (als_CP.cxx:678-694)
                    for b in range(a + 1, N):
                        others = [j for j in range(N) if j not in (a, b)]
                        subs = [LET[:N]] + [LET[j] + "z" for j in others]
                        ops[(a, b)] = np.einsum(",".join(subs) + "->" + LET[a] + LET[b] + "z", V,
                                                *[W[j] for j in others], optimize=True)
                for a in range(N):
                    M0[a] = _mttkrp(V, W, a)
            if it % resprint == 0 or it == maxiter or it == init_iter:
                gradnorm = print_block(it, 1)
                if gradnorm < tol:
                    break
            for i in range(N):
                M = M0[i].copy()
                for ii in range(i):        # M["jk"] += T["ijk"] dW[ii]["ik"]   (als_CP.cxx:785)
                    M += np.einsum("ijk,ik->jk", ops[(ii, i)], dW[ii])
                for ii in range(i + 1, N):  # M["ik"] += T["ijk"] dW[ii]["jk"]  (als_CP.cxx:793)
                    M += np.einsum("ijk,jk->ik", ops[(i, ii)], dW[ii])
                S = _S(W, i, lam)
                G[i] = -M + W[i] @ S
                Wn = M @ _svd_inverse(S)   # SVD_solve_mod (common.cxx:739-758)
                dW[i] = ratio_step * (Wn - W_init[i])
                W[i] = W_init[i] + dW[i] if ratio_step != 1.0 else Wn
            W = _normalize(W)
            it += 1
    return it, W, G, rows


# ---------------------------------------------------------------------------- Tucker (HOOI)
def _ttmc(V, W, skip):
    """TTMc (als_Tucker.cxx:76-110): V x_j W_j^T for every mode j != skip, mode positions kept"""
    Y = V
    for j, w in enumerate(W):
        if j == skip:
            continue
        Y = np.moveaxis(np.tensordot(w.T, Y, axes=(1, j)), 0, j)
    return Y


def _top_eigvecs(Y, i, r):
    """leading left singular vectors of the mode-i unfolding through its Gram (als_Tucker.cxx:399-406,
    common.cxx:205-223), LAPACK's symmetric solver; columns sorted by descending eigenvalue"""
    Yi = np.moveaxis(Y, i, 0).reshape(Y.shape[i], -1)
    w, Q = np.linalg.eigh(Yi @ Yi.T)
    return Q[:, ::-1][:, :r].copy()


def tucker_hosvd(V, ranks):
    """hosvd (als_Tucker.cxx:12-70)"""
    W = [_top_eigvecs(V, i, r) for i, r in enumerate(ranks)]
    return W, _ttmc(V, W, -1)


def tucker_hooi(V, W, sweeps):
    """`sweeps` sweeps of alsTucker_DT (als_Tucker.cxx:340-408): every mode in turn from the TTMc of
    the others; returns the factors and the core of the last sweep (Y_end x W_{N-1}^T)"""
    W = [w.copy() for w in W]
    N = V.ndim
    core = None
    for _ in range(sweeps):
        for i in range(N):
            Y = _ttmc(V, W, i)
            W[i] = _top_eigvecs(Y, i, W[i].shape[1])
            if i == N - 1:
                core = np.moveaxis(np.tensordot(W[i].T, Y, axes=(1, i)), 0, i)
    return W, core
