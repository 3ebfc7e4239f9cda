"""Closed-form ALS on exact-rank tensors (TEST INFRASTRUCTURE) — the size-independent checker for
BASELINE's full sizes.

For V = [[A_0, ..., A_{N-1}]] (the `-tensor r` input, test_ALS.cxx:275-286) every quantity of an
exact sweep is a function of s x R and R x R matrices only:

    MTTKRP      M_i = A_i * Hadamard_{j != i} (A_j^T W_j)
    tree node   T[keep modes, n] = KhatriRao_{j kept}(A_j) * Hadamard_{j contracted} (A_j^T W_j)
    <[[A]],[[B]]> = sum_{r,s} prod_j (A_j^T B_j)[r,s]         (norms, residual)

so alsCP_DT (als_CP.cxx:215-303: S, gradient with the pre-update W, W = M S^-1 through the SVD
of S, Normalize) can be iterated in fp64 numpy at any s in milliseconds. tests/test_rank_structured.py
pins this file against the full oracle at small sizes; tests/test_gpu_fullsize.py uses it at
s = 200 and s = 400."""
import numpy as np


def inner(A, B):
    """<[[A]], [[B]]>"""
    H = np.ones((A[0].shape[1], B[0].shape[1]))
    for a, b in zip(A, B):
        H = H * (a.T @ b)
    return float(H.sum())


def norm(A):
    return float(np.sqrt(max(inner(A, A), 0.0)))


def residual(A, W):
    """||[[A]] - [[W]]||_F (als_CP.cxx:183-187) without forming either tensor"""
    return float(np.sqrt(max(inner(A, A) + inner(W, W) - 2.0 * inner(A, W), 0.0)))


def mttkrp(A, W, i):
    R = W[0].shape[1]
    H = np.ones((A[0].shape[1], R))
    for j in range(len(A)):
        if j != i:
            H = H * (A[j].T @ W[j])
    return A[i] @ H


def tree_node(A, W, keep):
    """T[keep..., n]: V contracted with W_j for every mode j not in `keep` (ascending)"""
    R = W[0].shape[1]
    H = np.ones((A[0].shape[1], R))
    for j in range(len(A)):
        if j not in keep:
            H = H * (A[j].T @ W[j])
    K = None  # Khatri-Rao of the kept true factors, first kept mode fastest
    for j in keep:
        K = A[j] if K is None else (A[j][:, None, :] * K[None, :, :]).reshape(-1, A[j].shape[1])
    return (K @ H).reshape([A[j].shape[0] for j in keep] + [R], order="F")


def svd_inverse(S):
    """SVD_solve's S_reverse = V diag(1/s) U^T, no truncation (common.cxx:717-722)"""
    U, s, Vt = np.linalg.svd(S)
    return Vt.T @ np.diag(1.0 / s) @ U.T


def normalize(W):
    """Normalize (common.cxx:680-688)"""
    N = len(W)
    nrm = 1.0
    for w in W:
        nrm = nrm * np.linalg.norm(w)
    nrm = nrm ** (1.0 / N)
    return [w * (nrm / np.linalg.norm(w)) for w in W]


def als_cp_dt(A, W, gradW, sweeps, lam=0.0):
    """`sweeps` iterations of alsCP_DT's loop body. Returns (W, gradW)."""
    W = [w.copy() for w in W]
    G = [g.copy() for g in gradW]
    N, R = len(W), W[0].shape[1]
    for _ in range(sweeps):
        for i in range(N):
            M = mttkrp(A, W, i)
            S = np.ones((R, R))
            for j in range(N):
                if j != i:
                    S = S * (W[j].T @ W[j])
            S = S + lam * np.eye(R)
            G[i] = -M + W[i] @ S
            W[i] = M @ svd_inverse(S)
        W = normalize(W)
    return W, G


def _sweep_body(A, W, G, lam):
    """one pass of the mode loop of alsCP_DT / alsCP_DT_sub (als_CP.cxx:217-302 / :499-590)"""
    N, R = len(W), W[0].shape[1]
    for i in range(N):
        M = mttkrp(A, W, i)
        S = np.ones((R, R))
        for j in range(N):
            if j != i:
                S = S * (W[j].T @ W[j])
        S = S + lam * np.eye(R)
        G[i] = -M + W[i] @ S
        W[i] = M @ svd_inverse(S)


def als_cp_pp(A, W, gradW, tol, tol_init, maxiter, resprint=10, ratio_step=1.0, lam=0.0):
    """alsCP_PP (als_CP.cxx:1082-1137) with alsCP_DT_sub (:418-612) and alsCP_PP_sub (:621-833) in
    closed form. For V = [[A]] the PP cache is never formed: with H_ij = Hadamard_{m != i,j}
    (A_m^T W_m^init),
        pair operator    T_ij[x,y,r] = sum_k A_i[x,k] A_j[y,k] H_ij[k,r]          (:352-409)
        correction       (T_ij x_j dW_j)[x,r] = (A_i ((A_j^T dW_j) * H_ij))[x,r]   (:778-794)
        M_i^0            = A_i Hadamard_{m != i} (A_m^T W_m^init)
    Returns (rows, iter, W, gradW); rows = (iter, pp_update, gradnorm, diffV) of every print."""
    W = [w.copy() for w in W]
    G = [g.copy() for g in gradW]
    N, R = len(W), W[0].shape[1]
    rows = []
    dW = [np.zeros_like(w) for w in W]
    it, gradnorm = 0, 10.0

    def print_block(pp_flag):
        gn = float(np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G)))
        rows.append((it, pp_flag, gn, residual(A, W)))
        return gn

    while gradnorm > tol and it <= maxiter:
        # ---- alsCP_DT_sub
        W_prev = [np.zeros_like(w) for w in W]
        while it <= maxiter:
            if it % resprint == 0 or it == maxiter:
                gradnorm = print_block(0)
                if gradnorm < tol:
                    break
            _sweep_body(A, W, G, lam)
            W[:] = normalize(W)
            nbreak = 0
            for i in range(N):
                dW[i] = W[i] - W_prev[i]
                W_prev[i] = W[i].copy()
                if abs(np.linalg.norm(dW[i]) / np.linalg.norm(W[i])) < tol_init:
                    nbreak += 1
            if nbreak == N:
                break            # returns WITHOUT incrementing iter (:604-605)
            it += 1
        # ---- alsCP_PP_sub
        init_it = it
        W_init, AtW = None, None
        while it <= maxiter:
            nbreak = sum(1 for i in range(N)
                         if abs(np.linalg.norm(dW[i]) / np.linalg.norm(W[i])) > tol_init)
            if (it - init_it) % 15 == 0 or nbreak > 0:
                if nbreak > 0 or it != init_it:
                    break
                W_init = [w.copy() for w in W]
                dW = [np.zeros_like(w) for w in W]
                AtW = [A[m].T @ W_init[m] for m in range(N)]
            if it % resprint == 0 or it == maxiter or it == init_it:
                gradnorm = print_block(1)
                if gradnorm < tol:
                    break
            for i in range(N):
                H = np.ones((A[0].shape[1], R))
                for m in range(N):
                    if m != i:
                        H = H * AtW[m]
                M = A[i] @ H
                for j in range(N):
                    if j == i:
                        continue
                    Hij = np.ones((A[0].shape[1], R))
                    for m in range(N):
                        if m != i and m != j:
                            Hij = Hij * AtW[m]
                    M = M + A[i] @ ((A[j].T @ dW[j]) * Hij)
                S = np.ones((R, R))
                for j in range(N):
                    if j != i:
                        S = S * (W[j].T @ W[j])
                S = S + lam * np.eye(R)
                G[i] = -M + W[i] @ S
                W[i] = M @ svd_inverse(S)
                dW[i] = ratio_step * (W[i] - W_init[i])      # SVD_solve_mod, common.cxx:753-756
                if ratio_step != 1.0:
                    W[i] = W_init[i] + dW[i]
            W[:] = normalize(W)                               # dW is NOT rescaled (:824-825)
            it += 1
    return rows, it, W, G
