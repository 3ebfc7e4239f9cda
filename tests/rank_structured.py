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
