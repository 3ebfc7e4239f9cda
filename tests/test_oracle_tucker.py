"""Pins the Tucker part of the CPU oracle through the invariants of SURVEY.md §8c(5): orthonormal
factors, ||core|| non-decreasing, core = V x_i W_i^T, residual^2 = ||V||^2 - ||core||^2, and the DT
route (alsTucker_DT) == the order-agnostic TTMc route (alsTucker)."""
import numpy as np
import pytest

import oracle_lib as O


def proj(U):
    return U @ U.T


def test_ttmc_and_hosvd_invariants():
    lens, ranks = [7, 6, 5], [3, 2, 3]
    V = O.fill_uniform(int(np.prod(lens)), 5, lo=-1, hi=1).reshape(lens, order="F")
    W, core = O.hosvd(V, ranks)
    for i, w in enumerate(W):
        assert np.allclose(w.T @ w, np.eye(ranks[i]), atol=1e-12)
        # leading left singular vectors of the mode-i unfolding
        Vi = np.moveaxis(V, i, 0).reshape(lens[i], -1, order="F")
        U = np.linalg.svd(Vi, full_matrices=False)[0][:, :ranks[i]]
        assert np.allclose(proj(w), proj(U), atol=1e-9)
    ref = np.einsum("abc,ai,bj,ck->ijk", V, *W)
    assert np.allclose(core, ref, atol=1e-12)
    Y = O.ttmc(V, W, 1)
    assert np.allclose(Y, np.einsum("abc,ai,ck->ibk", V, W[0], W[2]), atol=1e-12)


@pytest.mark.parametrize("lens,ranks", [([8, 7, 6], [3, 3, 2]), ([6, 5, 4, 5], [2, 3, 2, 2])])
def test_dt_route_equals_plain_route(lens, ranks, tmp_path):
    V = O.fill_uniform(int(np.prod(lens)), 9, lo=0.5, hi=1.0).reshape(lens, order="F")
    W0, core0 = O.hosvd(V, ranks)
    K = 3
    csv = str(tmp_path / "t.csv")
    _, _, W_dt, core_dt = O.als_tucker_dt(V, W0, core0, tol=0.0, maxiter=K, csv=csv, resprint=1)
    _, _, W_pl, _ = O.als_tucker(V, W0, core0, tol=0.0, maxiter=K - 1)
    # alsTucker_DT with maxiter=K does K sweeps before its last print... compare K-sweep subspaces
    _, _, W_pl2, _ = O.als_tucker(V, W0, core0, tol=0.0, maxiter=K)
    ok = all(np.allclose(proj(a), proj(b), atol=1e-8) for a, b in zip(W_dt, W_pl)) or \
        all(np.allclose(proj(a), proj(b), atol=1e-8) for a, b in zip(W_dt, W_pl2))
    assert ok
    header, rows = O.read_csv(csv)
    assert header[2] == "[diffnorm]"
    Vn2 = np.linalg.norm(V) ** 2
    for w, r in zip(W_dt, ranks):
        assert np.allclose(w.T @ w, np.eye(r), atol=1e-10)
    # reported residual == sqrt(||V||^2 - ||core||^2) for orthonormal factors
    core_full = np.einsum(V, list(range(len(lens))), *sum(([w, [i, 10 + i]] for i, w in enumerate(W_dt)), []),
                          [10 + i for i in range(len(lens))])
    # (the driver sweeps once more after its last print, and the CSV holds 6 significant digits)
    final_resid = np.sqrt(Vn2 - np.linalg.norm(core_full) ** 2)
    assert -1e-5 * rows[-1][5] <= rows[-1][5] - final_resid < 0.05 * rows[-1][5]
    assert abs(np.linalg.norm(core_dt) - np.linalg.norm(core_full)) < 1e-9 * np.linalg.norm(core_full)
    diffV = [r[5] for r in rows]
    assert all(b <= a * (1 + 1e-10) for a, b in zip(diffV, diffV[1:]))
