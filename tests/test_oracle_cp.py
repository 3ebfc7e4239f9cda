"""Pins the CPU oracle (oracle/ppals_oracle.cpp): reference-derived dimension-tree table, the
reference's internal double-route identities, numpy einsum cross-checks."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O


def rand_problem(lens, R, seed, exact=True):
    Wt = O.init_factors(lens, R, 1000 + seed)
    if exact:
        V = O.build_V(Wt)
    else:
        V = O.fill_uniform(int(np.prod(lens)), 77 + seed, lo=0.5, hi=1.0).reshape(lens, order="F")
    W = O.init_factors(lens, R, 2000 + seed)
    return V, W


def einsum_mttkrp(V, Ws, mode):
    N = V.ndim
    letters = "abcdefgh"[:N]
    ops, subs = [V], [letters]
    for j in range(N):
        if j != mode:
            ops.append(Ws[j])
            subs.append(letters[j] + "z")
    return np.einsum(",".join(subs) + "->" + letters[mode] + "z", *ops, optimize=True)


@pytest.mark.parametrize("N", range(2, 9))
def test_dimension_tree_matches_reference(N, golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "dimension_tree.json")))["trees"][str(N)]
    assert O.dimension_tree(N) == gold


def test_fill_uniform_is_counter_based():
    a = O.fill_uniform(1000, 5)
    b = O.fill_uniform(400, 5, offset=600)
    assert np.array_equal(a[600:], b)
    assert 0.0 <= a.min() and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.05


def test_build_V_matches_einsum():
    Ws = O.init_factors([5, 6, 7, 4], 3, 1)
    V = O.build_V(Ws)
    ref = np.einsum("az,bz,cz,dz->abcd", *Ws)
    assert np.allclose(V, ref, rtol=1e-13, atol=1e-13)
    assert O.residual(V, Ws) < 1e-12
    assert abs(O.residual(V + 1.0, Ws) - np.sqrt(V.size)) < 1e-9


@pytest.mark.parametrize("lens,R", [([6, 5, 4], 3), ([5, 6, 4, 7], 3), ([4, 3, 5, 2, 3], 2),
                                    ([3, 4, 2, 3, 2, 3], 2)])
def test_dt_route_equals_naive_route(lens, R):
    """DT MTTKRP == KhatriRao_contract for every mode"""
    V, W = rand_problem(lens, R, 3, exact=False)
    for mode in range(len(lens)):
        a = O.mttkrp(V, W, mode, 0)
        b = O.mttkrp(V, W, mode, 1)
        c = einsum_mttkrp(V, W, mode)
        assert np.allclose(a, c, rtol=1e-12, atol=1e-12)
        assert np.allclose(b, c, rtol=1e-12, atol=1e-12)


def test_tree_node_and_pp_operator():
    lens, R = [5, 4, 6, 3], 3
    V, W = rand_problem(lens, R, 4, exact=False)
    T_ab = O.tree_node(V, W, "ab")
    assert np.allclose(T_ab, np.einsum("abcd,cz,dz->abz", V, W[2], W[3]))
    T_cd = O.tree_node(V, W, "cd")
    assert np.allclose(T_cd, np.einsum("abcd,az,bz->cdz", V, W[0], W[1]))
    T_ac = O.pp_operator(V, W, "bd")
    assert np.allclose(T_ac, np.einsum("abcd,bz,dz->acz", V, W[1], W[3]))
    M_b = O.pp_operator(V, W, "acd")
    assert np.allclose(M_b, einsum_mttkrp(V, W, 1))


def test_gram_solve_normalize():
    lens, R = [7, 6, 5, 8], 4
    W = O.init_factors(lens, R, 9)
    S = O.gram_hadamard(W, 1, 0.25)
    ref = (W[0].T @ W[0]) * (W[2].T @ W[2]) * (W[3].T @ W[3]) + 0.25 * np.eye(R)
    assert np.allclose(S, ref, rtol=1e-13)
    M = O.fill_uniform(6 * R, 3).reshape((6, R), order="F")
    X = O.svd_solve(M, S)
    assert np.allclose(X @ S, M, rtol=1e-9, atol=1e-11)
    Wn = O.normalize(W)
    norms = [np.linalg.norm(w) for w in Wn]
    assert np.allclose(norms, norms[0], rtol=1e-13)
    assert np.allclose(np.einsum("az,bz,cz,dz->abcd", *Wn), np.einsum("az,bz,cz,dz->abcd", *W),
                       rtol=1e-12)


def test_svd():
    A = O.fill_uniform(12 * 5, 11, lo=-1, hi=1).reshape((12, 5), order="F")
    U, s, Vm = O.svd(A)
    assert np.allclose(U * s @ Vm.T, A, atol=1e-13)
    assert np.allclose(s, np.linalg.svd(A, compute_uv=False), rtol=1e-12)
    assert np.allclose(U.T @ U, np.eye(5), atol=1e-13)


def test_dt_sweep_equals_plain_als_sweep():
    """one alsCP_DT sweep == one alsCP sweep (both routes share S, solve, Normalize)"""
    lens, R = [6, 5, 7, 4], 3
    V, W = rand_problem(lens, R, 5)
    G = O.init_factors(lens, R, 99)
    _, _, W_dt, _ = O.als_cp_dt(V, W, G, tol=0.0, maxiter=0, resprint=1000)
    _, _, W_pl, _ = O.als_cp(V, W, G, tol=0.0, maxiter=0)
    # maxiter=0 performs exactly one sweep in both drivers
    for a, b in zip(W_dt, W_pl):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-12)


def test_exact_rank_recovery_and_csv(tmp_path):
    lens, R = [8, 7, 6, 5], 2
    V, W = rand_problem(lens, R, 6)
    G = O.init_factors(lens, R, 55)
    csv = str(tmp_path / "dt.csv")
    Vnorm = np.linalg.norm(V)
    rc, iters, Wn, Gn = O.als_cp_dt(V, W, G, tol=1e-10 * Vnorm, maxiter=300, csv=csv, resprint=10)
    header, rows = O.read_csv(csv)
    assert header == ["[dim]", "[iter]", "[gradnorm]", "[tol]", "[pp_update]", "[diffV]", "[dtime]"]
    assert rows[0][0] == 8 and rows[0][1] == 0
    # iter 0 reports the norm of the *initial* grad_W (test_ALS.cxx:338, als_CP.cxx:174-181)
    assert abs(rows[0][2] - np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G))) < 1e-4 * rows[0][2]
    diffs = [r[5] for r in rows]
    assert diffs[-1] < 1e-6 * Vnorm
    assert all(d2 <= d1 * (1 + 1e-9) for d1, d2 in zip(diffs, diffs[1:]))
    assert O.residual(V, Wn) < 1e-6 * Vnorm


def test_pp_identities():
    lens, R = [6, 5, 4, 7], 3
    N = 4
    V, W = rand_problem(lens, R, 8, exact=False)
    # dW = 0: PP MTTKRP equals the exact MTTKRP
    for i in range(N):
        key = "".join(chr(97 + m) for m in range(N) if m != i)
        assert np.allclose(O.pp_operator(V, W, key), O.mttkrp(V, W, i, 0), rtol=1e-12)

    # first-order PP correction is O(|dW|^2) accurate: halving dW quarters the error
    def pp_error(scale):
        dW = [scale * O.fill_uniform(s * R, 40 + j, lo=-1, hi=1).reshape((s, R), order="F")
              for j, s in enumerate(lens)]
        Wp = [w + d for w, d in zip(W, dW)]
        i = 1
        M = O.pp_operator(V, W, "acd").copy()
        for j in range(N):
            if j == i:
                continue
            key = "".join(chr(97 + m) for m in range(N) if m not in (i, j))
            T = O.pp_operator(V, W, key)
            M += np.einsum("xyz,yz->xz", T, dW[j]) if j > i else np.einsum("yxz,yz->xz", T, dW[j])
        return np.linalg.norm(M - O.mttkrp(V, Wp, i, 0))

    e1, e2 = pp_error(1e-2), pp_error(5e-3)
    assert 3.5 < e1 / e2 < 4.5


def test_pp_driver_converges_like_dt(tmp_path):
    lens, R = [9, 8, 7, 6], 2
    V, W = rand_problem(lens, R, 12)
    G = O.init_factors(lens, R, 56)
    Vnorm = np.linalg.norm(V)
    csv = str(tmp_path / "pp.csv")
    rc, iters, Wn, _ = O.als_cp_pp(V, W, G, tol=1e-8 * Vnorm, tol_init=0.1, maxiter=400, csv=csv,
                                   resprint=1)
    _, rows = O.read_csv(csv)
    assert any(r[4] == 1 for r in rows), "PP phase never entered"
    assert O.residual(V, Wn) < 1e-5 * Vnorm


def test_sort_indexes_matches_reference(golden_dir):
    """the update order of alsCP_PP_partupdate: the oracle's rule against the reference's own
    STL-only sort_indexes (oracle/_ref/sortidx_ref -> tests/golden/sort_indexes.json)"""
    import json
    import os
    cases = json.load(open(os.path.join(golden_dir, "sort_indexes.json")))["cases"]
    assert len(cases) >= 40
    for c in cases:
        assert O.sort_indexes(c["v"]) == c["order"], c


def test_numpy_tucker_reading_matches_the_oracle():
    """tests/numpy_ref.py's HOOI (einsum + LAPACK eigh) against the C++ oracle (Jacobi SVD) on a small
    problem: the large-mode GPU tests (rank 100 on 1344 rows) use the numpy reading, which the
    oracle's O(J^3 sweeps) solver cannot serve in test time"""
    import numpy_ref as NR
    lens, ranks = [14, 11, 9], [4, 3, 2]
    V = O.fill_uniform(int(np.prod(lens)), 7, lo=0.5, hi=1.0).reshape(lens, order="F")
    W0, c0 = O.hosvd(V, ranks)
    Wn, cn = NR.tucker_hosvd(V, ranks)
    for a, b in zip(W0, Wn):
        assert np.linalg.norm(a @ a.T - b @ b.T) < 1e-9
    assert abs(np.linalg.norm(c0) - np.linalg.norm(cn)) < 1e-10 * np.linalg.norm(c0)
    _, _, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=4, resprint=10 ** 9)
    W_np, core_np = NR.tucker_hooi(V, W0, 5)
    for a, b in zip(W_ref, W_np):
        assert np.linalg.norm(a @ a.T - b @ b.T) < 1e-9
    assert abs(np.linalg.norm(core_ref) - np.linalg.norm(core_np)) < 1e-10 * np.linalg.norm(core_ref)
