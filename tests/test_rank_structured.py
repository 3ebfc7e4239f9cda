"""The closed-form checker for exact-rank tensors (tests/rank_structured.py) against the full
oracle at sizes where the oracle can form the tensor."""
import numpy as np
import pytest

import oracle_lib as O
import rank_structured as RS


@pytest.mark.parametrize("lens,R", [([7, 6, 5, 4], 3), ([9, 8, 7], 2), ([5, 4, 6, 3, 4], 2)])
def test_closed_forms_match_oracle(lens, R):
    A = O.init_factors(lens, R, 1000)
    W = O.init_factors(lens, R, 2000)
    G = O.init_factors(lens, R, 3000)
    V = O.build_V(A)
    assert abs(RS.norm(A) - np.linalg.norm(V)) < 1e-12 * np.linalg.norm(V)
    assert abs(RS.residual(A, W) - O.residual(V, W)) < 1e-10 * O.residual(V, W)
    for i in range(len(lens)):
        assert np.allclose(RS.mttkrp(A, W, i), O.mttkrp(V, W, i, 0), rtol=1e-11, atol=1e-12)
    if len(lens) == 4:
        assert np.allclose(RS.tree_node(A, W, [0, 1]), O.tree_node(V, W, "ab"), rtol=1e-11)
        assert np.allclose(RS.tree_node(A, W, [2, 3]), O.tree_node(V, W, "cd"), rtol=1e-11)
    K = 4
    _, _, W_ref, G_ref = O.als_cp_dt(V, W, G, tol=0.0, maxiter=K - 1, resprint=1000)
    W_cf, G_cf = RS.als_cp_dt(A, W, G, K)
    for a, b in zip(W_cf, W_ref):
        assert np.linalg.norm(a - b) < 1e-9 * np.linalg.norm(b)
    for a, b in zip(G_cf, G_ref):
        assert np.linalg.norm(a - b) < 1e-8 * (1 + np.linalg.norm(b))


@pytest.mark.parametrize("lens,R,ratio", [([12, 11, 10, 9], 3, 1.0), ([14, 12, 10], 3, 1.0),
                                          ([7, 6, 6, 5, 5], 2, 1.0), ([12, 11, 10, 9], 3, 0.8)])
def test_closed_form_pp_driver_matches_oracle(lens, R, ratio, tmp_path):
    """alsCP_PP in closed form (the checker of BASELINE configs[2] at s = 200) against the oracle's
    literal restatement: the same print rows (iteration, DT/PP flag, gradnorm, diffV), final
    iteration count and factors"""
    A = O.init_factors(lens, R, 1005)
    W = O.init_factors(lens, R, 2005)
    G = O.init_factors(lens, R, 97)
    V = O.build_V(A)
    Vn = np.linalg.norm(V)
    csv = str(tmp_path / "ref.csv")
    kw = dict(tol=1e-7 * Vn, tol_init=0.1, maxiter=45, resprint=1)
    _, it_ref, W_ref, G_ref = O.als_cp_pp(V, W, G, ratio_step=ratio, csv=csv, **kw)
    rows, it, W_cf, G_cf = RS.als_cp_pp(A, W, G, ratio_step=ratio, **kw)
    _, ref_rows = O.read_csv(csv)
    assert it == it_ref and len(rows) == len(ref_rows)
    assert any(r[1] == 1 for r in rows)
    for a, b in zip(rows, ref_rows):
        assert (a[0], a[1]) == (b[1], b[4])
        assert abs(a[2] - b[2]) <= 2e-5 * abs(b[2]) + 1e-9 * Vn        # CSV keeps 6 digits
        assert abs(a[3] - b[5]) <= 2e-5 * abs(b[5]) + 1e-7 * Vn
    for a, b in zip(W_cf, W_ref):
        assert np.linalg.norm(a - b) < 1e-7 * np.linalg.norm(b)
