"""The oracle's CP drivers against an independent numpy/LAPACK evaluation of the reference's own
index expressions (tests/numpy_ref.py): print rows (iteration, phase flag), gradient norms,
residuals and final factors of alsCP_DT and alsCP_PP."""
import numpy as np
import pytest

import numpy_ref as NR
import oracle_lib as O


def _problem(lens, R, seed):
    V = O.build_V(O.init_factors(lens, R, 1000 + seed))
    return V, O.init_factors(lens, R, 2000 + seed), O.init_factors(lens, R, 3000 + seed)


@pytest.mark.parametrize("lens,R", [([8, 7, 6, 5], 3), ([9, 8, 7], 2), ([5, 4, 4, 3, 4], 2)])
def test_dt_driver(lens, R, tmp_path):
    V, W, G = _problem(lens, R, 1)
    csv = str(tmp_path / "dt.csv")
    _, it_o, W_o, G_o = O.als_cp_dt(V, W, G, tol=1e-9 * np.linalg.norm(V), maxiter=25, csv=csv,
                                    resprint=5)
    it_n, W_n, G_n, rows_n = NR.als_cp_dt(V, W, G, 1e-9 * np.linalg.norm(V), 25, resprint=5)
    _, rows_o = O.read_csv(csv)
    assert it_o == it_n
    assert [(int(r[1]), int(r[4])) for r in rows_o] == [(r[0], r[1]) for r in rows_n]
    for a, b in zip(rows_o, rows_n):
        assert abs(a[2] - b[2]) <= 1e-5 * abs(b[2]) + 1e-9      # CSV keeps 6 significant digits
        assert abs(a[5] - b[3]) <= 1e-5 * abs(b[3]) + 1e-9
    for a, b in zip(W_o, W_n):
        assert np.linalg.norm(a - b) < 1e-8 * np.linalg.norm(b)
    for a, b in zip(G_o, G_n):
        assert np.linalg.norm(a - b) < 1e-7 * (1 + np.linalg.norm(b))


@pytest.mark.parametrize("lens,R,tol_init", [([8, 7, 6, 5], 3, 0.1), ([9, 8, 7, 6], 2, 0.05),
                                             ([6, 5, 5, 4, 4], 2, 0.1)])
def test_pp_driver(lens, R, tol_init, tmp_path):
    V, W, G = _problem(lens, R, 2)
    Vn = np.linalg.norm(V)
    csv = str(tmp_path / "pp.csv")
    _, it_o, W_o, _ = O.als_cp_pp(V, W, G, tol=1e-7 * Vn, tol_init=tol_init, maxiter=40, csv=csv,
                                  resprint=1)
    it_n, W_n, _, rows_n = NR.als_cp_pp(V, W, G, 1e-7 * Vn, tol_init, 40, resprint=1)
    _, rows_o = O.read_csv(csv)
    assert any(r[1] == 1 for r in rows_n), "PP phase never entered"
    assert it_o == it_n
    assert [(int(r[1]), int(r[4])) for r in rows_o] == [(r[0], r[1]) for r in rows_n]
    for a, b in zip(rows_o, rows_n):
        assert abs(a[5] - b[3]) <= 1e-4 * abs(b[3]) + 1e-7 * Vn
    for a, b in zip(W_o, W_n):
        assert np.linalg.norm(a - b) < 1e-6 * np.linalg.norm(b)
