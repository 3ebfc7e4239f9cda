"""Committed golden vectors (tests/golden/oracle_golden.npz, made by make_oracle_golden.py):
the oracle must keep reproducing them (CPU), and the HIP engine must match them through the C ABI
(GPU) — fp64 storage to 1e-8, fp32 storage to the north-star 1e-5 on factor matrices."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_oracle_golden as MG  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                            "oracle_golden.npz"))


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_oracle_reproduces_golden_cp(name):
    lens, R, seed = MG.CASES[name]
    V, W, G = MG.cp_problem(lens, R, seed)
    for mode in range(len(lens)):
        assert relerr(O.mttkrp(V, W, mode, 1), GOLD[f"{name}/mttkrp{mode}"]) < 1e-12
    _, _, W5, _ = O.als_cp_dt(V, W, G, tol=0.0, maxiter=4, resprint=10 ** 6)
    for i in range(len(lens)):
        assert relerr(W5[i], GOLD[f"{name}/dt5_W{i}"]) < 1e-10


def test_oracle_reproduces_golden_tucker():
    for name, (lens, ranks, seed) in MG.TUCKER.items():
        V = O.fill_uniform(int(np.prod(lens)), seed, lo=0.5, hi=1.0).reshape(lens, order="F")
        W0, c0 = O.hosvd(V, ranks)
        for i, w in enumerate(W0):
            assert np.linalg.norm(w @ w.T - GOLD[f"{name}/hosvd_P{i}"]) < 1e-9
        assert abs(np.linalg.norm(c0) - GOLD[f"{name}/hosvd_corenorm"][0]) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_engine_matches_golden_cp(name, dtype, tmp_path):
    import ppals
    lens, R, seed = MG.CASES[name]
    V, W, G = MG.cp_problem(lens, R, seed)
    ctx = ppals.Context(0)
    t = ppals.Tensor(ctx, lens, dtype).upload(V)
    s = ppals.CP(ctx, t, R)
    s.set_factors(W, G)
    for mode in range(len(lens)):
        assert relerr(s.mttkrp(mode), GOLD[f"{name}/mttkrp{mode}"]) < (2e-6 if dtype == 0 else 1e-11)
    s.sweeps_dt(5)
    Wg = s.get_factors()
    for i in range(len(lens)):
        assert relerr(Wg[i], GOLD[f"{name}/dt5_W{i}"]) < (1e-5 if dtype == 0 else 1e-8)
    assert abs(s.gradnorm() - GOLD[f"{name}/dt5_gradnorm"][0]) < 1e-3 * GOLD[f"{name}/dt5_gradnorm"][0] + 1e-9
    if dtype == 1:
        csv = str(tmp_path / "pp.csv")
        Vn = np.linalg.norm(V)
        s.set_factors(W, G)
        rc, it = s.run_pp(tol=1e-7 * Vn, tol_init=0.1, maxiter=40, csv=csv, resprint=1)
        _, rows = O.read_csv(csv)
        gold = GOLD[f"{name}/pp_rows"]
        assert it == int(GOLD[f"{name}/pp_iters"][0]) and len(rows) == len(gold)
        for r, g in zip(rows, gold):
            assert r[1] == g[0] and r[4] == g[1]
            assert abs(r[5] - g[2]) <= 1e-4 * abs(g[2]) + 1e-9 * Vn
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [0, 1])
def test_engine_matches_golden_tucker(dtype):
    import ppals
    for name, (lens, ranks, seed) in MG.TUCKER.items():
        V = O.fill_uniform(int(np.prod(lens)), seed, lo=0.5, hi=1.0).reshape(lens, order="F")
        ctx = ppals.Context(0)
        t = ppals.Tensor(ctx, lens, dtype).upload(V)
        s = ppals.Tucker(ctx, t, ranks)
        s.hosvd()
        W, core = s.get_factors()
        tol = 1e-3 if dtype == 0 else 1e-7
        for i, w in enumerate(W):
            assert np.linalg.norm(w @ w.T - GOLD[f"{name}/hosvd_P{i}"]) < tol
        assert abs(np.linalg.norm(core) - GOLD[f"{name}/hosvd_corenorm"][0]) < tol * np.linalg.norm(core)
        s.sweeps_dt(4)
        W, core = s.get_factors()
        for i, w in enumerate(W):
            assert np.linalg.norm(w @ w.T - GOLD[f"{name}/dt_P{i}"]) < tol * 10
        ctx.close()
