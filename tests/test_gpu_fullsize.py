"""Parity at BASELINE.json's FULL sizes (configs[1] = CP order-4 s=200 R=10, configs[3] = s=400 R=20
on one GPU), through the size-independent property the domain offers: for the exact-rank `-tensor r`
input every quantity of an exact sweep has a closed form in s x R matrices
(tests/rank_structured.py, pinned against the full oracle at small sizes by
tests/test_rank_structured.py). The HIP engine is called through the C ABI on the tensor it built
in HBM; nothing here forms an s^4 object on the host."""
import numpy as np
import pytest

import rank_structured as RS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    import ppals
    return ppals


@pytest.fixture(scope="module")
def ctx(pp):
    c = pp.Context(0)
    yield c
    c.close()


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _problem(pp, s, R, N=4):
    lens = [s] * N
    A = pp.init_factors(lens, R, 1000)      # W_true
    W = pp.init_factors(lens, R, 2000)
    G = pp.init_factors(lens, R, 3000)
    return lens, A, W, G


@pytest.mark.parametrize("dtype,schedule", [(0, "msdt"), (0, "dt"), (1, "msdt")])
def test_cfg2_full_size(pp, ctx, dtype, schedule):
    """s = 200, R = 10 (6.4 GB fp32 / 12.8 GB fp64 in HBM): ||V||, MTTKRP of every mode, the
    first-level tree nodes, the streaming residual, K exact sweeps (factors within the north-star
    1e-5 for fp32 storage), the gradient norm, and both sweep schedules."""
    lens, A, W, G = _problem(pp, 200, 10)
    V = pp.Tensor(ctx, lens, dtype).fill_cp(A)
    ktol = 2e-6 if dtype == 0 else 1e-10
    assert abs(V.norm() - RS.norm(A)) < ktol * RS.norm(A)
    s = pp.CP(ctx, V, 10)
    s.set_schedule(schedule)
    s.set_factors(W, G)
    for i in range(4):
        assert relerr(s.mttkrp(i), RS.mttkrp(A, W, i)) < ktol, i
    assert relerr(s.tree_node("ab", (200, 200, 10)), RS.tree_node(A, W, [0, 1])) < ktol
    assert relerr(s.tree_node("cd", (200, 200, 10)), RS.tree_node(A, W, [2, 3])) < ktol
    assert abs(s.residual() - RS.residual(A, W)) < 10 * ktol * RS.residual(A, W)
    K = 6
    s.sweeps_dt(K)
    W_ref, G_ref = RS.als_cp_dt(A, W, G, K)
    W_got, G_got = s.get_factors(with_grad=True)
    ftol = 1e-5 if dtype == 0 else 1e-9
    for a, b in zip(W_got, W_ref):
        assert relerr(a, b) < ftol, relerr(a, b)
    gn = np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G_ref))
    assert abs(s.gradnorm() - gn) < 100 * ftol * gn
    assert abs(s.residual() - RS.residual(A, W_ref)) < 1e-4 * RS.norm(A)
    # a sweep started AT the solution stays there: the model is a fixed point of ALS
    s.set_factors(RS.normalize(A), G)
    s.sweeps_dt(2)
    assert s.residual() < (1e-6 if dtype == 0 else 1e-12) * RS.norm(A)
    s.close()
    V.close()


def test_cfg2_pp_driver_full_size(pp, ctx):
    """alsCP_PP at full size: PP phases are entered, and the run ends on the same residual floor
    as exact sweeps (fp32 storage: ~3e-8 relative)"""
    lens, A, W, G = _problem(pp, 200, 10)
    V = pp.Tensor(ctx, lens, 0).fill_cp(A)
    Vn = RS.norm(A)
    s = pp.CP(ctx, V, 10)
    s.set_factors(W, G)
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        csv = os.path.join(d, "pp.csv")
        s.run_pp(tol=1e-10 * Vn, maxiter=250, tol_init=0.01, resprint=10, csv=csv)
        rows = [ln.split(",") for ln in open(csv).read().splitlines()[1:] if ln]
    assert any(r[4] == "1" for r in rows), "PP phase never entered"
    assert float(rows[-1][5]) < 1e-6 * Vn
    W_got = s.get_factors()
    assert RS.residual(A, W_got) < 1e-6 * Vn
    s.close()
    V.close()


@pytest.mark.parametrize("dtype", [0, 1])
def test_cfg3_pp_phase_pattern_and_factors_full_size(pp, ctx, dtype, tmp_path):
    """BASELINE configs[2] (`-pp 1`, s = 200, R = 10) against alsCP_PP in closed form
    (rank_structured.als_cp_pp, pinned to the oracle at small sizes): the SAME sequence of print
    rows — iteration numbers and the DT/PP flag of every one, i.e. where each exact phase hands over
    to pairwise perturbation and where PP restarts — gradnorm / diffV trajectories while above the
    storage floor, and the final factor matrices (1e-5 for fp32 storage)."""
    lens, A, W, G = _problem(pp, 200, 10)
    V = pp.Tensor(ctx, lens, dtype).fill_cp(A)
    Vn = RS.norm(A)
    kw = dict(tol=1e-10 * Vn, tol_init=0.01, maxiter=120, resprint=1)
    rows_ref, it_ref, W_ref, _ = RS.als_cp_pp(A, W, G, **kw)
    s = pp.CP(ctx, V, 10)
    s.set_factors(W, G)
    csv = str(tmp_path / "pp.csv")
    _, it = s.run_pp(csv=csv, **kw)
    rows = [[float(x) for x in ln.split(",")] for ln in open(csv).read().splitlines()[1:] if ln]
    assert it == it_ref and len(rows) == len(rows_ref)
    assert [(int(r[1]), int(r[4])) for r in rows] == [(r[0], r[1]) for r in rows_ref]
    assert sum(1 for r in rows_ref if r[1] == 1) > len(rows_ref) // 2   # mostly PP sweeps
    floor = (2e-5 if dtype == 0 else 1e-9) * Vn
    for got, ref in zip(rows, rows_ref):
        if ref[2] > 100 * floor:
            assert abs(got[2] - ref[2]) < 2e-3 * ref[2], (got, ref)
        if ref[3] > floor:
            assert abs(got[5] - ref[3]) < 2e-3 * ref[3] + floor, (got, ref)
    for a, b in zip(s.get_factors(), W_ref):
        assert relerr(a, b) < (1e-5 if dtype == 0 else 1e-8), relerr(a, b)
    s.close()
    V.close()


def test_cfg2_long_run_fp32_factor_parity(pp, ctx):
    """north_star's bar at full size on a long run: 200 exact sweeps (s = 200, R = 10, fp32 tensor
    storage), factors within 1e-5 relative Frobenius of the fp64 closed form at sweeps 50 and 200"""
    lens, A, W, G = _problem(pp, 200, 10)
    V = pp.Tensor(ctx, lens, 0).fill_cp(A)
    s = pp.CP(ctx, V, 10)
    s.set_factors(W, G)
    W_ref, G_ref, done = W, G, 0
    for upto in (50, 200):
        W_ref, G_ref = RS.als_cp_dt(A, W_ref, G_ref, upto - done)
        s.sweeps_dt(upto - done)
        done = upto
        for a, b in zip(s.get_factors(), W_ref):
            assert relerr(a, b) < 1e-5, (upto, relerr(a, b))
    assert s.residual() < 1e-6 * RS.norm(A)
    s.close()
    V.close()


def test_cfg4_full_size(pp, ctx):
    """s = 400, R = 20 on ONE GPU (102 GB fp32 + the second resident layout): two n-tiles, 64-bit
    offsets everywhere. MTTKRPs and exact sweeps against the closed form."""
    lens, A, W, G = _problem(pp, 400, 20)
    try:
        V = pp.Tensor(ctx, lens, 0).fill_cp(A)
    except pp.PpalsError as e:  # a GPU with less free HBM than an MI355X
        pytest.skip(f"cannot hold the 102 GB tensor: {e}")
    assert abs(V.norm() - RS.norm(A)) < 2e-6 * RS.norm(A)
    s = pp.CP(ctx, V, 20)
    s.set_factors(W, G)
    for i in (0, 3):
        assert relerr(s.mttkrp(i), RS.mttkrp(A, W, i)) < 2e-6, i
    K = 3
    s.sweeps_dt(K)
    W_ref, _ = RS.als_cp_dt(A, W, G, K)
    for a, b in zip(s.get_factors(), W_ref):
        assert relerr(a, b) < 1e-5, relerr(a, b)
    s.close()
    V.close()


@pytest.mark.parametrize("roots", [1, 2])
def test_script_order6_full_size(pp, ctx, roots, monkeypatch, tmp_path):
    """the shape of the reference's own job scripts (script/*.py: -dim 6 -size 50 -rank 6; 62.5 GB
    in fp32) with one and with two modes contracted per tensor scan, against the closed form.
    Its column strides (50^k * 4 B) are not multiples of 128 B: the scans read the padded resident
    layouts (when the three copies fit) and write compact results."""
    monkeypatch.setenv("PPALS_MSDT_ROOTS", str(roots))
    trace = tmp_path / "steps.txt"
    monkeypatch.setenv("PPALS_TRACE_STEPS", str(trace))
    lens, A, W, G = _problem(pp, 50, 6, N=6)
    try:
        V = pp.Tensor(ctx, lens, 0).fill_cp(A)
    except pp.PpalsError as e:
        pytest.skip(f"cannot hold the 62.5 GB tensor: {e}")
    assert abs(V.norm() - RS.norm(A)) < 2e-6 * RS.norm(A)
    s = pp.CP(ctx, V, 6)
    s.set_factors(W, G)
    K = 4
    s.sweeps_dt(K)
    W_ref, _ = RS.als_cp_dt(A, W, G, K)
    for a, b in zip(s.get_factors(), W_ref):
        assert relerr(a, b) < 1e-5, (roots, relerr(a, b))
    assert abs(s.residual() - RS.residual(A, W_ref)) < 1e-4 * RS.norm(A)
    used = {ln.split("layout=")[1].split()[0] for ln in trace.read_text().splitlines()}
    assert "VTpad" in used, used   # (the third copy, "Vpad", only if 3 x 62.5 GB fit)
    s.close()
    V.close()


def test_cfg5_tucker_full_size(pp, ctx):
    """configs[4]: Tucker order-3 s = 400, core 20^3. Input = a CP rank-10 tensor, whose
    multilinear rank (<= 10) is below the requested core size, so HOSVD and HOOI must reproduce it
    exactly: orthonormal factors, ||core|| = ||V||, core = V x_i W_i^T (checked through the norm
    identity ||V - [[core; W]]||^2 = ||V||^2 - ||core||^2 that alsTucker_DT prints)."""
    lens, ranks = [400, 400, 400], [20, 20, 20]
    A = pp.init_factors(lens, 10, 1000)
    V = pp.Tensor(ctx, lens, 0).fill_cp(A)
    Vn = RS.norm(A)
    tk = pp.Tucker(ctx, V, ranks)
    tk.hosvd()
    W, core = tk.get_factors()
    for w in W:
        assert np.allclose(w.T @ w, np.eye(20), atol=1e-9)
    assert abs(np.linalg.norm(core) - Vn) < 1e-5 * Vn
    tk.sweeps_dt(2)
    W, core = tk.get_factors()
    for w in W:
        assert np.allclose(w.T @ w, np.eye(20), atol=1e-9)
    assert abs(np.linalg.norm(core) - Vn) < 1e-5 * Vn
    # the factor subspaces contain the true mode subspaces: projecting A_i changes nothing
    for w, a in zip(W, A):
        assert np.linalg.norm(a - w @ (w.T @ a)) < 1e-4 * np.linalg.norm(a)
    tk.close()
    V.close()
