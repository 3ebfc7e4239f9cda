"""GPU parity tests of the Tucker (HOOI) path: TTMc (K11), Gram of the unfolding + leading
eigenvectors (K12/K13), hosvd, alsTucker_DT — against the fp64 oracle. Eigenvector signs are
LAPACK-defined in the reference, so factors are compared as subspaces (projectors)."""
import os

import numpy as np
import pytest

import oracle_lib as O

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


def proj(U):
    return U @ U.T


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def core_from_factors(V, W):
    """V x_i W_i^T entry by entry (als_Tucker.cxx:408): what the returned core must be for the RETURNED
    factors — a permuted or transposed core has the same norm"""
    core = V
    for m, w in enumerate(W):
        core = np.moveaxis(np.tensordot(w.T, core, axes=(1, m)), 0, m)
    return core


CASES = [([12, 10, 9], [3, 4, 2]), ([9, 8, 7, 6], [3, 2, 3, 2]), ([20, 16, 24], [5, 5, 5])]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("lens,ranks", CASES)
def test_ttmc_matches_oracle(pp, ctx, lens, ranks, dtype):
    V = O.fill_uniform(int(np.prod(lens)), 3, lo=0.5, hi=1.0).reshape(lens, order="F")
    W = [np.linalg.qr(O.fill_uniform(s * r, 40 + i, lo=-1, hi=1).reshape((s, r), order="F"))[0]
         for i, (s, r) in enumerate(zip(lens, ranks))]
    t = pp.Tensor(ctx, lens, dtype).upload(V)
    s = pp.Tucker(ctx, t, ranks)
    s.set_factors(W)
    for skip in [-1] + list(range(len(lens))):
        got = s.ttmc(skip)
        want = O.ttmc(V, W, skip)
        assert relerr(got, want) < (2e-6 if dtype == 0 else 1e-11), (skip, relerr(got, want))


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("lens,ranks", CASES)
def test_hosvd_and_dt_sweeps(pp, ctx, lens, ranks, dtype, tmp_path):
    V = O.fill_uniform(int(np.prod(lens)), 4, lo=0.5, hi=1.0).reshape(lens, order="F")
    tol = 1e-4 if dtype == 0 else 1e-8
    t = pp.Tensor(ctx, lens, dtype).upload(V)
    s = pp.Tucker(ctx, t, ranks)
    s.hosvd()
    W, core = s.get_factors()
    W_ref, core_ref = O.hosvd(V, ranks)
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
        assert np.linalg.norm(proj(a) - proj(b)) < tol * 10
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < tol * np.linalg.norm(core_ref)
    # full driver from the same (oracle) initialisation
    s.set_factors(W_ref)
    c_ref, c_got = str(tmp_path / "r.csv"), str(tmp_path / "g.csv")
    _, it_ref, W_dt_ref, core_dt_ref = O.als_tucker_dt(V, W_ref, core_ref, tol=0.0, maxiter=4,
                                                       csv=c_ref, resprint=1)
    rc, it = s.run_dt(tol=0.0, maxiter=4, csv=c_got, resprint=1)
    W_dt, core_dt = s.get_factors()
    assert it == it_ref
    for a, b in zip(W_dt, W_dt_ref):
        assert np.linalg.norm(proj(a) - proj(b)) < tol * 100
    # orders 3 and 4: the returned core is V x_i W_i^T of the returned factors, entry by entry
    want = core_from_factors(V, W_dt)
    assert core_dt.shape == want.shape
    assert np.abs(core_dt - want).max() < (2e-6 if dtype == 0 else 1e-10) * np.abs(want).max()
    h1, r1 = O.read_csv(c_ref)
    h2, r2 = O.read_csv(c_got)
    assert h1 == h2 and len(r1) == len(r2)
    for a, b in zip(r1, r2):
        assert a[1] == b[1]
        assert abs(a[5] - b[5]) < (1e-4 if dtype == 0 else 1e-8) * np.linalg.norm(V)


def test_large_mode_uses_vendor_eig(pp, ctx):
    lens, ranks = [72, 10, 9], [4, 3, 3]
    V = O.fill_uniform(int(np.prod(lens)), 8, lo=0.5, hi=1.0).reshape(lens, order="F")
    t = pp.Tensor(ctx, lens, 1).upload(V)
    s = pp.Tucker(ctx, t, ranks)
    s.hosvd()
    W, core = s.get_factors()
    W_ref, core_ref = O.hosvd(V, ranks)
    for a, b in zip(W, W_ref):
        assert np.linalg.norm(proj(a) - proj(b)) < 1e-7


@pytest.mark.parametrize("dtype", [0, 1])
def test_tucker_pp_driver_matches_oracle(pp, ctx, dtype, tmp_path):
    """alsTucker_PP (als_Tucker.cxx:906-962): DT/PP phase pattern, residual trajectory and final
    subspaces vs the oracle's restatement"""
    lens, ranks = [10, 9, 8], [3, 3, 2]
    V = O.fill_uniform(int(np.prod(lens)), 4, lo=0.5, hi=1.0).reshape(lens, order="F")
    W0, c0 = O.hosvd(V, ranks)
    c_ref, c_got = str(tmp_path / "r.csv"), str(tmp_path / "g.csv")
    kw = dict(tol=0.0, tol_init=0.1, maxiter=12, resprint=1)
    _, it_ref, W_ref, core_ref = O.als_tucker_pp(V, W0, c0, csv=c_ref, **kw)
    t = pp.Tensor(ctx, lens, dtype).upload(V)
    s = pp.Tucker(ctx, t, ranks)
    s.hosvd()          # fills the core; then start from the oracle's factors
    s.set_factors(W0)
    rc, it = s.run_pp(csv=c_got, **kw)
    _, r1 = O.read_csv(c_ref)
    _, r2 = O.read_csv(c_got)
    assert any(r[4] == 1 for r in r2), "PP phase never entered"
    if dtype == 1:
        assert it == it_ref and [r[1:2] + r[4:5] for r in r1] == [r[1:2] + r[4:5] for r in r2]
    n = min(len(r1), len(r2))
    for a, b in zip(r1[:n], r2[:n]):
        assert abs(a[5] - b[5]) < (1e-4 if dtype == 0 else 1e-7) * np.linalg.norm(V)
    W, core = s.get_factors()
    for a, b in zip(W, W_ref):
        assert np.linalg.norm(proj(a) - proj(b)) < (5e-2 if dtype == 0 else 1e-5)


@pytest.mark.parametrize("maxiter,resprint", [(1, 1), (3, 2)])
def test_tucker_bench_mode_matches_oracle(pp, ctx, maxiter, resprint, tmp_path):
    """`bool bench = true` of alsTucker_DT / alsTucker_PP as pp_bench.cxx:321-345 drives them: random
    (non-orthonormal) factors — hosvd is commented out there — a zero core on the first call, the
    SAME core object carried from call to call, factors restored before each. [DTtime] /
    [PPfirst] / [PPsecond] labels, iteration counts, return values, projectors and ||core||."""
    lens, ranks = [12, 10, 9], [3, 3, 3]
    V = O.fill_uniform(int(np.prod(lens)), 4, lo=-1.0, hi=1.0).reshape(lens, order="F")
    W0 = [O.fill_uniform(s * r, 50 + i).reshape((s, r), order="F")
          for i, (s, r) in enumerate(zip(lens, ranks))]
    t = pp.Tensor(ctx, lens, 1).upload(V)
    s = pp.Tucker(ctx, t, ranks)
    core_ref = np.zeros(ranks)
    kw = dict(tol=1e-12, maxiter=maxiter, resprint=resprint)
    for rep, phase in enumerate(["dt", "dt", "pp", "pp"]):
        c_ref, c_got = str(tmp_path / f"ref{rep}.csv"), str(tmp_path / f"got{rep}.csv")
        for c in (c_ref, c_got):
            open(c, "w").write("[timetype],[dtime]\n")
        s.set_factors(W0)
        if phase == "dt":
            rc_ref, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, core_ref, csv=c_ref, bench=1, **kw)
            rc, it = s.run_dt(csv=c_got, csv_append=1, bench=1, **kw)
        else:
            rc_ref, it_ref, W_ref, core_ref = O.als_tucker_pp(V, W0, core_ref, tol_init=0.05,
                                                              csv=c_ref, bench=1, **kw)
            rc, it = s.run_pp(tol_init=0.05, csv=c_got, csv_append=1, bench=1, **kw)
        assert (rc, it) == (rc_ref, it_ref), (rep, phase)
        got = [ln.split(",")[0] for ln in open(c_got).read().splitlines() if ln.strip()]
        ref = [ln.split(",")[0] for ln in open(c_ref).read().splitlines() if ln.strip()]
        assert got == ref and len(got) > 1, (rep, got, ref)
        if phase == "pp":
            assert got[1:] == ["  [PPfirst]  ", "  [PPsecond]  "] and it == maxiter + 1
        W_got, core_got = s.get_factors()
        for a, b in zip(W_got, W_ref):
            assert relerr(proj(a), proj(b)) < 1e-7, (rep, phase)
        assert abs(np.linalg.norm(core_got) - np.linalg.norm(core_ref)) < 1e-8 * np.linalg.norm(core_ref)
    s.close()
    t.close()


def _decaying_tensor(lens, inner, seed, noise):
    """a tensor of multilinear rank `inner` with a decaying core plus relative noise: the spectra
    of its unfolding Grams have a clear gap below every requested rank <= inner"""
    rng = np.random.default_rng(seed)
    U = [np.linalg.qr(rng.standard_normal((s, r)))[0] for s, r in zip(lens, inner)]
    core = rng.standard_normal(inner)
    for m, r in enumerate(inner):   # decay along every mode
        shape = [1] * len(inner)
        shape[m] = r
        core = core * (0.7 ** np.arange(r)).reshape(shape)
    V = core
    for m, u in enumerate(U):
        V = np.moveaxis(np.tensordot(u, V, axes=(1, m)), 0, m)
    E = rng.standard_normal(lens)
    return np.asfortranarray(V + noise * np.linalg.norm(V) / np.linalg.norm(E) * E)


@pytest.mark.parametrize("env", [{"PPALS_SYM_LDS_MIN": "64"}, {"PPALS_ROCTX": "1"}])
def test_projector_route_under_switches(pp, env, tmp_path, monkeypatch):
    """PPALS_SYM_LDS_MIN=64: every symmetric product of the sign iteration (and the leaf Grams) on the
    LDS-tiled kernel that production uses from 768 rows on (k_dgemm_nt_sym_lds<32>: ragged 32 x 32
    tiles at 96 / 80 / 72 rows, the check sums per workgroup); PPALS_ROCTX=1: the named ranges around
    the kernels switched on. Same iterates as the oracle either way."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")
    lens, ranks = [96, 80, 72], [5, 6, 4]
    V = _decaying_tensor(lens, [10, 9, 8], 5, 0.05)
    W0, c0 = O.hosvd(V, ranks)
    _, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=6, resprint=10 ** 9)
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    s.hosvd()
    W_h, _ = s.get_factors()
    for a, b in zip(W_h, W0):
        assert relerr(proj(a), proj(b)) < 1e-8
    s.set_factors(W0)
    s.set_core(c0)
    s.sweeps_dt(7)
    W, core = s.get_factors()
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-10)
        assert relerr(proj(a), proj(b)) < 1e-7, (env, relerr(proj(a), proj(b)))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    s.close()
    t.close()
    c2.close()


def _slow_decay_tensor(lens, inner, decay, seed, noise):
    """multilinear rank `inner`, singular values of mode m falling like decay[m]**k, plus relative noise"""
    rng = np.random.default_rng(seed)
    U = [np.linalg.qr(rng.standard_normal((s, r)))[0] for s, r in zip(lens, inner)]
    core = rng.standard_normal(inner)
    for m, r in enumerate(inner):
        shape = [1] * len(inner)
        shape[m] = r
        core = core * (decay[m] ** np.arange(r)).reshape(shape)
    V = core
    for m, u in enumerate(U):
        V = np.moveaxis(np.tensordot(u, V, axes=(1, m)), 0, m)
    E = rng.standard_normal(lens)
    return np.asfortranarray(V + noise * np.linalg.norm(V) / np.linalg.norm(E) * E)


@pytest.mark.parametrize("lens,ranks,inner", [
    ([300, 24, 20], [70, 20, 16], [110, 22, 18]),       # the coil-100 run's rank 70 (test_ALS.cxx:366-371)
    ([1344, 40, 36], [100, 12, 10], [150, 20, 18]),   # the hyperspectral run's rank 100 on 1344 rows (:373-379)
])
def test_core_rank_above_64_stays_on_the_projector_route(pp, lens, ranks, inner, tmp_path, monkeypatch, capfd):
    """Core ranks above 64 (the reference's own real-data runs use 70 and 100, test_ALS.cxx:366-379):
    the eigen-step of such a mode — cold start by block subspace iteration, warm steps by the sign
    iteration on the LDS-tiled symmetric product, tails by plain launches with the one-sided Jacobi
    on rank + 16 columns — must not call the vendor eigensolver. hosvd + 5 HOOI sweeps against
    numpy's LAPACK reading (tests/numpy_ref.py, pinned to the oracle in tests/test_oracle_cp.py):
    projectors, ||core||; the step log must hold accepted projector steps and no full solver."""
    import numpy_ref as NR
    V = _slow_decay_tensor(lens, inner, [0.985, 0.8, 0.8], 21, 1e-4)
    W0, c0 = NR.tucker_hosvd(V, ranks)
    W_ref, core_ref = NR.tucker_hooi(V, W0, 5)
    # (the long mode keeps its s x s Gram: the product of the other core ranks is not below its extent
    # in the first case — as in the reference's runs — and the thin route is switched off in the second)
    if lens[0] > np.prod(ranks[1:]):
        monkeypatch.setenv("PPALS_TUCKER_THIN", "0")
    monkeypatch.setenv("PPALS_EIG_DEBUG", "1")
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    capfd.readouterr()
    s.hosvd()
    W_h, _ = s.get_factors()
    for a, b in zip(W_h, W0):
        assert relerr(proj(a), proj(b)) < 1e-7, relerr(proj(a), proj(b))
    s.set_factors(W0)
    s.set_core(c0)
    s.sweeps_dt(5)
    W, core = s.get_factors()
    err = capfd.readouterr().err
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
        assert relerr(proj(a), proj(b)) < 1e-6, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    log = [ln for ln in err.splitlines() if "[ppals eig]" in ln]
    big = [ln for ln in log if f"rank {ranks[0]}:" in ln or "wide tail" in ln]
    # (cold starts end in a wide tail of rank + 1 or 2 columns, warm steps are accepted as they are)
    assert sum("-> accepted" in ln for ln in big) >= 5, "\n".join(big[-30:])
    assert not any("full solver" in ln for ln in log), "\n".join(log)
    s.close()
    t.close()
    c2.close()


@pytest.mark.parametrize("lens,ranks,inner,route", [
    ([200, 44, 40], [40, 8, 6], [60, 12, 10], "chol 03"),
    ([200, 44, 40], [48, 8, 8], [70, 12, 12], "chol 03"),   # the widest tail the fused step takes (rank + 16 <= 64)
])
def test_deferred_tail_between_32_and_64_columns(pp, lens, ranks, inner, route, tmp_path, monkeypatch, capfd):
    """The deferred tail's one-launch orthonormalisation (k_rmult_chol) beyond the 21 columns of the
    benchmark: 33 to 48 columns run with 1024 threads. 12 HOOI sweeps on a mode of 200 rows at core
    ranks 40 and 48 against numpy's LAPACK reading: projectors, orthonormal factors, ||core||; the step log must show
    deferred checks accepted on the route in question and no full solver after the start."""
    import numpy_ref as NR
    V = _slow_decay_tensor(lens, inner, [0.93, 0.8, 0.8], 33, 1e-5)
    W0, c0 = NR.tucker_hosvd(V, ranks)
    W_ref, core_ref = NR.tucker_hooi(V, W0, 12)
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")
    monkeypatch.setenv("PPALS_EIG_DEBUG", "1")
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    s.hosvd()
    s.set_factors(W0)
    s.set_core(c0)
    capfd.readouterr()
    s.sweeps_dt(12)
    W, core = s.get_factors()
    err = capfd.readouterr().err
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-10)
        assert relerr(proj(a), proj(b)) < 1e-6, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    big = [ln for ln in err.splitlines() if f"rank {ranks[0]}:" in ln]
    ok = [ln for ln in big if "accepted (deferred check)" in ln and "NOT" not in ln]
    assert len(ok) >= 4, "\n".join(big[-20:])
    assert sum(route in ln for ln in ok) >= 3, "\n".join(ok[-20:])
    assert not any("full solver" in ln for ln in big[2:]), "\n".join(big)
    s.close()
    t.close()
    c2.close()


def test_core_rank_above_64_on_a_flat_spectrum(pp, tmp_path, monkeypatch, capfd):
    """A noise tensor (`-tensor r2`: U(0.5, 1), test_ALS.cxx:272) at core rank 70: below the mean
    component the Gram's spectrum is a flat bulk, Ritz values cannot place a shift, and the cold start
    must COUNT eigenvalues with the sign iteration (cold_bisect) and finish with a tail of up to 86
    columns by plain launches (block Gram-Schmidt + the one-sided Jacobi). hosvd + 3 sweeps against
    numpy's reading — projectors to what the tiny gaps allow, ||core|| and the fit tightly — and a
    step log without the vendor solver."""
    import numpy_ref as NR
    lens, ranks = [1000, 36, 28], [70, 20, 16]   # (eigenvalues 70 and 75 of the mode-0 Gram: 1.5 % apart)
    V = O.fill_uniform(int(np.prod(lens)), 13, lo=0.5, hi=1.0).reshape(lens, order="F")
    W0, c0 = NR.tucker_hosvd(V, ranks)
    W_ref, core_ref = NR.tucker_hooi(V, W0, 3)
    monkeypatch.setenv("PPALS_EIG_DEBUG", "1")
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    capfd.readouterr()
    s.hosvd()
    W_h, core_h = s.get_factors()
    Vn = np.linalg.norm(V)
    # (the 70th and 71st eigenvalue of a flat bulk lie 1e-3 apart relative to the bulk: the SUBSPACE is
    # defined to eps * lambda_1 / gap only; the captured energy is defined sharply)
    assert abs(np.linalg.norm(core_h) - np.linalg.norm(c0)) < 1e-9 * Vn
    for a, r in zip(W_h, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
    s.set_factors(W0)
    s.set_core(c0)
    s.sweeps_dt(3)
    W, core = s.get_factors()
    err = capfd.readouterr().err
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * Vn
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
        assert relerr(proj(a), proj(b)) < 1e-5, relerr(proj(a), proj(b))
    log = [ln for ln in err.splitlines() if "[ppals eig]" in ln]
    assert any("by counting" in ln and "accepted" in ln for ln in log), "\n".join(log[-40:])
    assert not any("dsyevd" in ln for ln in log), "\n".join(log)
    s.close()
    t.close()
    c2.close()


def test_thin_route_with_more_than_64_columns_needs_no_vendor_solver(pp, tmp_path, monkeypatch, capfd):
    """A tall unfolding whose small side has 64 < c <= 128 columns (rank 70 of 90): the c x c
    eigen-problem is solved whole by the one-workgroup one-sided Jacobi (k_jacobi_onesided), not by
    rocSOLVER. hosvd + 3 sweeps against numpy's reading."""
    import numpy_ref as NR
    lens, ranks = [500, 12, 10], [70, 10, 9]
    V = _slow_decay_tensor(lens, [100, 12, 10], [0.97, 0.9, 0.9], 5, 1e-4)
    W0, c0 = NR.tucker_hosvd(V, ranks)
    W_ref, core_ref = NR.tucker_hooi(V, W0, 3)
    monkeypatch.setenv("PPALS_EIG_DEBUG", "1")
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    capfd.readouterr()
    s.set_factors(W0)
    s.set_core(c0)
    s.sweeps_dt(3)
    W, core = s.get_factors()
    err = capfd.readouterr().err
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
        assert relerr(proj(a), proj(b)) < 1e-6, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    assert "dsyevd" not in err, err[-2000:]
    s.close()
    t.close()
    c2.close()


@pytest.mark.parametrize("fail_every,handover", [(0, "value"), (2, "value"), (5, "value"), (0, "event"), (3, "event")])
def test_deferred_eigen_step_checks(pp, fail_every, handover, tmp_path, monkeypatch, capfd):
    """Deferred acceptance (Ops::eig_defer / eig_verify, hip_ops.hip; TuckerEngine::settle_mode /
    rollback_and_redo): once a slot's steps go through as scheduled, a plain HOOI sweep no longer
    waits for their checks — they are read when the engine comes back to the mode, or before a row
    is printed. 14 sweeps on mode extents above 64 against the oracle's full eigen-decompositions
    (projectors, ||core||, CSV rows), with PPALS_EIG_DEFER_FAIL=n turning every n-th deferred check
    into a failure: the engine must put back every factor stepped since and repeat those steps. The
    step log must show deferred checks (and the forced failures). The second stream that finishes the
    checks is released by a value the last kernel of the step stores (hipStreamWaitValue64) or, with
    PPALS_EIG_DEFER=2, by an event."""
    lens, ranks = [96, 80, 72], [5, 6, 4]
    V = _decaying_tensor(lens, [10, 9, 8], 5, 0.05)
    W0, c0 = O.hosvd(V, ranks)
    c_ref, c_got = str(tmp_path / "r.csv"), str(tmp_path / "g.csv")
    kw = dict(tol=0.0, maxiter=13, resprint=5)
    _, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, csv=c_ref, **kw)
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")
    monkeypatch.setenv("PPALS_EIG_DEBUG", "1")
    if fail_every:
        monkeypatch.setenv("PPALS_EIG_DEFER_FAIL", str(fail_every))
    if handover == "event":
        monkeypatch.setenv("PPALS_EIG_DEFER", "2")
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    s.hosvd()
    s.set_factors(W0)
    s.set_core(c0)
    capfd.readouterr()
    rc, it = s.run_dt(csv=c_got, **kw)
    assert it == it_ref
    W, core = s.get_factors()
    err = capfd.readouterr().err
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-10)
        assert relerr(proj(a), proj(b)) < 1e-7, (fail_every, relerr(proj(a), proj(b)))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    want = core_from_factors(V, W)   # after forced rollbacks too: the core of the RETURNED factors
    assert np.abs(core - want).max() < 1e-10 * np.abs(want).max(), fail_every
    _, r1 = O.read_csv(c_ref)
    _, r2 = O.read_csv(c_got)
    assert len(r1) == len(r2)
    for a, b in zip(r1, r2):
        assert a[1] == b[1] and abs(a[5] - b[5]) < 1e-8 * np.linalg.norm(V)
    ndef = err.count("accepted (deferred check)")
    nfail = err.count("NOT accepted (deferred check)")
    assert ndef >= 6, err[-3000:]
    assert (nfail >= 2) if fail_every else (nfail == 0), (fail_every, nfail, err[-3000:])
    # slowly turning subspaces: the orthonormalising matrix comes from the series of S^-1/2
    # (k_rmult_chol route 3 = the second status word), not from the elimination
    assert err.count("chol 03") >= 3, err[-3000:]
    s.close()
    t.close()
    c2.close()


def test_tucker_bench_pp_after_dt_large_mode(pp, ctx, tmp_path, monkeypatch):
    """pp_bench's call order on ONE session with a mode extent above 64 (advisor, round 3): run_dt
    leaves the slots in 'any basis of the subspace' mode, and run_pp(bench=1) skips the DT_sub that
    used to switch it off — the PP sweeps must still difference sorted eigenvectors. Compared with the
    oracle's alsTucker_DT + alsTucker_PP(bench) from the same factors: projectors, ||core||, and dW
    small enough that the PP corrections are corrections (the factors stay orthonormal)."""
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")  # the s x s route is the one with lazy eigenvectors
    lens, ranks = [96, 20, 18], [4, 3, 3]
    V = _decaying_tensor(lens, [8, 7, 6], 11, 0.02)
    W0, c0 = O.hosvd(V, ranks)
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    s.hosvd()
    s.set_factors(W0)
    kw = dict(tol=1e-14, maxiter=3, resprint=1)
    c_ref, c_got = str(tmp_path / "ref.csv"), str(tmp_path / "got.csv")
    for c in (c_ref, c_got):
        open(c, "w").write("[timetype],[dtime]\n")
    rc_ref, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, csv=c_ref, bench=1, **kw)
    rc, it = s.run_dt(csv=c_got, csv_append=1, bench=1, **kw)
    assert (rc, it) == (rc_ref, it_ref)
    # NO get_factors() here: it would finalise the rotations and hide the bug
    rc_ref, it_ref, W_ref, core_ref = O.als_tucker_pp(V, W_ref, core_ref, tol_init=0.05, csv=c_ref,
                                                      bench=1, **kw)
    rc, it = s.run_pp(tol_init=0.05, csv=c_got, csv_append=1, bench=1, **kw)
    assert (rc, it) == (rc_ref, it_ref)
    W_got, core_got = s.get_factors()
    for a, b, r in zip(W_got, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
        assert relerr(proj(a), proj(b)) < 1e-6
    assert abs(np.linalg.norm(core_got) - np.linalg.norm(core_ref)) < 1e-8 * np.linalg.norm(core_ref)
    s.close()
    t.close()
    c2.close()


@pytest.mark.parametrize("lens,ranks", [([96, 80, 72], [5, 6, 4]), ([130, 70, 66], [8, 3, 5])])
def test_eigen_step_projector_route_matches_oracle(pp, ctx, lens, ranks, tmp_path, monkeypatch):
    """mode extents above 64: from the second HOOI sweep on, the eigen-step is the spectral
    projector by Newton-Schulz sign iteration on the matrix cores (kernels_eig.hip.h), checked by
    trace(P) == rank, with the full solver as the first call and as fallback. Same iterates as the
    oracle's full eigen-decomposition: projectors, ||core||, CSV rows of alsTucker_DT; and the same
    as the engine with the route switched off (PPALS_EIG_FAST=0, fresh context)."""
    V = _decaying_tensor(lens, [10, 9, 8], 5, 0.05)
    W0, c0 = O.hosvd(V, ranks)
    c_ref, c_got = str(tmp_path / "r.csv"), str(tmp_path / "g.csv")
    _, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=6, csv=c_ref, resprint=1)
    # (these unfoldings are tall: without this the factors would come from the small Gram and the
    # s x s route under test would never run)
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")
    results = []
    for fast in ("1", "0"):
        monkeypatch.setenv("PPALS_EIG_FAST", fast)
        c2 = pp.Context(0)
        t = pp.Tensor(c2, lens, 1).upload(V)
        s = pp.Tucker(c2, t, ranks)
        s.hosvd()
        W_h, _ = s.get_factors()
        for a, b in zip(W_h, W0):
            assert relerr(proj(a), proj(b)) < 1e-8
        s.set_factors(W0)
        s.set_core(c0)
        rc, it = s.run_dt(tol=0.0, maxiter=6, csv=c_got, resprint=1)
        assert it == it_ref
        W, core = s.get_factors()
        for a, b, r in zip(W, W_ref, ranks):
            assert np.allclose(a.T @ a, np.eye(r), atol=1e-10)
            assert relerr(proj(a), proj(b)) < 1e-7, (fast, relerr(proj(a), proj(b)))
        assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
        _, r1 = O.read_csv(c_ref)
        _, r2 = O.read_csv(c_got)
        assert len(r1) == len(r2)
        for a, b in zip(r1, r2):
            assert a[1] == b[1] and abs(a[5] - b[5]) < 1e-8 * np.linalg.norm(V)
        results.append(W)
        s.close()
        t.close()
        c2.close()
    # eigenvectors one by one (sorted, up to sign): the two routes agree column by column
    for a, b in zip(*results):
        for k in range(a.shape[1]):
            assert min(np.linalg.norm(a[:, k] - b[:, k]), np.linalg.norm(a[:, k] + b[:, k])) < 1e-6


TALL = [([300, 6, 5], [5, 3, 2]),        # leading mode tall (L = 1: no re-ordering needed)
        ([6, 200, 5], [3, 8, 4]),        # middle mode tall (batched transposition)
        ([5, 4, 150, 3], [3, 2, 9, 2]),  # order 4
        ([20, 12, 400], [10, 8, 70])]    # last mode tall, more than 64 columns to orthonormalise


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("lens,ranks", TALL)
def test_tall_unfolding_thin_route_matches_oracle(pp, ctx, lens, ranks, dtype, tmp_path, monkeypatch):
    """a mode longer than the product of the other modes' ranks (the 7200-frame mode of the
    coil-100 shape): the factor comes from the SMALL Gram of the unfolding, Y^T Y, and one product
    with Y (TuckerEngine::factor_update) instead of the s x s Gram the reference always forms
    (als_Tucker.cxx:399-406). Same iterates as the oracle (which follows the reference) and as
    the engine with the route switched off."""
    inner = [min(s, r + 3) for s, r in zip(lens, ranks)]
    V = _decaying_tensor(lens, inner, 11, 0.02)
    W0, c0 = O.hosvd(V, ranks)
    c_ref, c_got = str(tmp_path / "r.csv"), str(tmp_path / "g.csv")
    _, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=4, csv=c_ref, resprint=1)
    tol = 2e-4 if dtype == 0 else 1e-7
    for thin in ("1", "0"):
        monkeypatch.setenv("PPALS_TUCKER_THIN", thin)
        t = pp.Tensor(ctx, lens, dtype).upload(V)
        s = pp.Tucker(ctx, t, ranks)
        s.set_factors(W0)
        s.set_core(c0)
        rc, it = s.run_dt(tol=0.0, maxiter=4, csv=c_got, resprint=1)
        assert it == it_ref
        W, core = s.get_factors()
        for a, b, r in zip(W, W_ref, ranks):
            assert np.allclose(a.T @ a, np.eye(r), atol=1e-10), thin
            assert relerr(proj(a), proj(b)) < tol, (thin, relerr(proj(a), proj(b)))
        assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < tol * np.linalg.norm(core_ref)
        _, r1 = O.read_csv(c_ref)
        _, r2 = O.read_csv(c_got)
        assert len(r1) == len(r2)
        for a, b in zip(r1, r2):
            assert a[1] == b[1] and abs(a[5] - b[5]) < tol * np.linalg.norm(V)
        s.close()
        t.close()


def test_tall_unfolding_rank_deficient_falls_back(pp, ctx):
    """requested rank above the tensor's multilinear rank: Y^T Y has zero eigenvalues among the
    wanted ones, Y v has null columns, the Cholesky QR reports it and the s x s route completes
    the basis — orthonormal factors and the oracle's residual either way"""
    lens, ranks = [120, 5, 4], [4, 3, 3]
    V = _decaying_tensor(lens, [2, 2, 2], 3, 0.0)
    W0, c0 = O.hosvd(V, ranks)
    _, _, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=2)
    t = pp.Tensor(ctx, lens, 1).upload(V)
    s = pp.Tucker(ctx, t, ranks)
    s.set_factors(W0)
    s.set_core(c0)
    s.run_dt(tol=0.0, maxiter=2)
    W, core = s.get_factors()
    for a, r in zip(W, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    s.close()
    t.close()


@pytest.mark.parametrize("order", ["asc", "desc"])
def test_chain_order_of_the_first_level_products(pp, ctx, order, monkeypatch):
    """the first-level node that keeps the left half removes the right-half modes one by one; the
    engine starts with the LAST of them when only that makes the tensor scan's columns 128-B
    aligned (the products commute). Both orders against the oracle's TTMc."""
    monkeypatch.setenv("PPALS_TUCKER_CHAIN", order)
    lens, ranks = [6, 5, 4, 7, 5, 6], [2, 3, 2, 3, 2, 2]
    V = O.fill_uniform(int(np.prod(lens)), 3, lo=0.5, hi=1.0).reshape(lens, order="F")
    W = [np.linalg.qr(O.fill_uniform(s * r, 40 + i, lo=-1, hi=1).reshape((s, r), order="F"))[0]
         for i, (s, r) in enumerate(zip(lens, ranks))]
    for dtype in (1, 0):
        t = pp.Tensor(ctx, lens, dtype).upload(V)
        s = pp.Tucker(ctx, t, ranks)
        s.set_factors(W)
        for skip in [-1] + list(range(len(lens))):
            assert relerr(s.ttmc(skip), O.ttmc(V, W, skip)) < (2e-6 if dtype == 0 else 1e-11), skip
        s.close()
        t.close()


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("lens,ranks", [([96, 80, 72], [5, 6, 4]), ([40, 17, 33], [4, 4, 5])])
def test_order3_multi_sweep_schedule(pp, lens, ranks, dtype, monkeypatch):
    """Order 3 on one GPU: the multi-sweep dimension tree (TuckerEngine::ms3_leaf) — ONE first-level
    intermediate X_r = V x_r W_r serves the two mode updates after mode r's, the root rotates 2, 1, 0
    over three resident rotations of the tensor: 3 tensor scans per 2 HOOI sweeps, the same factor
    versions in every product as alsTucker_DT's per-sweep tree (als_Tucker.cxx:340-408), hence the same
    iterates. Both schedules (PPALS_TUCKER_CHAIN=tree: the per-sweep tree) against the oracle after 1 to
    4 sweeps, the scan launches counted by the profile slots."""
    V = _decaying_tensor(lens, [r + 4 for r in ranks], 17, 0.02)
    W0, c0 = O.hosvd(V, ranks)
    scans = {}
    for sched in ("ms", "tree"):
        if sched == "tree":
            monkeypatch.setenv("PPALS_TUCKER_CHAIN", "tree")
        c2 = pp.Context(0)
        t = pp.Tensor(c2, lens, dtype).upload(V)
        s = pp.Tucker(c2, t, ranks)
        for sweeps in (1, 2, 3, 4, 5):
            s.set_factors(W0)
            s.set_core(c0)
            c2.profile_enable(1)
            c2.profile_reset()
            s.sweeps_dt(sweeps)
            c2.sync()
            scans[(sched, sweeps)] = c2.profile_read(0)[0]
            c2.profile_enable(0)
            W, core = s.get_factors()
            _, _, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=sweeps - 1, resprint=10 ** 6)
            for a, b, r in zip(W, W_ref, ranks):
                assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
                assert relerr(proj(a), proj(b)) < (1e-7 if dtype == 1 else 2e-4), (sched, sweeps)
            assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < (1e-9 if dtype == 1 else 1e-5) * np.linalg.norm(core_ref)
            # the lazily computed core (TuckerEngine::ensure_core) entry by entry, from the returned factors
            want = core_from_factors(V, W)
            assert core.shape == want.shape
            assert np.abs(core - want).max() < (1e-10 if dtype == 1 else 2e-6) * np.abs(want).max(), (sched, sweeps)
        s.close()
        t.close()
        c2.close()
    big = min(lens) >= 16  # (the leading-mode product of the back end wants 16 rows: else the per-sweep tree)
    for sweeps in (1, 2, 3, 4, 5):
        assert scans[("tree", sweeps)] == 2 * sweeps, scans
        assert scans[("ms", sweeps)] == ((3 * sweeps + 1) // 2 if big else 2 * sweeps), scans


@pytest.mark.parametrize("lens,inner,ranks", [([96, 80, 72], [10, 9, 8], [8, 7, 6]),
                                              ([130, 120, 12], [75, 75, 12], [70, 70, 10])])
def test_exactly_low_rank_tensor_block_loses_rank(pp, lens, inner, ranks):
    """A tensor of EXACT multilinear rank `inner` (no noise): the Gram of an unfolding has fewer non-zero
    eigenvalues than the cold start's block has columns (rank + 16), the block loses rank, its
    factorisation leaves NaNs and the Rayleigh-Ritz Jacobi sees them before the host has read the
    status — it once ranked the NaN eigenvalues all at position 0 and gathered through an unwritten
    permutation (a memory fault on the time-lapse extents with an image-like rank-100 tensor, round 5).
    The step must be rejected and the fallback deliver the oracle's subspaces, for a block of <= 64
    columns (in-LDS Jacobi) and of 86 (k_jacobi_onesided)."""
    rng = np.random.default_rng(3)
    U = [np.linalg.qr(rng.standard_normal((s, r)))[0] for s, r in zip(lens, inner)]
    core = rng.standard_normal(inner)
    for m, r in enumerate(inner):
        shape = [1] * len(inner)
        shape[m] = r
        core = core * (0.95 ** np.arange(r)).reshape(shape)
    V = core
    for m, u in enumerate(U):
        V = np.moveaxis(np.tensordot(u, V, axes=(1, m)), 0, m)
    V = np.asfortranarray(V)
    W_ref, core_ref = O.hosvd(V, ranks)
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    s.hosvd()
    W, core_got = s.get_factors()
    for a, b, r in zip(W, W_ref, ranks):
        assert np.isfinite(a).all()
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
        assert relerr(proj(a), proj(b)) < 1e-7, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core_got) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    s.sweeps_dt(3)
    W3, core3 = s.get_factors()
    _, _, W3_ref, core3_ref = O.als_tucker_dt(V, W_ref, core_ref, tol=0.0, maxiter=2, resprint=10 ** 9)
    for a, b in zip(W3, W3_ref):
        assert np.isfinite(a).all() and relerr(proj(a), proj(b)) < 1e-6, relerr(proj(a), proj(b))
    s.close()
    t.close()
    c2.close()


def test_eigen_step_wide_tail(pp, tmp_path, monkeypatch, capfd):
    """a shift that has slipped below a few more eigenvalues than wanted (forced here:
    PPALS_EIG_SIGMA_SCALE puts it at 0.4 x the next eigenvalue, under the two that follow in this
    0.49-per-index spectrum): the projector then covers rank + 2 dimensions, the Rayleigh-Ritz step
    runs on that many columns and the leading `rank` eigenpairs are still those of the full
    solver — same iterates as the oracle, and the log shows the wide tail was taken."""
    lens, ranks = [96, 80, 72], [5, 6, 4]
    V = _decaying_tensor(lens, [10, 9, 8], 5, 0.001)
    W0, c0 = O.hosvd(V, ranks)
    _, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=5)
    monkeypatch.setenv("PPALS_EIG_SIGMA_SCALE", "0.4")
    monkeypatch.setenv("PPALS_EIG_DEBUG", "1")
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")   # the s x s route (the unfoldings here are tall)
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    s.set_factors(W0)
    s.set_core(c0)
    s.run_dt(tol=0.0, maxiter=5)
    W, core = s.get_factors()
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-10)
        assert relerr(proj(a), proj(b)) < 1e-7, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    s.close()
    t.close()
    c2.close()
    err = capfd.readouterr().err
    assert "wide tail" in err and "accepted" in err.split("wide tail", 1)[1], err[-2000:]


@pytest.mark.parametrize("lens,ranks", [([100, 68, 76], [6, 5, 4]), ([128, 64, 72], [8, 4, 6]),
                                        ([90, 68, 76], [5, 5, 5])])
def test_hosvd_gram_syrk_f32(pp, lens, ranks, monkeypatch):
    """K13 for fp32 tensor storage at a size where the tiled SYRK runs (k_unfold_syrk_f32: upper
    triangle of 64 x 64 tiles, ragged edges, both unfolding layouts — mode in front / mode behind —
    and reduction splits): the HOSVD factors as subspaces and ||core|| against the oracle on the
    SAME fp32-representable values (the products are exact in fp64, so only rounding order differs).
    The third shape has no 4-aligned rows and takes the 32 x 32-tile kernel."""
    V = _decaying_tensor(lens, [min(s, r + 4) for s, r in zip(lens, ranks)], 7, 0.05)
    V = np.asfortranarray(V.astype(np.float32).astype(np.float64))
    W_ref, core_ref = O.hosvd(V, ranks)
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 0).upload(V)
    s = pp.Tucker(c2, t, ranks)
    s.hosvd()
    W, core = s.get_factors()
    for a, b, r in zip(W, W_ref, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-10)
        assert relerr(proj(a), proj(b)) < 1e-8, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-6 * np.linalg.norm(core_ref)
    s.close()
    t.close()
    c2.close()


def test_strict_preload_refuses_a_late_eigensolver_load():
    """include/ppals.h: with PPALS_STRICT_PRELOAD=1 a session that needs the vendor eigensolver in a
    process that did not call ppals_preload_eigensolver() before the HIP runtime came up fails with
    PPALS_ERR_UNSUPPORTED (-5) at once instead of stalling for minutes in dlopen (fresh child: this
    process has preloaded the solver already, tests/conftest.py)"""
    import subprocess
    import sys
    import time
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pairwise-perturbation_amd")
    code = f"""
import os, sys
os.environ['PPALS_STRICT_PRELOAD'] = '1'
os.environ['PPALS_EIG_FAST'] = '0'          # every eigen-step on the full solver
sys.path.insert(0, {pkg!r})
import ppals
ctx = ppals.Context(0)                       # the HIP runtime is up now, the solver is not loaded
t = ppals.Tensor(ctx, [160, 10, 8], 1).fill_uniform(3)
tk = ppals.Tucker(ctx, t, [4, 3, 3])
try:
    tk.hosvd()
    print('RAN')
except ppals.PpalsError as e:
    print('REFUSED', e)
"""
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "REFUSED ppals error -5" in out.stdout, out.stdout + out.stderr[-2000:]
    assert "ppals_preload_eigensolver" in out.stdout
    assert time.time() - t0 < 120


@pytest.mark.parametrize("lens,ranks,inner,from_rows", [
    ([2400, 24, 20], [30, 8, 7], [60, 14, 12], None),     # long enough for the route by default (2048)
    ([500, 24, 20], [20, 8, 7], [48, 14, 12], "65"),      # PPALS_COLD_SUBSPACE_FROM lowers the threshold
])
def test_long_mode_cold_start_by_subspace_iteration(pp, lens, ranks, inner, from_rows, monkeypatch, capfd):
    """hosvd on a LONG mode (als_Tucker.cxx:12-23; coil-100's image mode has 7200 rows,
    test_ALS.cxx:296-299): the cold start runs block subspace iteration with Rayleigh-Ritz on thin
    blocks (cold_subspace: two products of G with J x (rank + 16) per step) instead of ~46 products of
    J^3 on the projector route. The step log must show that route accepted and neither a projector
    step nor the full solver for that mode during hosvd; projectors against numpy's LAPACK reading;
    the HOOI sweeps that follow start warm from the state it left."""
    import numpy_ref as NR
    # (ranks: 8 x 7 > rank + 16, so that the sweeps' Grams keep a full block of directions)
    if from_rows:
        monkeypatch.setenv("PPALS_COLD_SUBSPACE_FROM", from_rows)
    V = _slow_decay_tensor(lens, inner, [0.9, 0.8, 0.8], 5, 1e-5)
    W0, c0 = NR.tucker_hosvd(V, ranks)
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")     # (the long mode keeps its s x s Gram in the sweeps too)
    monkeypatch.setenv("PPALS_EIG_DEBUG", "1")
    c2 = pp.Context(0)
    t = pp.Tensor(c2, lens, 1).upload(V)
    s = pp.Tucker(c2, t, ranks)
    capfd.readouterr()
    s.hosvd()
    err = capfd.readouterr().err
    W_h, core_h = s.get_factors()
    for a, b, r in zip(W_h, W0, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9)
        assert relerr(proj(a), proj(b)) < 1e-7, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core_h) - np.linalg.norm(c0)) < 1e-9 * np.linalg.norm(c0)
    log = [ln for ln in err.splitlines() if "[ppals eig]" in ln and f"J {lens[0]}" in ln]
    steps = [ln for ln in log if "cold subspace step" in ln]
    assert 1 <= len(steps) <= 16, "\n".join(log)
    assert not any("||X^2-I||" in ln for ln in log), "\n".join(log)    # no projector step from cold
    assert not any("full solver" in ln for ln in log), "\n".join(log)
    # warm from there
    W_ref, core_ref = NR.tucker_hooi(V, W0, 3)
    s.set_factors(W0)
    s.set_core(c0)
    s.sweeps_dt(3)
    W, core = s.get_factors()
    err = capfd.readouterr().err
    for a, b in zip(W, W_ref):
        assert relerr(proj(a), proj(b)) < 1e-6, relerr(proj(a), proj(b))
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-9 * np.linalg.norm(core_ref)
    log2 = [ln for ln in err.splitlines() if "[ppals eig]" in ln]
    assert not any("full solver" in ln for ln in log2), "\n".join(log2[:60])
    s.close()
    t.close()
    c2.close()


@pytest.mark.parametrize("lens,ranks", [([128, 160, 64], [8, 70, 6]),       # the mode product with 70 columns is a batched wide scan
                                        ([96, 72, 144, 16], [6, 5, 100, 4])])  # 100 columns (the time-lapse run's rank), order 4
def test_mode_products_above_64_columns(pp, lens, ranks, monkeypatch):
    """Tucker mode products with more than 64 core columns (the reference's real-data ranks 70 and 100,
    test_ALS.cxx:366-379; als_Tucker.cxx:102): on fp32 storage the first-level product is ONE pass of the
    LDS-staged wide scan in its batched, keep-the-mode-in-place form (k_scan_wide with T > 1 and an fp64
    result); TTMc for every skipped mode against the oracle, and the same with the wide scan switched
    off (64-column chunks) to rounding level."""
    V = O.fill_uniform(int(np.prod(lens)), 3, lo=0.5, hi=1.0).reshape(lens, order="F")
    W = [np.linalg.qr(O.fill_uniform(s * r, 40 + i, lo=-1, hi=1).reshape((s, r), order="F"))[0]
         for i, (s, r) in enumerate(zip(lens, ranks))]
    got = {}
    for wide in ("1", "0"):
        monkeypatch.setenv("PPALS_SCAN_WIDE", wide)
        c2 = pp.Context(0)
        t = pp.Tensor(c2, lens, 0).upload(V)
        s = pp.Tucker(c2, t, ranks)
        s.set_factors(W)
        c2.profile_enable(1)
        c2.profile_reset()
        got[wide] = [s.ttmc(skip) for skip in [-1] + list(range(len(lens)))]
        n_scans, _, _ = c2.profile_read(0)
        c2.profile_enable(0)
        got[wide + "n"] = n_scans
        for skip, g in zip([-1] + list(range(len(lens))), got[wide]):
            assert relerr(g, O.ttmc(V, W, skip)) < 2e-6, (wide, skip, relerr(g, O.ttmc(V, W, skip)))
        s.close()
        t.close()
        c2.close()
    for a, b in zip(got["1"], got["0"]):
        assert relerr(a, b) < 5e-7
    # fewer tensor passes with the wide scan wherever a product of more than 64 columns meets the fp32
    # tensor itself (order 3 here; the order-4 chain contracts a short-rank mode first and its 100-column
    # product acts on an fp64 intermediate, which keeps the 64-column chunks)
    assert got["1n"] <= got["0n"], (got["1n"], got["0n"])
    if len(lens) == 3:
        assert got["1n"] < got["0n"], (got["1n"], got["0n"])
