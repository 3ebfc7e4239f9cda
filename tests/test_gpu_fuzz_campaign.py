"""Seeded random campaign over the knobs that choose code paths: order, extents (tile edges,
unaligned row counts, one long mode among short ones), rank (one / two / four n-tiles), storage type, sweep schedule, number of
root modes, padded resident layouts on or off — CP exact sweeps, the PP driver and Tucker HOOI
against the fp64 oracle. The default run is a dozen cases; a campaign sets PPALS_FUZZ_CASES (and
PPALS_FUZZ_SEED) — e.g. 2000 (profiles/README.md)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import test_gpu_cp as G
import test_gpu_tucker as GT

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("PPALS_FUZZ_CASES", "12"))
SEED = int(os.environ.get("PPALS_FUZZ_SEED", "20261004"))


@pytest.fixture(scope="module")
def pp():
    import ppals
    return ppals


@pytest.fixture(scope="module")
def ctx(pp):
    c = pp.Context(0)
    yield c
    c.close()


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        N = int(rng.integers(3, 7))
        big = rng.random() < 0.5
        lens = [int(rng.integers(2, 70 if big else 12)) for _ in range(N)]
        if rng.random() < 0.2:
            # skewed, as the reference's real-data extents (3 x 128 x 128 x 7200, 33 x 1344 x 1024 x 9): one
            # LONG mode among short ones — never-root modes, the row-parallel update, one-wave planes and
            # j-split reductions of the cached-intermediate contractions, longest-mode-first PP chains
            N = int(rng.integers(3, 5))
            lens = [int(rng.integers(3, 14)) for _ in range(N)]
            lens[int(rng.integers(0, N))] = int(rng.integers(600, 3000))
        size = int(np.prod(lens))
        if size > 2.5e6 or size < 500:
            continue
        R = min(int(rng.choice([1, 2, 3, 5, 8, 10, 16, 17, 24, 33, 40])), min(lens))
        out.append(dict(lens=lens, R=R, dtype=int(rng.integers(0, 2)),
                        sched=str(rng.choice(["dt", "msdt"])), roots=int(rng.integers(0, 4)),
                        pad=int(rng.integers(0, 2)), kind=str(rng.choice(["r", "r2"])),
                        seed=int(rng.integers(0, 1 << 30))))
    return out


def _setenv(monkeypatch, c):
    if c["roots"] > 0 and c["roots"] <= len(c["lens"]) - 2:
        monkeypatch.setenv("PPALS_MSDT_ROOTS", str(c["roots"]))
    monkeypatch.setenv("PPALS_PAD_LAYOUT", str(c["pad"]))


@pytest.mark.parametrize("c", _cases(NCASES, SEED), ids=lambda c: "-".join(map(str, c["lens"])) + f"-R{c['R']}")
def test_cp_sweeps(pp, ctx, c, monkeypatch):
    _setenv(monkeypatch, c)
    lens, R, dtype = c["lens"], c["R"], c["dtype"]
    V, W = G.problem(lens, R, c["seed"] % 1000, c["kind"])
    Gr = O.init_factors(lens, R, 7 + c["seed"] % 100)
    K = 3
    _, _, W_ref, G_ref = O.als_cp_dt(V, W, Gr, tol=0.0, maxiter=K - 1, resprint=1000)
    t = pp.Tensor(ctx, lens, dtype).upload(V)
    s = pp.CP(ctx, t, R)
    s.set_schedule(c["sched"])
    s.set_factors(W, Gr)
    for m in range(len(lens)):
        assert G.relerr(s.mttkrp(m), O.mttkrp(V, W, m, 0)) < G.KTOL[dtype], (c, m)
    s.sweeps_dt(K)
    W_got, _ = s.get_factors(with_grad=True)
    # ill-conditioned random problems amplify storage rounding: compare through the residual too
    r_got, r_ref = O.residual(V, W_got), O.residual(V, W_ref)
    assert abs(r_got - r_ref) < (1e-6 if dtype == 1 else 1e-3) * max(r_ref, 1e-3 * np.linalg.norm(V)), c
    if c["kind"] == "r" and dtype == 1:
        for a, b in zip(W_got, W_ref):
            assert G.relerr(a, b) < 1e-6, (c, G.relerr(a, b))
    s.close()
    t.close()


@pytest.mark.parametrize("c", _cases(max(2, NCASES // 3), SEED + 1),
                         ids=lambda c: "-".join(map(str, c["lens"])) + f"-R{c['R']}")
def test_cp_pp_driver(pp, ctx, c, monkeypatch, tmp_path):
    _setenv(monkeypatch, c)
    lens, R = c["lens"], min(c["R"], 6)
    V, W = G.problem(lens, R, c["seed"] % 1000, "r")
    Gr = O.init_factors(lens, R, 97)
    Vn = np.linalg.norm(V)
    c_ref, c_got = str(tmp_path / "ref.csv"), str(tmp_path / "got.csv")
    kw = dict(tol=1e-6 * Vn, tol_init=0.1, maxiter=25, resprint=1)
    O.als_cp_pp(V, W, Gr, csv=c_ref, **kw)
    # Some of these random problems pass through nearly singular normal equations, where the ALS
    # trajectory amplifies rounding by ten orders of magnitude (the oracle ITSELF moves by 5e-5 at
    # such an iteration when its input is perturbed by 1e-15). The yardstick for "same trajectory"
    # is therefore the oracle's own sensitivity, row by row.
    c_pert = str(tmp_path / "pert.csv")
    rng = np.random.default_rng(5)
    Vp = np.asfortranarray(V * (1.0 + 1e-15 * rng.standard_normal(V.shape)))
    O.als_cp_pp(Vp, W, Gr, csv=c_pert, **kw)
    t = pp.Tensor(ctx, lens, 1).upload(V)
    Wl = [w.copy(order="F") for w in W]
    Gl = [g.copy(order="F") for g in Gr]
    pp.alsCP_PP(t, Wl, Gl, kw["tol"], kw["tol_init"], 5e3, kw["maxiter"], 0.0, 1.0, c_got, 1,
                False, ctx)
    _, r1 = O.read_csv(c_ref)
    _, r2 = O.read_csv(c_got)
    _, r3 = O.read_csv(c_pert)
    n = min(len(r1), len(r2), len(r3))
    for a, b, p in zip(r1[:n], r2[:n], r3[:n]):
        if a[5] < 1e-4 * Vn:
            break
        own = abs(a[5] - p[5]) if (a[1] == p[1] and a[4] == p[4]) else float("inf")
        if own > 1e-3 * abs(a[5]):
            break   # from here on the problem does not define its own trajectory any more
        assert a[1] == b[1] and a[4] == b[4], (c, a, b)
        assert abs(a[5] - b[5]) <= 1e-5 * abs(a[5]) + 1e-9 * Vn + 100 * own, (c, a, b)
    t.close()


@pytest.mark.parametrize("c", [x for x in _cases(max(2, NCASES // 3), SEED + 2) if len(x["lens"]) <= 5],
                         ids=lambda c: "-".join(map(str, c["lens"])))
def test_tucker_sweeps(pp, ctx, c, monkeypatch, tmp_path):
    # (a long mode of the skewed class is cut to 200-300 rows: the oracle's full eigen-decomposition of a
    # 3000 x 3000 Gram takes minutes)
    lens = [s if s <= 300 else 200 + s % 100 for s in c["lens"]]
    rng = np.random.default_rng(c["seed"])
    if c["seed"] % 3 == 0 and len(lens) <= 4 and int(np.prod(lens)) < 4e5:
        # one mode above 64: the projector route of the eigen-step, with its cold start from Ritz
        # values and its wide tail, instead of the in-LDS Jacobi
        k = int(np.argmax(lens))
        lens[k] = 70 + c["seed"] % 90
    ranks = [int(rng.integers(1, max(2, min(s, 6)))) for s in lens]
    for i, r in enumerate(ranks):   # a rank above the product of the others is ill-posed
        ranks[i] = min(r, int(np.prod([q for j, q in enumerate(ranks) if j != i])))
    inner = [min(s, r + 2) for s, r in zip(lens, ranks)]
    V = GT._decaying_tensor(lens, inner, c["seed"] % 997, 0.02)
    W0, c0 = O.hosvd(V, ranks)
    _, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=3)
    t = pp.Tensor(ctx, lens, c["dtype"]).upload(V)
    s = pp.Tucker(ctx, t, ranks)
    s.set_factors(W0)
    s.set_core(c0)
    s.run_dt(tol=0.0, maxiter=3)
    W, core = s.get_factors()
    tol = 5e-4 if c["dtype"] == 0 else 1e-7
    for a, r in zip(W, ranks):
        assert np.allclose(a.T @ a, np.eye(r), atol=1e-9), c
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < tol * np.linalg.norm(core_ref), c
    s.close()
    t.close()
