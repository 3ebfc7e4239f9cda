"""CPU tests of the engine's HOST LOGIC: the very same cases as tests/test_gpu_cp.py, but with the
engine + C ABI linked against the host stand-in ops (tests/hostsim). They check the sweep / tree /
PP-restart / CSV control flow of pairwise-perturbation_amd/csrc/engine.cpp against the oracle on a
box without a GPU. They say nothing about the HIP kernels — that is what `-m gpu` is for."""
import os

import pytest

import hostsim_util
import test_gpu_cp as G


def core_from_factors(V, W):
    """V x_i W_i^T entry by entry (als_Tucker.cxx:408: the core belongs to the RETURNED factors)"""
    import numpy as np
    core = V
    for m, w in enumerate(W):
        core = np.moveaxis(np.tensordot(w.T, core, axes=(1, m)), 0, m)
    return core


@pytest.fixture(scope="module")
def pp():
    return hostsim_util.load()


@pytest.fixture(scope="module")
def ctx(pp):
    c = pp.Context(0)
    yield c
    c.close()


test_fill_and_norm = G.test_fill_and_norm
test_tree_nodes_and_mttkrp = G.test_tree_nodes_and_mttkrp
test_pp_operators = G.test_pp_operators
test_gram_system = G.test_gram_system
test_jacobi_fallback_path = G.test_jacobi_fallback_path
test_dt_sweeps_match_oracle = G.test_dt_sweeps_match_oracle
test_normalize_and_owed_scale = G.test_normalize_and_owed_scale
test_pp_operator_after_msdt_sweeps = G.test_pp_operator_after_msdt_sweeps
test_short_modes_are_never_roots = G.test_short_modes_are_never_roots
test_long_and_short_modes_exact_and_pp = G.test_long_and_short_modes_exact_and_pp
test_update_reads_gathered_row_blocks = G.test_update_reads_gathered_row_blocks
test_driver_dt_csv_matches_oracle = G.test_driver_dt_csv_matches_oracle
test_driver_pp_matches_oracle = G.test_driver_pp_matches_oracle
test_driver_pp_partupdate_matches_oracle = G.test_driver_pp_partupdate_matches_oracle
test_driver_pp_with_magni = G.test_driver_pp_with_magni
test_schedule_switch_mid_run = G.test_schedule_switch_mid_run
test_msdt_root_counts = G.test_msdt_root_counts
test_edge_shapes = G.test_edge_shapes
test_random_shapes_against_oracle = G.test_random_shapes_against_oracle
test_random_larger_shapes_against_oracle = G.test_random_larger_shapes_against_oracle
test_class_api_als_matches_oracle = G.test_class_api_als_matches_oracle
test_class_api_reference_test_case = G.test_class_api_reference_test_case
test_class_api_low_rank_optimizers = G.test_class_api_low_rank_optimizers
test_class_api_low_rank_optimizers_randomsvd = G.test_class_api_low_rank_optimizers_randomsvd
test_tensor_refill_while_session_alive = G.test_tensor_refill_while_session_alive
test_bench_mode_matches_oracle = G.test_bench_mode_matches_oracle
test_long_run_factor_parity = G.test_long_run_factor_parity
test_rank_stream_on_matrix_cores = G.test_rank_stream_on_matrix_cores

import test_gpu_tucker as GT  # noqa: E402

test_ttmc_matches_oracle = GT.test_ttmc_matches_oracle
test_hosvd_and_dt_sweeps = GT.test_hosvd_and_dt_sweeps
test_tucker_pp_driver_matches_oracle = GT.test_tucker_pp_driver_matches_oracle
test_tucker_bench_mode_matches_oracle = GT.test_tucker_bench_mode_matches_oracle
test_eigen_step_projector_route_matches_oracle = GT.test_eigen_step_projector_route_matches_oracle
test_tensor_p_laplacian = G.test_tensor_p_laplacian
test_tensor_c_collinear = G.test_tensor_c_collinear
test_tall_unfolding_thin_route_matches_oracle = GT.test_tall_unfolding_thin_route_matches_oracle
test_tall_unfolding_rank_deficient_falls_back = GT.test_tall_unfolding_rank_deficient_falls_back
test_chain_order_of_the_first_level_products = GT.test_chain_order_of_the_first_level_products


def test_context_destroyed_before_its_children(pp):
    """a garbage-collected binding (or an exception on the way out) may destroy the context while
    sessions / tensors are still alive: ppals_ctx_destroy tears them down itself and leaves DEAD
    handles — destroying them later is a no-op, using them an error, never a use-after-free"""
    import ctypes as C
    c = pp.Context(0)
    t = pp.Tensor(c, [6, 5, 4], 1).fill_uniform(3)
    s = pp.CP(c, t, 2)
    k = pp.Tucker(c, t, [2, 2, 2])
    s.set_factors(pp.init_factors([6, 5, 4], 2, 1))
    s.sweeps_dt(1)
    pp.lib().ppals_ctx_destroy(c._h)   # behind the binding's back: children still alive
    c._h = C.c_void_p()
    with pytest.raises(pp.PpalsError):
        s.sweeps_dt(1)
    with pytest.raises(pp.PpalsError):
        k.hosvd()
    with pytest.raises(pp.PpalsError):
        t.norm()
    k.close()
    s.close()
    t.close()


def test_placement_measurement_leaves_results_alone(pp, monkeypatch):
    """the ONLINE placement choice of the multi-sweep schedule (engine.cpp ms_place_pick: the first
    ~20 visits of every root run the sweep's own scan at a different offset / store kind of the
    result, under a stopwatch) only chooses WHERE a buffer lies: with the size threshold lowered so
    that a small tensor goes through all of it, sweeps give bit-identical factors to a session that
    explores nothing, every root settles, and nothing was spent at set-up — and the machinery runs
    under the sanitizer build (test_sanitizers.py)"""
    lens, R = [9, 8, 7, 6], 3
    W0 = pp.init_factors(lens, R, 5)
    reports = {}

    def run(name, env, sweeps):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = pp.Context(0)
        t = pp.Tensor(c, lens, 1).fill_uniform(11)
        s = pp.CP(c, t, R)
        s.set_factors(W0)
        s.sweeps_dt(sweeps)
        out = [w.copy() for w in s.get_factors()]
        reports[name] = s.placement_report()
        s.close()
        t.close()
        c.close()
        for k in env:
            monkeypatch.delenv(k)
        return out

    for sweeps in (4, 80):      # in the middle of the exploration / after every root has settled
        plain = run("plain", {"PPALS_PLACE_TUNE": "0"}, sweeps)
        tuned = run("tuned", {"PPALS_PLACE_MIN_MB": "0"}, sweeps)
        for a, b in zip(plain, tuned):
            assert (a == b).all()
    assert reports["plain"]["mode"] == "off" and reports["plain"]["roots"] == []
    rep = reports["tuned"]
    assert rep["mode"] == "online" and rep["setup_s"] == 0.0 and len(rep["roots"]) in (2, 4)  # (root sets of 2 or 1 modes)
    for r in rep["roots"]:
        assert r["settled"] and r["visits"] == 14 + 4 + 6 and r["worst_ms"] >= r["best_ms"] > 0
    # (the stand-in's stopwatch is a hash of the call count: the roots do not all agree)
    assert len({(r["block"], r["offset_mb"], r["store"]) for r in rep["roots"]}) > 1


def test_placement_exploration_ends_early_without_evidence():
    """the evidence gate of the online placement choice (engine.cpp ms_place_pick, round 6): when a
    root's first 8 samples lie within 4 % of each other the exploration ends there — the root keeps its
    fastest candidate of the FIRST block and the second candidate block is released — instead of
    walking all 24 candidates for nothing (what five rounds of driver-box headlines showed). A fresh
    process: the stand-in reads its stopwatch spread once."""
    import subprocess
    import sys
    code = """
import os, sys, json
os.environ["PPALS_HOSTSIM_TIMER_SPREAD"] = "0.01"
os.environ["PPALS_PLACE_MIN_MB"] = "0"
sys.path.insert(0, %r)
import hostsim_util
pp = hostsim_util.load()
lens, R = [9, 8, 7, 6], 3
c = pp.Context(0)
t = pp.Tensor(c, lens, 1).fill_uniform(11)
s = pp.CP(c, t, R)
s.set_factors(pp.init_factors(lens, R, 5))
s.sweeps_dt(40)
print("REPORT " + json.dumps(s.placement_report()))
""" % os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    rep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("REPORT ")][0][7:])
    assert rep["mode"] == "online" and rep["candidate_blocks_held"] == 1 and rep["roots"]
    for r in rep["roots"]:
        # one sample is read a visit late: the gate sees 8 samples at the head of the 10th visit
        assert r["settled"] and r["gated"] and r["visits"] <= 10 and r["block"] == 0, r


@pytest.mark.parametrize("sched", ["ms", "tree"])
@pytest.mark.parametrize("lens,ranks", [([14, 14, 14], [3, 3, 3]), ([9, 12, 10], [2, 3, 4])])
def test_tucker_order3_multi_sweep_schedule(pp, lens, ranks, sched, monkeypatch):
    """Order 3, one rank: the multi-sweep dimension tree (TuckerEngine::ms3_leaf; three rotations of the
    tensor, one first-level intermediate serving two mode updates, the root rotating) and the per-sweep
    tree (PPALS_TUCKER_CHAIN=tree) give the oracle's iterates after 1 .. 5 sweeps — with deferred steps
    that fail every third check on the way, so that the schedule is rebuilt in the middle of its cycle."""
    import numpy as np
    import oracle_lib as O
    if sched == "tree":
        monkeypatch.setenv("PPALS_TUCKER_CHAIN", "tree")
    monkeypatch.setenv("PPALS_HOSTSIM_DEFER", "1")
    monkeypatch.setenv("PPALS_HOSTSIM_DEFER_FAIL", "3")
    V = O.fill_uniform(int(np.prod(lens)), 13, lo=0.5, hi=1.0).reshape(lens, order="F")
    W0, c0 = O.hosvd(V, ranks)
    c = pp.Context(0)
    t = pp.Tensor(c, lens, 1).upload(V)
    s = pp.Tucker(c, t, ranks)
    for n in (1, 2, 3, 4, 5, 6, 7):   # (every position of the 2-sweep root cycle ends a call once)
        s.set_factors(W0)
        s.set_core(c0)
        s.sweeps_dt(n)
        W, core = s.get_factors()
        _, _, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=n - 1, resprint=10 ** 9)
        for a, b in zip(W, W_ref):
            assert np.linalg.norm(a @ a.T - b @ b.T) < 1e-10, (sched, n)
        assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-11 * np.linalg.norm(core_ref)
        # the lazily computed core (TuckerEngine::ensure_core: the live leaf borrowed, a transposition for
        # front-stored leaves, a recompute when the multi-sweep root permuted the leaf) entry by entry
        # against V x_i W_i^T of the RETURNED factors — the norm alone would not see a permuted core
        want = core_from_factors(V, W)
        assert core.shape == want.shape and np.abs(core - want).max() < 1e-10 * np.abs(want).max(), (sched, n)
    s.close()
    t.close()
    c.close()


@pytest.mark.parametrize("lens,ranks", [([12, 10, 9], [3, 4, 2]), ([9, 8, 7, 6], [3, 2, 3, 2])])
@pytest.mark.parametrize("fail_every", [0, 1, 2, 3, 7])
def test_tucker_deferred_eigen_steps_roll_back(pp, lens, ranks, fail_every, monkeypatch, tmp_path):
    """Deferred acceptance of eigen-steps (Ops::eig_defer / eig_verify, tucker.cpp settle_mode /
    rollback_and_redo): the stand-in returns every warm step of a plain sweep UNCHECKED and, every
    n-th time, with a wrong basis that the later check reports. The engine must notice before it
    steps the mode again or prints a row, put back every factor stepped since and repeat the work:
    CSV rows, projectors and ||core|| equal to the oracle's whatever the failure pattern."""
    import numpy as np
    import oracle_lib as O
    monkeypatch.setenv("PPALS_HOSTSIM_DEFER", "1")
    monkeypatch.setenv("PPALS_HOSTSIM_DEFER_FAIL", str(fail_every))
    monkeypatch.setenv("PPALS_TUCKER_THIN", "0")
    V = O.fill_uniform(int(np.prod(lens)), 4, lo=0.5, hi=1.0).reshape(lens, order="F")
    W0, c0 = O.hosvd(V, ranks)
    c_ref, c_got = str(tmp_path / "r.csv"), str(tmp_path / "g.csv")
    kw = dict(tol=0.0, maxiter=7, resprint=3)
    _, it_ref, W_ref, core_ref = O.als_tucker_dt(V, W0, c0, csv=c_ref, **kw)
    c = pp.Context(0)
    t = pp.Tensor(c, lens, 1).upload(V)
    s = pp.Tucker(c, t, ranks)
    s.hosvd()
    s.set_factors(W0)
    rc, it = s.run_dt(csv=c_got, **kw)
    assert it == it_ref
    W, core = s.get_factors()
    for a, b in zip(W, W_ref):
        assert np.linalg.norm(a @ a.T - b @ b.T) < 1e-9
    assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < 1e-10 * np.linalg.norm(core_ref)
    want = core_from_factors(V, W)     # entry by entry, from the returned factors (orders 3 and 4)
    assert core.shape == want.shape and np.abs(core - want).max() < 1e-10 * np.abs(want).max()
    _, r1 = O.read_csv(c_ref)
    _, r2 = O.read_csv(c_got)
    assert len(r1) == len(r2)
    for a, b in zip(r1, r2):
        assert a[1] == b[1] and abs(a[2] - b[2]) < 1e-9 and abs(a[5] - b[5]) < 1e-9 * np.linalg.norm(V)
    # the sweep entry point: every step behind the factors it returns has been checked
    # (one session, call after call: the failure pattern walks over every mode and sweep position —
    # this is where a leaf left valid by a repeated step was once taken for the next sweep's)
    for n in (2, 5, 6, 7):
        s.set_factors(W0)
        s.set_core(None)
        s.sweeps_dt(n)
        Wn, core_n = s.get_factors()
        _, _, Wn_ref, _ = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=n - 1, resprint=10 ** 9)
        for a, b in zip(Wn, Wn_ref):
            assert np.linalg.norm(a @ a.T - b @ b.T) < 1e-9, (n, fail_every)
        want = core_from_factors(V, Wn)
        assert np.abs(core_n - want).max() < 1e-10 * np.abs(want).max(), (n, fail_every)
    s.close()
    t.close()
    c.close()
