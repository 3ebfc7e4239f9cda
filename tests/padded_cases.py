"""Shared body of the padded-layout tests (tests/test_padded_hostsim.py on the CPU stand-in,
tests/test_gpu_padded.py on the device): the very same CP cases as tests/test_gpu_cp.py, run with
the padded resident layouts forced on (PPALS_PAD_LAYOUT=1, any padding cost accepted), so that
every first-level scan that can reads a padded layout and writes its result compact
(CpEngine::plan_scan, RowPad). Same iterates as the oracle — nothing downstream of a scan may
notice the padding."""
import numpy as np
import pytest

import oracle_lib as O
import test_gpu_cp as G


@pytest.fixture(autouse=True)
def force_padding(monkeypatch):
    monkeypatch.setenv("PPALS_PAD_LAYOUT", "1")


test_tree_nodes_and_mttkrp = G.test_tree_nodes_and_mttkrp
test_pp_operators = G.test_pp_operators
test_dt_sweeps_match_oracle = G.test_dt_sweeps_match_oracle
test_driver_pp_matches_oracle = G.test_driver_pp_matches_oracle
test_driver_pp_partupdate_matches_oracle = G.test_driver_pp_partupdate_matches_oracle
test_schedule_switch_mid_run = G.test_schedule_switch_mid_run
test_msdt_root_counts = G.test_msdt_root_counts
test_edge_shapes = G.test_edge_shapes
test_random_shapes_against_oracle = G.test_random_shapes_against_oracle
test_random_larger_shapes_against_oracle = G.test_random_larger_shapes_against_oracle
test_tensor_refill_while_session_alive = G.test_tensor_refill_while_session_alive


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("lens,R,roots", [([6, 5, 7, 6], 3, 1), ([5, 6, 4, 5, 3, 4], 2, 2),
                                          ([50, 50, 40, 36], 4, 1), ([9, 40, 33, 7, 5], 17, 2)])
def test_padded_layouts_are_read(pp, ctx, lens, R, roots, dtype, tmp_path, monkeypatch):
    """the trace of a few sweeps shows scans on padded layouts (both of them when the order allows
    it), and the iterates are those of the oracle's alsCP_DT"""
    trace = tmp_path / "trace.txt"
    monkeypatch.setenv("PPALS_TRACE_STEPS", str(trace))
    monkeypatch.setenv("PPALS_MSDT_ROOTS", str(roots))
    V, W = G.problem(lens, R, 5, "r")
    Gr = O.init_factors(lens, R, 99)
    K = 5
    _, _, W_ref, G_ref = O.als_cp_dt(V, W, Gr, tol=0.0, maxiter=K - 1, resprint=1000)
    t = pp.Tensor(ctx, lens, dtype).upload(V)
    s = pp.CP(ctx, t, R)
    s.set_schedule("msdt")
    s.set_factors(W, Gr)
    s.sweeps_dt(K)
    W_got, _ = s.get_factors(with_grad=True)
    for a, b in zip(W_got, W_ref):
        assert G.relerr(a, b) < G.FTOL[dtype], G.relerr(a, b)
    s.close()
    t.close()
    lines = trace.read_text().splitlines()
    used = {ln.split("layout=")[1].split()[0] for ln in lines}
    assert "VTpad" in used, used
    if len(lens) >= 4:
        assert "Vpad" in used, used
