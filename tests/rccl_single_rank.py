"""Single-rank RCCL rehearsal on the GPU (TEST INFRASTRUCTURE; launched by test_gpu_rccl.py).

A 1-GPU box cannot run two RCCL ranks, but with PPALS_FORCE_COMM=1 the engine takes its sharded
code paths (pack -> reduce-scatter -> row-block update -> all-gather, or all-reduce + redundant
update; all-reduced Tucker leaves and Grams) through a real one-rank RCCL communicator on the
engine's stream: library loading, symbol binding, datatype/op enums, stream ordering and in-place
buffers are exercised exactly as with N ranks. Results must match the unsharded fp64 oracle."""
import os
import sys

import numpy as np

os.environ["PPALS_FORCE_COMM"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402


def relerr(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def run(ppals, ctx):
    """the checks proper; `ppals` is the product binding (GPU) or the host stand-in (CPU test)"""
    assert ctx.nranks == 1 and ctx.rank == 0
    for case_no, (lens, R, dtype) in enumerate([([24, 20, 16, 12], 5, ppals.F64),
                                                ([40, 30, 20, 10], 8, ppals.F32),
                                                ([30, 20, 25], 4, ppals.F64)]):
        for plan in ("0", str(1 << 20)):  # reduce-scatter + all-gather / one all-reduce
            os.environ["PPALS_COMM_SMALL_BYTES"] = plan
            Wt = O.init_factors(lens, R, 1234)
            V = O.build_V(Wt)
            W = O.init_factors(lens, R, 4321)
            G = O.init_factors(lens, R, 99)
            t = ppals.Tensor(ctx, lens, dtype).upload(V)
            assert abs(t.norm() - np.linalg.norm(V)) < 1e-6 * np.linalg.norm(V)
            s = ppals.CP(ctx, t, R)
            s.set_factors(W, G)
            tol = 1e-10 if dtype == ppals.F64 else 1e-5
            for mode in range(len(lens)):
                assert relerr(s.mttkrp(mode), O.mttkrp(V, W, mode, 0)) < tol, (lens, mode)
            assert abs(s.residual() - O.residual(V, W)) < 1e-5 * O.residual(V, W)
            K = 4
            _, _, W_ref, G_ref = O.als_cp_dt(V, W, G, tol=0.0, maxiter=K - 1, resprint=1000)
            s.sweeps_dt(K)
            W_dev, G_dev = s.get_factors(with_grad=True)
            for i in range(len(lens)):
                assert relerr(W_dev[i], W_ref[i]) < tol, (lens, plan, i, relerr(W_dev[i], W_ref[i]))
                assert relerr(G_dev[i], G_ref[i]) < 50 * tol, (lens, plan, i)
            gn = np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G_ref))
            assert abs(s.gradnorm() - gn) < 50 * tol * gn
            # PP driver through the sharded path
            s.set_factors(W, G)
            stop, it = s.run_pp(tol=1e-7 * np.linalg.norm(V), maxiter=30, tol_init=0.1)
            _, it_ref, W_pp, _ = O.als_cp_pp(V, W, G, tol=1e-7 * np.linalg.norm(V), maxiter=30,
                                             tol_init=0.1)
            W_dev = s.get_factors()
            if dtype == ppals.F64:
                assert it == it_ref, (it, it_ref)
                for i in range(len(lens)):
                    assert relerr(W_dev[i], W_pp[i]) < 1e-6, (lens, plan, i, relerr(W_dev[i], W_pp[i]))
            if dtype == ppals.F64 and plan != "0":  # -pp 2 needs the complete-matrix plan
                s.set_factors(W, G)
                _, it2 = s.run_pp_partupdate(tol=1e-7 * np.linalg.norm(V), maxiter=20, tol_init=0.1,
                                             update_percentage=0.5)
                _, it2_ref, W_pu, _ = O.als_cp_pp_partupdate(V, W, G, tol=1e-7 * np.linalg.norm(V),
                                                             tol_init=0.1, maxiter=20,
                                                             update_percentage=0.5)
                assert it2 == it2_ref, (it2, it2_ref)
                for a, b in zip(s.get_factors(), W_pu):
                    assert relerr(a, b) < 1e-6
            s.close()
            t.close()
    # Tucker through the all-reduced leaves / Grams and the gathered HOSVD
    lens, ranks = [20, 16, 12], [4, 3, 2]
    V = O.fill_uniform(int(np.prod(lens)), 4, lo=0.5, hi=1.0).reshape(lens, order="F")
    t = ppals.Tensor(ctx, lens, ppals.F64).upload(V)
    tk = ppals.Tucker(ctx, t, ranks)
    tk.hosvd()
    W0, core0 = O.hosvd(V, ranks)
    Wd, cd = tk.get_factors()
    assert abs(np.linalg.norm(cd) - np.linalg.norm(core0)) < 1e-9 * np.linalg.norm(core0)
    tk.set_factors(W0)
    rc, it = tk.run_dt(tol=0.0, maxiter=3, resprint=1000)
    _, it_ref, Wr, cr = O.als_tucker_dt(V, W0, core0, tol=0.0, maxiter=3, resprint=1000)
    Wd, cd = tk.get_factors()
    assert it == it_ref
    assert abs(np.linalg.norm(cd) - np.linalg.norm(cr)) < 1e-8 * np.linalg.norm(cr)
    for a, b in zip(Wd, Wr):
        assert np.linalg.norm(a @ a.T - b @ b.T) < 1e-6
    # alsTucker_PP through the same completions
    tk.hosvd()
    tk.set_factors(W0)
    _, itp = tk.run_pp(tol=0.0, tol_init=0.1, maxiter=8, resprint=1000)
    _, itp_ref, Wp, _ = O.als_tucker_pp(V, W0, core0, tol=0.0, tol_init=0.1, maxiter=8,
                                        resprint=1000)
    assert itp == itp_ref
    for a, b in zip(tk.get_factors()[0], Wp):
        assert np.linalg.norm(a @ a.T - b @ b.T) < 1e-5


def main():
    import torch  # noqa: F401  (first: libppals shares torch's HIP runtime and librccl)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "pairwise-perturbation_amd"))
    import ppals
    ctx = ppals.Context(0)
    ctx.init_comm(0, 1, ppals.Context.unique_id())
    run(ppals, ctx)
    print("RCCL single-rank rehearsal: OK")


if __name__ == "__main__":
    main()
