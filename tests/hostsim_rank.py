"""One rank of the world_size-N CPU rehearsal of the sharded engine (TEST INFRASTRUCTURE).

Runs the product's engine + C ABI over the host stand-in ops with torch.distributed (gloo) behind
the communicator callbacks, on a tensor block-partitioned along its leading mode, and checks the
result against the unsharded fp64 oracle. Launched by tests/test_multirank_gloo.py."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hostsim_util  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # PPALS_RANK_BACKEND=hipsim: the same cases over the product's HIP kernels with the staged
    # callback communicator (tests/hipsim), P processes sharing the one GPU (test_gpu_multirank.py)
    if os.environ.get("PPALS_RANK_BACKEND") == "hipsim":
        import hipsim_util as util
    else:
        util = hostsim_util
    pp = util.load()
    ctx = pp.Context(0)

    uid, cbs, calls = util.gloo_comm_uid(rank, world)  # (cbs: kept alive with the context)
    ctx.init_comm(rank, world, uid)
    assert ctx.nranks == world and ctx.rank == rank

    mode = os.environ.get("PPALS_RANK_MODE")
    body = {"rs_unequal": rs_unequal_cases, "rs_plan": rs_plan_cases}.get(mode, default_cases)
    body(pp, ctx, rank, world, calls, relerr)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: OK", calls)


def relerr(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def default_cases(pp, ctx, rank, world, calls, relerr):
    """CP (both shard plans alternating, both schedules' traces, PP, -magni, -pp 2) and Tucker
    (hosvd, DT, PP) on a leading-mode block partition against the unsharded oracle"""
    cases = [([10, 7, 6, 5], 3, 1), ([9, 8, 7, 6], 4, 0), ([11, 6, 5], 2, 1),
             ([8, 4, 5, 4, 3, 3], 2, 1)]
    # the degenerate partition (row blocks of ceil(s0/P) leave the last rank empty) is refused on
    # every rank alike
    if world >= 3:
        bad = [world + 1 if world > 3 else 4, 4, 3]    # e.g. s0=5, P=4: blocks of 2 -> 2,2,1,0
        if -(-bad[0] // world) * (world - 1) >= bad[0]:
            try:
                pp.Tensor(ctx, bad, 1)
                raise AssertionError("degenerate partition accepted")
            except pp.PpalsError as e:
                assert "no rows" in str(e)
    for case_no, (lens, R, dtype) in enumerate(cases):
        lens = list(lens)
        while -(-lens[0] // world) * (world - 1) >= lens[0]:
            lens[0] += 1  # keep every rank non-empty
        # alternate the two shard plans of a mode update: one all-reduce + redundant update
        # (small s x R) vs reduce-scatter + row-block update + all-gather
        os.environ["PPALS_COMM_SMALL_BYTES"] = "0" if case_no % 2 else str(1 << 20)
        Wt = O.init_factors(lens, R, 1234)
        V = O.build_V(Wt)
        W = O.init_factors(lens, R, 4321)
        G = O.init_factors(lens, R, 99)
        t = pp.Tensor(ctx, lens, dtype).upload(V)
        lo, n = t.local_rows()
        blk = -(-lens[0] // world)
        assert lo == rank * blk and n == min(blk, lens[0] - lo)
        assert abs(t.norm() - np.linalg.norm(V)) < 1e-6 * np.linalg.norm(V)
        s = pp.CP(ctx, t, R)
        s.set_factors(W, G)
        tol = 1e-10 if dtype == 1 else 2e-6
        for mode in range(len(lens)):
            assert relerr(s.mttkrp(mode), O.mttkrp(V, W, mode, 0)) < tol
        assert abs(s.residual() - O.residual(V, W)) < 1e-6 * O.residual(V, W)
        K = 4
        _, _, W_ref, G_ref = O.als_cp_dt(V, W, G, tol=0.0, maxiter=K - 1, resprint=1000)
        # shard-aware multi-sweep schedule: trace the root set of every first-level scan
        trace = f"/tmp/ppals_gloo_steps_{os.getpid()}_{case_no}.txt"
        if os.path.exists(trace):
            os.remove(trace)
        os.environ["PPALS_TRACE_STEPS"] = trace
        s.sweeps_dt(K)
        del os.environ["PPALS_TRACE_STEPS"]
        steps = [dict(kv.split("=") for kv in ln.split()) for ln in open(trace).read().splitlines()]
        os.remove(trace)
        N = len(lens)
        assert steps, "no first-level scan traced"
        for st in steps:   # no step contracts the partitioned mode 0 first (its X would be a
            root, k = int(st["root"]), int(st["k"])   # partial sum of full global size)
            assert all((root + q) % N != 0 for q in range(k)), st
        # order 4, one root: 3 scans serve 8 mode updates (runs 3, 3, 2) instead of 8/3
        if N == 4 and int(steps[0]["k"]) == 1:
            assert len(steps) == 6 and [int(st["root"]) for st in steps] == [3, 2, 1, 3, 2, 1], steps
        W_got, G_got = s.get_factors(with_grad=True)
        for a, b in zip(W_got, W_ref):
            assert relerr(a, b) < (1e-8 if dtype == 1 else 1e-5), relerr(a, b)
        for a, b in zip(G_got, G_ref):
            assert np.linalg.norm(a - b) < (1e-7 if dtype == 1 else 1e-3) * (1 + np.linalg.norm(b))
        gn_ref = np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G_ref))
        assert abs(s.gradnorm() - gn_ref) < 1e-4 * gn_ref + 1e-9
        # PP driver, sharded, against the unsharded oracle
        Vn = np.linalg.norm(V)
        csv = f"/tmp/ppals_gloo_{os.getpid()}.csv"
        kw = dict(tol=1e-6 * Vn, tol_init=0.1, maxiter=30, resprint=1)
        _, _, W_pp_ref, _ = O.als_cp_pp(V, W, G, csv=csv + ".ref", **kw)
        s.set_factors(W, G)
        s.run_pp(csv=csv if rank == 0 else None, **kw)
        W_pp = s.get_factors()
        if dtype == 1:
            for a, b in zip(W_pp, W_pp_ref):
                assert relerr(a, b) < 1e-6, relerr(a, b)
            if rank == 0:
                _, r1 = O.read_csv(csv + ".ref")
                _, r2 = O.read_csv(csv)
                assert [r[:2] + [r[4]] for r in r1] == [r[:2] + [r[4]] for r in r2]
                assert any(r[4] == 1 for r in r2)
        # -magni != 1 (ratio_step of SVD_solve_mod), sharded, on whichever plan this case runs
        if dtype == 1:
            kwm = dict(tol=1e-6 * Vn, tol_init=0.1, maxiter=20, resprint=1000)
            _, itm_ref, W_m_ref, _ = O.als_cp_pp(V, W, G, ratio_step=0.8, **kwm)
            s.set_factors(W, G)
            _, itm = s.run_pp(ratio_step=0.8, **kwm)
            assert itm == itm_ref, (itm, itm_ref)
            for a, b in zip(s.get_factors(), W_m_ref):
                assert relerr(a, b) < 1e-6, relerr(a, b)
        # partial-update PP (-pp 2), sharded (the plan with complete s x R matrices on every rank)
        if case_no % 2 == 0 and dtype == 1:
            kw2 = dict(tol=1e-6 * Vn, tol_init=0.1, maxiter=25, resprint=1)
            _, it_ref, W_pu_ref, _ = O.als_cp_pp_partupdate(V, W, G, update_percentage=0.5,
                                                            csv=csv + ".ref2", **kw2)
            s.set_factors(W, G)
            _, it_got = s.run_pp_partupdate(csv=csv + "2" if rank == 0 else None,
                                            update_percentage=0.5, **kw2)
            assert it_got == it_ref, (it_got, it_ref)
            for a, b in zip(s.get_factors(), W_pu_ref):
                assert relerr(a, b) < 1e-6, relerr(a, b)
            if rank == 0:
                _, r1 = O.read_csv(csv + ".ref2")
                _, r2 = O.read_csv(csv + "2")
                assert [r[:2] + [r[4]] for r in r1] == [r[:2] + [r[4]] for r in r2]
        s.close()
        t.close()
    # ---- Tucker (HOOI), sharded: hosvd + the DT driver against the unsharded oracle
    def proj(U):
        return U @ U.T

    for lens, ranks, dtype in [([9, 8, 7], [3, 2, 3], 1), ([7, 6, 5, 6], [2, 3, 2, 2], 1),
                               ([10, 6, 8], [3, 3, 2], 0)]:
        lens = list(lens)
        while -(-lens[0] // world) * (world - 1) >= lens[0]:
            lens[0] += 1  # keep every rank non-empty
        V = O.fill_uniform(int(np.prod(lens)), 21, lo=0.5, hi=1.0).reshape(lens, order="F")
        t = pp.Tensor(ctx, lens, dtype).upload(V)
        tk = pp.Tucker(ctx, t, ranks)
        tk.hosvd()
        W, core = tk.get_factors()
        W_ref, core_ref = O.hosvd(V, ranks)
        tol = 1e-8 if dtype == 1 else 1e-4
        for a, b in zip(W, W_ref):
            assert np.linalg.norm(proj(a) - proj(b)) < tol * 10
        assert abs(np.linalg.norm(core) - np.linalg.norm(core_ref)) < tol * np.linalg.norm(core_ref)
        tk.set_factors(W_ref)
        csv = f"/tmp/ppals_gloo_tk_{os.getpid()}.csv"
        _, it_ref, W2_ref, core2_ref = O.als_tucker_dt(V, W_ref, core_ref, tol=0.0, maxiter=3,
                                                       csv=csv + ".ref", resprint=1)
        rc, it = tk.run_dt(tol=0.0, maxiter=3, csv=csv if rank == 0 else None, resprint=1)
        W2, core2 = tk.get_factors()
        assert it == it_ref
        for a, b in zip(W2, W2_ref):
            assert np.linalg.norm(proj(a) - proj(b)) < tol * 100
        assert abs(np.linalg.norm(core2) - np.linalg.norm(core2_ref)) < tol * 10 * np.linalg.norm(core2_ref)
        if rank == 0:
            _, r1 = O.read_csv(csv + ".ref")
            _, r2 = O.read_csv(csv)
            assert len(r1) == len(r2)
            for a, b in zip(r1, r2):
                assert a[1] == b[1] and abs(a[5] - b[5]) < (1e-6 if dtype == 1 else 1e-3) * np.linalg.norm(V)
        for skip in (-1, 1):
            assert np.linalg.norm(tk.ttmc(skip) - O.ttmc(V, W2, skip)) < tol * 10 * np.linalg.norm(V)
        # alsTucker_PP, sharded, against the unsharded oracle (fp64 storage: same phase pattern)
        if dtype == 1:
            kwp = dict(tol=0.0, tol_init=0.1, maxiter=10, resprint=1)
            _, itp_ref, Wp_ref, corep_ref = O.als_tucker_pp(V, W_ref, core_ref, csv=csv + ".ppref",
                                                            **kwp)
            tk.hosvd()
            tk.set_factors(W_ref)
            _, itp = tk.run_pp(csv=csv + ".pp" if rank == 0 else None, **kwp)
            Wp, corep = tk.get_factors()
            assert itp == itp_ref, (itp, itp_ref)
            for a, b in zip(Wp, Wp_ref):
                assert np.linalg.norm(proj(a) - proj(b)) < 1e-5
            if rank == 0:
                _, r1 = O.read_csv(csv + ".ppref")
                _, r2 = O.read_csv(csv + ".pp")
                assert [r[1:2] + r[4:5] for r in r1] == [r[1:2] + r[4:5] for r in r2]
                assert any(r[4] == 1 for r in r2)
        tk.close()
        t.close()
    assert calls["rs"] > 0 and calls["ag"] > 0 and calls["ar"] > 0


def rs_plan_cases(pp, ctx, rank, world, calls, relerr):
    """the plan north_star names — reduce-scatter of the s x R partial MTTKRP rows, row-block solve,
    all-gather of the new rows — for EVERY mode of cfg-shaped problems (cubic, s divisible by the
    world size: BASELINE configs[3] has s = 400 on 8 GPUs), at any world size up to 8; both sweep
    schedules, the PP driver, and the collective counts a sweep must issue."""
    os.environ["PPALS_COMM_SMALL_BYTES"] = "0"   # no message is "small": never the all-reduce plan
    for lens, R, dtype in [([2 * world] * 4, 3, 1), ([world, 2 * world, world + 3], 2, 1),
                           ([2 * world] * 4, 4, 0)]:
        N = len(lens)
        Wt = O.init_factors(lens, R, 1234)
        V = O.build_V(Wt)
        W = O.init_factors(lens, R, 4321)
        G = O.init_factors(lens, R, 99)
        t = pp.Tensor(ctx, lens, dtype).upload(V)
        lo, n = t.local_rows()
        assert n == lens[0] // world and lo == rank * n     # equal shards
        K = 3
        _, _, W_ref, G_ref = O.als_cp_dt(V, W, G, tol=0.0, maxiter=K - 1, resprint=1000)
        for schedule in ("msdt", "dt"):
            s = pp.CP(ctx, t, R)
            s.set_schedule(schedule)
            s.set_factors(W, G)
            before = dict(calls)
            s.sweeps_dt(K)
            # per sweep: a reduce-scatter for every mode but the partitioned one (its rows are
            # complete on their owner) and an all-gather for every mode
            assert calls["rs"] - before["rs"] == K * (N - 1), (calls, before)
            assert calls["ag"] - before["ag"] == K * N, (calls, before)
            W_got, G_got = s.get_factors(with_grad=True)
            for a, b in zip(W_got, W_ref):
                assert relerr(a, b) < (1e-8 if dtype == 1 else 1e-5), (schedule, relerr(a, b))
            for a, b in zip(G_got, G_ref):
                assert np.linalg.norm(a - b) < (1e-7 if dtype == 1 else 1e-3) * (1 + np.linalg.norm(b))
            s.close()
        if dtype == 1:
            Vn = np.linalg.norm(V)
            kw = dict(tol=1e-6 * Vn, tol_init=0.1, maxiter=20, resprint=1000)
            _, it_ref, W_pp_ref, _ = O.als_cp_pp(V, W, G, **kw)
            s = pp.CP(ctx, t, R)
            s.set_factors(W, G)
            _, it = s.run_pp(**kw)
            assert it == it_ref
            for a, b in zip(s.get_factors(), W_pp_ref):
                assert relerr(a, b) < 1e-6, relerr(a, b)
            s.close()
        t.close()
    assert calls["rs"] > 0 and calls["ag"] > 0


def rs_unequal_cases(pp, ctx, rank, world, calls, relerr):
    """the reduce-scatter / all-gather plan for every mode with UNEQUAL shards: the leading extent is
    not a multiple of the world size, so the last rank holds fewer rows and the row blocks of every
    other mode's s x R partials are padded to a uniform height — the case a one-rank communicator on
    the GPU box can never reach. fp64 storage, both schedules, factors and gradients against the
    unsharded oracle, collective counts per sweep."""
    os.environ["PPALS_COMM_SMALL_BYTES"] = "0"
    for lens, R in [([2 * world + 1, 2 * world + 1, world + 2, world + 1], 3),
                    ([3 * world - 1, 2 * world + 1, 2 * world + 3], 2)]:
        N = len(lens)
        V = O.build_V(O.init_factors(lens, R, 1234))
        W = O.init_factors(lens, R, 4321)
        G = O.init_factors(lens, R, 99)
        t = pp.Tensor(ctx, lens, 1).upload(V)
        lo, n = t.local_rows()
        blk = -(-lens[0] // world)
        assert lo == rank * blk and n == min(blk, lens[0] - lo)
        assert lens[0] % world != 0 and (rank < world - 1 or n < blk)      # the last rank is short
        K = 3
        _, _, W_ref, G_ref = O.als_cp_dt(V, W, G, tol=0.0, maxiter=K - 1, resprint=1000)
        for schedule in ("msdt", "dt"):
            s = pp.CP(ctx, t, R)
            s.set_schedule(schedule)
            s.set_factors(W, G)
            before = dict(calls)
            s.sweeps_dt(K)
            assert calls["rs"] - before["rs"] == K * (N - 1), (calls, before)
            assert calls["ag"] - before["ag"] == K * N, (calls, before)
            W_got, G_got = s.get_factors(with_grad=True)
            for a, b in zip(W_got, W_ref):
                assert relerr(a, b) < 1e-8, (schedule, relerr(a, b))
            for a, b in zip(G_got, G_ref):
                assert np.linalg.norm(a - b) < 1e-7 * (1 + np.linalg.norm(b))
            gn_ref = np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G_ref))
            assert abs(s.gradnorm() - gn_ref) < 1e-8 * gn_ref
            assert abs(s.residual() - O.residual(V, W_ref)) < 1e-8 * np.linalg.norm(V)
            s.close()
        t.close()
    assert calls["rs"] > 0 and calls["ag"] > 0


if __name__ == "__main__":
    main()
