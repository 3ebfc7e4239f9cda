"""Ranks of the P > 1 run of the HIP kernels on ONE GPU (TEST INFRASTRUCTURE; launched by
tests/test_gpu_multirank.py, never by the product).

The product's engine + C ABI + HIP kernels (tests/hipsim: libppals_hipsim.so, a staged callback
communicator in place of RCCL) on a tensor block-partitioned along its leading mode, P ranks sharing
the one device, every result against the UNSHARDED fp64 oracle (or the closed form of
tests/rank_structured.py at BASELINE's full size).

  process mode   RANK / WORLD_SIZE / MASTER_* in the environment, one rank per process, gloo behind
                 the callbacks:   python hipsim_rank.py <case>
  thread mode    one process, P threads, hipsim_util.ThreadWorld behind the callbacks (a GPU box
                 allows 6 processes on its card; world 8 needs this):
                                  python hipsim_rank.py <case> --threads P

cases: cp_mid, cp_plans, tucker_mid, cfg4 (BASELINE configs[3] at full size, P = 8: 8 x (12.8 GB
shard + its second layout) on one MI355X), cfg5 (configs[4], Tucker s = 400), tiny (the hostsim
cases of tests/hostsim_rank.py over the HIP kernels; process mode only)."""
import argparse
import os
import sys
import threading
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hipsim_util  # noqa: E402
import oracle_lib as O  # noqa: E402
import rank_structured as RS  # noqa: E402


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


class ProcWorld:
    """one rank per process: torch.distributed / gloo"""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank0, self.size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=self.rank0, world_size=self.size)
        self.calls = {}

    def comm_uid(self, rank):
        uid, cbs, calls = hipsim_util.gloo_comm_uid(rank, self.size)
        self.calls[rank] = calls
        return uid, cbs

    def barrier(self):
        self.dist.barrier()

    def setenv(self, rank, key, val):
        if val is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = str(val)

    def once(self, rank, key, fn):
        return fn()

    def calls_of(self, rank):
        return self.calls[rank]


class ThreadRanks(hipsim_util.ThreadWorld):
    """P ranks as threads; what the oracle computes is computed ONCE (rank 0) and shared"""

    def __init__(self, size):
        super().__init__(size)
        self._once = {}

    def once(self, rank, key, fn):
        if rank == 0:
            self._once[key] = fn()
        self.barrier()
        return self._once[key]

    def calls_of(self, rank):
        return self.calls[rank]


def nonempty(lens, P):
    lens = list(lens)
    while -(-lens[0] // P) * (P - 1) >= lens[0]:
        lens[0] += 1
    return lens


# ------------------------------------------------------------------------------------------------
def cp_cases(pp, ctx, w, rank, shapes, pp_driver=True):
    """CP on mid-size shapes (the real scan kernels: persistent one-tile, two n-tiles, fp64, the
    second resident layout, padded layouts where the strides ask for them): MTTKRP of every mode,
    residual, exact sweeps under both schedules with factors AND gradients, the PP driver, `-pp 2`
    — on both collective plans — against the unsharded oracle."""
    P = w.size
    for case_no, (lens, R, dtype) in enumerate(shapes):
        lens = nonempty(lens, P)
        N = len(lens)

        def problem():
            V = O.build_V(O.init_factors(lens, R, 1234))
            W = O.init_factors(lens, R, 4321)
            G = O.init_factors(lens, R, 99)
            K = 3
            _, _, W_ref, G_ref = O.als_cp_dt(V, W, G, tol=0.0, maxiter=K - 1, resprint=1000)
            M_ref = [O.mttkrp(V, W, m, 0) for m in range(N)]
            Vn = np.linalg.norm(V)
            kw = dict(tol=1e-6 * Vn, tol_init=0.1, maxiter=24, resprint=1000)
            pp_ref = O.als_cp_pp(V, W, G, **kw)[1:3] if pp_driver else None
            kw2 = dict(tol=1e-6 * Vn, tol_init=0.1, maxiter=16, resprint=1000)
            pu_ref = (O.als_cp_pp_partupdate(V, W, G, update_percentage=0.5, **kw2)[1:3]
                      if pp_driver and dtype == 1 else None)
            return dict(V=V, W=W, G=G, K=K, W_ref=W_ref, G_ref=G_ref, M_ref=M_ref, Vn=Vn, kw=kw,
                        kw2=kw2, pp_ref=pp_ref, pu_ref=pu_ref, res0=O.residual(V, W),
                        res_ref=O.residual(V, W_ref))

        pr = w.once(rank, ("cp", case_no), problem)
        V, W, G, K = pr["V"], pr["W"], pr["G"], pr["K"]
        t = pp.Tensor(ctx, lens, dtype).upload(V)
        lo, n = t.local_rows()
        blk = -(-lens[0] // P)
        assert lo == rank * blk and n == min(blk, lens[0] - lo), (lo, n, blk)
        assert abs(t.norm() - pr["Vn"]) < 1e-6 * pr["Vn"]
        ktol = 1e-10 if dtype == 1 else 2e-6
        ftol = 1e-8 if dtype == 1 else 1e-5
        for plan in ("0", str(1 << 20)):   # reduce-scatter + all-gather / one all-reduce
            w.setenv(rank, "PPALS_COMM_SMALL_BYTES", plan)
            for schedule in ("msdt", "dt"):
                s = pp.CP(ctx, t, R)
                s.set_schedule(schedule)
                s.set_factors(W, G)
                if schedule == "msdt":
                    for m in range(N):
                        e = relerr(s.mttkrp(m), pr["M_ref"][m])
                        assert e < ktol, (lens, plan, m, e)
                    assert abs(s.residual() - pr["res0"]) < 1e-5 * pr["res0"]
                before = dict(w.calls_of(rank))
                s.sweeps_dt(K)
                after = w.calls_of(rank)
                if plan == "0":   # every mode: reduce-scatter (not the partitioned one) + all-gather
                    assert after["rs"] - before["rs"] == K * (N - 1), (before, after)
                    assert after["ag"] - before["ag"] == K * N, (before, after)
                else:
                    assert after["rs"] == before["rs"], (before, after)
                W_got, G_got = s.get_factors(with_grad=True)
                for i, (a, b) in enumerate(zip(W_got, pr["W_ref"])):
                    assert relerr(a, b) < ftol, (lens, plan, schedule, i, relerr(a, b))
                for a, b in zip(G_got, pr["G_ref"]):
                    assert np.linalg.norm(a - b) < (1e-7 if dtype == 1 else 1e-3) * (1 + np.linalg.norm(b))
                gn_ref = np.sqrt(sum(np.linalg.norm(g) ** 2 for g in pr["G_ref"]))
                assert abs(s.gradnorm() - gn_ref) < (1e-8 if dtype == 1 else 1e-3) * gn_ref + 1e-9
                assert abs(s.residual() - pr["res_ref"]) < 1e-5 * pr["Vn"]
                s.close()
            if pp_driver:
                s = pp.CP(ctx, t, R)
                s.set_factors(W, G)
                _, it = s.run_pp(**pr["kw"])
                it_ref, W_pp_ref = pr["pp_ref"]
                if dtype == 1:
                    assert it == it_ref, (it, it_ref)
                    for a, b in zip(s.get_factors(), W_pp_ref):
                        assert relerr(a, b) < 1e-6, (lens, plan, relerr(a, b))
                else:
                    # fp32 storage may switch phases at other sweeps: the fit, not the iterates
                    res_pp = O.residual(V, W_pp_ref)
                    assert s.residual() < 1.2 * res_pp + 1e-3 * pr["Vn"], (s.residual(), res_pp)
                if pr["pu_ref"] is not None and plan != "0":
                    s.set_factors(W, G)
                    _, it2 = s.run_pp_partupdate(update_percentage=0.5, **pr["kw2"])
                    assert it2 == pr["pu_ref"][0], (it2, pr["pu_ref"][0])
                    for a, b in zip(s.get_factors(), pr["pu_ref"][1]):
                        assert relerr(a, b) < 1e-6, relerr(a, b)
                s.close()
        w.setenv(rank, "PPALS_COMM_SMALL_BYTES", None)
        t.close()
        if rank == 0:
            print(f"  cp case {case_no} {lens} R={R} dt={dtype}: ok", flush=True)
    c = w.calls_of(rank)
    assert c["rs"] > 0 and c["ag"] > 0 and c["ar"] > 0, c


def tucker_cases(pp, ctx, w, rank, shapes):
    """Tucker on mid-size shapes (modes above 64 rows: the projector eigen-step, deferred checks
    off on the sharded path): hosvd, alsTucker_DT, TTMc, alsTucker_PP against the unsharded oracle"""
    P = w.size

    def proj(U):
        return U @ U.T

    for case_no, (lens, ranks, dtype) in enumerate(shapes):
        lens = nonempty(lens, P)

        def problem():
            V = O.fill_uniform(int(np.prod(lens)), 21, lo=0.5, hi=1.0).reshape(lens, order="F")
            W_ref, core_ref = O.hosvd(V, ranks)
            _, it_ref, W2_ref, core2_ref = O.als_tucker_dt(V, W_ref, core_ref, tol=0.0, maxiter=3,
                                                           resprint=1000)
            Y = {skip: O.ttmc(V, W2_ref, skip) for skip in (-1, 1)}
            ppr = None
            if dtype == 1:
                kwp = dict(tol=0.0, tol_init=0.1, maxiter=8, resprint=1000)
                ppr = O.als_tucker_pp(V, W_ref, core_ref, **kwp)[1:3]
            return dict(V=V, W_ref=W_ref, core_ref=core_ref, it_ref=it_ref, W2_ref=W2_ref,
                        core2_ref=core2_ref, Y=Y, ppr=ppr)

        pr = w.once(rank, ("tk", case_no), problem)
        V = pr["V"]
        Vn = np.linalg.norm(V)
        t = pp.Tensor(ctx, lens, dtype).upload(V)
        tk = pp.Tucker(ctx, t, ranks)
        tk.hosvd()
        Wg, core = tk.get_factors()
        tol = 1e-8 if dtype == 1 else 1e-4
        for a, b in zip(Wg, pr["W_ref"]):
            assert np.linalg.norm(proj(a) - proj(b)) < tol * 10, np.linalg.norm(proj(a) - proj(b))
        assert abs(np.linalg.norm(core) - np.linalg.norm(pr["core_ref"])) < tol * np.linalg.norm(pr["core_ref"])
        tk.set_factors(pr["W_ref"])
        rc, it = tk.run_dt(tol=0.0, maxiter=3, resprint=1000)
        W2, core2 = tk.get_factors()
        assert it == pr["it_ref"]
        for a, b in zip(W2, pr["W2_ref"]):
            assert np.linalg.norm(proj(a) - proj(b)) < tol * 100, np.linalg.norm(proj(a) - proj(b))
        assert abs(np.linalg.norm(core2) - np.linalg.norm(pr["core2_ref"])) < tol * 10 * np.linalg.norm(pr["core2_ref"])
        # the core belongs to the returned factors, entry by entry
        assert np.linalg.norm(core2 - O.ttmc(V, W2, -1)) < tol * 10 * Vn
        tk.set_factors(pr["W2_ref"])
        for skip in (-1, 1):
            assert np.linalg.norm(tk.ttmc(skip) - pr["Y"][skip]) < tol * 10 * Vn
        if pr["ppr"] is not None:
            kwp = dict(tol=0.0, tol_init=0.1, maxiter=8, resprint=1000)
            tk.hosvd()
            tk.set_factors(pr["W_ref"])
            _, itp = tk.run_pp(**kwp)
            Wp, _ = tk.get_factors()
            assert itp == pr["ppr"][0], (itp, pr["ppr"][0])
            for a, b in zip(Wp, pr["ppr"][1]):
                assert np.linalg.norm(proj(a) - proj(b)) < 1e-5
        tk.close()
        t.close()
        if rank == 0:
            print(f"  tucker case {case_no} {lens} ranks={ranks} dt={dtype}: ok", flush=True)


def cfg4_full(pp, ctx, w, rank, s=400, R=20, K=3):
    """BASELINE configs[3] (script/script_strongscaling.py:10,45-46: order 4, s = 400, R = 20) at
    FULL size over P ranks on one MI355X: every rank fills its own shard of the 102 GB fp32 tensor,
    MTTKRP of the partitioned mode and of the last mode, K exact sweeps on the default plan (one
    all-reduce per mode at 64 KB) and 2 more sessions' sweeps on the reduce-scatter plan, against
    the closed form. Timing is ignored: P ranks time-share one device."""
    P = w.size
    lens = [s] * 4
    A = pp.init_factors(lens, R, 1000)
    W = pp.init_factors(lens, R, 2000)
    G = pp.init_factors(lens, R, 3000)
    t0 = time.time()
    V = pp.Tensor(ctx, lens, 0).fill_cp(A)
    lo, n = V.local_rows()
    assert n == s // P and lo == rank * n
    vn = V.norm()
    assert abs(vn - RS.norm(A)) < 2e-6 * RS.norm(A), (vn, RS.norm(A))
    if rank == 0:
        print(f"  cfg4: shards of {n} rows filled in {time.time() - t0:.1f} s", flush=True)
    ref = w.once(rank, "cfg4_ref", lambda: dict(
        M0=RS.mttkrp(A, W, 0), M3=RS.mttkrp(A, W, 3), Wk=RS.als_cp_dt(A, W, G, K)[0],
        W2=RS.als_cp_dt(A, W, G, 2)[0]))
    for plan, sweeps, key in ((str(1 << 20), K, "Wk"), ("0", 2, "W2")):
        w.setenv(rank, "PPALS_COMM_SMALL_BYTES", plan)
        t0 = time.time()
        cp = pp.CP(ctx, V, R)
        cp.set_factors(W, G)
        if plan != "0":
            for i, Mr in ((0, ref["M0"]), (3, ref["M3"])):
                e = relerr(cp.mttkrp(i), Mr)
                assert e < 2e-6, (i, e)
        before = dict(w.calls_of(rank))
        cp.sweeps_dt(sweeps)
        after = w.calls_of(rank)
        if plan == "0":
            assert after["rs"] - before["rs"] == sweeps * 3 and after["ag"] - before["ag"] == sweeps * 4
        worst = max(relerr(a, b) for a, b in zip(cp.get_factors(), ref[key]))
        assert worst < 1e-5, (plan, worst)
        if rank == 0:
            print(f"  cfg4 plan small_bytes={plan}: {sweeps} sweeps, worst factor error {worst:.2e}, "
                  f"{time.time() - t0:.1f} s", flush=True)
        cp.close()
    w.setenv(rank, "PPALS_COMM_SMALL_BYTES", None)
    V.close()


def cfg5_full(pp, ctx, w, rank, s=400, r=20):
    """BASELINE configs[4] (Tucker order 3, s = 400, core 20^3) over P ranks on one device: HOSVD +
    HOOI sweeps on an exact-multilinear-rank tensor + noise-free closed form: the fit must be exact
    (residual ~ 0), factors orthonormal and the core equal to V x_i W_i^T of the returned factors."""
    P = w.size
    lens = [s] * 3
    A = pp.init_factors(lens, r, 1000)
    V = pp.Tensor(ctx, lens, 0).fill_cp(A)      # CP rank 20 => multilinear rank (20, 20, 20)
    Vn = RS.norm(A)
    assert abs(V.norm() - Vn) < 2e-6 * Vn
    tk = pp.Tucker(ctx, V, [r] * 3)
    tk.hosvd()
    tk.sweeps_dt(3)
    Wg, core = tk.get_factors()
    for U, Ai in zip(Wg, A):
        assert np.linalg.norm(U.T @ U - np.eye(r)) < 1e-9
        # span(U) = span(A_i): the projector leaves A_i alone
        assert np.linalg.norm(U @ (U.T @ Ai) - Ai) < 2e-5 * np.linalg.norm(Ai)
    # ||core|| = ||V|| for an exact fit (als_Tucker.cxx:291-294's metric)
    assert abs(np.linalg.norm(core) - Vn) < 1e-5 * Vn, (np.linalg.norm(core), Vn)
    # core = [[U_0^T A_0, U_1^T A_1, U_2^T A_2]] entry by entry
    B = [U.T @ Ai for U, Ai in zip(Wg, A)]
    core_ref = np.einsum("ar,br,cr->abc", *B)
    assert np.linalg.norm(core - core_ref) < 1e-5 * Vn
    if rank == 0:
        print("  cfg5: HOSVD + 3 HOOI sweeps, exact fit recovered", flush=True)
    tk.close()
    V.close()


CP_MID = [([96, 64, 48, 40], 10, 0), ([100, 56, 48, 36], 8, 1), ([160, 120, 96], 16, 1),
          ([80, 64, 48, 40], 20, 0), ([18, 12, 10, 8, 8, 6], 4, 1)]
CP_PLANS = [([64, 64, 64, 64], 10, 0), ([48, 40, 36, 50], 6, 1)]
CP_SMALL = [([12, 8, 6, 5], 3, 1), ([10, 9, 7], 4, 0)]            # CPU rehearsal of the script
TUCKER_SMALL = [([10, 9, 8], [3, 2, 3], 1)]
TUCKER_MID = [([96, 80, 72], [8, 6, 7], 1), ([72, 48, 40, 36], [4, 3, 4, 3], 0),
              ([128, 96, 80], [12, 10, 8], 0)]


def run_case(case, pp, ctx, w, rank):
    if case == "cp_mid":
        cp_cases(pp, ctx, w, rank, CP_MID)
    elif case == "cp_plans":
        cp_cases(pp, ctx, w, rank, CP_PLANS, pp_driver=False)
    elif case == "tucker_mid":
        tucker_cases(pp, ctx, w, rank, TUCKER_MID)
    elif case == "cp_small":
        cp_cases(pp, ctx, w, rank, CP_SMALL)
    elif case == "tucker_small":
        tucker_cases(pp, ctx, w, rank, TUCKER_SMALL)
    elif case == "cfg4":
        cfg4_full(pp, ctx, w, rank)
    elif case == "cfg4_small":      # the same body at s = 64 (CPU-side rehearsal of the script)
        cfg4_full(pp, ctx, w, rank, s=64, R=20)
    elif case == "cfg5":
        cfg5_full(pp, ctx, w, rank)
    elif case == "tiny":
        import hostsim_rank
        hostsim_rank.default_cases(pp, ctx, rank, w.size, w.calls_of(rank), hostsim_rank.relerr)
    elif case == "tiny_rs_plan":
        import hostsim_rank
        hostsim_rank.rs_plan_cases(pp, ctx, rank, w.size, w.calls_of(rank), hostsim_rank.relerr)
    elif case == "tiny_rs_unequal":
        import hostsim_rank
        hostsim_rank.rs_unequal_cases(pp, ctx, rank, w.size, w.calls_of(rank), hostsim_rank.relerr)
    else:
        raise SystemExit(f"unknown case {case}")


def one_rank(case, pp, w, rank):
    ctx = pp.Context(0)
    uid, keep = w.comm_uid(rank)
    ctx.init_comm(rank, w.size, uid)
    assert ctx.nranks == w.size and ctx.rank == rank
    try:
        run_case(case, pp, ctx, w, rank)
        w.barrier()
    finally:
        ctx.close()
    del keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--backend", default="hipsim", choices=["hipsim", "hostsim"],
                    help="hostsim: the CPU rehearsal of THIS script (tests/test_hipsim_script_cpu.py)")
    a = ap.parse_args()
    if not a.threads:
        import torch  # noqa: F401  (first: the library then shares torch's HIP runtime, as in bench.py)
    if a.backend == "hostsim":
        import hostsim_util
        pp = hostsim_util.load()
    else:
        pp = hipsim_util.load(make=False)
        pp.preload_eigensolver()    # before anything initialises the HIP runtime (include/ppals.h)
    if a.threads:
        assert not a.case.startswith("tiny"), "the hostsim cases set the environment per rank"
        w = ThreadRanks(a.threads)
        errors = []

        def body(rank):
            try:
                one_rank(a.case, pp, w, rank)
            except BaseException:
                errors.append((rank, traceback.format_exc()))
                w.abort()

        ths = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(a.threads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        real = [e for e in errors if "BrokenBarrierError" not in e[1]] or errors
        if real or w.failed:
            for r, tb in real[:2]:
                print(f"rank {r} FAILED:\n{tb}")
            print("callback failures:", w.failed[:4])
            sys.exit(1)
        print(f"all {a.threads} ranks: OK", w.calls[0])
    else:
        w = ProcWorld()
        one_rank(a.case, pp, w, w.rank0)
        w.dist.destroy_process_group()
        print(f"rank {w.rank0}: OK", w.calls_of(w.rank0))


if __name__ == "__main__":
    main()
