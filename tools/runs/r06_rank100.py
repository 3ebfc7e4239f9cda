"""probe: exact CP sweeps above 64 columns (the reference's default rank is s/2, test_ALS.cxx:119-125)
at s = 200: ms per sweep per schedule, the scans' share (HIP events on the engine's stream), parity of
the factors against the closed form. Usage: r06_rank100.py [R=100] [s=200] [sweeps=6]"""
import os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # (BLAS worker threads disturb the host side of the timed loop)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ppals  # noqa: E402
import rank_structured as RS  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s = int(sys.argv[2]) if len(sys.argv) > 2 else 200
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
lens = [s] * 4
ctx = ppals.Context(0)
A = ppals.init_factors(lens, R, 1000)
W = ppals.init_factors(lens, R, 2000)
G = ppals.init_factors(lens, R, 3000)
V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(A)
for sched in ("msdt", "dt"):
    cp = ppals.CP(ctx, V, R)
    cp.set_schedule(sched)
    cp.set_factors(W, G)
    cp.sweeps_dt(2)
    W_got = cp.get_factors()
    cp.sweeps_dt(2)
    ctx.sync()
    t0 = time.perf_counter()
    cp.sweeps_dt(K)
    ctx.sync()
    dt = (time.perf_counter() - t0) / K
    ctx.profile_enable(1)      # (the scans' own time in a second run: the event bracket perturbs the wall clock)
    ctx.profile_reset()
    cp.sweeps_dt(K)
    ctx.sync()
    n, ms, by = ctx.profile_read(0)
    ctx.profile_enable(0)
    flops = 2.0 * s ** 4 * R
    W_ref, _ = RS.als_cp_dt(A, W, G, 2)   # (after the timing: BLAS worker threads disturb the host's waits)
    err = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(W_got, W_ref))
    print(f"R={R} s={s} {sched}: {dt * 1e3:.3f} ms/sweep; scans {n / K:.2f}/sweep, {ms / max(n, 1):.3f} ms each "
          f"({by / max(ms, 1e-9) / 1e6:.0f} GB/s algorithmic, "
          f"{flops / (ms / max(n, 1) * 1e-3) / 1e12:.1f} TFLOP/s if full-tensor scans); "
          f"factor error after 2 sweeps {err:.2e}", flush=True)
    cp.close()
