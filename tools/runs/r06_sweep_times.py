"""probe: wall time of individual sweeps (sync after each) per schedule. Usage: r06_sweep_times.py [R] [s] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import ppals
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lens = [s] * 4
ctx = ppals.Context(0)
V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, R, 1000))
W, G = ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000)
for sched in ("dt", "msdt", "dt"):
    cp = ppals.CP(ctx, V, R)
    cp.set_schedule(sched)
    cp.set_factors(W, G)
    ts = []
    for _ in range(n):
        ctx.sync()
        t0 = time.perf_counter()
        cp.sweeps_dt(1)
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(sched, " ".join(f"{t:.2f}" for t in ts), flush=True)
    cp.close()
