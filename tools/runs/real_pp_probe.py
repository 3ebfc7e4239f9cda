#!/usr/bin/env python3
"""Where `-pp 1` spends its time on the coil-100 extents (3 x 128 x 128 x 7200, synthetic values):
bench-mode lines ([DTtime] / [PPfirst] / [PPsecond], als_CP.cxx:204-208,736-747), the driver's own
[dtime] of a 60-iteration run, number and duration of the operator builds. Run under
`rocprofv3 --kernel-trace --stats` for the kernel split.   usage: tools/runs/real_pp_probe.py [coil100|timelapse]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402

SHAPES = {"coil100": [3, 128, 128, 7200], "timelapse": [33, 1344, 1024, 9]}


def main():
    lens, R = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "coil100"], 10
    ctx = ppals.Context(0)
    import numpy as np
    Wt = ppals.init_factors(lens, 100, 1000)   # image-like: a rank-100 model with decaying weights
    decay = (0.93 ** np.arange(100)) ** (1.0 / len(lens))
    V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp([np.asfortranarray(w * decay[None, :]) for w in Wt])
    vn = V.norm()
    W0, G0 = ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000)
    cp = ppals.CP(ctx, V, R)
    with tempfile.TemporaryDirectory() as tmp:
        b = os.path.join(tmp, "b.csv")
        open(b, "w").write("[timetype],[dtime]\n")
        for _ in range(3):
            cp.set_factors(W0, G0)
            cp.run_dt(tol=1e-10 * vn, maxiter=1, resprint=1, bench=1, csv=b, csv_append=1)
        for _ in range(3):
            cp.set_factors(W0, G0)
            cp.run_pp(tol=1e-10 * vn, tol_init=0.05, maxiter=1, resprint=1, bench=1, csv=b, csv_append=1)
        print(open(b).read())
        c = os.path.join(tmp, "pp.csv")
        cp.set_factors(W0, G0)
        cp.pp_build_stats(+1)
        cp.run_pp(tol=1e-10 * vn, tol_init=0.05, maxiter=60, resprint=1, csv=c)
        n, s = cp.pp_build_stats(-1)
        print(f"builds {n}, {1e3 * s / max(n, 1):.3f} ms each")
        rows = [ln.split(",") for ln in open(c).read().splitlines()[1:] if ln.strip()]
        prev = None
        for r in rows:
            dt = float(r[6])
            print(f"iter {r[1]} pp_update {r[4]} dtime {dt:.5f}" + (f"  (+{1e3 * (dt - prev):.3f} ms)" if prev is not None else ""))
            prev = dt
    cp.close()


if __name__ == "__main__":
    main()
