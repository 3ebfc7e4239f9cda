#!/usr/bin/env python3
"""Time the streaming residual (K10, `[diffV]`) of the cfg2 problem: tools/k10_probe.py [s=200] [R=10] [reps=7]

One process per setting (the launch geometry is read from the environment once): prints the median
milliseconds of `ppals_cp_residual` between stream synchronisations and the fraction of the 8 TB/s peak
one read of the tensor comes to.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pairwise-perturbation_amd"))
import ppals  # noqa: E402


def main():
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    rng = np.random.default_rng(0)
    ctx = ppals.Context(0)
    lens = [s] * 4
    Wtrue = [rng.random((n, R)) for n in lens]
    V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(Wtrue)
    cp = ppals.CP(ctx, V, R)
    cp.set_factors([rng.random((n, R)) for n in lens], [np.zeros((n, R)) for n in lens])
    cp.residual()
    ts = []
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        cp.residual()
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    gb = 4.0 * s ** 4 / 1e9
    print(f"s={s} R={R} "
          f"residual {med * 1e3:.3f} ms (min {ts[0] * 1e3:.3f}) = {gb / med / 1e3:.2f} TB/s = {gb / med / 8e3:.3f} of peak")
    ctx.close()


if __name__ == "__main__":
    main()
