#!/usr/bin/env python3
"""Tucker (hosvd + alsTucker_DT sweeps) on the reference's real-data extents with its ranks
(test_ALS.cxx:366-379), synthetic values; run under `rocprofv3 --kernel-trace --stats` for the kernel
split.   usage: tools/runs/real_tucker_probe.py coil100|timelapse [sweeps=20]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402

SHAPES = {"coil100": ([3, 128, 128, 7200], [3, 10, 10, 70]), "timelapse": ([33, 1344, 1024, 9], [10, 100, 100, 5])}


def main():
    lens, ranks = SHAPES[sys.argv[1]]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ppals.preload_eigensolver()
    ctx = ppals.Context(0)
    import numpy as np
    Wt = ppals.init_factors(lens, 100, 1000)   # image-like: a rank-100 model with decaying weights
    decay = (0.93 ** np.arange(100)) ** (1.0 / len(lens))
    V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp([np.asfortranarray(w * decay[None, :]) for w in Wt])
    tk = ppals.Tucker(ctx, V, ranks)
    ctx.sync()
    t0 = time.perf_counter()
    tk.hosvd()
    ctx.sync()
    print(f"hosvd {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    tk.sweeps_dt(3)
    ctx.sync()
    t0 = time.perf_counter()
    tk.sweeps_dt(n)
    ctx.sync()
    print(f"{1e3 * (time.perf_counter() - t0) / n:.3f} ms per HOOI sweep ({n} sweeps)", flush=True)
    tk.close()
    V.close()
    ctx.close()


if __name__ == "__main__":
    main()
