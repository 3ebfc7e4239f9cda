#!/usr/bin/env python3
"""Time the per-rank compute of a strong-scaled sweep on ONE GPU: a [s/P, s, s, s] tensor is what
each of P ranks holds of the s^4 problem (leading-mode block), so its sweep time is the no-comm
floor of the P-GPU run. usage: tools/shard_probe.py [s] [R]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402


def main():
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ctx = ppals.Context(0)
    for P in (1, 2, 4, 8):
        lens = [s // P, s, s, s]
        V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, R, 1000))
        cp = ppals.CP(ctx, V, R)
        cp.set_factors(ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000))
        cp.sweeps_dt(3)
        ctx.sync()
        t0 = time.perf_counter()
        K = 30
        cp.sweeps_dt(K)
        ctx.sync()
        dt = (time.perf_counter() - t0) / K
        ctx.profile_reset()
        ctx.profile_enable(1)  # what bench.py's timed region pays for its HIP-event brackets
        t0 = time.perf_counter()
        cp.sweeps_dt(K)
        ctx.sync()
        dte = (time.perf_counter() - t0) / K
        ctx.profile_enable(0)
        n, ms, _ = ctx.profile_read(0)
        print(f"P={P} shard {lens}: {dt * 1e3:.3f} ms/sweep -> no-comm ceiling {1 / dt:.0f} sweeps/s "
              f"(ideal {P}x of P=1); with scan events {dte * 1e3:.3f} ms/sweep, "
              f"scan avg {ms / max(n, 1):.4f} ms x {n / K:.2f}/sweep", flush=True)
        cp.close()
        V.close()


if __name__ == "__main__":
    main()
