#!/usr/bin/env python3
"""Does the rank-local scan of a P = 8 shard lose its last, partial round of workgroups? The scan's
achieved bandwidth for shards of a0 x 200 x 200 x 200 (a0 = 20 .. 32: 3125 .. 5000 tiles of 256 rows on
768 resident workgroups), R = 10, fp32, sharded code path on a one-rank communicator.
usage: tools/runs/shard_rows_probe.py [s=200] [R=10]"""
import os
import sys
import time

os.environ["PPALS_FORCE_COMM"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402


def main():
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ctx = ppals.Context(0)
    ctx.init_comm(0, 1, ppals.Context.unique_id())
    for a0 in (20, 21, 22, 23, 24, 25, 26, 27, 28, 30, 32):
        lens = [a0, s, s, s]
        V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, R, 1000))
        cp = ppals.CP(ctx, V, R)
        cp.set_factors(ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000))
        cp.sweeps_dt(4)
        ctx.sync()
        K = 24
        t0 = time.perf_counter()
        cp.sweeps_dt(K)
        ctx.sync()
        dt = (time.perf_counter() - t0) / K
        ctx.profile_reset()
        ctx.profile_enable(1)
        cp.sweeps_dt(K)
        ctx.sync()
        ctx.profile_enable(0)
        n, ms, by = ctx.profile_read(0)
        tiles = a0 * s * s / 256.0
        print(f"a0={a0}: {tiles:.0f} tiles = {tiles / 768:.2f} rounds of 768; scan {1e3 * ms / n:.1f} us at "
              f"{by / ms / 1e6:.0f} GB/s = {by / ms / 1e6 / 8000:.3f} of peak; sweep {1e3 * dt:.3f} ms "
              f"({1e6 * dt / a0:.2f} us per row of mode 0)", flush=True)
        cp.close()
        V.close()


if __name__ == "__main__":
    main()
