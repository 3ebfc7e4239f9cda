#!/usr/bin/env python3
"""Does the online placement choice pay on a run of the reference's length? ONE process, the cfg2
tensor resident, sessions of 250 exact sweeps (script/script_synthetic.py:18) created alternately with
PPALS_PLACE_TUNE=1 and =0: wall time from ppals_cp_create to the last sweep done, every run printed,
medians at the end.   usage: tools/runs/place_ab.py [pairs=8] [sweeps=250]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    lens, R = [200] * 4, 10
    ctx = ppals.Context(0)
    V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, R, 1000))
    W0, G0 = ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000)
    t = {"1": [], "0": []}
    tail = {"1": [], "0": []}
    for p in range(pairs):
        for env in (("1", "0") if p % 2 == 0 else ("0", "1")):
            os.environ["PPALS_PLACE_TUNE"] = env
            ctx.sync()
            t0 = time.perf_counter()
            cp = ppals.CP(ctx, V, R)
            cp.set_factors(W0, G0)
            cp.sweeps_dt(sweeps - 50)
            ctx.sync()
            t1 = time.perf_counter()
            cp.sweeps_dt(50)          # the last 50 sweeps: the settled state
            ctx.sync()
            t2 = time.perf_counter()
            cp.close()
            t[env].append(t2 - t0)
            tail[env].append((t2 - t1) / 50)
            print(f"pair {p} PLACE_TUNE={env}: {t2 - t0:.4f} s to {sweeps} sweeps, last 50 at {1e3 * (t2 - t1) / 50:.4f} ms/sweep",
                  flush=True)
    for env, name in (("1", "online choice"), ("0", "off")):
        print(f"{name}: median {statistics.median(t[env]):.4f} s (min {min(t[env]):.4f}, max {max(t[env]):.4f}); "
              f"settled sweeps median {1e3 * statistics.median(tail[env]):.4f} ms")
    print(f"ratio of medians (on / off): {statistics.median(t['1']) / statistics.median(t['0']):.4f} total, "
          f"{statistics.median(tail['1']) / statistics.median(tail['0']):.4f} settled")


if __name__ == "__main__":
    main()
