#!/usr/bin/env python3
"""The side lane of the multi-sweep step (the next leaf's first contraction of X beside the mode update in
flight): sweeps/s with it and without, ONE process, alternating fresh contexts (PPALS_LANES / PPALS_LANE_CUS are
read when a context is created), cfg2 at P = 1 and the P = 8 shard on the sharded code path.
(needs profiles/r05o_side_lane_experiment.patch applied: the side lane was measured and NOT kept —
profiles/r05o_side_lane_ab.txt.)   usage: tools/runs/lanes_ab.py [pairs=4]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402


def run(lens, R, env, sharded, sweeps):
    for k in ("PPALS_LANES", "PPALS_LANE_CUS", "PPALS_FORCE_COMM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if sharded:
        os.environ["PPALS_FORCE_COMM"] = "1"
    ctx = ppals.Context(0)
    if sharded:
        ctx.init_comm(0, 1, ppals.Context.unique_id())
    V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, R, 1000))
    cp = ppals.CP(ctx, V, R)
    cp.set_factors(ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000))
    cp.sweeps_dt(96 if not sharded else 12)     # (P = 1: the online placement choice settles first)
    ctx.sync()
    t0 = time.perf_counter()
    cp.sweeps_dt(sweeps)
    ctx.sync()
    dt = (time.perf_counter() - t0) / sweeps
    g = cp.gradnorm()
    cp.close()
    V.close()
    ctx.close()
    return dt, g


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    variants = [("off", {"PPALS_LANES": "0"}), ("on", {"PPALS_LANES": "1"}),
                ("on, 16 CUs masked", {"PPALS_LANES": "1", "PPALS_LANE_CUS": "16"})]
    for name, lens, R, sharded, sweeps in (("cfg2 P=1", [200] * 4, 10, False, 60),
                                           ("cfg2 P=8 shard", [25, 200, 200, 200], 10, True, 48)):
        res = {v[0]: [] for v in variants}
        gn = {}
        for p in range(pairs):
            order = variants if p % 2 == 0 else variants[::-1]
            for vname, env in order:
                dt, g = run(lens, R, env, sharded, sweeps)
                res[vname].append(dt)
                gn[vname] = g
                print(f"{name} pair {p} lanes {vname}: {1e3 * dt:.4f} ms per sweep (gradnorm {g:.9e})", flush=True)
        for vname in res:
            print(f"{name}: lanes {vname}: median {1e3 * statistics.median(res[vname]):.4f} ms "
                  f"(min {1e3 * min(res[vname]):.4f}, max {1e3 * max(res[vname]):.4f})")
        print(f"{name}: same iterates: gradnorm {gn}")


if __name__ == "__main__":
    main()
