#!/usr/bin/env python3
"""Per-rank compute of a strong-scaled sweep, measured on ONE GPU: a [s/P, s, s, s] tensor is what
each of P ranks holds of the s^4 problem (leading-mode block). The session runs the SHARDED code
path (PPALS_FORCE_COMM=1: shard-aware multi-sweep schedule, packed partials, collectives issued on
a one-rank RCCL communicator), so its sweep time is the floor of the P-GPU run before any real
exchange.   usage: tools/shard_probe.py [s=200] [R=10] [P,P,...=1,2,4,8] [dt|msdt]"""
import os
import sys
import time

os.environ["PPALS_FORCE_COMM"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402


def main():
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    Ps = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
    sched = sys.argv[4] if len(sys.argv) > 4 else None
    ctx = ppals.Context(0)
    ctx.init_comm(0, 1, ppals.Context.unique_id())
    base = None
    for P in Ps:
        lens = [s // P, s, s, s]
        V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, R, 1000))
        cp = ppals.CP(ctx, V, R)
        if sched:
            cp.set_schedule(sched)
        cp.set_factors(ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000))
        K = 24 if s <= 200 else 8      # a multiple of 2 sweeps: the sharded cycle is 3 scans / 2 sweeps
        cp.sweeps_dt(4)
        ctx.sync()
        t0 = time.perf_counter()
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
        n, ms, by = ctx.profile_read(0)
        if base is None:
            base = dt * P
        print(f"P={P} shard {lens} ({cp.schedule}): {dt * 1e3:.3f} ms/sweep = {dt * P / base:.2f} x "
              f"(P={Ps[0]} time)/{P // Ps[0] if P >= Ps[0] else 1}; no-exchange ceiling {1 / dt:.0f} "
              f"sweeps/s; with scan events {dte * 1e3:.3f} ms/sweep: scan avg {ms / max(n, 1):.4f} ms "
              f"x {n / K:.2f}/sweep at {by / max(ms, 1e-9) / 1e6:.0f} GB/s", flush=True)
        cp.close()
        V.close()


if __name__ == "__main__":
    main()
