#!/usr/bin/env python3
"""What would a ONE-PASS build of the six PP pair operators cost (als_CP.cxx:678-694; SURVEY K8,
'init bytes = 1 s^4 sizeof if fused')? Every tensor element then feeds six rank-R accumulations
(T_ab, T_ac, T_ad, T_bc, T_bd, T_cd): 6 x 16 = 96 MFMA columns per element at R = 10 padded to 16.
This measures the friendliest possible form of that work — ONE row-contiguous scan of the resident
tensor against 96 packed columns (all six operators pretended to be suffix contractions; the real
thing needs strided gathers for four of them and cross-workgroup slab sums for five) — against the
scans the build actually runs. A lower bound of the direct route, measured with the product's own
kernel.   usage: tools/pp_direct_bound.py [s=200]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
import torch  # noqa: E402,F401
import ppals  # noqa: E402


def main():
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lens = [s] * 4
    ctx = ppals.Context(0)
    V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, 10, 1000))
    print(f"tensor {lens} fp32 = {4e-9 * s ** 4:.2f} GB; one first-level tree node (suffix scan over (c,d)) per R:")
    base = None
    for R in (10, 16, 32, 48, 64, 96):
        cp = ppals.CP(ctx, V, R)
        cp.set_schedule("dt")
        cp.set_factors(ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000))
        cp.tree_node("ab")
        ctx.sync()
        ctx.profile_reset()
        ctx.profile_enable(1)
        n = 5
        for _ in range(n):
            cp.tree_node("ab")
        ctx.sync()
        ctx.profile_enable(0)
        launches, ms, by = ctx.profile_read(0)
        per = ms / (2 * n)   # (the binding's tree_node() computes the node twice: size query + data)
        base = base or per
        print(f"  R = {R:3d} ({(R + 15) // 16} n-tiles): {per:.3f} ms per pass over the tensor "
              f"({launches / (2 * n):.0f} launch(es)), {per / base:.2f} x the R = 10 scan", flush=True)
        cp.close()
    print("the build of a PP phase today: 3 such R = 10 scans from cold factors, 2 inside a run "
          "(bench.py sub_records.cfg3_pp.pp_build_ms)")


if __name__ == "__main__":
    main()
