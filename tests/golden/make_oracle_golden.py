"""Generate tests/golden/oracle_golden.npz: seeded inputs -> outputs of the fp64 CPU oracle
(oracle/ppals_oracle.cpp), so that (a) drift of the oracle itself is caught on CPU and (b) the GPU
engine is compared against committed numbers, not only against an oracle built on the spot.
Inputs are regenerated from seeds (counter-based generator), only outputs are stored.

Run in the authoring container:  python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

CASES = {
    # name: (lens, R, seed)
    "cp4": ([12, 10, 9, 11], 4, 3),
    "cp3": ([14, 9, 11], 3, 5),
    "cp4_r10": ([16, 16, 16, 16], 10, 7),
}
TUCKER = {"tk3": ([12, 10, 9], [3, 4, 2], 4)}


def cp_problem(lens, R, seed):
    V = O.build_V(O.init_factors(lens, R, 1000 + seed))
    W = O.init_factors(lens, R, 2000 + seed)
    G = O.init_factors(lens, R, 3000 + seed)
    return V, W, G


def main():
    out = {}
    for name, (lens, R, seed) in CASES.items():
        V, W, G = cp_problem(lens, R, seed)
        N = len(lens)
        for mode in range(N):
            out[f"{name}/mttkrp{mode}"] = O.mttkrp(V, W, mode, 0)
        _, _, W5, G5 = O.als_cp_dt(V, W, G, tol=0.0, maxiter=4, resprint=10 ** 6)
        for i in range(N):
            out[f"{name}/dt5_W{i}"] = W5[i]
        out[f"{name}/dt5_gradnorm"] = np.array([np.sqrt(sum(np.linalg.norm(g) ** 2 for g in G5))])
        out[f"{name}/dt5_residual"] = np.array([O.residual(V, W5)])
        csv = "/tmp/_golden_pp.csv"
        Vn = np.linalg.norm(V)
        _, it, Wpp, _ = O.als_cp_pp(V, W, G, tol=1e-7 * Vn, tol_init=0.1, maxiter=40, csv=csv,
                                    resprint=1)
        _, rows = O.read_csv(csv)
        out[f"{name}/pp_rows"] = np.array([[r[1], r[4], r[5]] for r in rows])
        out[f"{name}/pp_iters"] = np.array([it])
    for name, (lens, ranks, seed) in TUCKER.items():
        V = O.fill_uniform(int(np.prod(lens)), seed, lo=0.5, hi=1.0).reshape(lens, order="F")
        W0, c0 = O.hosvd(V, ranks)
        for i, w in enumerate(W0):
            out[f"{name}/hosvd_P{i}"] = w @ w.T
        out[f"{name}/hosvd_corenorm"] = np.array([np.linalg.norm(c0)])
        _, _, W3, c3 = O.als_tucker_dt(V, W0, c0, tol=0.0, maxiter=3, resprint=10 ** 6)
        for i, w in enumerate(W3):
            out[f"{name}/dt_P{i}"] = w @ w.T
        out[f"{name}/dt_corenorm"] = np.array([np.linalg.norm(c3)])
    path = os.path.join(HERE, "oracle_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
