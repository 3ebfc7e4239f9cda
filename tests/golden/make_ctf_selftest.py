#!/usr/bin/env python3
"""Writes tests/golden/ctf_selftest/cp4_small: a fixture in the layout tests/golden/ctf/README.md
describes, made by THIS repository's fp64 oracle (not by the reference — it only exercises the
reader and the comparison of tests/test_ctf_fixtures.py).   usage: python tests/golden/make_ctf_selftest.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def flat(mats):
    return np.concatenate([np.asfortranarray(m).ravel(order="F") for m in mats])


def main():
    out = os.path.join(HERE, "ctf_selftest", "cp4_small")
    os.makedirs(out, exist_ok=True)
    lens, R = [7, 6, 5, 8], 3
    V = O.build_V(O.init_factors(lens, R, 1000))
    W0, G0 = O.init_factors(lens, R, 2000), O.init_factors(lens, R, 3000)
    Vn = np.linalg.norm(V)
    meta = {"made_by": "oracle/ppals_oracle.cpp of this repository (NOT the reference)", "model": "CP",
            "lens": lens, "rank": R, "pp": 0, "maxiter": 10, "resprint": 2, "tol": 1e-12, "lambda": 0.0,
            "factor_tol": 1e-8, "csv": "out.csv"}
    csv = os.path.join(out, "out.csv")
    _, _, W, G = O.als_cp_dt(V, W0, G0, tol=meta["tol"] * Vn, maxiter=meta["maxiter"], resprint=meta["resprint"],
                             csv=csv)
    np.asfortranarray(V).ravel(order="F").astype("<f8").tofile(os.path.join(out, "V.bin"))
    np.concatenate([flat(W0), flat(G0)]).astype("<f8").tofile(os.path.join(out, "W0.bin"))
    np.concatenate([flat(W), flat(G)]).astype("<f8").tofile(os.path.join(out, "W.bin"))
    json.dump(meta, open(os.path.join(out, "meta.json"), "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
