#!/usr/bin/env python3
"""Write a synthetic stand-in for the reference's image datasets in ITS file format (raw
little-endian fp64, first index fastest — what script/imageloader.py / matloader.py dump and
V.read_dense_from_file reads, test_ALS.cxx:289-325): a low-multilinear-rank tensor plus noise.

    tools/make_o_file.py o1|o2|a,b,c,d out.bin [rank=12] [noise=0.05]

o1 = coil-100 extents 3x128x128x7200 (2.8 GB), o2 = time-lapse 33x1344x1024x9 (3.3 GB)."""
import sys

import numpy as np


def main():
    shape = {"o1": [3, 128, 128, 7200], "o2": [33, 1344, 1024, 9]}.get(
        sys.argv[1], None) or [int(x) for x in sys.argv[1].split(",")]
    out = sys.argv[2]
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    noise = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
    rng = np.random.default_rng(7)
    W = [np.abs(rng.standard_normal((s, min(rank, s)))) for s in shape]
    r = min(w.shape[1] for w in W)
    W = [w[:, :r] for w in W]
    lead = int(np.prod(shape[:-1]))
    # Khatri-Rao of all modes but the last, first index fastest
    K = W[0]
    for w in W[1:-1]:
        K = (w[:, None, :] * K[None, :, :]).reshape(-1, r)
    assert K.shape[0] == lead
    scale = np.linalg.norm(K) * np.linalg.norm(W[-1]) / np.sqrt(lead * shape[-1]) + 1e-300
    with open(out, "wb") as f:
        for l in range(shape[-1]):        # one slice of the last mode at a time
            sl = K @ W[-1][l]
            sl += noise * scale * rng.standard_normal(lead)
            sl.astype("<f8").tofile(f)
    print(f"wrote {out}: lens {shape} ({8e-9 * lead * shape[-1]:.2f} GB), rank {r} + {noise} noise")


if __name__ == "__main__":
    main()
