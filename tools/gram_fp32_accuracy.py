#!/usr/bin/env python3
"""What the HOSVD Gram would lose on the fp32 matrix cores (CPU simulation, numpy).

k_unfold_syrk_f32 widens the fp32 tensor values to fp64 and multiplies on v_mfma_f64_16x16x4_f64: the
products of fp32 values are exact in fp64, so the Gram carries fp64 rounding only. The alternative —
v_mfma_f32_16x16x4_f32 chains of 64 reduction indices flushed to fp64, as the tensor scan does — rounds
every partial sum to 24 bits. This script forms the unfolding Grams both ways (the fp32 chain simulated
term by term: 4 reduction indices per MFMA step, fp32 accumulator, fp64 across chains) for the inputs of
tests/test_gpu_tucker.py::test_hosvd_gram_syrk_f32 and for a `-tensor r2` cube, and prints the relative
error of the Gram and of the projector onto its leading eigenvectors (what the tests bound by 1e-8).
usage: tools/gram_fp32_accuracy.py > profiles/r05_gram_fp32_accuracy.txt"""
import numpy as np


def decaying(lens, inner, seed, noise):
    rng = np.random.default_rng(seed)
    U = [np.linalg.qr(rng.standard_normal((s, r)))[0] for s, r in zip(lens, inner)]
    core = rng.standard_normal(inner)
    for m, r in enumerate(inner):
        shape = [1] * len(inner)
        shape[m] = r
        core = core * (0.7 ** np.arange(r)).reshape(shape)
    V = core
    for m, u in enumerate(U):
        V = np.moveaxis(np.tensordot(u, V, axes=(1, m)), 0, m)
    E = rng.standard_normal(lens)
    return V + noise * np.linalg.norm(V) / np.linalg.norm(E) * E


def proj(a):
    return a @ a.T


def top(G, r):
    return np.linalg.eigh(G)[1][:, ::-1][:, :r]


def gram_f32_chains(A, chain=64):
    A32 = A.astype(np.float32)
    J, C = A.shape
    G = np.zeros((J, J))
    for c0 in range(0, C, chain):
        acc = np.zeros((J, J), np.float32)
        for c in range(c0, min(C, c0 + chain), 4):
            blk = A32[:, c:c + 4]
            acc = (acc + (blk @ blk.T).astype(np.float32)).astype(np.float32)
        G += acc
    return G


def main():
    print("unfolding Gram: exact fp32 products summed in fp64 (k_unfold_syrk_f32) vs fp32 MFMA chains of 64 "
          "terms flushed to fp64 (simulated)")
    for lens, ranks in (([100, 68, 76], [6, 5, 4]), ([128, 64, 72], [8, 4, 6])):
        V = decaying(lens, [min(s, r + 4) for s, r in zip(lens, ranks)], 7, 0.05)
        V = V.astype(np.float32).astype(np.float64)
        for m in range(3):
            A = np.moveaxis(V, m, 0).reshape(lens[m], -1)
            G, G32 = A @ A.T, gram_f32_chains(A)
            p, p32 = proj(top(G, ranks[m])), proj(top(G32, ranks[m]))
            print(f"decaying {lens} mode {m} rank {ranks[m]}: Gram {np.linalg.norm(G - G32) / np.linalg.norm(G):.2e}, "
                  f"projector {np.linalg.norm(p - p32) / np.linalg.norm(p):.2e}   (test bar 1e-8)")
    rng = np.random.default_rng(0)
    s, r = 96, 20
    V = rng.uniform(0.5, 1, (s, s, s)).astype(np.float32).astype(np.float64)
    A = V.reshape(s, -1)
    G, G32 = A @ A.T, gram_f32_chains(A)
    w = np.linalg.eigvalsh(G)[::-1]
    p, p32 = proj(top(G, r)), proj(top(G32, r))
    print(f"-tensor r2 cube s={s} rank {r}: lambda_1/lambda_2 {w[0] / w[1]:.1e}, gap below rank / lambda_1 "
          f"{(w[r - 1] - w[r]) / w[0]:.1e}; Gram {np.linalg.norm(G - G32) / np.linalg.norm(G):.2e}, projector "
          f"{np.linalg.norm(p - p32) / np.linalg.norm(p):.2e}")


if __name__ == "__main__":
    main()
