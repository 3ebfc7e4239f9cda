"""probe: core rank above 64 (tests/test_gpu_tucker.py::test_core_rank_above_64...) with the step log on stderr"""
import os, sys
os.environ["PPALS_TUCKER_THIN"] = "0"
os.environ["PPALS_EIG_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ppals as pp
import numpy_ref as NR
import test_gpu_tucker as T
case = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lens, ranks, inner = [([300, 24, 20], [70, 20, 16], [110, 22, 18]), ([1344, 40, 36], [100, 12, 10], [150, 20, 18])][case]
V = T._slow_decay_tensor(lens, inner, [0.985, 0.8, 0.8], 21, 1e-4)
W0, c0 = NR.tucker_hosvd(V, ranks)
c2 = pp.Context(0)
t = pp.Tensor(c2, lens, 1).upload(V)
s = pp.Tucker(c2, t, ranks)
s.hosvd()
W_h, _ = s.get_factors()
print("hosvd", [T.relerr(T.proj(a), T.proj(b)) for a, b in zip(W_h, W0)], flush=True)
for n in range(1, 6):
    s.set_factors(W0); s.set_core(c0); s.sweeps_dt(n)
    W, core = s.get_factors()
    W_ref, core_ref = NR.tucker_hooi(V, W0, n)
    print("sweeps", n, [T.relerr(T.proj(a), T.proj(b)) for a, b in zip(W, W_ref)],
          [float(np.abs(a.T @ a - np.eye(a.shape[1])).max()) for a in W], flush=True)
