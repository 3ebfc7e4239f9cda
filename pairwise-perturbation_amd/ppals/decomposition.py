"""Host-side mirror of the reference's class API (src/decomposition.h, src/CP.h, src/optimizer/):
same class names, constructor arguments, Init / als signatures and return values, over the C ABI.

    decom = CPD(6, 13, 5, ctx, optimizer=CPDTOptimizer)        # CPD<double, CPDTOptimizer<double>>
    decom.Init(V, W)                                           # tests/test_decomposition.cxx:55
    decom.als(1e-5, 1000, 30, 100, "results/test.csv")         # tests/test_decomposition.cxx:64

The optimizer classes only name the step cadence the reference class has (one full sweep, two
half-sweep subtrees, N-1 modes per tensor contraction); every one of them runs on the engine's
multi-sweep contraction schedule, and the ALS iterates are identical for all three."""
import sys

import numpy as np


def _binding(ctx):
    """the ctypes binding module the context belongs to (the product's, or a test stand-in build
    of the same file loaded under another module name)"""
    mod = sys.modules.get(type(ctx).__module__)
    if mod is None or not hasattr(mod, "CP"):
        import ppals as mod
    return mod


class CPSimpleOptimizer:  # src/optimizer/cp_simple_optimizer.h
    kind = 0
    sweeps_per_step = staticmethod(lambda order: 1.0)


class CPDTOptimizer:  # src/optimizer/cp_dt_optimizer.h
    kind = 1
    sweeps_per_step = staticmethod(lambda order: 0.5)


class CPMSDTOptimizer:  # src/optimizer/cp_msdt_optimizer.h
    kind = 2
    sweeps_per_step = staticmethod(lambda order: (order - 1) / order)


class Decomposition:
    """src/decomposition.h:8-38. `dw` (the CTF World) becomes the ppals Context."""

    def __init__(self, order, size, r, ctx):
        self.world = ctx
        self.order = int(order)
        self.size = [int(size)] * order if np.isscalar(size) else [int(x) for x in size]
        self.rank = [int(r)] * order if np.isscalar(r) else [int(x) for x in r]
        if len(self.size) != order or len(self.rank) != order:
            raise ValueError("size / rank must have `order` entries")
        self.V = None
        self.W = None

    def Init(self, input, mat):  # src/decomposition.cxx:55-70
        if list(input.lens) != self.size:
            raise ValueError(f"tensor extents {input.lens} != {self.size}")
        if len(mat) != self.order or any(np.asarray(m).shape != (s, r)
                                         for m, s, r in zip(mat, self.size, self.rank)):
            raise ValueError("factor matrices must be size[i] x rank[i]")
        self.V = input
        self.W = [np.asfortranarray(m, dtype=np.float64) for m in mat]

    def print_V(self):
        raise NotImplementedError("the tensor lives in HBM; download it explicitly if needed")

    def print_W(self, i):
        print(self.W[i])


class CPD(Decomposition):
    """CPD<dtype, Optimizer> (src/CP.h:10-48)."""

    def __init__(self, order, size, r, ctx, optimizer=CPDTOptimizer):
        super().__init__(order, size, r, ctx)
        if any(x != self.rank[0] for x in self.rank):
            raise ValueError("CP needs one rank for all modes (src/CP.cxx:58-61)")
        self.optimizer = optimizer
        self.grad_W = None
        self.gradnorm = 0.0
        self._cp = None
        self._lambda = 0.0

    def Init(self, input, mat, lambda_=0.0, grad_seed=3000):
        """src/CP.cxx:72-87: grad_W[i] is filled with U(0,1) values (the reference draws them from
        CTF's stream; here from the engine's counter-based generator, seed `grad_seed`)."""
        b = _binding(self.world)
        super().Init(input, mat)
        self.grad_W = b.init_factors(self.size, self.rank[0], grad_seed)
        self._lambda = float(lambda_)
        if self._cp is not None:
            self._cp.close()
        self._cp = b.CP(self.world, input, self.rank[0])
        self._cp.set_factors(self.W, self.grad_W)

    def print_grad(self, i):
        print(self.grad_W[i])

    def update_gradnorm(self):  # src/CP.cxx:89-96
        self.gradnorm = float(np.sqrt(sum(np.linalg.norm(g) ** 2 for g in self.grad_W)))
        return self.gradnorm

    def als(self, tol, timelimit, maxsweep, resprint, Plot_File=None, bench=False, verbose=False):
        """src/CP.cxx:100-186. Plot_File is a path (the callee opens, writes and closes it, as the
        reference closes the stream it was handed). Returns the reference's bool."""
        if self._cp is None:
            raise RuntimeError("call Init first")
        rc, sweeps, iters = self._cp.cpd_als(self.optimizer.kind, tol=tol, timelimit=timelimit,
                                             maxiter=maxsweep, lam=self._lambda,
                                             resprint=resprint, csv=Plot_File, bench=int(bench),
                                             csv_append=int(bench), verbose=int(verbose))
        self.sweeps, self.iters = sweeps, iters
        self.W, self.grad_W = self._cp.get_factors(with_grad=True)
        self.update_gradnorm()
        return bool(rc)
