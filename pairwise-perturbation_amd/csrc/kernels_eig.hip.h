// kernels_eig.hip.h — the Tucker eigen-step (K12) on the matrix cores: leading eigenvectors of the
// s x s Gram of an unfolding (als_Tucker.cxx:399-406, `MTM.svd(U,S,VT,rank)`) WITHOUT a full
// eigen-decomposition.
//
// A full symmetric eigensolver is a chain of thousands of dependent tiny launches on a GPU
// (rocSOLVER dsyevd at 400 x 400: ~13 000 launches, 9.3 ms — the whole cost of a HOOI sweep of
// BASELINE config 5). HOOI only needs the invariant subspace of the `rank` largest eigenvalues, and
// from the second sweep on it knows where the gap below them is. So:
//   P = (I + sign(G - sigma I)) / 2,  sigma inside the gap below the rank-th eigenvalue,
// is the orthogonal projector onto that subspace, and sign() is computed with the scaled
// Newton-Schulz iteration X <- 1/2 mu X (3 I - mu^2 X^2), which is nothing but dense fp64 GEMMs
// (v_mfma_f64_16x16x4_f64) — ~30 launches of a few microseconds each, enqueued without a read-back. trace(P) must come out as
// `rank`: that check makes the result exact (to the conditioning eps*||G||/gap every solver has) or
// the call falls back to the full solver. The eigenvectors inside the subspace (the reference
// returns them one by one, sorted) come from a Rayleigh-Ritz step on a rank x rank matrix.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_small.hip.h"

namespace ppals {

// C[M x N] = alpha * A[M x K] * B[K x N] + beta * D[M x N]   (fp64, column-major, ld = rows)
// B is handed over as Bt = B^T (N x K, column-major, ldb): element B[k][j] = Bt[j + ldb*k], so the
// 16 lanes of an MFMA column group read 128 contiguous bytes for both operands. For the symmetric
// iterates of the sign iteration Bt IS B. The matrices here are a few hundred rows: everything is
// L2-resident and a launch is pure latency, so ONE 16 x 16 output tile is spread over a whole
// workgroup — its 4 waves split K, each keeps UN steps (2*UN loads per lane) in flight, and the
// four partial tiles meet in LDS in a fixed order.
// BN = true: the second operand is handed over as B itself (K x N, column-major, ldb) — for the
// thin operands of the eigen-step's tail (a few dozen columns, L2-resident) this saves the
// transposition launch; the lanes of a column group then read 16 different lines instead of one.
template <bool BN>
__global__ __launch_bounds__(256) void k_dgemm_nx(const double *__restrict__ A, int64_t lda,
                                                  const double *__restrict__ Bt, int64_t ldb,
                                                  const double *__restrict__ D, int64_t ldd,
                                                  double *__restrict__ C, int64_t ldc, int M, int N,
                                                  int K, double alpha, double beta) {
  constexpr int UN = 13;
  __shared__ double part[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  // clamped operand rows: lanes past the edge re-read the last row, their results are not stored
  const int ia = min(i0 + l16, M - 1), jb = min(j0 + l16, N - 1);
  const double *__restrict__ ap = A + ia;
  const double *__restrict__ bp = BN ? Bt + (int64_t)ldb * jb : Bt + jb;
  const int64_t bstep = BN ? 1 : ldb;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int ksteps = (K + 3) / 4;
  const int spw = (ksteps + 3) / 4;
  const int s_begin = wave * spw, s_end = min(ksteps, s_begin + spw);
  for (int s0 = s_begin; s0 < s_end; s0 += UN) {
    double av[UN], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int k = (s0 + u) * 4 + g;
      const bool ok = (s0 + u) < s_end && k < K;  // beyond the range: multiply by zero
      const int kc = ok ? k : 0;
      const double a = ap[(int64_t)lda * kc], b = bp[bstep * kc];
      av[u] = ok ? a : 0.0;
      bv[u] = ok ? b : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UN; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) part[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave > 0) return;
  // D-matrix map of the f64 16x16x4 shape: lane holds column j = lane&15, rows (lane>>4) + 4*reg
  const int j = j0 + l16;
  if (j < N) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = i0 + g + 4 * r;
      if (i < M) {
        double v = alpha * (((acc[r] + part[0][r][lane]) + part[1][r][lane]) + part[2][r][lane]);
        if (D) v += beta * D[i + ldd * (int64_t)j];
        C[i + ldc * (int64_t)j] = v;
      }
    }
  }
}

// The same product for a SYMMETRIC result (the iterates of the sign iteration: X^2 and
// X (a I + b X^2) with X symmetric): only the tiles on and above the diagonal are computed
// (blockIdx.x enumerates them), every value is stored at (i, j) and at (j, i). The result is
// symmetric bit for bit — k_dgemm_nt reads its second operand transposed, where an antisymmetric
// rounding residue would double per iteration — without a separate symmetrisation pass. NW waves
// split K (8 at J = 400: one round of 13 steps per wave instead of two).
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_dgemm_nt_sym(const double *__restrict__ A, int64_t lda,
                                                      const double *__restrict__ Bt, int64_t ldb,
                                                      const double *__restrict__ D, int64_t ldd,
                                                      double *__restrict__ C, int64_t ldc, int M,
                                                      int K, double alpha, double beta,
                                                      int chk_mode = 0,
                                                      double *__restrict__ chk_part = nullptr) {
  // chk_mode 1: chk_part[blockIdx.x] = this tile's share of ||C - I||_F^2 (off-diagonal tiles count
  // twice: both triangles); 2: its share of trace(C). The convergence check of the sign iteration
  // rides on the products it has to do anyway — no extra product, no reduction launch.
  constexpr int UN = 13;
  __shared__ double part[NW - 1][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  // blockIdx.x -> (ti <= tj): row ti of the upper triangle holds nt - ti tiles
  const int nt = (M + 15) / 16;
  int ti = 0, rem = blockIdx.x;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ti++;
  }
  const int tj = ti + rem;
  const int i0 = ti * 16, j0 = tj * 16;
  const int ia = min(i0 + l16, M - 1), jb = min(j0 + l16, M - 1);
  const double *__restrict__ ap = A + ia;
  const double *__restrict__ bp = Bt + jb;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int ksteps = (K + 3) / 4;
  const int spw = (ksteps + NW - 1) / NW;
  const int s_begin = wave * spw, s_end = min(ksteps, s_begin + spw);
  for (int s0 = s_begin; s0 < s_end; s0 += UN) {
    double av[UN], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int k = (s0 + u) * 4 + g;
      const bool ok = (s0 + u) < s_end && k < K;
      const int kc = ok ? k : 0;
      const double a = ap[(int64_t)lda * kc], b = bp[(int64_t)ldb * kc];
      av[u] = ok ? a : 0.0;
      bv[u] = ok ? b : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UN; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) part[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave > 0) return;
  const int j = j0 + l16;
  double chk = 0.0;
  if (j < M) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = i0 + g + 4 * r;
      if (i < M && i <= j) {  // (a diagonal tile stores its upper half only, mirrored)
        double v = acc[r];
#pragma unroll
        for (int w = 0; w < NW - 1; w++) v += part[w][r][lane];  // fixed order
        v *= alpha;
        if (D) v += beta * D[i + ldd * (int64_t)j];
        C[i + ldc * (int64_t)j] = v;
        if (i != j) C[j + ldc * (int64_t)i] = v;
        if (chk_mode == 1) {
          const double d = v - (i == j ? 1.0 : 0.0);
          chk += (i == j ? 1.0 : 2.0) * d * d;
        } else if (chk_mode == 2 && i == j) {
          chk += v;
        }
      }
    }
  }
  if (chk_mode) {  // (wave 0 only is left: a wave-level sum in a fixed order)
    chk = wave_sum(chk);
    if (lane == 0) chk_part[blockIdx.x] = chk;
  }
}

// K13 on the matrix cores: Gram of the mode-`pos` unfolding (unroll_tensor_contraction,
// common.cxx:205-223; the HOSVD initialisation runs it on the full tensor: N * 2 s^(N+1) flops, the
// largest single cost of a Tucker run and MFMA-bound).
//   G[p + J*q] (slab blockIdx.z) = sum_{c in chunk} A[p][c] * A[q][c],  A[p][c] = X[l + L*(p + J*t)]
// 32 x 32 output tile per workgroup, the reduction index staged through LDS 32 at a time AS fp64
// (an fp32 tensor is widened there: its products are exact in fp64, so the Gram carries fp64
// rounding only — eigenvectors of a matrix whose lambda_1 is 1e6 x the rest need that), four waves
// = four 16 x 16 v_mfma_f64_16x16x4_f64 tiles. Loads are arranged so that consecutive threads touch
// consecutive addresses (p fastest when L == 1, c fastest otherwise), as in k_unfold_gram.
template <typename TV>
__global__ __launch_bounds__(256) void k_unfold_gram_mfma(const TV *__restrict__ X, int64_t L,
                                                          int64_t J, int64_t T, int64_t c_per_split,
                                                          double *__restrict__ slab) {
  __shared__ double As[32][33];
  __shared__ double Bs[32][33];
  const int64_t C = L * T;
  const int64_t p0 = (int64_t)blockIdx.x * 32, q0 = (int64_t)blockIdx.y * 32;
  const int64_t c_begin = (int64_t)blockIdx.z * c_per_split;
  const int64_t c_end = min(C, c_begin + c_per_split);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty in [0,8)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int wi = (wave & 1) * 16, wj = (wave >> 1) * 16;
  const bool p_fast = (L == 1);
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int64_t c0 = c_begin; c0 < c_end; c0 += 32) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int a = tx, b = ty + 8 * i;
      const int pi = p_fast ? a : b, ci = p_fast ? b : a;
      const int64_t c = c0 + ci;
      double va = 0, vb = 0;
      if (c < c_end) {
        const int64_t l = c % L, t = c / L;
        if (p0 + pi < J) va = (double)X[l + L * ((p0 + pi) + J * t)];
        if (q0 + pi < J) vb = (double)X[l + L * ((q0 + pi) + J * t)];
      }
      As[pi][ci] = va;
      Bs[pi][ci] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int cq = 0; cq < 8; cq++)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[wi + l16][4 * cq + g], Bs[wj + l16][4 * cq + g],
                                                 acc, 0, 0, 0);
    __syncthreads();
  }
  double *gs = slab + (int64_t)blockIdx.z * J * J;
  const int64_t q = q0 + wj + l16;  // D: lane holds column j = lane&15, rows (lane>>4) + 4*reg
  if (q < J) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int64_t pp_ = p0 + wi + g + 4 * r;
      if (pp_ < J) gs[pp_ + J * q] = acc[r];
    }
  }
}

// X = (G - sigma I) / rho   (J x J)
__global__ void k_shift_scale(const double *__restrict__ G, int64_t J, double sigma, double inv_rho,
                              double *__restrict__ X) {
  const int64_t total = J * J;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % J, j = e / J;
    X[e] = (G[e] - (i == j ? sigma : 0.0)) * inv_rho;
  }
}

// X = G - sum_{d < m} (lam[d] - tau) q_d q_d^T - sigma I   (deflation of the dominant eigenpairs:
// their eigenvalues are moved down to tau, everything else is untouched), partial[blk] = the
// block's share of ||X||_F^2. Q: J x m column-major.
// The same with the scale of the sign iteration applied at once, X = (G - deflation - sigma I) *
// inv_rho, and the dominant eigenpair taken straight from the power iteration's last product:
// q = y / ||y||, lambda = the Rayleigh quotient left in the partial sums `pp` (np pairs: ||y||^2
// share, q_in^T y share). Workgroup 0 also writes q (J) and lambda for the tail. m is 0 or 1.
__global__ __launch_bounds__(256) void k_ns_prepare(const double *__restrict__ G, int64_t J,
                                                    const double *__restrict__ y,
                                                    const double *__restrict__ pp, int np, int m,
                                                    double tau, double sigma, double inv_rho,
                                                    double *__restrict__ X,
                                                    double *__restrict__ q_out,
                                                    double *__restrict__ lam_out) {
  __shared__ double lds[17];
  double inv = 0, lam = 0;
  if (m > 0) {
    double n2 = 0, rq = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
      n2 += pp[2 * i];
      rq += pp[2 * i + 1];
    }
    n2 = block_sum(n2, lds);
    rq = block_sum(rq, lds);
    inv = 1.0 / sqrt(n2);
    lam = rq;
    if (blockIdx.x == 0) {
      for (int64_t i = threadIdx.x; i < J; i += blockDim.x) q_out[i] = y[i] * inv;
      if (threadIdx.x == 0) *lam_out = lam;
    }
  }
  const double w = (lam - tau) * inv * inv;
  const int64_t total = J * J;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % J, j = e / J;
    double v = G[e] - (i == j ? sigma : 0.0);
    if (m > 0) v -= w * y[i] * y[j];
    X[e] = v * inv_rho;
  }
}

__global__ __launch_bounds__(256) void k_deflate_shift(const double *__restrict__ G, int64_t J,
                                                       const double *__restrict__ Q, int m,
                                                       const double *__restrict__ lam, double tau,
                                                       double sigma, double *__restrict__ X,
                                                       double *__restrict__ partial) {
  __shared__ double lds[17];
  double s = 0;
  const int64_t total = J * J;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % J, j = e / J;
    double v = G[e] - (i == j ? sigma : 0.0);
    for (int d = 0; d < m; d++) v -= (lam[d] - tau) * Q[i + J * d] * Q[j + J * d];
    X[e] = v;
    s += v * v;
  }
  s = block_sum(s, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// *out = || GU - U diag(ev) ||_F^2  (residual of the returned eigenpairs; one block)
__global__ __launch_bounds__(1024) void k_eig_residual(const double *__restrict__ GU,
                                                       const double *__restrict__ U,
                                                       const double *__restrict__ ev, int64_t J,
                                                       int r, double *__restrict__ out) {
  __shared__ double lds[17];
  double s = 0;
  for (int64_t e = threadIdx.x; e < J * r; e += blockDim.x) {
    const double d = GU[e] - U[e] * ev[e / J];
    s += d * d;
  }
  s = block_sum(s, lds);
  if (threadIdx.x == 0) *out = s;
}

// partial[blk] = sum over the block's elements of (G - sigma I)^2
__global__ __launch_bounds__(256) void k_frob_shifted(const double *__restrict__ G, int64_t J,
                                                      double sigma, double *__restrict__ partial) {
  __shared__ double lds[17];
  double s = 0;
  const int64_t total = J * J;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % J, j = e / J;
    const double d = G[e] - (i == j ? sigma : 0.0);
    s += d * d;
  }
  s = block_sum(s, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// partial[2*blk] += ||Y - I||_F^2 share, partial[2*blk+1] = trace(X) share (Y = X^2 of the last
// iterate); summed by k_sum_pairs
__global__ __launch_bounds__(256) void k_sign_check(const double *__restrict__ Y,
                                                    const double *__restrict__ X, int64_t J,
                                                    double *__restrict__ partial) {
  __shared__ double lds[17];
  double e2 = 0, tr = 0;
  const int64_t total = J * J;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % J, j = e / J;
    const double d = Y[e] - (i == j ? 1.0 : 0.0);
    e2 += d * d;
    if (i == j) tr += X[e];
  }
  e2 = block_sum(e2, lds);
  tr = block_sum(tr, lds);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = e2;
    partial[2 * blockIdx.x + 1] = tr;
  }
}
__global__ __launch_bounds__(256) void k_sum_pairs(const double *__restrict__ partial, int n,
                                                   double *__restrict__ out) {
  __shared__ double lds[17];
  double a = 0, b = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    a += partial[2 * i];
    b += partial[2 * i + 1];
  }
  a = block_sum(a, lds);
  b = block_sum(b, lds);
  if (threadIdx.x == 0) {
    out[0] = a;
    out[1] = b;
  }
}

// One power step on a single vector (the deflation of a dominant eigenpair), spread over the
// device (one workgroup reads the 1.3 MB of a 400 x 400 Gram at the bandwidth of ONE CU: 25 us; 50
// workgroups: ~5 us): y = G q with q = y_in * s, s = 1 / ||y_in||
// taken from the partial sums the previous step left (pin == nullptr: y_in is already a unit
// vector). 8 columns per workgroup (G symmetric: row i = column i, contiguous).
//   pout[2*blk] = the block's share of ||y||^2, pout[2*blk+1] = its share of q^T y
__global__ __launch_bounds__(256) void k_power_mv(const double *__restrict__ G, int64_t J,
                                                  const double *__restrict__ y_in,
                                                  const double *__restrict__ pin, int npin,
                                                  double *__restrict__ y_out,
                                                  double *__restrict__ pout) {
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double scale = 1.0;
  if (pin) {
    double n2 = 0;
    for (int i = lane; i < npin; i += 64) n2 += pin[2 * i];
    n2 = __shfl(wave_sum(n2), 0, 64);  // (lane 0 holds the sum)
    scale = 1.0 / sqrt(n2);
  }
  const int64_t i0 = (int64_t)blockIdx.x * 8 + 2 * wave;
  const double *c0 = G + J * min(i0, J - 1), *c1 = G + J * min(i0 + 1, J - 1);
  double s0 = 0, s1 = 0;
  for (int64_t k = lane; k < J; k += 64) {
    const double qk = y_in[k] * scale;
    s0 += c0[k] * qk;
    s1 += c1[k] * qk;
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  double n2 = 0, rq = 0;
  if (i0 < J) {
    n2 += s0 * s0;
    rq += s0 * (y_in[i0] * scale);
  }
  if (i0 + 1 < J) {
    n2 += s1 * s1;
    rq += s1 * (y_in[i0 + 1] * scale);
  }
  if (lane == 0) {
    if (i0 < J) y_out[i0] = s0;
    if (i0 + 1 < J) y_out[i0 + 1] = s1;
    red[0][wave] = n2;
    red[1][wave] = rq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    pout[2 * blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    pout[2 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}
// q_out = y / ||y||, *lam = the Rayleigh quotient of the vector that WENT INTO the last k_power_mv
__global__ __launch_bounds__(256) void k_power_finish(const double *__restrict__ y, int64_t J,
                                                      const double *__restrict__ p, int np,
                                                      double *__restrict__ q_out,
                                                      double *__restrict__ lam) {
  __shared__ double red[17];
  double n2 = 0, rq = 0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    n2 += p[2 * i];
    rq += p[2 * i + 1];
  }
  n2 = block_sum(n2, red);
  rq = block_sum(rq, red);
  const double inv = 1.0 / sqrt(n2);
  for (int64_t i = threadIdx.x; i < J; i += blockDim.x) q_out[i] = y[i] * inv;
  if (threadIdx.x == 0) *lam = rq;
}
// x[e] = a reproducible pseudo-random value in (-1, 1) (the start block of a cold subspace iteration)
__global__ void k_fill_hash(double *__restrict__ x, int64_t n, uint64_t seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    x[e] = 2.0 * u01(seed, (uint64_t)e) - 1.0;
}
// E (J x n, zeroed by the caller): column t = unit vector e_j with j = (t + 1) J / (n + 1)
__global__ void k_set_unit_cols(double *__restrict__ E, int64_t J, int n) {
  const int t = threadIdx.x;
  if (t < n) E[(int64_t)(t + 1) * J / (n + 1) + J * t] = 1.0;
}
// X *= 1 / (1.0001 * sqrt(*fro2)): the scaling of the sign iteration by a norm that stays on the
// device (no read-back between the deflation and the first product)
__global__ void k_scale_by_frob(double *__restrict__ X, int64_t n, const double *__restrict__ fro2) {
  const double f = 1.0 / (1.0001 * sqrt(*fro2));
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    X[e] *= f;
}

// C = A^T B for two tall matrices (rows x r, column-major, ld = rows): one wave per entry (p, q)
__global__ void k_tn_small(const double *__restrict__ A, const double *__restrict__ B, int64_t rows,
                           int r, double *__restrict__ C) {
  const int lane = threadIdx.x & 63;
  const int wid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nw = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  for (int e = wid; e < r * r; e += nw) {
    const int p = e % r, q = e / r;
    const double *a = A + rows * p, *b = B + rows * q;
    double s = 0;
    for (int64_t i = lane; i < rows; i += 64) s += a[i] * b[i];
    s = wave_sum(s);
    if (lane == 0) C[p + r * q] = s;
  }
}

// T = A^T B, A: rows x ra, B: rows x rb (column-major, ld = rows): one wave per entry
__global__ void k_tn_rect(const double *__restrict__ A, int ra, const double *__restrict__ B, int rb,
                          int64_t rows, double *__restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int wid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nw = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  for (int e = wid; e < ra * rb; e += nw) {
    const int p = e % ra, q = e / ra;
    const double *a = A + rows * p, *b = B + rows * q;
    double s = 0;
    for (int64_t i = lane; i < rows; i += 64) s += a[i] * b[i];
    s = wave_sum(s);
    if (lane == 0) T[p + ra * q] = s;
  }
}
// Z (rows x rz) -= Q (rows x m) * T (m x rz)
__global__ void k_sub_mult(double *__restrict__ Z, int64_t rows, int rz, const double *__restrict__ Q,
                           int m, const double *__restrict__ T) {
  const int64_t total = rows * rz;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % rows;
    const int c = (int)(e / rows);
    double acc = 0;
    for (int d = 0; d < m; d++) acc += Q[i + rows * d] * T[d + m * c];
    Z[e] -= acc;
  }
}

// In place: C (r x r, SPD, column-major) -> Rinv = L^{-T} with C = L L^T, so that Z * Rinv has
// orthonormal columns when C = Z^T Z (Cholesky QR). One wave, r <= 64, in LDS. *status = 1 when a
// pivot is not safely positive (Z numerically rank deficient: the caller falls back), 2 when the
// pivots spread by more than a factor 4.
__global__ __launch_bounds__(64) void k_chol_rinv(double *__restrict__ C, int r,
                                                  int *__restrict__ status) {
  extern __shared__ double lds[];
  double *L = lds;            // r x r, column-major, lower
  double *X = lds + r * r;    // inverse of L (lower)
  const int lane = threadIdx.x;
  for (int e = lane; e < r * r; e += 64) {
    L[e] = C[e];
    X[e] = 0.0;
  }
  wave_sync();
  double dmax = 0;
  for (int k = 0; k < r; k++) dmax = fmax(dmax, L[k + r * k]);
  bool bad = false;
  double pmin = dmax;  // smallest pivot: dmax / pmin <= cond(C)
  for (int k = 0; k < r; k++) {
    const double d = L[k + r * k];
    if (!(d > 1e-14 * dmax)) {
      bad = true;
      break;
    }
    pmin = fmin(pmin, d);
    const double sk = sqrt(d);
    wave_sync();
    for (int i = k + lane; i < r; i += 64) L[i + r * k] = (i == k) ? sk : L[i + r * k] / sk;
    wave_sync();
    for (int e = lane; e < (r - k - 1) * (r - k - 1); e += 64) {  // trailing update, lower part
      const int i = k + 1 + e % (r - k - 1), j = k + 1 + e / (r - k - 1);
      if (i >= j) L[i + r * j] -= L[i + r * k] * L[j + r * k];
    }
    wave_sync();
  }
  // status 2: factorised, but C is not close to the identity (pivots spread by more than 4): ONE
  // Cholesky-QR pass leaves an orthogonality error of eps * cond(C), a caller that runs a single
  // pass treats that as a failure
  if (lane == 0) *status = bad ? 1 : (dmax > 4.0 * pmin ? 2 : 0);
  if (bad) return;
  // X = L^{-1} by forward substitution, one column per lane
  for (int c = lane; c < r; c += 64) {
    for (int i = c; i < r; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; k++) s -= L[i + r * k] * X[k + r * c];
      X[i + r * c] = s / L[i + r * i];
    }
  }
  wave_sync();
  for (int e = lane; e < r * r; e += 64) {  // Rinv = X^T (upper triangular)
    const int i = e % r, j = e / r;
    C[e] = (i <= j) ? X[j + r * i] : 0.0;
  }
}

// out[i, q] = sum_p Z[i, p] * T[p, q]   (rows x r times r x r; out may NOT alias Z)
__global__ void k_right_mult(const double *__restrict__ Z, int64_t rows, int r,
                             const double *__restrict__ T, double *__restrict__ out) {
  extern __shared__ double sT[];
  for (int e = threadIdx.x; e < r * r; e += blockDim.x) sT[e] = T[e];
  __syncthreads();
  const int64_t total = rows * r;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % rows;
    const int q = (int)(e / rows);
    double acc = 0;
    for (int p = 0; p < r; p++) acc += Z[i + rows * p] * sT[p + r * q];
    out[e] = acc;
  }
}

// Z = 0.5 * (Omega + XO)   (projector applied to the previous basis)
__global__ void k_half_sum(const double *__restrict__ a, const double *__restrict__ b, int64_t n,
                           double *__restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    out[e] = 0.5 * (a[e] + b[e]);
}

// ---------------------------------------------------------------------------------------------
// The tail of a projector step in TWO single-workgroup launches (was ~17: projections, Gram,
// Cholesky, triangular product, transposition, H = B^T G B, Jacobi, two back-products, residual,
// reductions of the check sums). Everything here is a few hundred rows by a few dozen columns.
//
// k_block_orth: Z (J x cols, global) -> B (J x cols, global) with orthonormal columns spanning the
// same space: [optional] the m leading columns are REPLACED by QD (J x m, orthonormal: the deflated
// eigenvectors from the power iteration) and the others are cleared of them (twice); then Cholesky
// QR, `npass` passes (2 unless Z is known to be nearly orthonormal). Z lives in LDS as it goes
// (J * cols * 8 bytes of dynamic LDS + 2 cols^2 + 64). status[0..npass): as k_chol_rinv.
__global__ __launch_bounds__(1024) void k_block_orth(const double *__restrict__ Z, int64_t J, int cols,
                                                     const double *__restrict__ QD, int m, int npass,
                                                     double *__restrict__ B, int *__restrict__ status) {
  extern __shared__ double lds[];
  double *Zs = lds;                           // J x cols, column-major
  double *Cm = Zs + (size_t)J * cols;         // cols x cols
  double *Ri = Cm + cols * cols;              // cols x cols
  double *red = Ri + cols * cols;             // 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int Ji = (int)J;
  for (int e = tid; e < Ji * cols; e += blockDim.x) {
    const int c = e / Ji;
    Zs[e] = (c < m) ? QD[e] : Z[e];
  }
  __syncthreads();
  for (int pass = 0; pass < 2 && m > 0; pass++) {
    // t[d][c] = q_d^T z_c, then z_c -= sum_d q_d t[d][c]   (c >= m)
    for (int e = wave; e < m * (cols - m); e += nw) {
      const int d = e % m, c = m + e / m;
      double s = 0;
      for (int i = lane; i < Ji; i += 64) s += Zs[i + Ji * d] * Zs[i + Ji * c];
      s = wave_sum(s);
      if (lane == 0) Cm[d + m * (c - m)] = s;
    }
    __syncthreads();
    for (int e = tid; e < Ji * (cols - m); e += blockDim.x) {
      const int i = e % Ji, c = m + e / Ji;
      double a = 0;
      for (int d = 0; d < m; d++) a += Zs[i + Ji * d] * Cm[d + m * (c - m)];
      Zs[i + Ji * c] -= a;
    }
    __syncthreads();
  }
  for (int pass = 0; pass < npass; pass++) {
    // C = Z^T Z (upper triangle computed, mirrored)
    const int ntri = cols * (cols + 1) / 2;
    for (int e = wave; e < ntri; e += nw) {
      int p = 0, rem = e;
      while (rem >= cols - p) {
        rem -= cols - p;
        p++;
      }
      const int q = p + rem;
      double s = 0;
      for (int i = lane; i < Ji; i += 64) s += Zs[i + Ji * p] * Zs[i + Ji * q];
      s = wave_sum(s);
      if (lane == 0) {
        Cm[p + cols * q] = s;
        Cm[q + cols * p] = s;
      }
    }
    __syncthreads();
    if (wave == 0) {  // Cholesky C = L L^T in place (lower), Ri = L^{-T}: one wave, as k_chol_rinv
      const int r = cols;
      double *L = Cm, *X = Ri;
      for (int e = lane; e < r * r; e += 64) X[e] = 0.0;
      wave_sync();
      double dmax = 0;
      for (int k = 0; k < r; k++) dmax = fmax(dmax, L[k + r * k]);
      bool bad = false;
      double pmin = dmax;
      for (int k = 0; k < r; k++) {
        const double d = L[k + r * k];
        if (!(d > 1e-14 * dmax)) {
          bad = true;
          break;
        }
        pmin = fmin(pmin, d);
        const double sk = sqrt(d);
        wave_sync();
        for (int i = k + lane; i < r; i += 64) L[i + r * k] = (i == k) ? sk : L[i + r * k] / sk;
        wave_sync();
        for (int e = lane; e < (r - k - 1) * (r - k - 1); e += 64) {
          const int i = k + 1 + e % (r - k - 1), j = k + 1 + e / (r - k - 1);
          if (i >= j) L[i + r * j] -= L[i + r * k] * L[j + r * k];
        }
        wave_sync();
      }
      if (lane == 0) status[pass] = bad ? 1 : (dmax > 4.0 * pmin ? 2 : 0);
      if (lane == 0) red[0] = bad ? 1.0 : 0.0;
      if (!bad) {
        for (int c = lane; c < r; c += 64) {  // X = L^{-1}, one column per lane
          for (int i = c; i < r; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; k++) s -= L[i + r * k] * X[k + r * c];
            X[i + r * c] = s / L[i + r * i];
          }
        }
      }
      wave_sync();
    }
    __syncthreads();
    if (red[0] != 0.0) break;  // rank deficient: the caller falls back (status says so)
    // Z <- Z * L^{-T}: column q of the result = sum_{p <= q} z_p * Linv[q][p] needs the OLD columns
    // p <= q only, so the columns are replaced from the last one down, every thread its own rows
    for (int i = tid; i < Ji; i += blockDim.x) {
      for (int q = cols - 1; q >= 0; q--) {
        double a = 0;
        for (int p = 0; p <= q; p++) a += Zs[i + Ji * p] * Ri[q + cols * p];
        Zs[i + Ji * q] = a;
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < Ji * cols; e += blockDim.x) B[e] = Zs[e];
}

// out[0] = sum a[0..n), out[1] = sum b[0..n)  (the check sums of a counting trial)
__global__ __launch_bounds__(256) void k_chk_sums(const double *__restrict__ a,
                                                  const double *__restrict__ b, int n,
                                                  double *__restrict__ out) {
  __shared__ double lds[17];
  double x = 0, y = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    x += a[i];
    y += b[i];
  }
  x = block_sum(x, lds);
  y = block_sum(y, lds);
  if (threadIdx.x == 0) {
    out[0] = x;
    out[1] = y;
  }
}

// k_rr_tail: Rayleigh-Ritz of G on the orthonormal basis B (J x cols) given GB = G B: H = B^T GB
// (cols x cols), its eigen-decomposition by the in-LDS Jacobi, eigenvalues sorted descending ->
// evW[0..cols), U = B Y[:, :rank] and the residual ||G U - U diag(ev)||_F^2 of the leading `rank`
// pairs -> chk[4]; on the way the partial check sums the sign iteration's last two products left
// (np each) are added up: chk[0] = ||X_prev^2 - I||_F^2, chk[1] = trace(X).
// dynamic LDS: A[cols][cols+1] | Q[cols][cols+1] | cs[64] | pq[64 ints] | red[17] | ord[64 ints]
__global__ __launch_bounds__(1024) void k_rr_tail(const double *__restrict__ Bm,
                                                  const double *__restrict__ GB, int64_t J, int cols,
                                                  int rank, const double *__restrict__ part_e2,
                                                  const double *__restrict__ part_tr, int np,
                                                  double *__restrict__ U, double *__restrict__ evW,
                                                  double *__restrict__ chk) {
  extern __shared__ double lds[];
  const int ldA = cols + 1;
  double *A = lds;
  double *Q = A + cols * ldA;
  double *cs = Q + cols * ldA;
  int *pq = (int *)(cs + 64);
  double *red = (double *)(pq + 64);
  int *ord = (int *)(red + 17);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int Ji = (int)J;
  // H = B^T (G B), symmetrised from its upper triangle
  const int ntri = cols * (cols + 1) / 2;
  for (int e = wave; e < ntri; e += nw) {
    int p = 0, rem = e;
    while (rem >= cols - p) {
      rem -= cols - p;
      p++;
    }
    const int q = p + rem;
    const double *a = Bm + J * p, *b = GB + J * q;
    double s = 0;
    for (int i = lane; i < Ji; i += 64) s += a[i] * b[i];
    s = wave_sum(s);
    if (lane == 0) {
      A[p * ldA + q] = s;
      A[q * ldA + p] = s;
    }
  }
  __syncthreads();
  if (cols <= 32) {  // one wave: wave-level barriers only (rounds of <= 16 rotations x 32 rows)
    if (wave == 0) jacobi_eig_t<64>(A, Q, cs, pq, cols, nullptr);
    __syncthreads();
  } else {
    jacobi_eig_t<1024>(A, Q, cs, pq, cols, red);
  }
  if (tid < cols) {
    const double wk = A[tid * ldA + tid];
    int pos = 0;
    for (int j = 0; j < cols; j++) {
      const double wj = A[j * ldA + j];
      if (wj > wk || (wj == wk && j < tid)) pos++;
    }
    ord[pos] = tid;
    evW[pos] = wk;
  }
  __syncthreads();
  // U = B Y, residual of the leading pairs: r_i,k = (GB Y)_ik - U_ik * ev_k
  double res = 0;
  for (int e = tid; e < Ji * rank; e += blockDim.x) {
    const int i = e % Ji, k = e / Ji;
    const int col = ord[k];
    double u = 0, gu = 0;
    for (int p = 0; p < cols; p++) {
      const double y = Q[p * ldA + col];
      u += Bm[i + J * p] * y;
      gu += GB[i + J * p] * y;
    }
    U[e] = u;
    const double d = gu - u * A[col * ldA + col];
    res += d * d;
  }
  res = block_sum(res, red);
  double e2 = 0, tr = 0;
  for (int i = tid; i < np; i += blockDim.x) {
    e2 += part_e2[i];
    tr += part_tr[i];
  }
  e2 = block_sum(e2, red);
  tr = block_sum(tr, red);
  if (tid == 0) {
    chk[0] = e2;
    chk[1] = tr;
    chk[4] = res;
  }
}

}  // namespace ppals
