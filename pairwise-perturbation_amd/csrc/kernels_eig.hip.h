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

// Ranking key of an eigenvalue: a NaN (a Rayleigh-Ritz matrix built from a block that lost rank — an
// exactly low-rank tensor, whose Gram has fewer non-zero eigenvalues than the block has columns —
// reaches the Jacobi before the host has read the factorisation's status) compares false with
// everything, so several columns would take rank 0 and the other entries of the permutation stay
// unwritten: an index out of nowhere, a memory fault in the gather behind it. NaNs rank last, ties by index.
__device__ inline double eig_rank_key(double v) { return v == v ? v : -1.0e308; }


// Layout of the block a projector step leaves for its one read-back (device workspace, the slot's
// pinned copy): check words | eigenvalues (rank + wide columns) | status | deflated eigenvalues |
// per-workgroup residual shares.
constexpr int kEigEvMax = 128;  // eigenvalues a step can return: core rank + 16 <= 128
constexpr int kEigOffEv = 16, kEigOffStatus = kEigOffEv + kEigEvMax, kEigOffLamD = kEigOffStatus + 4,
              kEigOffResp = kEigOffLamD + 64, kEigChkDoubles = kEigOffResp + 64;

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
                                                  int K, double alpha, double beta,
                                                  const double *__restrict__ Qov = nullptr, int mov = 0) {
  // Qov / mov: the first `mov` result columns are REPLACED by the columns of Qov (M x mov, ld = M) —
  // the deflated eigenvector takes its place in front of the projected basis without a copy launch
  constexpr int UN = 13;
  __shared__ double part[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  // clamped operand rows: lanes past the edge re-read the last row, their results are not stored
  const int ia = min(i0 + l16, M - 1), jb = min(j0 + l16, N - 1);
  // 32-bit element offsets from the (uniform) matrix bases: one v_mad per load instead of a 64-bit
  // multiply-add chain — the address arithmetic of 26 loads per lane was ~1 us of a 8 us launch
  // (tools/nsprod_bench.hip, round 4). The matrices here are far below 2^32 bytes.
  const unsigned la = (unsigned)lda, lb = (unsigned)ldb;
  const unsigned a0 = (unsigned)ia, b0 = BN ? lb * (unsigned)jb : (unsigned)jb;
  const unsigned bstep = BN ? 1u : lb;
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
      const double a = A[a0 + la * (unsigned)kc], b = Bt[b0 + bstep * (unsigned)kc];
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
        if (j < mov) v = Qov[i + (int64_t)M * j];
        C[i + ldc * (int64_t)j] = v;
      }
    }
  }
}

// Batched mode product of a SMALL tensor (the second-level products of a Tucker chain: a 400 x 400
// x 20 intermediate against a 400 x 20 factor): out[l + L*(k + R*t)] = sum_j X[l + L*(j + J*t)] *
// W[j + ldw*k], one batch per t (blockIdx.z). The streaming scan kernel serves this with a packed
// operand, split-K slabs and their combine — three launches of 5 + 9 + 5 us at cfg5 for 25 MB of
// input; this is ONE launch of the thin GEMM above (16 x 16 tile per workgroup, its four waves split
// J), the input widened to fp64 on its way into the matrix cores when the chain keeps it in fp32.
template <typename TA, int NT, int NW = 8>
__global__ __launch_bounds__(64 * NW) void k_mode_product_small(const TA *__restrict__ X, int L, int J,
                                                            const double *__restrict__ W, int64_t ldw, int R,
                                                            double *__restrict__ out) {
  // NT column tiles of 16 per workgroup: the tensor operand is loaded once for all of them; NW waves
  // split J (eight at J = 400: ONE round of 13 k-steps per wave, every load of a wave in flight at once)
  constexpr int UN = NT == 1 ? 13 : (NT == 2 ? 13 : 7);
  __shared__ double part[NW - 1][NT][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16 * NT;
  const TA *__restrict__ A = X + (size_t)blockIdx.z * L * J;        // this batch: L x J, ld = L
  double *__restrict__ C = out + (size_t)blockIdx.z * L * R;         // L x R, ld = L
  const unsigned a0 = (unsigned)min(i0 + l16, L - 1);
  unsigned b0[NT];
#pragma unroll
  for (int n = 0; n < NT; n++) b0[n] = (unsigned)ldw * (unsigned)min(j0 + 16 * n + l16, R - 1);
  f64x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int ksteps = (J + 3) / 4;
  const int spw = (ksteps + NW - 1) / NW;
  const int s_begin = wave * spw, s_end = min(ksteps, s_begin + spw);
  for (int s0 = s_begin; s0 < s_end; s0 += UN) {
    double av[UN], bv[NT][UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int k = (s0 + u) * 4 + g;
      const bool ok = (s0 + u) < s_end && k < J;
      const int kc = ok ? k : 0;
      const double a = (double)A[a0 + (unsigned)L * (unsigned)kc];
      av[u] = ok ? a : 0.0;
#pragma unroll
      for (int n = 0; n < NT; n++) {
        const double b = W[b0[n] + (unsigned)kc];
        bv[n][u] = ok ? b : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; u++)
#pragma unroll
      for (int n = 0; n < NT; n++)
        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[n][u], acc[n], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) part[wave - 1][n][r][lane] = acc[n][r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int n = 0; n < NT; n++) {
    const int j = j0 + 16 * n + l16;
    if (j < R) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = i0 + g + 4 * r;
        if (i < L) {
          double v = acc[n][r];
#pragma unroll
          for (int w = 0; w < NW - 1; w++) v += part[w][n][r][lane];  // fixed order
          C[i + (size_t)L * j] = v;
        }
      }
    }
  }
}

// The product with the LEADING mode of a small tensor, result with the next mode in front
// (Ops::ttm_lead_front): per batch t, C_t (S x R) = A_t^T W with A_t (J x S, ld = J) — both operands are
// read along the contraction index (16 lines per operand load: the data is L2-resident, 25 MB at
// cfg5). Same tiling as k_mode_product_small: 16 result rows x NT column tiles per workgroup, NW waves
// split J.
template <typename TA, int NT, int NW = 8>
__global__ __launch_bounds__(64 * NW) void k_mode_product_lead(const TA *__restrict__ X, int J, int S,
                                                               const double *__restrict__ W, int64_t ldw, int R,
                                                               double *__restrict__ out) {
  constexpr int UN = NT <= 2 ? 13 : 7;
  __shared__ double part[NW - 1][NT][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16 * NT;
  const TA *__restrict__ A = X + (size_t)blockIdx.z * J * S;         // this batch: J x S, ld = J
  double *__restrict__ C = out + (size_t)blockIdx.z * S * R;          // S x R, ld = S
  const unsigned a0 = (unsigned)J * (unsigned)min(i0 + l16, S - 1);
  unsigned b0[NT];
#pragma unroll
  for (int n = 0; n < NT; n++) b0[n] = (unsigned)ldw * (unsigned)min(j0 + 16 * n + l16, R - 1);
  f64x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int ksteps = (J + 3) / 4;
  const int spw = (ksteps + NW - 1) / NW;
  const int s_begin = wave * spw, s_end = min(ksteps, s_begin + spw);
  for (int s0 = s_begin; s0 < s_end; s0 += UN) {
    double av[UN], bv[NT][UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int k = (s0 + u) * 4 + g;
      const bool ok = (s0 + u) < s_end && k < J;
      const int kc = ok ? k : 0;
      const double a = (double)A[a0 + (unsigned)kc];
      av[u] = ok ? a : 0.0;
#pragma unroll
      for (int n = 0; n < NT; n++) {
        const double b = W[b0[n] + (unsigned)kc];
        bv[n][u] = ok ? b : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; u++)
#pragma unroll
      for (int n = 0; n < NT; n++)
        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[n][u], acc[n], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) part[wave - 1][n][r][lane] = acc[n][r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int n = 0; n < NT; n++) {
    const int j = j0 + 16 * n + l16;
    if (j < R) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = i0 + g + 4 * r;
        if (i < S) {
          double v = acc[n][r];
#pragma unroll
          for (int w = 0; w < NW - 1; w++) v += part[w][n][r][lane];  // fixed order
          C[i + (size_t)S * j] = v;
        }
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
  const unsigned la = (unsigned)lda, lb = (unsigned)ldb;  // (32-bit offsets: see k_dgemm_nx)
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
      const double a = A[(unsigned)ia + la * (unsigned)kc], b = Bt[(unsigned)jb + lb * (unsigned)kc];
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

// The symmetric product for LARGE matrices (J of many hundreds: the Tucker modes of the reference's
// own data sets, 1344 rows at rank 100, test_ALS.cxx:366-379), where a product is no longer a
// latency problem but 2 J^3 flops: TS x TS output tile per workgroup (TS = 64: four waves of
// 32 x 32 = 2 x 2 MFMA tiles each, one LDS read per MFMA; TS = 32: four waves of 16 x 16), both
// operand panels staged through LDS 16 reduction indices at a time, the next chunk's global loads
// (16-byte, two per thread and panel) in flight while the current one is multiplied. Row stride of
// the panels = TS + 16 doubles: the four k-groups of an operand read fall into different bank
// halves. Same contract as k_dgemm_nt_sym: upper-triangle tiles only, every value stored at (i, j)
// and (j, i), the optional check sums per workgroup. 231 workgroups at J = 1344.
template <int TS>
__global__ __launch_bounds__(256) void k_dgemm_nt_sym_lds(const double *__restrict__ A, int64_t lda,
                                                          const double *__restrict__ Bt, int64_t ldb,
                                                          const double *__restrict__ D, int64_t ldd,
                                                          double *__restrict__ C, int64_t ldc, int M,
                                                          int K, double alpha, double beta,
                                                          int chk_mode = 0,
                                                          double *__restrict__ chk_part = nullptr) {
  constexpr int KC = 16, LD = TS + 16, NT = TS / 32;  // NT x NT MFMA tiles per wave
  constexpr int PT = TS * KC / 256 / 2;               // double2 loads per thread and panel (2 or 1)
  __shared__ double As[KC][LD];
  __shared__ double Bs[KC][LD];
  __shared__ double red[17];
  const int nt = (M + TS - 1) / TS;
  int ti = 0, rem = blockIdx.x;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ti++;
  }
  const int tj = ti + rem;
  const bool diag = (ti == tj) && (A == Bt) && (lda == ldb);
  const int p0 = ti * TS, q0 = tj * TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int wi = (wave & 1) * (TS / 2), wj = (wave >> 1) * (TS / 2);
  f64x4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  // a thread's slice of a TS x 16 panel: rows pr, pr + 1 of columns cc + (16 / PT) * h
  constexpr int RT = TS / 2;  // threads along the rows
  const int pr = (tid % RT) * 2, cc = tid / RT;
  const bool vec_ok = ((lda | ldb) & 1) == 0 && (((uintptr_t)A | (uintptr_t)Bt) & 15) == 0;
  double ra[2 * PT], rb[2 * PT];
  auto fetch = [&](int k0, int r0, const double *__restrict__ X, int64_t ldx, double *r) {
#pragma unroll
    for (int h = 0; h < PT; h++) {
      const int k = k0 + cc + (KC / PT) * h, row = r0 + pr;
      double v0 = 0.0, v1 = 0.0;
      if (k < K) {
        if (vec_ok && row + 1 < M) {
          const double2 v = *reinterpret_cast<const double2 *>(X + row + ldx * (int64_t)k);
          v0 = v.x;
          v1 = v.y;
        } else {
          if (row < M) v0 = X[row + ldx * (int64_t)k];
          if (row + 1 < M) v1 = X[row + 1 + ldx * (int64_t)k];
        }
      }
      r[2 * h] = v0;
      r[2 * h + 1] = v1;
    }
  };
  auto stash = [&](double (*S)[LD], const double *r) {
#pragma unroll
    for (int h = 0; h < PT; h++) {
      S[cc + (KC / PT) * h][pr] = r[2 * h];
      S[cc + (KC / PT) * h][pr + 1] = r[2 * h + 1];
    }
  };
  fetch(0, p0, A, lda, ra);
  if (!diag) fetch(0, q0, Bt, ldb, rb);
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();  // the previous chunk's fragments have been read
    stash(As, ra);
    if (!diag) stash(Bs, rb);
    __syncthreads();
    if (k0 + KC < K) {  // next chunk on its way while this one is multiplied
      fetch(k0 + KC, p0, A, lda, ra);
      if (!diag) fetch(k0 + KC, q0, Bt, ldb, rb);
    }
    double (*Bp)[LD] = diag ? As : Bs;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ks++) {
      double av[NT], bv[NT];
#pragma unroll
      for (int a = 0; a < NT; a++) av[a] = As[4 * ks + g][wi + 16 * a + l16];
#pragma unroll
      for (int b = 0; b < NT; b++) bv[b] = Bp[4 * ks + g][wj + 16 * b + l16];
#pragma unroll
      for (int a = 0; a < NT; a++)
#pragma unroll
        for (int b = 0; b < NT; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  double chk = 0.0;
#pragma unroll
  for (int a = 0; a < NT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) {
      const int j = q0 + wj + 16 * b + l16;  // D map: lane holds column l16, rows g + 4 r
      if (j < M) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = p0 + wi + 16 * a + g + 4 * r;
          if (i < M && i <= j) {  // (tiles on the diagonal store their upper half only, mirrored)
            double v = alpha * acc[a][b][r];
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
    }
  if (chk_mode) {
    chk = block_sum(chk, red);
    if (tid == 0) chk_part[blockIdx.x] = chk;
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

// K13, second generation (fp32 tensor storage): the same Gram as a real SYRK.
//   * only the tiles on and above the diagonal are computed (blockIdx.x enumerates them), every
//     value is written at (p, q) and (q, p) of the slab;
//   * 64 x 64 tile per workgroup, each of its four waves a 32 x 32 quarter = four 16 x 16
//     v_mfma_f64_16x16x4_f64 accumulators fed from two A and two B fragments per reduction step
//     (one LDS read per MFMA instead of two);
//   * 32 reduction indices per barrier pair as before, but the NEXT chunk's global loads are in
//     flight (16 floats per thread in registers) while the current one is multiplied;
//   * operands are widened to fp64 on their way into LDS: products of fp32 values are exact in
//     fp64, so the Gram carries fp64 rounding only (lambda_1 is 1e4 x the bulk for a tensor with
//     a mean component; fp32 products would smear it over the eigenvectors that matter).
// Needs 4-element aligned rows: J % 4 == 0 when L == 1 (p fastest), L % 4 == 0 otherwise, and
// c_per_split % 32 == 0. LDS: As, Bs [32][80] doubles (row stride 80: the four k-groups of an MFMA
// operand read land in different bank groups).
__global__ __launch_bounds__(256) void k_unfold_syrk_f32(const float *__restrict__ X, int64_t L,
                                                         int64_t J, int64_t T, int64_t c_per_split,
                                                         double *__restrict__ slab) {
  constexpr int TS = 64, KC = 32, LD = 80;
  __shared__ double As[KC][LD];
  __shared__ double Bs[KC][LD];
  const int64_t C = L * T;
  const int nt = (int)((J + TS - 1) / TS);
  int ti = 0, rem = blockIdx.x;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ti++;
  }
  const int tj = ti + rem;
  const bool diag = ti == tj;
  const int64_t p0 = (int64_t)ti * TS, q0 = (int64_t)tj * TS;
  const int64_t c_begin = (int64_t)blockIdx.z * c_per_split;
  const int64_t c_end = min(C, c_begin + c_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
  const bool p_fast = (L == 1);
  f64x4 acc[2][2];
  // which of the wave's four 16 x 16 sub-tiles are worth the matrix cores (wave-uniform): not those
  // past the edge (J = 400: the seventh tile of 64 holds 16 rows) and, in a tile on the diagonal, not
  // those strictly below it — they are the mirror images of the ones above, written from there. At
  // J = 400: 325 sub-tiles instead of 448 (the other waves of a SIMD get the pipe meanwhile).
  bool use[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
      const int64_t r0 = p0 + wi + 16 * a, c0 = q0 + wj + 16 * b;
      use[a][b] = r0 < J && c0 < J && (!diag || c0 >= r0);
    }
  // per-thread slice of a 64 x 32 panel: p-fast: rows pq..pq+3 of columns cc, cc + 16;
  // c-fast: row pp, columns cq..cq+7
  const int pq = (tid & 15) * 4, cc = tid >> 4;
  const int pp = tid >> 2, cq = (tid & 3) * 8;
  float ra[8], rb[8];
  auto fetch = [&](int64_t c0, int64_t r0, float *r) {
    if (p_fast) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int64_t c = c0 + cc + 16 * h;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < c_end && r0 + pq < J) v = *reinterpret_cast<const float4 *>(X + (r0 + pq) + J * c);
        r[4 * h + 0] = v.x;
        r[4 * h + 1] = v.y;
        r[4 * h + 2] = v.z;
        r[4 * h + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int64_t c = c0 + cq + 4 * h;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < c_end && r0 + pp < J) {
          const int64_t l = c % L, t = c / L;
          v = *reinterpret_cast<const float4 *>(X + l + L * ((r0 + pp) + J * t));
        }
        r[4 * h + 0] = v.x;
        r[4 * h + 1] = v.y;
        r[4 * h + 2] = v.z;
        r[4 * h + 3] = v.w;
      }
    }
  };
  auto stash = [&](double (*S)[LD], const float *r) {
    if (p_fast) {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 4; k++) S[cc + 16 * h][pq + k] = (double)r[4 * h + k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) S[cq + k][pp] = (double)r[k];
    }
  };
  fetch(c_begin, p0, ra);
  if (!diag) fetch(c_begin, q0, rb);
  for (int64_t c0 = c_begin; c0 < c_end; c0 += KC) {
    __syncthreads();  // the previous chunk's fragments have been read
    stash(As, ra);
    if (!diag) stash(Bs, rb);
    __syncthreads();
    if (c0 + KC < c_end) {  // next chunk on its way while this one is multiplied
      fetch(c0 + KC, p0, ra);
      if (!diag) fetch(c0 + KC, q0, rb);
    }
    double (*Bp)[LD] = diag ? As : Bs;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ks++) {
      const double a0 = As[4 * ks + g][wi + l16], a1 = As[4 * ks + g][wi + 16 + l16];
      const double b0 = Bp[4 * ks + g][wj + l16], b1 = Bp[4 * ks + g][wj + 16 + l16];
      if (use[0][0]) acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      if (use[0][1]) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      if (use[1][0]) acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      if (use[1][1]) acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  double *gs = slab + (int64_t)blockIdx.z * J * J;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int64_t q = q0 + wj + 16 * b + l16;  // D: lane holds column l16, rows g + 4 r
      // (a sub-tile ON the diagonal holds both of its triangles; every other one is mirrored)
      const bool mirror = !diag || (q0 + wj + 16 * b) > (p0 + wi + 16 * a);
      if (use[a][b] && q < J) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int64_t pr = p0 + wi + 16 * a + g + 4 * r;
          if (pr < J) {
            gs[pr + J * q] = acc[a][b][r];
            if (mirror) gs[q + J * pr] = acc[a][b][r];
          }
        }
      }
    }
}

// the probe of the stored-value hand-over (HipOps::lazy_prepare): publishes `seq` the way the last
// workgroup of k_rmult_chol does
__global__ void k_handover_probe(unsigned long long *flag, unsigned long long seq) {
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
                                                    double *__restrict__ lam_out,
                                                    int *__restrict__ zero8 = nullptr,
                                                    const double *__restrict__ q_prev = nullptr,
                                                    double *__restrict__ move_out = nullptr) {
  // q_prev / move_out: *move_out = 1 - (q_prev^T q)^2, the squared sine between the vector the power
  // steps started from and the one they ended in — how far the dominant eigenvector moved since the
  // slot's previous step; the host counts the next step's power launches from it
  __shared__ double lds[17];
  // (the step's status words are cleared here instead of by a memset launch of their own)
  if (zero8 && blockIdx.x == 0 && threadIdx.x < 8) zero8[threadIdx.x] = 0;
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
      double dot = 0;
      for (int64_t i = threadIdx.x; i < J; i += blockDim.x) {
        const double qi = y[i] * inv;
        q_out[i] = qi;
        if (q_prev) dot += qi * q_prev[i];
      }
      if (threadIdx.x == 0) *lam_out = lam;
      if (move_out) {
        dot = block_sum(dot, lds);
        if (threadIdx.x == 0) *move_out = fmax(0.0, 1.0 - dot * dot);
      }
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
                                                       double *__restrict__ partial,
                                                       int *__restrict__ zero8 = nullptr) {
  __shared__ double lds[17];
  if (zero8 && blockIdx.x == 0 && threadIdx.x < 8) zero8[threadIdx.x] = 0;
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

// Z[:, d] /= ev[d] (ev[d] <= 0: left alone): the block G * (Ritz vectors) brought back to unit scale
// column by column, so that the Cholesky QR behind it sees a nearly orthonormal block whatever the
// spread of the Ritz values (cold_subspace)
__global__ void k_scale_cols_inv(double *__restrict__ Z, int64_t J, int r, const double *__restrict__ ev) {
  const int64_t total = J * r;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const double v = ev[e / J];
    if (v > 0) Z[e] /= v;
  }
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
                           int r, double *__restrict__ C, double *__restrict__ C2 = nullptr) {
  const int nblocks = gridDim.x;
  const int lane = threadIdx.x & 63;
  const int wid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nw = (int)(((int64_t)nblocks * blockDim.x) >> 6);
  for (int e = wid; e < r * r; e += nw) {
    const int p = e % r, q = e / r;
    const double *a = A + rows * p, *b = B + rows * q;
    double s = 0;
    for (int64_t i = lane; i < rows; i += 64) s += a[i] * b[i];
    s = wave_sum(s);
    if (lane == 0) {
      C[p + r * q] = s;
      if (C2) C2[p + r * q] = s;  // (a second copy for a consumer on another stream: no blit launch)
    }
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
// The tail of a projector step: thin products and reductions over the J rows stay multi-workgroup
// (a lone workgroup waits a full memory round trip per dependent access: the first fused version,
// two one-workgroup kernels, took 320 us where these six launches take ~60); what is only
// cols x cols — the Cholesky factor of the basis' Gram, the eigen-decomposition of H — is done by
// EVERY workgroup of the launch that needs it, redundantly, in its own LDS, instead of a launch of
// its own plus a global round trip.
//
// k_tn_two: C[p + cols q] = z_p^T z_q (cols x cols) and, behind it, t[d + m c] = q_d^T z_c (m x cols):
// one wave per entry.
__global__ void k_tn_two(const double *__restrict__ Z, const double *__restrict__ QD, int64_t rows,
                         int cols, int m, double *__restrict__ C) {
  const int lane = threadIdx.x & 63;
  const int wid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nw = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int n1 = cols * cols, ntot = n1 + m * cols;
  for (int e = wid; e < ntot; e += nw) {
    const double *a, *b;
    if (e < n1) {
      a = Z + rows * (e % cols);
      b = Z + rows * (e / cols);
    } else {
      a = QD + rows * ((e - n1) % m);
      b = Z + rows * ((e - n1) / m);
    }
    double s = 0;
    for (int64_t i = lane; i < rows; i += 64) s += a[i] * b[i];
    s = wave_sum(s);
    if (lane == 0) C[e] = s;
  }
}

// ---- the tail of a deferred step (round 4). On the sweep's stream only what the NEXT mode waits for:
//   Z' = [q_D | P Omega_rest]  (thin GEMM)  |  k_tn_gram: C1 = Z'^T Z'  |  k_rmult_chol: B = Z' M
// With the deflated vector in front of Z', ONE Cholesky QR of all columns keeps it (the first column
// of a QR is the first column, normalised) and clears the others of it. Everything the CHECKS read —
// G B, H = B^T G B, the residual, the Jacobi of H — is formed from B on the second stream, from the
// slot's own copy of the Gram (Ops::eig_gram), beside the next mode's tensor scan.
// k_tn_gram: C1[p + cols q] = z_p^T z_q — one wave per entry. The last workgroup instead adds up the
// sign iteration's check sums (np partials each) into chk_out[0], chk_out[1] and carries the step's norm
// word [8] over into the slot's check block.
__global__ void k_tn_gram(const double *__restrict__ Z, int64_t rows, int cols, double *__restrict__ C1,
                          const double *__restrict__ part_e2, const double *__restrict__ part_tr, int np,
                          const double *__restrict__ chk_src, double *__restrict__ chk_out) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x >= 64) return;
    double e2 = 0, tr = 0;
    for (int i = lane; i < np; i += 64) {
      e2 += part_e2[i];
      tr += part_tr[i];
    }
    e2 = wave_sum(e2);
    tr = wave_sum(tr);
    if (lane == 0) {
      chk_out[0] = e2;
      chk_out[1] = tr;
      chk_out[8] = chk_src[8];
      chk_out[9] = chk_src[9];  // (how far the dominant eigenvector moved: k_ns_prepare)
    }
    return;
  }
  const int wid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nw = (int)(((int64_t)(gridDim.x - 1) * blockDim.x) >> 6);
  for (int e = wid; e < cols * cols; e += nw) {
    const int p = e % cols, q = e / cols;
    const double *a = Z + rows * p, *b = Z + rows * q;
    double s1 = 0;
    for (int64_t i = lane; i < rows; i += 64) s1 += a[i] * b[i];
    s1 = wave_sum(s1);
    if (lane == 0) C1[e] = s1;
  }
}
// k_rmult_chol: B = Z M with M^T (Z^T Z) M = I — the small factorisation INSIDE the multiplication:
// every workgroup (1024 threads, 64 rows of Z) forms M for itself in LDS while its rows of Z are on
// their way into the cache, then multiplies. One launch instead of factorisation + multiplication,
// and no round trip of the n x n matrix through memory.
//
// Two ways to M. Z = [q_D | P Omega'] (m = 1: the deflated unit vector in front) or P Omega (m = 0)
// with the PREVIOUS orthonormal basis Omega, so S = the Gram of the projected columns (after the
// one-pivot elimination of q_D) is I - E with E = Omega^T (I - P) Omega: the squared sines of the
// angles between the old and the new subspace — a HOOI sweep of cfg5 turns it by 0.03-0.07 rad,
// ||E||_F = 5e-4 ... 5e-3.
//  (a) ||E||_F <= kSeriesTol: S^-1/2 = sum_k c_k E^k, c_k = binom(2k, k) / 4^k, up to
//      E^9 by Paterson-Stockmeyer — E^2, E^3 and two more products: FOUR dependent stages of (n-1)^2
//      elements instead of ten elimination rounds; remainder 0.18 ||E||^10 < 1e-16.
//      B = [z_0 c00^-1/2 | (Z' - z_0 t) S^-1/2], t = C1[0, 1:] / c00: q_D stays the first column (the
//      power steps of the slot's next call start from it). status[1] = 3 marks the route in the log.
//  (b) otherwise: the elimination of k_chol_m on ALL columns (two pivots per barrier: ten dependent
//      rounds at n = 20): C1 -> D L_1^T, I -> L_1^-1, M[p][q] = L_1^-1[q][p] / sqrt(d_q) (upper
//      triangular: Cholesky QR).
// Workgroup 0 leaves the step's status words in the slot's check block: 1 = a pivot not safely
// positive (NOTHING is written to `out` then: the step is not accepted), 2 = pivots spread by more
// than 4, else 0. n <= kSeriesMax (the callers' tails have at most 48 columns: rank + 16 <= 64).
// Launch: rmult_chol_threads(n) threads, rmult_chol_lds(n) bytes of dynamic LDS, one workgroup per
// kRmultRows rows (the workgroup's rows of Z wait in LDS for M).
// flag != nullptr: see the end of the kernel (tools/waitvalue_bench.hip: the producing stream goes on
// 2-4 us after the launch instead of 7.7 us after launch + event marker, the waiting stream starts
// 3-4 us after the data is there instead of 11.8).
// (Measured and not kept, round 4: the factorisation by ONE wave in registers — lane j holds column
// j, a pivot is a v_readlane per remaining row, fully unrolled for constant register indices: 59 KB
// of straight-line code that runs once per launch out of a cold instruction cache, 22.7 us per
// launch: profiles/r04z_cfg5_tail_on_second_stream.txt.)
constexpr int kSeriesMax = 48;
constexpr double kSeriesTol = 0.03;
// (workgroup size by the number of columns, measured with tools/rmult_chol_bench.hip at n = 21: the launch
// chain copy + Gram + this kernel 23.2 us with 256 threads, 19.1 with 512, 20.5 with 1024 — the stages of the
// series are (n-1)^2 elements and a barrier each, a barrier of 16 waves costs ~0.4 us; the elimination of
// many columns wants the threads)
__host__ __device__ inline int rmult_chol_threads(int n) { return n <= 32 ? 512 : 1024; }
constexpr int kRmultRows = 64;  // rows of Z per workgroup
__host__ __device__ inline size_t rmult_chol_lds(int n) {
  return sizeof(double) * (6 * (size_t)n * n + (size_t)kRmultRows * n + 96);
}
#ifndef PPALS_RC_STAMP  // (tools/rmult_chol_bench.hip defines it: phase stamps of workgroup 0)
#define PPALS_RC_STAMP(k)
#endif
// sum_k A[i + nn k] B[k + nn j] for symmetric nn x nn matrices in LDS (polynomials of one matrix: they
// commute). The loads of eight terms are in flight together: a term at a time costs an LDS round trip each.
__device__ __forceinline__ double lds_symm_dot(const double *A, const double *B, int nn, int i, int j) {
  const double *ap = A + i, *bp = B + nn * j;
  double a0 = 0, a1 = 0;
  int k = 0;
  for (; k + 8 <= nn; k += 8) {
    double x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x[u] = ap[nn * (k + u)];
      y[u] = bp[k + u];
    }
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      a0 += x[u] * y[u];
      a1 += x[u + 1] * y[u + 1];
    }
  }
  // (a branch-free last batch — clamped addresses, zero factors — was measured slower: 4.2 vs 3.6 us for the
  // four stages at n = 21, tools/rmult_chol_bench.hip)
  for (; k < nn; k++) a0 += ap[nn * k] * bp[k];
  return a0 + a1;
}
__global__ __launch_bounds__(1024) void k_rmult_chol(const double *__restrict__ Z, int64_t rows, int n, int m,
                                                     const double *__restrict__ C1,
                                                     double *__restrict__ out, int *__restrict__ status,
                                                     const double *__restrict__ chk_src,
                                                     double *__restrict__ chk_out,
                                                     unsigned *__restrict__ done_count,
                                                     unsigned long long *__restrict__ flag,
                                                     unsigned long long seq) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  constexpr int RB = kRmultRows;
  const int64_t row0 = (int64_t)blockIdx.x * RB;
  const int nrow = (int)min((int64_t)RB, rows - row0);
  const bool small = n - m >= 4;
  double *sc = lds + 6 * n * n;  // [0..16] reductions | [17..19] | t[64] from 32
  double *Zs = sc + 96;                                   // RB x n: this workgroup's rows of Z
  // this workgroup's rows of Z: requested now (<= 8 per thread), parked in LDS once M is there
  double zreg[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int e = tid + u * nthr;
    zreg[u] = e < RB * n ? Z[min(row0 + (e % RB), rows - 1) + rows * (e / RB)] : 0.0;
  }
  const double *Mf = nullptr;  // what the rows are multiplied by, column-major in LDS
  bool bad = false;
  int st = 0, route = 0;
  PPALS_RC_STAMP(0);
  if (small) {
    const int nn = n - m;
    double *Cs = lds, *E = Cs + n * n, *E2 = E + nn * nn, *E3 = E2 + nn * nn, *T1 = E3 + nn * nn,
           *T2 = T1 + nn * nn;
    for (int e = tid; e < n * n; e += nthr) {
      const int i = e % n, j = e / n;
      Cs[e] = 0.5 * (C1[i + n * j] + C1[j + n * i]);
    }
    lds_barrier();
    PPALS_RC_STAMP(1);
    const double c00 = m ? Cs[0] : 1.0;
    const double rc00 = 1.0 / c00;
    double e2 = 0;
    for (int e = tid; e < nn * nn; e += nthr) {
      const int a2 = e % nn, b2 = e / nn;
      double sab = Cs[(a2 + m) + n * (b2 + m)];
      if (m) sab -= Cs[n * (a2 + m)] * Cs[n * (b2 + m)] * rc00;
      const double v = (a2 == b2 ? 1.0 : 0.0) - sab;
      E[e] = v;
      e2 += v * v;
    }
    if (m && tid < nn) sc[32 + tid] = Cs[n * (tid + m)] * rc00;  // t = C1[0, 1:] / c00
    e2 = block_sum_lds(e2, sc);  // (its barrier also publishes E and t)
    PPALS_RC_STAMP(2);
    if (e2 <= kSeriesTol * kSeriesTol && c00 > 0.25 && c00 < 4.0) {  // (uniform)
      route = 3;
      constexpr double c0 = 1.0, c1 = 1.0 / 2, c2 = 3.0 / 8, c3 = 5.0 / 16, c4 = 35.0 / 128, c5 = 63.0 / 256,
                       c6 = 231.0 / 1024, c7 = 429.0 / 2048, c8 = 6435.0 / 32768, c9 = 12155.0 / 65536;
      // S^-1/2 = A0 + E^3 (A1 + E^3 (A2 + c9 E^3)), A_i = c_3i I + c_3i+1 E + c_3i+2 E^2: four stages
      for (int e = tid; e < nn * nn; e += nthr) E2[e] = lds_symm_dot(E, E, nn, e % nn, e / nn);
      lds_barrier();
      for (int e = tid; e < nn * nn; e += nthr) {
        const double e3 = lds_symm_dot(E2, E, nn, e % nn, e / nn);
        E3[e] = e3;
        T1[e] = ((e % nn == e / nn) ? c6 : 0.0) + c7 * E[e] + c8 * E2[e] + c9 * e3;
      }
      lds_barrier();
      for (int e = tid; e < nn * nn; e += nthr)
        T2[e] = ((e % nn == e / nn) ? c3 : 0.0) + c4 * E[e] + c5 * E2[e] + lds_symm_dot(E3, T1, nn, e % nn, e / nn);
      lds_barrier();
      for (int e = tid; e < nn * nn; e += nthr)  // (T1 is free again: the last stage read T2)
        T1[e] = ((e % nn == e / nn) ? c0 : 0.0) + c1 * E[e] + c2 * E2[e] + lds_symm_dot(E3, T2, nn, e % nn, e / nn);
      Mf = T1;  // = S^-1/2 (nn x nn); the barrier that parks the rows of Z below publishes it
      if (m && tid == 0) sc[20] = 1.0 / sqrt(c00);
    }
  }
  if (!Mf) {
    const int w2 = 2 * n;
    double *E0 = lds, *E1 = E0 + n * w2;  // row-major n x 2n: [C | I] being eliminated
    lds_barrier();
    for (int e = tid; e < n * w2; e += nthr) {
      const int i = e / w2, j = e - i * w2;
      E0[e] = j < n ? 0.5 * (C1[i + n * j] + C1[j + n * i]) : ((j - n == i) ? 1.0 : 0.0);
    }
    lds_barrier();
    if (tid == 0) {
      double dmax = 0;
      for (int k = 0; k < n; k++) dmax = fmax(dmax, E0[k * w2 + k]);
      sc[17] = dmax;
      sc[18] = dmax;
      sc[19] = 0.0;
    }
    lds_barrier();
    double *src = E0, *dst = E1;
    for (int k = 0; k < n; k += 2) {
      const double d = src[k * w2 + k];
      if (!(d > 1e-14 * sc[17])) {  // (uniform)
        if (tid == 0) sc[19] = 1.0;
        break;
      }
      const double dinv = 1.0 / d;
      const bool two = k + 1 < n;
      double d1 = d, d1inv = 0.0, lk1 = 0.0;
      bool bad1 = false;
      if (two) {
        lk1 = src[(k + 1) * w2 + k] * dinv;
        d1 = src[(k + 1) * w2 + (k + 1)] - lk1 * src[k * w2 + (k + 1)];
        bad1 = !(d1 > 1e-14 * sc[17]);
        d1inv = bad1 ? 0.0 : 1.0 / d1;
      }
      if (tid == 0) sc[18] = fmin(sc[18], (two && !bad1) ? fmin(d, d1) : d);
      for (int e = tid; e < n * w2; e += nthr) {
        const int i = e / w2, j = e - i * w2;
        double v = src[e];
        const double akj = src[k * w2 + j];
        if (i > k) v -= (src[i * w2 + k] * dinv) * akj;
        if (two && !bad1 && i > k + 1) {
          const double aik1 = src[i * w2 + (k + 1)] - (src[i * w2 + k] * dinv) * src[k * w2 + (k + 1)];
          const double ak1j = src[(k + 1) * w2 + j] - lk1 * akj;
          v -= (aik1 * d1inv) * ak1j;
        }
        dst[e] = v;
      }
      lds_barrier();
      double *t0 = src;
      src = dst;
      dst = t0;
      if (two && bad1) {
        if (tid == 0) sc[19] = 1.0;
        break;
      }
    }
    lds_barrier();
    bad = sc[19] != 0.0;
    st = bad ? 1 : (sc[17] > 4.0 * sc[18] ? 2 : 0);
    if (!bad) {
      // src = [D L_1^T | L_1^-1]; M[p][q] = L_1^-1[q][p] / sqrt(d_q) for p <= q -> dst[p + n q]
      for (int e = tid; e < n * n; e += nthr) {
        const int p2 = e % n, q = e / n;
        dst[e] = (p2 <= q) ? src[q * w2 + n + p2] / sqrt(src[q * w2 + q]) : 0.0;
      }
      Mf = dst;
    }
  }
  PPALS_RC_STAMP(3);
  if (blockIdx.x == 0) {
    if (tid == 0) {
      status[0] = st;
      status[1] = route;
      union {
        int w[2];
        double d;
      } u;
      u.w[0] = st;
      u.w[1] = route;
      chk_out[kEigOffStatus] = u.d;
    } else if (tid < 4) {
      chk_out[kEigOffStatus + tid] = chk_src[kEigOffStatus + tid];
    }
  }
  if (!bad) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = tid + u * nthr;
      if (e < RB * n) Zs[e] = zreg[u];
    }
    lds_barrier();
    const int r = tid % RB, part = tid / RB, nparts = nthr / RB;
    if (r < nrow) {
      const double *zr = Zs + r;
      if (route == 3) {
        // B = [z_0 c00^-1/2 | (Z' - z_0 t) S^-1/2]: q_D stays the first column, the others are cleared of it
        const int nn = n - m;
        const double z0 = m ? zr[0] : 0.0;
        for (int q = part; q < n; q += nparts) {
          double a;
          if (m && q == 0) {
            a = z0 * sc[20];
          } else {
            const double *mq = Mf + nn * (q - m);
            double a0 = 0, a1 = 0;
            int i = 0;
            for (; i + 4 <= nn; i += 4) {
              double x[4], y[4], t[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                x[u] = zr[RB * (i + u + m)];
                y[u] = mq[i + u];
                t[u] = m ? sc[32 + i + u] : 0.0;
              }
#pragma unroll
              for (int u = 0; u < 4; u += 2) {
                a0 += (x[u] - t[u] * z0) * y[u];
                a1 += (x[u + 1] - t[u + 1] * z0) * y[u + 1];
              }
            }
            for (; i < nn; i++) a0 += (zr[RB * (i + m)] - (m ? sc[32 + i] : 0.0) * z0) * mq[i];
            a = a0 + a1;
          }
          out[row0 + r + rows * q] = a;
        }
      } else {
        for (int q = part; q < n; q += nparts) {  // (M upper triangular)
          const double *mq = Mf + n * q;
          double a0 = 0, a1 = 0;
          int p2 = 0;
          for (; p2 + 8 <= q + 1; p2 += 8) {
            double x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              x[u] = zr[RB * (p2 + u)];
              y[u] = mq[p2 + u];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
              a0 += x[u] * y[u];
              a1 += x[u + 1] * y[u + 1];
            }
          }
          for (; p2 <= q; p2++) a0 += zr[RB * p2] * mq[p2];
          out[row0 + r + rows * q] = a0 + a1;
        }
      }
    }
  }
  PPALS_RC_STAMP(4);
  // Hand-over to the second stream without a packet on this one: the workgroup that finishes last
  // publishes the launch's sequence number in `flag`; the second stream waits for that value
  // (hipStreamWaitValue64). Every workgroup releases its rows first (agent scope: its XCD's L2).
  if (flag) {
    // (every wave: its own stores are released at agent scope before the workgroup counts itself done)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gridDim.x - 1) {
        __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  PPALS_RC_STAMP(5);
}

// k_chol_m (ONE workgroup, cols x cols work only): the matrix M (cols x cols) with
//   B = [Q_D | Z[:, m:]] * M  =  orthonormal basis of [Q_D | (I - Q_D Q_D^T) Z[:, m:]]   (Cholesky QR),
// from C = Z^T Z and t = Q_D^T Z (k_tn_two). With C' = C[m:, m:] - t^T t (the Gram of the deflated
// block; Q_D is orthonormal) = L L^T and Rinv = L^-T:  M = [[I_m, -t[:, m:] Rinv], [0, Rinv]].
// Rinv comes from ONE elimination of the augmented matrix [C' | I] (row operations, all 2 n^2
// elements of a step in parallel, ping-pong buffers: one barrier per pivot): C' -> D L_1^T, I ->
// L_1^-1 (unit lower), Rinv[p][q] = L_1^-1[q][p] / sqrt(d_q). The thin product itself is a launch
// of the MFMA GEMM. Also copies Q_D into the first m columns of Z. *status: 1 = a pivot is not
// safely positive (numerically rank deficient), 2 = pivots spread by more than 4 (one Cholesky-QR
// pass leaves an orthogonality error of eps * cond), else 0.
// dynamic LDS: 2 buffers of n x 2n | T[max(m,1) x cols] | sc[8]
// Everything the threads exchange goes through LDS, so the barriers are lds_barrier (no wait for the
// copy of Q_D into Z). 17.7 us at n = 20: twenty dependent pivots of a barrier (~0.4 us with 16 waves)
// plus the update; a 256-thread workgroup (cheaper barriers, 3 elements per thread) was measured
// slower, 22.3 us (tools/runs/r03_ai.sh).
__global__ __launch_bounds__(1024) void k_chol_m(double *__restrict__ Z, const double *__restrict__ QD,
                                                 int64_t J, int cols, int m,
                                                 const double *__restrict__ C, double *__restrict__ M,
                                                 int *__restrict__ status) {
  extern __shared__ double lds[];
  const int n = cols - m, w2 = 2 * n;
  double *E0 = lds, *E1 = E0 + n * w2;   // row-major n x 2n: [C' | I] being eliminated
  double *T = E1 + n * w2;
  double *sc = T + (m > 0 ? m : 1) * cols;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int64_t e = tid; e < J * m; e += nthr) Z[e] = QD[e];
  const double *tg = C + cols * cols;
  for (int e = tid; e < m * cols; e += nthr) T[e] = tg[e];
  lds_barrier();
  for (int e = tid; e < n * w2; e += nthr) {
    const int i = e / w2, j = e - i * w2;
    double v;
    if (j < n) {
      v = C[(m + i) + cols * (m + j)];
      for (int d = 0; d < m; d++) v -= T[d + m * (m + i)] * T[d + m * (m + j)];
    } else {
      v = (j - n == i) ? 1.0 : 0.0;
    }
    E0[e] = v;
  }
  lds_barrier();
  if (tid == 0) {
    double dmax = 0;
    for (int k = 0; k < n; k++) dmax = fmax(dmax, E0[k * w2 + k]);
    sc[0] = dmax;
    sc[1] = dmax;
    sc[2] = 0.0;
  }
  lds_barrier();
  double *src = E0, *dst = E1;
  // TWO pivots per barrier: every element applies the row operations of pivots k and k + 1 from the
  // same source buffer, in the order and with the roundings of one pivot after the other (the
  // second pivot row and its multipliers are re-derived per thread from rows k and k + 1) — half
  // the barriers of a launch that is twenty dependent barrier + update rounds at n = 20.
  for (int k = 0; k < n; k += 2) {
    const double d = src[k * w2 + k];
    if (!(d > 1e-14 * sc[0])) {  // (uniform)
      if (tid == 0) sc[2] = 1.0;
      break;
    }
    const double dinv = 1.0 / d;
    const bool two = k + 1 < n;
    double d1 = d, d1inv = 0.0, lk1 = 0.0;
    bool bad1 = false;
    if (two) {
      lk1 = src[(k + 1) * w2 + k] * dinv;                          // multiplier of row k+1, pivot k
      d1 = src[(k + 1) * w2 + (k + 1)] - lk1 * src[k * w2 + (k + 1)];  // second pivot, after step k
      bad1 = !(d1 > 1e-14 * sc[0]);
      d1inv = bad1 ? 0.0 : 1.0 / d1;
    }
    if (tid == 0) sc[1] = fmin(sc[1], (two && !bad1) ? fmin(d, d1) : d);
    for (int e = tid; e < n * w2; e += nthr) {
      const int i = e / w2, j = e - i * w2;
      double v = src[e];
      const double akj = src[k * w2 + j];
      if (i > k) v -= (src[i * w2 + k] * dinv) * akj;  // pivot k
      if (two && !bad1 && i > k + 1) {
        // pivot k+1 on the matrix after step k: a'_{i,k+1} = a_{i,k+1} - l_ik a_{k,k+1},
        // a'_{k+1,j} = a_{k+1,j} - l_{k+1,k} a_kj
        const double aik1 = src[i * w2 + (k + 1)] - (src[i * w2 + k] * dinv) * src[k * w2 + (k + 1)];
        const double ak1j = src[(k + 1) * w2 + j] - lk1 * akj;
        v -= (aik1 * d1inv) * ak1j;
      }
      dst[e] = v;
    }
    lds_barrier();
    double *t0 = src;
    src = dst;
    dst = t0;
    if (two && bad1) {  // (uniform: the second pivot of the pair is not safely positive)
      if (tid == 0) sc[2] = 1.0;
      break;
    }
  }
  lds_barrier();
  const bool bad = sc[2] != 0.0;
  if (tid == 0) *status = bad ? 1 : (sc[0] > 4.0 * sc[1] ? 2 : 0);
  if (bad) return;
  // src = [D L_1^T | L_1^-1]; Rinv[p][q] = L_1^-1[q][p] / sqrt(d_q)  (upper triangular)
  for (int e = tid; e < cols * cols; e += nthr) {
    const int r = e % cols, c = e / cols;  // M[r + cols c]
    double v;
    if (c < m) {
      v = (r == c) ? 1.0 : 0.0;
    } else {
      const int q = c - m;
      const double isq = 1.0 / sqrt(src[q * w2 + q]);
      if (r >= m) {
        const int p = r - m;
        v = (p <= q) ? src[q * w2 + n + p] * isq : 0.0;
      } else {  // -(t[:, m:] Rinv)[r][q] = -sum_{p <= q} t[r][m + p] Rinv[p][q]
        double a = 0;
        for (int p = 0; p <= q; p++) a += T[r + m * (m + p)] * src[q * w2 + n + p];
        v = -a * isq;
      }
    }
    M[e] = v;
  }
}

// k_rr_apply: every workgroup loads H = B^T G B (cols x cols), diagonalises it (block Jacobi in
// LDS), ranks the eigenvalues descending and forms, for its own `rows_per` rows, U = B Y[:, :rank]
// and its share of || G U - U diag(ev) ||_F^2 (-> resp[blockIdx.x]; the host adds them up).
// Workgroup 0 also writes the eigenvalues (evW[0..cols)) and adds up the check sums the sign
// iteration's last two products left (np each): chk[0] = ||X_prev^2 - I||_F^2, chk[1] = trace(X).
// dynamic LDS: top_eig_small_lds(cols) + 64 ints (ord)
__global__ __launch_bounds__(1024) void k_rr_apply(const double *__restrict__ Bm,
                                                  const double *__restrict__ GB, int64_t J, int cols,
                                                  int rank, const double *__restrict__ H, int rows_per,
                                                  const double *__restrict__ part_e2,
                                                  const double *__restrict__ part_tr, int np,
                                                  double *__restrict__ U, double *__restrict__ U2,
                                                  double *__restrict__ evW, double *__restrict__ chk,
                                                  double *__restrict__ resp) {
  extern __shared__ double lds[];
  const int ldA = cols + 1;
  double *A0 = lds, *A1 = A0 + cols * ldA, *Q0 = A1 + cols * ldA, *Q1 = Q0 + cols * ldA;
  double *rc = Q1 + cols * ldA, *rs = rc + 128, *red = rs + 128;
  int *partner = (int *)(red + 17);
  int *ord = partner + 128;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int e = tid; e < cols * cols; e += nthr) {  // symmetrised on the way in
    const int i = e % cols, j = e / cols;
    A0[i * ldA + j] = 0.5 * (H[i + cols * j] + H[j + cols * i]);
  }
  __syncthreads();
  double *A, *Q;
  jacobi_eig_block(A0, A1, Q0, Q1, cols, rc, rs, partner, red, &A, &Q);
  if (tid < cols) {
    const double wk = eig_rank_key(A[tid * ldA + tid]);
    int pos = 0;
    for (int j = 0; j < cols; j++) {
      const double wj = eig_rank_key(A[j * ldA + j]);
      if (wj > wk || (wj == wk && j < tid)) pos++;
    }
    ord[pos] = tid;
    if (blockIdx.x == 0) evW[pos] = wk;
  }
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int nrow = (int)max((int64_t)0, min(J, r0 + rows_per) - r0);
  double res = 0;
  for (int e = tid; e < nrow * rank; e += nthr) {
    const int64_t i = r0 + e % nrow;
    const int k = e / nrow;
    const int col = ord[k];
    double u = 0, gu = 0;
    for (int p = 0; p < cols; p++) {
      const double y = Q[p * ldA + col];
      u += Bm[i + J * p] * y;
      gu += GB[i + J * p] * y;
    }
    U[i + J * k] = u;
    if (U2) U2[i + J * k] = u;
    const double d = gu - u * A[col * ldA + col];
    res += d * d;
  }
  res = block_sum(res, red);
  if (tid == 0) resp[blockIdx.x] = res;
  if (blockIdx.x == 0) {
    for (int b = gridDim.x + tid; b < 64; b += nthr) resp[b] = 0.0;  // (the host adds all 64)
    double e2 = 0, tr = 0;
    for (int i = tid; i < np; i += nthr) {
      e2 += part_e2[i];
      tr += part_tr[i];
    }
    e2 = block_sum(e2, red);
    tr = block_sum(tr, red);
    if (tid == 0) {
      chk[0] = e2;
      chk[1] = tr;
    }
  }
}

// ---- the tail WITHOUT the eigen-decomposition on the critical path (plain HOOI sweeps) ----------
// What the sweep needs from an eigen-step is the invariant SUBSPACE: every later contraction, the
// core's norm and the convergence measure depend on W_i W_i^T only. So the step hands out the
// orthonormal basis B itself and checks it as a subspace, ||G B - B H||_F with H = B^T G B, while
// the Jacobi of H (70 us in one workgroup) runs on a second stream beside the next mode's tensor
// scans; its rotation Y is applied when somebody asks for the factors (eigenvectors one by one,
// sorted, as the reference returns them) and its eigenvalues schedule the slot's next step.
//
// k_sub_residual: every workgroup loads H and forms, for its rows, r = G B - B H (-> resp[block])
// and copies B into U (and U2). Workgroup 0 adds up the sign iteration's check sums and leaves
// Gershgorin bounds of H's spectrum: chk[5] <= lambda_min(H), chk[6] >= lambda_max(H).
__global__ __launch_bounds__(256) void k_sub_residual(const double *__restrict__ Bm,
                                                      const double *__restrict__ GB, int64_t J, int cols,
                                                      const double *__restrict__ H, int rows_per,
                                                      const double *__restrict__ part_e2,
                                                      const double *__restrict__ part_tr, int np,
                                                      double *__restrict__ U, double *__restrict__ U2,
                                                      double *__restrict__ chk,
                                                      double *__restrict__ resp,
                                                      double *__restrict__ host = nullptr) {
  // host != nullptr (deferred acceptance): everything the step's checks read — the sums, the
  // Gershgorin bounds, the status words of the Cholesky kernel, the residual shares — is ALSO
  // written straight into that pinned block of the slot (same layout as chk: no copy launch)
  extern __shared__ double sH[];  // cols x cols | red[17] | lo[64] | hi[64]
  double *red = sH + cols * cols;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int e = tid; e < cols * cols; e += nthr) {
    const int i = e % cols, j = e / cols;
    sH[e] = 0.5 * (H[i + cols * j] + H[j + cols * i]);
  }
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int nrow = (int)max((int64_t)0, min(J, r0 + rows_per) - r0);
  double res = 0;
  for (int e = tid; e < nrow * cols; e += nthr) {
    const int64_t i = r0 + e % nrow;
    const int k = e / nrow;
    double bh = 0;
    for (int p = 0; p < cols; p++) bh += Bm[i + J * p] * sH[p + cols * k];
    const double b = Bm[i + J * k];
    if (U) U[i + J * k] = b;
    if (U2) U2[i + J * k] = b;
    const double d = GB[i + J * k] - bh;
    res += d * d;
  }
  res = block_sum(res, red);
  double *hresp = host ? host + (resp - chk) : nullptr;
  if (tid == 0) {
    resp[blockIdx.x] = res;
    if (hresp) hresp[blockIdx.x] = res;
  }
  if (blockIdx.x == 0) {
    for (int b = gridDim.x + tid; b < 64; b += nthr) {
      resp[b] = 0.0;
      if (hresp) hresp[b] = 0.0;
    }
    if (host && tid < 4) host[kEigOffStatus + tid] = chk[kEigOffStatus + tid];  // status words (8 ints)
    double e2 = 0, tr = 0;
    for (int i = tid; i < np; i += nthr) {
      e2 += part_e2[i];
      tr += part_tr[i];
    }
    e2 = block_sum(e2, red);
    tr = block_sum(tr, red);
    if (np < 0) {  // (the producer of H left the sums in place: k_tn_small with chk_out)
      e2 = chk[0];
      tr = chk[1];
    }
    double *lo = red + 17, *hi = lo + 64;
    if (tid < cols) {
      double off = 0;
      for (int j = 0; j < cols; j++)
        if (j != tid) off += fabs(sH[tid + cols * j]);
      lo[tid] = sH[tid + cols * tid] - off;
      hi[tid] = sH[tid + cols * tid] + off;
    }
    __syncthreads();
    if (tid == 0) {
      double l = lo[0], h = hi[0];
      for (int i = 1; i < cols; i++) {
        l = fmin(l, lo[i]);
        h = fmax(h, hi[i]);
      }
      chk[0] = e2;
      chk[1] = tr;
      chk[5] = l;
      chk[6] = h;
      if (host) {
        host[0] = e2;
        host[1] = tr;
        host[5] = l;
        host[6] = h;
        host[8] = chk[8];
        host[9] = chk[9];
      }
    }
  }
}
// k_rr_small (ONE workgroup, second stream): eigen-decomposition of H; Y (cols x cols, column k =
// the eigenvector of the k-th largest eigenvalue) and the eigenvalues, descending.
// dynamic LDS: top_eig_small_lds(cols) + 64 ints
__global__ __launch_bounds__(1024) void k_rr_small(const double *__restrict__ H, int cols,
                                                   double *__restrict__ Y, double *__restrict__ ev,
                                                   double *__restrict__ ev_host) {
  extern __shared__ double lds[];
  const int ldA = cols + 1;
  double *A0 = lds, *A1 = A0 + cols * ldA, *Q0 = A1 + cols * ldA, *Q1 = Q0 + cols * ldA;
  double *rc = Q1 + cols * ldA, *rs = rc + 128, *red = rs + 128;
  int *partner = (int *)(red + 17);
  int *ord = partner + 128;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int e = tid; e < cols * cols; e += nthr) {
    const int i = e % cols, j = e / cols;
    A0[i * ldA + j] = 0.5 * (H[i + cols * j] + H[j + cols * i]);
  }
  __syncthreads();
  double *A, *Q;
  jacobi_eig_block(A0, A1, Q0, Q1, cols, rc, rs, partner, red, &A, &Q);
  if (tid < cols) {
    const double wk = eig_rank_key(A[tid * ldA + tid]);
    int pos = 0;
    for (int j = 0; j < cols; j++) {
      const double wj = eig_rank_key(A[j * ldA + j]);
      if (wj > wk || (wj == wk && j < tid)) pos++;
    }
    ord[pos] = tid;
    ev[pos] = wk;
    if (ev_host) ev_host[pos] = wk;  // (pinned host memory: the host reads it after the event)
  }
  __syncthreads();
  for (int e = tid; e < cols * cols; e += nthr) {
    const int p = e % cols, k = e / cols;
    Y[p + cols * k] = Q[p * ldA + ord[k]];
  }
}

// Eigen-decomposition of a symmetric positive semi-definite H of 64 < n <= kJacobiBigMax rows (the
// Rayleigh-Ritz matrix of a Tucker mode whose core rank exceeds 64: the reference's own data sets
// run ranks 70 and 100, test_ALS.cxx:366-379) by ONE workgroup: one-sided (Hestenes) Jacobi. The
// two-sided block Jacobi of the small modes keeps A and Q twice in LDS (4 n^2 doubles: 64 rows at
// most); here only W = H V lives in LDS (n (n + 1) doubles, 129 KB at n = 128) and V — the product
// of the rotations, L2-resident — in global memory: a round rotates n / 2 disjoint column pairs of
// W (and V) so that the columns of W become orthogonal; at convergence W = V Lambda: the columns
// of V are the eigenvectors, the column norms of W the eigenvalues. Sixteen threads per pair (their
// three inner products meet by lane shuffles inside the 16-lane group), one barrier per round,
// round-robin pairing (circle method), <= 40 sweeps; a sweep without a rotation ends it.
//   Y (n x n) = eigenvectors, column k for the k-th largest eigenvalue; ev / ev_host: eigenvalues.
// dynamic LDS: n (n + 1) doubles + 256 doubles + 256 ints.
constexpr int kJacobiBigMax = 128;
__global__ __launch_bounds__(1024) void k_jacobi_onesided(const double *__restrict__ H, int n,
                                                          double *__restrict__ V, double *__restrict__ Y,
                                                          double *__restrict__ ev,
                                                          double *__restrict__ ev_host,
                                                          const int *__restrict__ gate = nullptr) {
  // (gate: the launch is a conditional one — it does nothing unless *gate is set; the R x R normal
  // equations' eigen-route, wanted only when the elimination met a non-positive pivot)
  if (gate && *gate == 0) return;
  extern __shared__ double lds[];
  const int ld = n + 1;
  double *W = lds;
  double *nrm = W + (size_t)n * ld;  // 128 column norms
  int *flag = (int *)(nrm + 256);    // [0]: rotations in this sweep
  int *ord = flag + 8;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int e = tid; e < n * n; e += nthr) {
    const int i = e % n, j = e / n;
    W[j * ld + i] = 0.5 * (H[i + n * j] + H[j + n * i]);  // column j
    V[i + n * j] = i == j ? 1.0 : 0.0;
  }
  if (tid == 0) flag[0] = 0;
  __syncthreads();
  const int ne = (n + 1) & ~1;  // players of the tournament (an odd n plays with a bye)
  const int grp = tid >> 4, gl = tid & 15;  // 64 groups of 16 threads
  for (int sweep = 0; sweep < 40; sweep++) {
    for (int r = 0; r < ne - 1; r++) {
      // pair of this group in round r (circle method): player ne - 1 stays, the others rotate
      int p = -1, q = -1;
      if (grp < ne / 2) {
        if (grp == 0) {
          p = ne - 1;
          q = r;
        } else {
          p = (r + grp) % (ne - 1);
          q = (r - grp + (ne - 1)) % (ne - 1);
        }
        if (p > q) {
          const int t = p;
          p = q;
          q = t;
        }
        if (q >= n) p = -1;  // the bye
      }
      if (p >= 0) {
        double *wp = W + p * ld, *wq = W + q * ld;
        double a = 0, b = 0, c = 0;
        for (int i = gl; i < n; i += 16) {
          const double x = wp[i], y = wq[i];
          a += x * x;
          b += y * y;
          c += x * y;
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
          a += __shfl_xor(a, o, 16);
          b += __shfl_xor(b, o, 16);
          c += __shfl_xor(c, o, 16);
        }
        if (fabs(c) > 1e-15 * sqrt(a * b) && a * b > 0.0) {
          const double zeta = (b - a) / (2.0 * c);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
          for (int i = gl; i < n; i += 16) {
            const double x = wp[i], y = wq[i];
            wp[i] = cs * x - sn * y;
            wq[i] = sn * x + cs * y;
            const double vx = V[i + n * p], vy = V[i + n * q];
            V[i + n * p] = cs * vx - sn * vy;
            V[i + n * q] = sn * vx + cs * vy;
          }
          if (gl == 0) flag[0] = 1;  // (benign race: everybody writes 1)
        }
      }
      __syncthreads();
    }
    const int rotated = flag[0];
    __syncthreads();
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    if (!rotated) break;
  }
  // eigenvalues = column norms of W (H is positive semi-definite), ranked descending
  for (int k = tid >> 4; k < n; k += nthr >> 4) {
    double a = 0;
    for (int i = gl; i < n; i += 16) a += W[k * ld + i] * W[k * ld + i];
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) a += __shfl_xor(a, o, 16);
    if (gl == 0) nrm[k] = sqrt(a);
  }
  __syncthreads();
  if (tid < n) {
    const double wk = eig_rank_key(nrm[tid]);
    int pos = 0;
    for (int j = 0; j < n; j++) {
      const double wj = eig_rank_key(nrm[j]);
      if (wj > wk || (wj == wk && j < tid)) pos++;
    }
    ord[pos] = tid;
    ev[pos] = wk;
    if (ev_host) ev_host[pos] = wk;
  }
  __syncthreads();
  __threadfence_block();
  for (int e = tid; e < n * n; e += nthr) {
    const int i = e % n, k = e / n;
    Y[i + n * k] = V[i + n * ord[k]];
  }
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

}  // namespace ppals
