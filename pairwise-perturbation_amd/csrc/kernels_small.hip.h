// kernels_small.hip.h — everything that is not a scan of the s^N tensor: tensor generation /
// residual (untimed in the reference), contraction of cached intermediates (K3/K9), the R x R
// normal-equation side (K4-K7) and the PP bookkeeping. All fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_scan.hip.h"

namespace ppals {

typedef double f64x4 __attribute__((ext_vector_type(4)));  // (also in kernels_scan.hip.h)


// ------------------------------------------------------------------ RNG (bit-identical to the
// host generator in ppals_api.cpp and to oracle/ppals_oracle.cpp)
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ inline double u01(uint64_t seed, uint64_t idx) {
  uint64_t h = splitmix64(splitmix64(seed) ^ idx);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------ reductions
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// sum over the block; result valid in thread 0 (and broadcast through LDS to everyone)
__device__ inline double block_sum(double v, double *lds /* >= 17 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int w = 0; w < nw; w++) s += lds[w];
    lds[16] = s;
  }
  __syncthreads();
  return lds[16];
}

// A workgroup barrier for phases that hand data over through LDS only: it does not wait for the
// workgroup's outstanding GLOBAL stores, which a __syncthreads() does (its fence covers every address
// space: ~1 us after a burst of stores, measured in k_cp_mode_update with tools/update_bench.hip).
__device__ inline void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// block_sum with ONE such barrier (a barrier of 16 waves costs ~0.4 us: update_bench phase stamps):
// every thread adds the waves' partial sums itself, in the same order. `slot` (16 doubles) must not
// be written again before another barrier has been passed.
__device__ inline double block_sum_lds(double v, double *slot /* 16 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  if (lane == 0) slot[wave] = v;
  lds_barrier();
  double s = 0;
  for (int w = 0; w < nw; w++) s += slot[w];
  return s;
}

__global__ void k_sum_partials(const double *__restrict__ part, int n, double *__restrict__ out) {
  __shared__ double lds[17];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  s = block_sum(s, lds);
  if (threadIdx.x == 0) *out = s;
}

// ------------------------------------------------------------------ tensor generation
template <typename TV>
__global__ void k_fill_uniform(TV *__restrict__ V, int64_t l0, int64_t g0, int64_t row0,
                               int64_t rest, uint64_t seed, double lo, double hi) {
  const int64_t total = l0 * rest;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = e % l0, r = e / l0;
    const uint64_t gi = (uint64_t)(row0 + a) + (uint64_t)g0 * (uint64_t)r;
    V[e] = (TV)(lo + (hi - lo) * u01(seed, gi));
  }
}

// `-tensor p / p2`: Laplacian (Poisson) operator tensor, elementwise closed form of
// laplacian_tensor (common.cxx:575-642)
__host__ __device__ inline double laplacian_value(uint64_t gi, int ndigits, int s) {
  int mismatches = 0;
  double dval = 0.0;
  for (int k = 0; k < ndigits / 2; k++) {
    const int a = (int)(gi % (uint64_t)s);
    gi /= (uint64_t)s;
    const int b = (int)(gi % (uint64_t)s);
    gi /= (uint64_t)s;
    if (a != b) {
      mismatches++;
      dval = (a - b == 1 || b - a == 1) ? -1.0 : 0.0;
    }
  }
  if (mismatches == 0) return 2.0 * (ndigits / 2);
  if (mismatches == 1) return dval;
  return 0.0;
}
template <typename TV>
__global__ void k_fill_laplacian(TV *__restrict__ V, int64_t l0, int64_t g0, int64_t row0,
                                 int64_t rest, int ndigits, int s) {
  const int64_t total = l0 * rest;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = e % l0, r = e / l0;
    const uint64_t gi = (uint64_t)(row0 + a) + (uint64_t)g0 * (uint64_t)r;
    V[e] = (TV)laplacian_value(gi, ndigits, s);
  }
}
// MODE 0: V[e] += alpha*u(e);  MODE 1: partial[blk] = sum u(e)^2  (u regenerated on the fly)
template <typename TV, int MODE>
__global__ __launch_bounds__(256) void k_uniform_noise(TV *__restrict__ V, int64_t l0, int64_t g0,
                                                       int64_t row0, int64_t rest, uint64_t seed,
                                                       double lo, double hi, double alpha,
                                                       double *__restrict__ partial) {
  __shared__ double lds[17];
  const int64_t total = l0 * rest;
  double acc = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = e % l0, r = e / l0;
    const uint64_t gi = (uint64_t)(row0 + a) + (uint64_t)g0 * (uint64_t)r;
    const double u = lo + (hi - lo) * u01(seed, gi);
    if (MODE == 0)
      V[e] = (TV)((double)V[e] + alpha * u);
    else
      acc += u * u;
  }
  if (MODE == 1) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
  }
}

// rank-R outer structure  vhat[m,k] = sum_r Q[m,r] P[k,r]; block = 256 rows m, KCH columns k.
// MODE 0: V = vhat (build_V, common.cxx:135-197). MODE 1: partial[blk] = sum (V - vhat)^2
// (als_CP.cxx:183-187, nothing materialised). MODE 2: partial[blk] = sum V^2.
template <typename TV, int RB, int MODE>
__global__ __launch_bounds__(256) void k_rank_stream(TV *__restrict__ V, int64_t M, int64_t K,
                                                     const double *__restrict__ Q,
                                                     const double *__restrict__ P, int R,
                                                     int kch, double *__restrict__ partial) {
  __shared__ double lds[17];
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k0 = (int64_t)blockIdx.y * kch;
  const int64_t k1 = min(K, k0 + (int64_t)kch);
  const bool ok = m < M;
  double q[RB];
  if (MODE != 2) {
#pragma unroll
    for (int r = 0; r < RB; r++) q[r] = (ok && r < R) ? Q[m + M * r] : 0.0;
  }
  double acc = 0;
  for (int64_t k = k0; k < k1; k++) {
    double v = 0;
    if (MODE != 2) {
#pragma unroll
      for (int r = 0; r < RB; r++)
        if (r < R) v += q[r] * P[k + K * r];
    }
    if (ok) {
      if (MODE == 0)
        V[m + M * k] = (TV)v;
      else {
        double d = (double)V[m + M * k] - v;
        acc += d * d;
      }
    }
  }
  if (MODE != 0) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
  }
}

// any rank (R > 64): the row's R coefficients do not fit in registers; they are re-read from Q for
// every column (L1/L2-resident: 256 rows x R doubles per block). Untimed paths only (tensor
// generation, [diffV]).
template <typename TV, int MODE>
__global__ __launch_bounds__(256) void k_rank_stream_any(TV *__restrict__ V, int64_t M, int64_t K,
                                                         const double *__restrict__ Q,
                                                         const double *__restrict__ P, int R,
                                                         int kch, double *__restrict__ partial) {
  __shared__ double lds[17];
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k0 = (int64_t)blockIdx.y * kch;
  const int64_t k1 = min(K, k0 + (int64_t)kch);
  const bool ok = m < M;
  double acc = 0;
  for (int64_t k = k0; k < k1; k++) {
    double v = 0;
    if (ok)
      for (int r = 0; r < R; r++) v += Q[m + M * r] * P[k + K * r];
    if (ok) {
      if (MODE == 0)
        V[m + M * k] = (TV)v;
      else {
        double d = (double)V[m + M * k] - v;
        acc += d * d;
      }
    }
  }
  if (MODE != 0) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
  }
}

// dst[c + cols*r] = src[r + rows*c]: 64x64 tiles through LDS (both sides coalesced); blockIdx.y
// walks over independent rows x cols blocks stored one after another
template <typename TV>
__global__ __launch_bounds__(256) void k_transpose(const TV *__restrict__ src, int64_t rows,
                                                   int64_t cols, TV *__restrict__ dst) {
  __shared__ TV tile[64][65];
  src += (int64_t)blockIdx.y * rows * cols;
  dst += (int64_t)blockIdx.y * rows * cols;
  const int64_t tiles_r = (rows + 63) / 64;
  const int64_t r0 = (int64_t)(blockIdx.x % tiles_r) * 64, c0 = (int64_t)(blockIdx.x / tiles_r) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // ty in [0,4)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * i][tx] = src[r + rows * c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
    if (r < rows && c < cols) dst[c + cols * r] = tile[tx][ty + 4 * i];
  }
}

// the same transposition into a padded layout (Ops::pad_layout): column index c of the source is
// the fast index of the destination, stored in blocks of `blk` padded to `ld`:
//   dst[(c % blk) + ld*(c / blk + (cols / blk)*r)] = src[r + rows*c]
template <typename TV>
__global__ __launch_bounds__(256) void k_transpose_pad(const TV *__restrict__ src, int64_t rows,
                                                       int64_t cols, int64_t blk, int64_t ld,
                                                       TV *__restrict__ dst) {
  __shared__ TV tile[64][65];
  const int64_t tiles_r = (rows + 63) / 64;
  const int64_t r0 = (int64_t)(blockIdx.x % tiles_r) * 64, c0 = (int64_t)(blockIdx.x / tiles_r) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t nblkc = cols / blk;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * i][tx] = src[r + rows * c];
  }
  __syncthreads();
  const int64_t c = c0 + tx;
  const int64_t cq = c / blk, cr = c - cq * blk;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int64_t r = r0 + ty + 4 * i;
    if (r < rows && c < cols) dst[cr + ld * (cq + nblkc * r)] = tile[tx][ty + 4 * i];
  }
}

template <typename TV>
__global__ void k_unpack_shards(const char *__restrict__ stage, int64_t s0, int64_t rest,
                                int64_t blk, int P, int64_t chunk_bytes, TV *__restrict__ full) {
  const int64_t total = s0 * rest;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = e % s0, c = e / s0;
    const int p = (int)(a / blk);
    const int64_t lp = min(blk, s0 - (int64_t)p * blk);
    const TV *src = reinterpret_cast<const TV *>(stage + (int64_t)p * chunk_bytes);
    full[e] = src[(a - (int64_t)p * blk) + lp * c];
  }
}

template <typename TV>
__global__ void k_convert_rows(TV *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    dst[e] = (TV)src[e];
}

template <typename TV>
__global__ void k_widen_rows(double *__restrict__ dst, const TV *__restrict__ src, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    dst[e] = (double)src[e];
}

// ------------------------------------------------------------------ plain Khatri-Rao product
__global__ void k_krp(double *__restrict__ out, KrpArgs a, int64_t J, int col0, int ncols) {
  const int64_t total = J * ncols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e % J;
    const int c = (int)(e / J);
    double v = 1.0;
    int64_t rem = j;
    for (int f = 0; f < a.nf; f++) {
      const int64_t jf = rem % a.rows[f];
      rem /= a.rows[f];
      v *= a.ptr[f][jf + a.ld[f] * (col0 + c)];
    }
    out[e] = v;
  }
}

// ------------------------------------------------------------------ mttv (K3, K9, deeper tree nodes)
// out[l + L*t + rs*r] (+)= scale * sum_j X[l + L*(j + J*(t + T*r))] * B[j + ldb*r]
// X is a cached intermediate that already carries the rank index: fp64 (tree nodes, PP operators)
// or fp32 (the first-level intermediate of the multi-sweep schedule). `scale` (device scalar,
// may be null) is the pending Normalize factor of a cached tensor (see engine.cpp, MSDT).
__device__ inline double mttv_scale(const double *scale) { return scale ? *scale : 1.0; }

// A normal-equation system to prepare on the side: the S and S^-1 of the mode update that FOLLOWS a
// contraction depend on the other modes' Grams only, which are final before the contraction starts.
// One extra workgroup of the contraction's launch computes them (Gall == nullptr: none), so the
// update kernel — one workgroup, a chain of dependent phases — starts with them in hand instead of
// spending its first 3-8 us on the Hadamard product and the R x R inverse.
__device__ inline void gram_system_wave(const double *__restrict__ Gall, int N, int mode, int R,
                                        double lambda, double *__restrict__ S,
                                        double *__restrict__ Sinv, int force_jacobi, double *lds);
struct SysArgs {
  const double *Gall = nullptr;
  int N = 0, mode = 0;
  double lambda = 0;
  double *S = nullptr, *Sinv = nullptr;
  int force_jacobi = 0;
};
// the extra workgroup of a contraction launch: true when THIS workgroup was it (and is done)
#define PPALS_SYS_BLOCK(sys_, R_, nblk_)                                                        \
  int nblk_ = (int)gridDim.x;                                                                   \
  if ((sys_).Gall) {                                                                            \
    nblk_ -= 1;                                                                                 \
    if ((int)blockIdx.x == nblk_) {                                                             \
      extern __shared__ double sys_lds_[];                                                      \
      if (threadIdx.x < 64)                                                                     \
        gram_system_wave((sys_).Gall, (sys_).N, (sys_).mode, R_, (sys_).lambda, (sys_).S,       \
                         (sys_).Sinv, (sys_).force_jacobi, sys_lds_);                           \
      return;                                                                                   \
    }                                                                                           \
  }

// variant V (large L, L % VL == 0): one wave per (256*VL/4.. l-tile, t, r), 16-byte loads along l,
// B[j,r] is wave-uniform. This is the streaming kernel for the s^(N-1) R intermediates.
template <typename TX>
__global__ __launch_bounds__(64) void k_mttv_vec(const TX *__restrict__ X, int64_t L, int64_t J,
                                                 int64_t T, int R, const double *__restrict__ B,
                                                 int64_t ldb, double *__restrict__ out, int64_t rs,
                                                 int accumulate, const double *__restrict__ scale,
                                                 SysArgs sys = SysArgs()) {
  PPALS_SYS_BLOCK(sys, R, nblk)
  constexpr int VL = 16 / sizeof(TX);
  typedef TX vecx __attribute__((ext_vector_type(VL)));
  const int lane = threadIdx.x;
  const int64_t ltiles = (L + 64 * VL - 1) / (64 * VL);
  const int64_t total = ltiles * T * R;
  const double sc = mttv_scale(scale);
  for (int64_t blk = blockIdx.x; blk < total; blk += nblk) {
    const int64_t lt = blk % ltiles;
    const int64_t t = (blk / ltiles) % T;
    const int r = (int)(blk / (ltiles * T));
    const int64_t l = lt * 64 * VL + (int64_t)lane * VL;
    if (l >= L) continue;
    const TX *x = X + l + L * J * (t + T * (int64_t)r);
    const double *b = B + ldb * r;
    double acc[VL];
#pragma unroll
    for (int e = 0; e < VL; e++) acc[e] = 0;
    int64_t j = 0;
    for (; j + 3 < J; j += 4) {
      vecx v0 = *reinterpret_cast<const vecx *>(x + L * j);
      vecx v1 = *reinterpret_cast<const vecx *>(x + L * (j + 1));
      vecx v2 = *reinterpret_cast<const vecx *>(x + L * (j + 2));
      vecx v3 = *reinterpret_cast<const vecx *>(x + L * (j + 3));
      const double b0 = b[j], b1 = b[j + 1], b2 = b[j + 2], b3 = b[j + 3];
#pragma unroll
      for (int e = 0; e < VL; e++)
        acc[e] += ((double)v0[e] * b0 + (double)v1[e] * b1) + ((double)v2[e] * b2 + (double)v3[e] * b3);
    }
    for (; j < J; j++) {
      vecx v0 = *reinterpret_cast<const vecx *>(x + L * j);
      const double b0 = b[j];
#pragma unroll
      for (int e = 0; e < VL; e++) acc[e] += (double)v0[e] * b0;
    }
    double *o = out + l + L * t + rs * r;
#pragma unroll
    for (int e = 0; e < VL; e++) o[e] = accumulate ? (o[e] + sc * acc[e]) : sc * acc[e];
  }
}

// variant L: block = 64 consecutive l for one (t,r); the NW waves split the j range and combine
// through LDS in a fixed order, so even a 200 x 200 x R leaf contraction spreads over hundreds of
// waves (NW = 16 for the small fp64 nodes, where the j loop is the whole latency of the launch).
// jsplit > 1: a LONG reduction over few row tiles (128 x 7200 x R: 20 workgroups) is cut into jsplit
// ranges of jchunk indices, one workgroup each; the raw partial sums go to `partial`
// [split][l + L*(t + T*r)] and k_mttv_combine adds them in a fixed order (no atomics).
template <typename TX, int NW>
__global__ __launch_bounds__(64 * NW) void k_mttv_l(const TX *__restrict__ X, int64_t L, int64_t J,
                                                    int64_t T, int R, const double *__restrict__ B,
                                                    int64_t ldb, double *__restrict__ out,
                                                    int64_t rs, int accumulate,
                                                    const double *__restrict__ scale,
                                                    SysArgs sys = SysArgs(), int jsplit = 1,
                                                    int64_t jchunk = 0,
                                                    double *__restrict__ partial = nullptr) {
  PPALS_SYS_BLOCK(sys, R, nblk)
  __shared__ double part[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ltiles = (L + 63) / 64;
  const int64_t total = ltiles * T * R * jsplit;
  const double sc = mttv_scale(scale);
  for (int64_t blk0 = blockIdx.x; blk0 < total; blk0 += nblk) {
    const int split = (int)(blk0 % jsplit);
    const int64_t blk = blk0 / jsplit;
    const int64_t jbeg = jsplit > 1 ? split * jchunk : 0;
    const int64_t jend = jsplit > 1 ? min(J, jbeg + jchunk) : J;
    const int64_t lt = blk % ltiles;
    const int64_t t = (blk / ltiles) % T;
    const int r = (int)(blk / (ltiles * T));
    // fewer than 64 rows (the 25 local rows of a cfg2 shard on 8 GPUs): a wave takes P = 64 / L
    // consecutive j at a time — lanes (l, p) read L*P contiguous elements — instead of idling
    // 64 - L lanes; the P partial sums of a row meet in LDS with those of the other waves
    const int P = L < 64 ? 64 / (int)L : 1;
    const int lp = L < 64 ? lane / (int)L : 0;
    const int64_t l = L < 64 ? lane % (int)L : lt * 64 + lane;
    const TX *x = X + l + L * J * (t + T * (int64_t)r);
    const double *b = B + ldb * r;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (l < L && lp < P) {
      const int64_t st = (int64_t)NW * P;
      int64_t j = jbeg + (int64_t)wave * P + lp;
      for (; j + 3 * st < jend; j += 4 * st) {
        s0 += (double)x[L * j] * b[j];
        s1 += (double)x[L * (j + st)] * b[j + st];
        s2 += (double)x[L * (j + 2 * st)] * b[j + 2 * st];
        s3 += (double)x[L * (j + 3 * st)] * b[j + 3 * st];
      }
      for (; j < jend; j += st) s0 += (double)x[L * j] * b[j];
    }
    part[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && l < L && lp == 0) {
      double s = 0;
#pragma unroll
      for (int w = 0; w < NW; w++)
        for (int q = 0; q < P; q++) s += part[w][lane + (int)L * q];
      if (jsplit > 1) {
        partial[(int64_t)split * (L * T * R) + l + L * (t + T * (int64_t)r)] = s;
      } else {
        s *= sc;
        double *o = out + l + L * t + rs * r;
        *o = accumulate ? (*o + s) : s;
      }
    }
    __syncthreads();
  }
}
// out[lt + rs*r] (+)= scale * sum_s partial[s][lt + LT*r]   (the second pass of a j-split k_mttv_l)
__global__ __launch_bounds__(256) void k_mttv_combine(const double *__restrict__ partial, int jsplit,
                                                      int64_t LT, int R, double *__restrict__ out,
                                                      int64_t rs, int accumulate,
                                                      const double *__restrict__ scale) {
  const int64_t n = LT * R;
  const double sc = mttv_scale(scale);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double s = 0;
    for (int q = 0; q < jsplit; q++) s += partial[(int64_t)q * n + e];
    s *= sc;
    double *o = out + (e % LT) + rs * (e / LT);
    *o = accumulate ? (*o + s) : s;
  }
}
// variant S (a short leading extent and a short reduction: L < 64, L*J a few hundred elements, T*R
// large — X_c of the coil-100 extents, [3, 128 | 7200 x R]): ONE WAVE per (t, r), no workgroup barrier.
// The wave walks the contiguous L*J plane P = 64 / L reduction indices at a time (lanes (l, p) read L*P
// contiguous elements), every lane keeps the sum of its own (l, p-class), and the P classes of a row
// meet by shuffles. k_mttv_l's per-workgroup LDS combine (NW x P serial adds by L lanes, two barriers)
// was the whole cost of such a launch: 79 us for 110 MB.
template <typename TX>
__global__ __launch_bounds__(256) void k_mttv_s(const TX *__restrict__ X, int L, int64_t J, int64_t T,
                                                int R, const double *__restrict__ B, int64_t ldb,
                                                double *__restrict__ out, int64_t rs, int accumulate,
                                                const double *__restrict__ scale,
                                                SysArgs sys = SysArgs()) {
  PPALS_SYS_BLOCK(sys, R, nblk)
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)nblk * blockDim.x) >> 6;
  const int P = 64 / L;
  const int l = lane % L, p = lane / L;
  const bool on = p < P;
  const double sc = mttv_scale(scale);
  const int64_t plane = (int64_t)L * J;
  for (int64_t w = wid; w < T * R; w += nw) {
    const int64_t t = w % T;
    const int r = (int)(w / T);
    const TX *x = X + plane * (t + T * (int64_t)r) + l;
    const double *b = B + ldb * r;
    double s0 = 0, s1 = 0;
    if (on) {
      int64_t j = p;
      for (; j + P < J; j += 2 * P) {
        s0 += (double)x[(int64_t)L * j] * b[j];
        s1 += (double)x[(int64_t)L * (j + P)] * b[j + P];
      }
      if (j < J) s0 += (double)x[(int64_t)L * j] * b[j];
    }
    double s = on ? s0 + s1 : 0.0;
    // classes p, p + h, ... of one row: halving tree over p (lane distance h * L)
    for (int h = (P + 1) / 2, cnt = P; cnt > 1; cnt = h, h = (h + 1) / 2) {
      const double o = __shfl_down(s, h * L, 64);
      if (p + h < cnt && on) s += o;
    }
    if (p == 0) {
      double *o = out + l + (int64_t)L * t + rs * r;
      const double v = sc * s;
      *o = accumulate ? (*o + v) : v;
    }
  }
}
// variant 1 (L == 1): one wave per 4 consecutive t of one r, lanes stride the contiguous j (four
// independent load / FMA chains per lane: a row of J values is too little to hide a round trip).
template <typename TX>
__global__ void k_mttv_1(const TX *__restrict__ X, int64_t J, int64_t T, int R,
                         const double *__restrict__ B, int64_t ldb, double *__restrict__ out,
                         int64_t rs, int accumulate, const double *__restrict__ scale,
                         SysArgs sys = SysArgs()) {
  PPALS_SYS_BLOCK(sys, R, nblk)
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)nblk * blockDim.x) >> 6;
  const int64_t T4 = (T + 3) / 4;
  const int64_t total = T4 * R;
  const double sc = mttv_scale(scale);
  for (int64_t e = wid; e < total; e += nw) {
    const int64_t t0 = (e % T4) * 4;
    const int r = (int)(e / T4);
    const double *b = B + ldb * r;
    const TX *x0 = X + J * (t0 + T * (int64_t)r);
    const TX *x1 = x0 + J * (t0 + 1 < T ? 1 : 0), *x2 = x0 + J * (t0 + 2 < T ? 2 : 0),
             *x3 = x0 + J * (t0 + 3 < T ? 3 : 0);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t j = lane; j < J; j += 64) {
      const double bj = b[j];
      s0 += (double)x0[j] * bj;
      s1 += (double)x1[j] * bj;
      s2 += (double)x2[j] * bj;
      s3 += (double)x3[j] * bj;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    if (lane == 0) {
      const double v[4] = {s0, s1, s2, s3};
      for (int q = 0; q < 4; q++)
        if (t0 + q < T) {
          double *o = out + (t0 + q) + rs * r;
          *o = accumulate ? (*o + sc * v[q]) : sc * v[q];
        }
    }
  }
}

// PP correction of one mode in ONE launch (K9, als_CP.cxx:778-794):
//   M[x + rows*r] = M0[x + rows*r] + sum_t sum_y T_t[x,y,r] * dW_t[y + lddw_t*r]
// block = 64 consecutive x for one r, NW waves. A term whose kept mode is stored fastest is read
// with lanes along x (each wave takes every NW-th y); a term stored the other way round is read
// with lanes along y (each wave takes every NW-th x and reduces its row across the lanes). The
// partial sums meet in LDS and are added in a fixed order.
struct PPTerms {
  const double *T[MAX_ORDER];
  const double *dW[MAX_ORDER];
  int64_t ny[MAX_ORDER];
  int64_t lddw[MAX_ORDER];
  int keep_first[MAX_ORDER];
  int n;
};
// block = 16 consecutive x for one r, 256 threads. Both storage orders are read with 16 lanes along
// the contiguous index (128-byte segments) and every thread keeps 4 independent partial sums, so a
// block has ~64 loads per thread in flight instead of a dependent chain: the launch is latency.
// (Measured and rejected, tools/runs/r03_ak.sh: all terms advancing together in one loop — four
// terms' loads in flight at once, per-term state in small arrays: 9.5 -> 17.1 us per launch,
// `[PPsecond]` 102 -> 138 us.)
__global__ __launch_bounds__(256) void k_pp_correct(const double *__restrict__ M0, int64_t rows,
                                                    int R, PPTerms tm, double *__restrict__ M,
                                                    SysArgs sys) {
  extern __shared__ double sys_lds[];
  if (blockIdx.x == gridDim.x - 1 && sys.Gall) {  // (the launcher added this column of blocks)
    if (blockIdx.y == 0 && threadIdx.x < 64)
      gram_system_wave(sys.Gall, sys.N, sys.mode, R, sys.lambda, sys.S, sys.Sinv, sys.force_jacobi,
                       sys_lds);
    return;
  }
  __shared__ double part[16][17];
  __shared__ double rowsum[16];
  const int t = threadIdx.x;
  const int lo = t & 15, hi = t >> 4;  // 16 x 16
  const int64_t x0 = (int64_t)blockIdx.x * 16;
  const int r = blockIdx.y;
  double acc = 0.0;   // keep_first terms: this thread's (x = x0 + lo, y = hi, hi + 16, ...) share
  double racc = 0.0;  // the other terms: row x0 + hi, y = lo, lo + 16, ...
  for (int tt = 0; tt < tm.n; tt++) {
    const double *__restrict__ T = tm.T[tt];
    const double *__restrict__ dw = tm.dW[tt] + tm.lddw[tt] * r;
    const int64_t ny = tm.ny[tt];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (tm.keep_first[tt]) {
      const int64_t x = x0 + lo;
      if (x < rows) {
        const double *__restrict__ tp = T + x + rows * ny * (int64_t)r;
        int64_t y = hi;
        for (; y + 48 < ny; y += 64) {
          s0 += tp[rows * y] * dw[y];
          s1 += tp[rows * (y + 16)] * dw[y + 16];
          s2 += tp[rows * (y + 32)] * dw[y + 32];
          s3 += tp[rows * (y + 48)] * dw[y + 48];
        }
        for (; y < ny; y += 16) s0 += tp[rows * y] * dw[y];
      }
      acc += (s0 + s1) + (s2 + s3);
    } else {
      const int64_t x = x0 + hi;
      if (x < rows) {
        const double *__restrict__ tp = T + ny * (x + rows * (int64_t)r);
        int64_t y = lo;
        for (; y + 48 < ny; y += 64) {
          s0 += tp[y] * dw[y];
          s1 += tp[y + 16] * dw[y + 16];
          s2 += tp[y + 32] * dw[y + 32];
          s3 += tp[y + 48] * dw[y + 48];
        }
        for (; y < ny; y += 16) s0 += tp[y] * dw[y];
      }
      racc += (s0 + s1) + (s2 + s3);
    }
  }
  // row sums: reduce racc over the 16 lanes that share a row (lanes lo = 0..15 of one hi)
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) racc += __shfl_xor(racc, off, 16);
  if (lo == 0) rowsum[hi] = racc;
  part[hi][lo] = acc;
  __syncthreads();
  if (t < 16 && x0 + t < rows) {
    double s = M0[x0 + t + rows * (int64_t)r];
#pragma unroll
    for (int h = 0; h < 16; h++) s += part[h][t];
    M[x0 + t + rows * (int64_t)r] = s + rowsum[t];
  }
}

// *dst *= prod_{m in mask} scales[m]  (pending Normalize factor of a cached tensor); set_one: 1.0
struct ScaleMasks {
  unsigned m[32];
};
__global__ void k_scale_update_many(double *__restrict__ dst, const double *__restrict__ scales,
                                    ScaleMasks masks, unsigned active, unsigned fresh) {
  const int k = threadIdx.x;
  if (k < 32 && ((active >> k) & 1u)) {
    double v = ((fresh >> k) & 1u) ? 1.0 : dst[k];
    for (int m = 0; m < MAX_ORDER; m++)
      if (masks.m[k] & (1u << m)) v *= scales[m];
    dst[k] = v;
  }
}
__global__ void k_scale_update(double *__restrict__ dst, const double *__restrict__ scales,
                               unsigned mask, int set_one) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double v = set_one ? 1.0 : *dst;
    for (int m = 0; m < MAX_ORDER; m++)
      if (mask & (1u << m)) v *= scales[m];
    *dst = v;
  }
}

// ------------------------------------------------------------------ Gram  G = W^T W
// one wave per (p<=q) pair, lanes stride the rows (coalesced); symmetric fill.
__global__ void k_gram(const double *__restrict__ W, int64_t rows, int64_t ld, int R,
                       double *__restrict__ G) {
  const int lane = threadIdx.x & 63;
  const int wid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nw = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int npairs = R * (R + 1) / 2;
  for (int e = wid; e < npairs; e += nw) {
    // decode e -> (p,q), p <= q, row-major over the upper triangle
    int p = 0, rem = e;
    while (rem >= R - p) {
      rem -= R - p;
      p++;
    }
    const int q = p + rem;
    const double *a = W + ld * p, *b = W + ld * q;
    // four independent chains per lane: a long mode (7200 rows: 112 dependent loads per lane in one
    // chain, 34 us) is latency, not bandwidth
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t i = lane;
    for (; i + 192 < rows; i += 256) {
      s0 += a[i] * b[i];
      s1 += a[i + 64] * b[i + 64];
      s2 += a[i + 128] * b[i + 128];
      s3 += a[i + 192] * b[i + 192];
    }
    for (; i < rows; i += 64) s0 += a[i] * b[i];
    const double s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) {
      G[p + R * q] = s;
      G[q + R * p] = s;
    }
  }
}

// ------------------------------------------------------------------ S and S^{-1}  (K4 + K6a)
// wave-level ordering point for LDS traffic between lanes of ONE wave (LDS executes a wave's DS
// instructions in order; this only stops the compiler from moving them)
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Hadamard product of the Grams of all modes but `mode` in the reference's index order
// (als_CP.cxx:219-232: modes 0..N-1 with `mode` and N-1 swapped, first N-1 of them) + lambda*I
__device__ inline double hadamard_entry(const double *__restrict__ Gall, int N, int mode, int R,
                                        double lambda, int e) {
  double v = 1.0;
  bool first = true;
  for (int ii = 0; ii < N - 1; ii++) {
    const int j = (ii == mode) ? (N - 1) : ii;
    const double gval = Gall[(int64_t)j * R * R + e];
    v = first ? gval : v * gval;
    first = false;
  }
  if (e % R == e / R) v += lambda;
  return v;
}

// ONE wave (lane = threadIdx.x & 63). In: A (R x R, ld R+1, row-major view of symmetric S).
// Out: Sinv[i + R*j]. Parallel-ordered cyclic Jacobi S = Q diag(w) Q^T, S^{-1} = Q diag(1/w) Q^T:
// for a symmetric matrix this equals the reference's V diag(1/sigma) U^T (common.cxx:717-722),
// including its behaviour of NOT truncating tiny singular values.
// LDS: A[R][R+1], Q[R][R+1], cs[64] doubles, pq[64] ints  (R <= 64)
// NTHR cooperating threads: 64 = one wave (wave-level barriers; callable from one wave of a larger
// block), otherwise the whole block of NTHR threads (block barriers; `red` = 17 doubles of LDS for
// the convergence sums).
template <int NTHR>
__device__ inline void jacobi_eig_t(double *A, double *Q, double *cs, int *pq, int R, double *red) {
  const int lane = NTHR == 64 ? (threadIdx.x & 63) : threadIdx.x;
  auto sync = [&]() {
    if constexpr (NTHR == 64)
      wave_sync();
    else
      __syncthreads();
  };
  const int ldA = R + 1;
  for (int e = lane; e < R * R; e += NTHR) {
    const int i = e % R, j = e / R;
    Q[i * ldA + j] = (i == j) ? 1.0 : 0.0;
  }
  sync();
  const int n2 = (R + 1) & ~1;  // players (even); index R (if odd) is a bye
  const int npair = n2 / 2;
  for (int sweep = 0; sweep < 40; sweep++) {
    double off = 0, diag = 0;
    for (int e = lane; e < R * R; e += NTHR) {
      const int i = e % R, j = e / R;
      const double a = A[i * ldA + j];
      if (i == j)
        diag += a * a;
      else
        off += a * a;
    }
    if constexpr (NTHR == 64) {
      off = wave_sum(off);
      diag = wave_sum(diag);
      off = __shfl(off, 0, 64);
      diag = __shfl(diag, 0, 64);
    } else {
      off = block_sum(off, red);
      diag = block_sum(diag, red);
    }
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int rd = 0; rd < n2 - 1; rd++) {
      if (lane < npair) {  // round-robin pairing: player n2-1 fixed, the others rotate
        int a, b;
        if (lane == 0) {
          a = n2 - 1;
          b = rd % (n2 - 1);
        } else {
          a = (rd + lane) % (n2 - 1);
          b = (rd - lane + (n2 - 1)) % (n2 - 1);
        }
        int p = min(a, b), q = max(a, b);
        double c = 1.0, sn = 0.0;
        if (q < R) {
          const double apq = A[p * ldA + q];
          if (apq != 0.0) {
            const double app = A[p * ldA + p], aqq = A[q * ldA + q];
            const double theta = (aqq - app) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(t * t + 1.0);
            sn = t * c;
          }
        } else {
          p = q = -1;  // bye
        }
        cs[2 * lane] = c;
        cs[2 * lane + 1] = sn;
        pq[2 * lane] = p;
        pq[2 * lane + 1] = q;
      }
      sync();
      for (int e = lane; e < npair * R; e += NTHR) {  // A <- A J, Q <- Q J
        const int pi = e / R, i = e % R;
        const int p = pq[2 * pi], q = pq[2 * pi + 1];
        if (p >= 0) {
          const double c = cs[2 * pi], sn = cs[2 * pi + 1];
          const double aip = A[i * ldA + p], aiq = A[i * ldA + q];
          A[i * ldA + p] = c * aip - sn * aiq;
          A[i * ldA + q] = sn * aip + c * aiq;
          const double qip = Q[i * ldA + p], qiq = Q[i * ldA + q];
          Q[i * ldA + p] = c * qip - sn * qiq;
          Q[i * ldA + q] = sn * qip + c * qiq;
        }
      }
      sync();
      for (int e = lane; e < npair * R; e += NTHR) {  // A <- J^T A
        const int pi = e / R, j = e % R;
        const int p = pq[2 * pi], q = pq[2 * pi + 1];
        if (p >= 0) {
          const double c = cs[2 * pi], sn = cs[2 * pi + 1];
          const double apj = A[p * ldA + j], aqj = A[q * ldA + j];
          A[p * ldA + j] = c * apj - sn * aqj;
          A[q * ldA + j] = sn * apj + c * aqj;
        }
      }
      sync();
    }
  }
  sync();
}
// The same eigen-decomposition for a whole workgroup, built for latency: ONE barrier per round.
//  * a round's R/2 rotations are applied to A and Q in one pass — every element of A' = J^T A J
//    and Q' = Q J is written from four (two) elements of the previous iterate into a second pair of
//    buffers (jacobi_eig_t: three passes and three barriers per round, div/mod per element);
//  * the NEXT round's rotations are computed in the same pass, by the last wave of the workgroup
//    (which owns no elements): each of its lanes forms the three entries a'_pp, a'_qq, a'_pq of
//    its next pair from the old buffer with the formula above and leaves (c, s, partner) in the
//    other half of the rotation tables — so nothing waits for the angles after the barrier.
// LDS: A0, A1, Q0, Q1 [R][R+1] each | rc[2][64], rs[2][64] doubles | red[17] | partner[2][64] ints.
// blockDim.x >= 128 (a multiple of 64). Result: *Aout (eigenvalues on its diagonal), *Qout.
__device__ inline void jacobi_pair(int n2, int rd, int t, int *p, int *q) {
  int a, b;
  if (t == 0) {  // round-robin pairing: player n2-1 fixed, the others rotate
    a = n2 - 1;
    b = rd % (n2 - 1);
  } else {
    a = (rd + t) % (n2 - 1);
    b = (rd - t + (n2 - 1)) % (n2 - 1);
  }
  *p = min(a, b);
  *q = max(a, b);
}
// 1 / sqrt(x), x > 0: the hardware estimate + two Newton steps (full fp64 accuracy) — a dozen
// instructions where an IEEE division or square root is a sequence of ~30
__device__ inline double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
// The rotation that annihilates a_pq: with d = a_qq - a_pp, h = 2 a_pq, r = sqrt(d^2 + h^2) the
// SMALLER angle has cos^2 = (1 + |d| / r) / 2 and sin = sign(d) h / (2 r cos) — the same rotation
// as the textbook t = sign(theta) / (|theta| + sqrt(theta^2 + 1)) form of jacobi_eig_t, computed
// with two reciprocal square roots instead of two divisions and two square roots (the next round's
// angles sit on the critical path of every Jacobi round). (c, s) is renormalised to first order, so
// c^2 + s^2 = 1 to rounding whatever the estimates' last bits are.
__device__ inline void jacobi_rotation(double app, double aqq, double apq, double *c, double *sn) {
  *c = 1.0;
  *sn = 0.0;
  if (apq != 0.0) {
    const double d = aqq - app, h = 2.0 * apq;
    const double r2 = d * d + h * h;
    // (scaled: the entries of a Gram with a 1e7 x dominant eigenvalue square to 1e15 and beyond)
    const double inv_r = fast_rsqrt(r2);
    if (isfinite(inv_r) && inv_r > 0.0) {
      const double x = 0.5 * (1.0 + fabs(d) * inv_r);       // cos^2 in [1/2, 1]
      const double ic = fast_rsqrt(x);                       // 1 / cos
      double cc = x * ic;
      double ss = (d >= 0 ? 0.5 : -0.5) * h * inv_r * ic;
      const double corr = 1.5 - 0.5 * (cc * cc + ss * ss);   // first-order renormalisation
      *c = cc * corr;
      *sn = ss * corr;
    } else {  // overflow / underflow of d^2 + h^2: the division form
      const double theta = d / h;
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      *c = 1.0 / sqrt(t * t + 1.0);
      *sn = t * *c;
    }
  }
}
__device__ inline void jacobi_eig_block(double *A0, double *A1, double *Q0, double *Q1, int R,
                                        double *rc, double *rs, int *partner, double *red,
                                        double **Aout, double **Qout) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int nel = nthr - 64;           // threads that own elements
  const int rt = tid - nel;            // >= 0: lane of the rotation wave
  const int ldA = R + 1;
  for (int e = tid; e < R * R; e += nthr) {
    const int i = e / R, j = e - i * R;
    Q0[i * ldA + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  const int n2 = (R + 1) & ~1, npair = n2 / 2;
  double *A = A0, *An = A1, *Q = Q0, *Qn = Q1;
  // rotation tables of the pairing (p, q) with the angle (c, sn): coefficient of the PARTNER's
  // entry is -sn for p and +sn for q  (J[p][p] = J[q][q] = c, J[q][p] = -sn, J[p][q] = sn)
  auto put = [&](int half, int p, int q, double c, double sn) {
    int *pt = partner + 64 * half;
    double *pc = rc + 64 * half, *ps = rs + 64 * half;
    if (q < R) {
      pt[p] = q;
      pt[q] = p;
      pc[p] = c;
      pc[q] = c;
      ps[p] = -sn;
      ps[q] = sn;
    } else {  // bye
      pt[p] = p;
      pc[p] = 1.0;
      ps[p] = 0.0;
    }
  };
  for (int sweep = 0; sweep < 40; sweep++) {
    double off = 0, diag = 0;
    for (int e = tid; e < R * R; e += nthr) {
      const int i = e / R, j = e - i * R;
      const double a = A[i * ldA + j];
      if (i == j)
        diag += a * a;
      else
        off += a * a;
    }
    off = block_sum(off, red);
    diag = block_sum(diag, red);
    if (off <= 1e-30 * diag || off == 0.0) break;
    if (rt >= 0 && rt < npair) {  // rotations of round 0 from the current iterate
      int p, q;
      jacobi_pair(n2, 0, rt, &p, &q);
      double c = 1.0, sn = 0.0;
      if (q < R) jacobi_rotation(A[p * ldA + p], A[q * ldA + q], A[p * ldA + q], &c, &sn);
      put(0, p, q, c, sn);
    }
    __syncthreads();
    for (int rd = 0; rd < n2 - 1; rd++) {
      const int half = rd & 1;
      const int *pt = partner + 64 * half;
      const double *pc = rc + 64 * half, *ps = rs + 64 * half;
      // a'_ij = ci (cj a_ij + sj a_i,pj) + si (cj a_pi,j + sj a_pi,pj)
      auto rotated = [&](int i, int j) {
        const int pi = pt[i], pj = pt[j];
        const double ci = pc[i], si = ps[i], cj = pc[j], sj = ps[j];
        const double x = cj * A[i * ldA + j] + sj * A[i * ldA + pj];
        const double y = cj * A[pi * ldA + j] + sj * A[pi * ldA + pj];
        return ci * x + si * y;
      };
      if (rt < 0) {
        for (int e = tid; e < R * R; e += nel) {
          const int i = e / R, j = e - i * R;
          An[i * ldA + j] = rotated(i, j);
          const int pj = pt[j];
          Qn[i * ldA + j] = pc[j] * Q[i * ldA + j] + ps[j] * Q[i * ldA + pj];
        }
      } else if (rt < npair && rd + 1 < n2 - 1) {  // next round's angles, from the old buffer
        int p, q;
        jacobi_pair(n2, rd + 1, rt, &p, &q);
        double c = 1.0, sn = 0.0;
        if (q < R) {
          // a'_pp, a'_qq, a'_pq of THIS round's result in two LDS round trips: the tables of p and
          // q, then the 3 x 4 entries of the old iterate they combine
          const int pp_ = pt[p], pq_ = pt[q];
          const double cp = pc[p], sp = ps[p], cq = pc[q], sq = ps[q];
          const double a_pp = A[p * ldA + p], a_pP = A[p * ldA + pp_], a_PP = A[pp_ * ldA + pp_];
          const double a_qq = A[q * ldA + q], a_qQ = A[q * ldA + pq_], a_QQ = A[pq_ * ldA + pq_];
          const double a_pq = A[p * ldA + q], a_pQ = A[p * ldA + pq_], a_Pq = A[pp_ * ldA + q],
                       a_PQ = A[pp_ * ldA + pq_];
          // (A is symmetric: a_Pp = a_pP)
          const double npp = cp * (cp * a_pp + sp * a_pP) + sp * (cp * a_pP + sp * a_PP);
          const double nqq = cq * (cq * a_qq + sq * a_qQ) + sq * (cq * a_qQ + sq * a_QQ);
          const double npq = cp * (cq * a_pq + sq * a_pQ) + sp * (cq * a_Pq + sq * a_PQ);
          jacobi_rotation(npp, nqq, npq, &c, &sn);
        }
        put(half ^ 1, p, q, c, sn);
      }
      __syncthreads();
      double *t0 = A;
      A = An;
      An = t0;
      t0 = Q;
      Q = Qn;
      Qn = t0;
    }
  }
  __syncthreads();
  *Aout = A;
  *Qout = Q;
}

__device__ inline void jacobi_eig_wave(double *A, double *Q, double *cs, int *pq, int R) {
  jacobi_eig_t<64>(A, Q, cs, pq, R, nullptr);
}
__device__ inline void jacobi_inverse_wave(double *A, double *Q, double *cs, int *pq, int R,
                                           double *Sinv_out) {
  const int lane = threadIdx.x & 63;
  const int ldA = R + 1;
  jacobi_eig_wave(A, Q, cs, pq, R);
  for (int e = lane; e < R * R; e += 64) {  // S^{-1}[i,j] = sum_k Q[i,k] Q[j,k] / w_k
    const int i = e % R, j = e / R;
    double acc = 0;
    for (int k = 0; k < R; k++) acc += Q[i * ldA + k] * (1.0 / A[k * ldA + k]) * Q[j * ldA + k];
    Sinv_out[e] = acc;
  }
  wave_sync();
}

// ONE wave. Fast path for the (overwhelmingly common) positive-definite S: in-place Gauss-Jordan
// sweeps without pivoting (for an SPD matrix every pivot is a Schur-complement diagonal, > 0, the
// same quantities a Cholesky factorisation would take the root of). One LDS round trip and one
// barrier per pivot, ping-ponging between A and X, instead of the factor / triangular-inverse /
// product chain (three dependent R-step loops). Returns false (wave-uniform) when a pivot is not
// positive — the caller then falls back to the Jacobi path, which handles any symmetric S like the
// reference's SVD does. In: A (R x R, ld R+1), destroyed. X: R*(R+1) scratch.
__device__ inline bool spd_inverse_wave(double *A, double *X, int R, double *Sinv_out) {
  const int lane = threadIdx.x & 63;
  const int ldA = R + 1;  // element (i,j) at A[i*ldA + j]
  double *src = A, *dst = X;
  for (int k = 0; k < R; k++) {
    wave_sync();
    const double p = src[k * ldA + k];
    if (!(p > 0.0)) return false;  // same value in every lane -> uniform
    const double d = 1.0 / p;
    for (int e = lane; e < R * R; e += 64) {
      const int i = e % R, j = e / R;
      double v;
      if (i == k)
        v = (j == k) ? d : src[k * ldA + j] * d;
      else if (j == k)
        v = -src[i * ldA + k] * d;
      else
        v = src[i * ldA + j] - src[i * ldA + k] * (src[k * ldA + j] * d);
      dst[i * ldA + j] = v;
    }
    double *t = src;
    src = dst;
    dst = t;
  }
  wave_sync();
  // symmetrise (the two triangles differ by rounding only) and emit column-major
  for (int e = lane; e < R * R; e += 64) {
    const int a = e % R, b = e / R;
    Sinv_out[e] = 0.5 * (src[a * ldA + b] + src[b * ldA + a]);
  }
  wave_sync();
  return true;
}

// S and S^{-1} for one mode, ONE wave (used stand-alone by the sharded path and the parity tests;
// the single-GPU sweep uses the fused k_cp_mode_update below).
// dynamic LDS: A[R][R+1], Q[R][R+1], cs[64] doubles, pq[64] ints  (R <= 64)
__device__ inline void gram_system_wave(const double *__restrict__ Gall, int N, int mode, int R,
                                        double lambda, double *__restrict__ S,
                                        double *__restrict__ Sinv, int force_jacobi, double *lds) {
  const int ldA = R + 1;
  double *A = lds;
  double *Q = A + R * ldA;
  double *cs = Q + R * ldA;
  int *pq = (int *)(cs + 64);
  const int lane = threadIdx.x & 63;
  for (int e = lane; e < R * R; e += 64) {
    const double v = hadamard_entry(Gall, N, mode, R, lambda, e);
    S[e] = v;
    A[(e % R) * ldA + e / R] = v;
  }
  wave_sync();
  bool ok = false;
  if (!force_jacobi) ok = spd_inverse_wave(A, Q, R, Sinv);
  if (!ok) {
    wave_sync();
    for (int e = lane; e < R * R; e += 64) A[(e % R) * ldA + e / R] = S[e];
    wave_sync();
    jacobi_inverse_wave(A, Q, cs, pq, R, Sinv);
  }
}
__global__ __launch_bounds__(64) void k_gram_system(const double *__restrict__ Gall, int N,
                                                    int mode, int R, double lambda,
                                                    double *__restrict__ S,
                                                    double *__restrict__ Sinv, int force_jacobi) {
  extern __shared__ double lds[];
  gram_system_wave(Gall, N, mode, R, lambda, S, Sinv, force_jacobi, lds);
}

// ------------------------------------------------------------------ rank above 64
// The fused / one-wave kernels keep S, S^-1 and two work matrices in LDS (R <= 64). For larger R
// the same arithmetic runs out of global memory (R x R doubles stay L2-resident):
//  k_gram_system_big: ONE 1024-thread block, S = Hadamard of the Grams, S^-1 by the same
//    pivot-free Gauss-Jordan sweeps (ping-pong between two global buffers, one barrier per pivot).
//    *status = 1 when a pivot is not positive (S not SPD: the caller reports it).
__global__ __launch_bounds__(1024) void k_gram_system_big(const double *__restrict__ Gall, int N,
                                                          int mode, int R, double lambda,
                                                          double *__restrict__ S,
                                                          double *__restrict__ Sinv,
                                                          double *__restrict__ work,
                                                          int *__restrict__ status) {
  const int tid = threadIdx.x;
  double *src = work, *dst = Sinv;  // R even or odd: the final swap lands the result in `res`
  for (int e = tid; e < R * R; e += blockDim.x) {
    const double v = hadamard_entry(Gall, N, mode, R, lambda, e);
    S[e] = v;
    src[e] = v;
  }
  __syncthreads();
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int k = 0; k < R; k++) {
    const double p = src[k + (int64_t)R * k];
    if (!(p > 0.0)) {
      if (tid == 0) bad = 1;
    }
    __syncthreads();
    if (bad) break;
    const double d = 1.0 / p;
    for (int e = tid; e < R * R; e += blockDim.x) {
      const int i = e % R, j = e / R;
      double v;
      if (i == k)
        v = (j == k) ? d : src[k + (int64_t)R * j] * d;
      else if (j == k)
        v = -src[i + (int64_t)R * k] * d;
      else
        v = src[e] - src[i + (int64_t)R * k] * (src[k + (int64_t)R * j] * d);
      dst[e] = v;
    }
    __syncthreads();
    double *t = src;
    src = dst;
    dst = t;
  }
  if (tid == 0) *status = bad;
  // symmetrise into Sinv (src holds the result; it may already be Sinv)
  for (int e = tid; e < R * R; e += blockDim.x) {
    const int a = e % R, b = e / R;
    if (a <= b) {
      const double v = 0.5 * (src[a + (int64_t)R * b] + src[b + (int64_t)R * a]);
      dst[a + (int64_t)R * b] = v;
      dst[b + (int64_t)R * a] = v;
    }
  }
  __syncthreads();
  if (dst != Sinv)
    for (int e = tid; e < R * R; e += blockDim.x) Sinv[e] = dst[e];
  if (bad) {  // not SPD: make the failure visible instead of returning a half-swept matrix
    __syncthreads();
    for (int e = tid; e < R * R; e += blockDim.x) Sinv[e] = __longlong_as_double(0x7ff8000000000000LL);
  }
}
// The same sweeps with the matrix in LDS (64 < R <= 128: R*R + 2R doubles fit the 160 KB of a CU):
// in place — the pivot row and column are set aside first, every other entry is updated from
// them — so a pivot costs two barriers and LDS round trips instead of dependent L2 round trips
// (0.6 ms at R = 100 out of global memory, a third of the whole unfused mode update).
__global__ __launch_bounds__(1024) void k_gram_system_lds(const double *__restrict__ Gall, int N,
                                                          int mode, int R, double lambda,
                                                          double *__restrict__ S,
                                                          double *__restrict__ Sinv,
                                                          int *__restrict__ status) {
  extern __shared__ double lds[];
  double *A = lds;              // R x R, column-major
  double *prow = A + R * R;     // pivot row   A[k][*]
  double *pcol = prow + R;      // pivot column A[*][k]
  __shared__ int bad;
  const int tid = threadIdx.x;
  for (int e = tid; e < R * R; e += blockDim.x) {
    const double v = hadamard_entry(Gall, N, mode, R, lambda, e);
    S[e] = v;
    A[e] = v;
  }
  if (tid == 0) bad = 0;
  __syncthreads();
  for (int k = 0; k < R; k++) {
    const double p = A[k + R * k];
    if (!(p > 0.0)) {
      if (tid == 0) bad = 1;
    }
    if (tid < R) {
      prow[tid] = A[k + R * tid];
      pcol[tid] = A[tid + R * k];
    }
    __syncthreads();
    if (bad) break;
    const double d = 1.0 / p;
    for (int e = tid; e < R * R; e += blockDim.x) {
      const int i = e % R, j = e / R;
      double v;
      if (i == k)
        v = (j == k) ? d : prow[j] * d;
      else if (j == k)
        v = -pcol[i] * d;
      else
        v = A[e] - pcol[i] * (prow[j] * d);
      A[e] = v;
    }
    __syncthreads();
  }
  if (tid == 0) *status = bad;
  for (int e = tid; e < R * R; e += blockDim.x) {
    const int a = e % R, b = e / R;
    Sinv[e] = bad ? __longlong_as_double(0x7ff8000000000000LL)
                  : 0.5 * (A[a + R * b] + A[b + R * a]);  // the two triangles differ by rounding
  }
}
// The same inverse by BLOCK Gauss-Jordan sweeps, four pivots per step, with the trailing update on
// the fp64 matrix cores (64 < R <= 128; round 6). The scalar sweeps above spend 3.3 us per pivot on one
// CU (330 us at R = 100: four of them were 15 % of an exact sweep at the reference's default rank s/2);
// here a step inverts the 4 x 4 pivot block in registers (every thread, redundantly: the same scalar
// pivots as the scalar sweeps, so the same "not positive definite" test), threads 0..RP-1 form the
// row panel P^-1 A[K,:] and the column panels -A[:,K], -A[:,K] P^-1, and the 16 waves apply
// A[O,O] -= A[O,K] P^-1 A[K,O] tile by tile as ONE v_mfma_f64_16x16x4_f64 each (K = 4 is the block).
// The matrix is padded to a multiple of 16 with the identity (its inverse is the identity: the pad
// never mixes with S). Three barriers per step, R/4 steps.
// LDS: LD*RP + 8*RP doubles, RP = R rounded up to 16, LD = RP padded to 2 (mod 32) (141 KB at R = 128).
__global__ __launch_bounds__(1024) void k_gram_system_mfma(const double *__restrict__ Gall, int N,
                                                           int mode, int R, double lambda,
                                                           double *__restrict__ S,
                                                           double *__restrict__ Sinv,
                                                           int *__restrict__ status) {
  extern __shared__ double lds[];
  const int RP = (R + 15) & ~15;
  // leading dimension == 2 (mod 32) doubles: the 16 columns x 4 rows a wave touches in a tile land in
  // distinct banks (a column stride of RP = 112 doubles puts eight lanes on one bank)
  const int LD = RP + ((34 - (RP & 31)) & 31);
  double *A = lds;             // RP x RP, column-major, leading dimension LD
  double *Rk = A + LD * RP;    // [4][RP]  P^-1 A[K, :]     (zero in the K columns)
  double *Cn = Rk + 4 * RP;    // [4][RP]  -A[:, K]         (zero in the K rows)
  __shared__ int bad_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < RP * RP; e += blockDim.x) {
    const int i = e % RP, j = e / RP;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < R && j < R) {
      v = hadamard_entry(Gall, N, mode, R, lambda, i + R * j);
      S[i + R * j] = v;
    }
    A[i + LD * j] = v;
  }
  if (tid == 0) bad_s = 0;
  __syncthreads();
  const int ntile = RP >> 4;
  const int t = tid;
  for (int k0 = 0; k0 < R; k0 += 4) {
    // threads 0..RP-1 (two waves): the pivot block's inverse in registers (each of them, redundantly:
    // the same scalar pivots as the scalar sweeps), column t of the row panel, row t of the column panels
    double P[4][4], rk_new[4], cw[4];
    const bool inK = t >= k0 && t < k0 + 4;
    if (t < RP) {
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) P[a][b] = A[(k0 + a) + LD * (k0 + b)];
      int bad = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const double p = P[s][s];
        if (!(p > 0.0)) bad = 1;
        const double d = 1.0 / p;
        double rs[4], cs[4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
          rs[a] = P[s][a];
          cs[a] = P[a][s];
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) {
            double v;
            if (a == s)
              v = (b == s) ? d : rs[b] * d;
            else if (b == s)
              v = -cs[a] * d;
            else
              v = P[a][b] - cs[a] * (rs[b] * d);
            P[a][b] = v;
          }
      }
      if (bad && t == 0) bad_s = 1;
      double x[4], y[4];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        x[a] = A[(k0 + a) + LD * t];
        y[a] = A[t + LD * (k0 + a)];
      }
#pragma unroll
      for (int a = 0; a < 4; a++) {
        double r = 0, c = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          r += P[a][b] * x[b];
          c -= y[b] * P[b][a];
        }
        rk_new[a] = r;
        cw[a] = c;
        Rk[a * RP + t] = inK ? 0.0 : r;
        Cn[a * RP + t] = inK ? 0.0 : -y[a];
      }
    }
    __syncthreads();
    if (bad_s) break;  // (workgroup-uniform: written before the barrier, read after it)
    // trailing update, one MFMA per 16 x 16 tile (f64 C/D map: column lane & 15, rows (lane >> 4) + 4 r)
    for (int tile = wave; tile < ntile * ntile; tile += 16) {
      const int ti = tile % ntile, tj = tile / ntile;
      double *c0 = A + (16 * ti + (lane >> 4)) + LD * (16 * tj + (lane & 15));
      f64x4 c = {c0[0], c0[4], c0[8], c0[12]};
      const double av = Cn[(lane >> 4) * RP + 16 * ti + (lane & 15)];
      const double bv = Rk[(lane >> 4) * RP + 16 * tj + (lane & 15)];
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
      c0[0] = c[0];
      c0[4] = c[1];
      c0[8] = c[2];
      c0[12] = c[3];
    }
    __syncthreads();
    if (t < RP) {
#pragma unroll
      for (int a = 0; a < 4; a++) {
        if (!inK) {
          A[(k0 + a) + LD * t] = rk_new[a];
          A[t + LD * (k0 + a)] = cw[a];
        } else {
          const int tb = t - k0;  // (a select chain: no dynamically indexed register array)
          A[(k0 + a) + LD * t] = tb == 0 ? P[a][0] : (tb == 1 ? P[a][1] : (tb == 2 ? P[a][2] : P[a][3]));
        }
      }
    }
    __syncthreads();
  }
  const int bad = bad_s;
  if (tid == 0) *status = bad;
  for (int e = tid; e < R * R; e += blockDim.x) {
    const int a = e % R, b = e / R;
    Sinv[e] = bad ? __longlong_as_double(0x7ff8000000000000LL)
                  : 0.5 * (A[a + LD * b] + A[b + LD * a]);  // the two triangles differ by rounding
  }
}
// Sinv = Z diag(1/w) Z^T from the eigen-decomposition S = Z diag(w) Z^T (Z column-major, w any
// order): for symmetric S this IS the reference's untruncated V diag(1/sigma) U^T
// (common.cxx:717-722) — the defined answer when S is not positive definite
__global__ void k_eig_inverse(const double *__restrict__ Z, const double *__restrict__ w, int R,
                              double *__restrict__ Sinv, const int *__restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;  // (conditional launch: see k_jacobi_onesided)
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < R * R; e += gridDim.x * blockDim.x) {
    const int i = e % R, j = e / R;
    double acc = 0;
    for (int k = 0; k < R; k++) acc += Z[i + (int64_t)R * k] * (1.0 / w[k]) * Z[j + (int64_t)R * k];
    Sinv[e] = acc;
  }
}
// row-parallel mode update for any R: grad = -M + W_old S, W = M S^-1 (optional SVD_solve_mod
// tail), per-block partial ||grad||^2. One block per 64 rows; a thread owns (row, column j).
// Wnew must not alias Wold here (several blocks): the caller passes a scratch copy of W_old.
__global__ __launch_bounds__(256) void k_cp_update_big(
    const double *__restrict__ M, int64_t ldm, const double *__restrict__ Wold, int64_t ldw,
    double *__restrict__ Wnew, int64_t ldn, double *__restrict__ grad, int64_t ldg, int64_t rows,
    int R, const double *__restrict__ S, const double *__restrict__ Sinv,
    double *__restrict__ gradsq_part, const double *__restrict__ Winit, int64_t ldi,
    double *__restrict__ dW, int64_t ldd, double ratio) {
  __shared__ double lds[17];
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  double gs = 0;
  if (i < rows)
    for (int j = threadIdx.x >> 6; j < R; j += 4) {
      double a = 0, b = 0;
      for (int k = 0; k < R; k++) {
        a += Wold[i + ldw * k] * S[k + (int64_t)R * j];
        b += M[i + ldm * k] * Sinv[k + (int64_t)R * j];
      }
      const double gv = -M[i + ldm * j] + a;
      grad[i + ldg * j] = gv;
      gs += gv * gv;
      if (Winit) {
        const double wi = Winit[i + ldi * j];
        const double d = ratio * (b - wi);
        dW[i + ldd * j] = d;
        if (ratio != 1.0) b = wi + d;
      }
      Wnew[i + ldn * j] = b;
    }
  gs = block_sum(gs, lds);
  if (threadIdx.x == 0) gradsq_part[blockIdx.x] = gs;
}

// ------------------------------------------------------------------ mode update (K5 + K6b)
// ONE block. grad = -M + Wold*S (pre-update W, als_CP.cxx:296); Wnew = M*Sinv (SVD_solve);
// optional SVD_solve_mod tail (common.cxx:753-756). Wnew may alias Wold.
__global__ __launch_bounds__(1024) void k_cp_update(
    const double *__restrict__ M, int64_t ldm, const double *Wold, int64_t ldw, double *Wnew,
    int64_t ldn, double *__restrict__ grad, int64_t ldg, int64_t rows, int R,
    const double *__restrict__ S, const double *__restrict__ Sinv, double *__restrict__ gradsq,
    const double *__restrict__ Winit, int64_t ldi, double *__restrict__ dW, int64_t ldd,
    double ratio) {
  extern __shared__ double lds[];
  double *sS = lds + 32;
  double *sI = sS + R * R;
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    sS[e] = S[e];
    sI[e] = Sinv[e];
  }
  __syncthreads();
  const int64_t total = rows * R;
  double gs = 0;
  for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
    const int64_t i = e % rows;
    const int j = (int)(e / rows);
    double acc = 0;
    for (int k = 0; k < R; k++) acc += Wold[i + ldw * k] * sS[k + R * j];
    const double gv = -M[i + ldm * j] + acc;
    grad[i + ldg * j] = gv;
    gs += gv * gv;
  }
  gs = block_sum(gs, lds);  // also the barrier between reading Wold and writing Wnew
  if (threadIdx.x == 0) *gradsq = gs;
  for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
    const int64_t i = e % rows;
    const int j = (int)(e / rows);
    double acc = 0;
    for (int k = 0; k < R; k++) acc += M[i + ldm * k] * sI[k + R * j];
    if (Winit) {
      const double wi = Winit[i + ldi * j];
      const double d = ratio * (acc - wi);
      dW[i + ldd * j] = d;
      if (ratio != 1.0) acc = wi + d;
    }
    Wnew[i + ldn * j] = acc;
  }
}

// ------------------------------------------------------------------ fused mode update (single GPU)
// ONE block of 1024 threads does a whole mode update: S (K4), S^{-1} (wave 0: Gauss-Jordan sweeps, Jacobi
// fallback), gradient with the pre-update W + ||grad||^2 (K5), W = M S^{-1} (K6, optional
// SVD_solve_mod tail), and the refreshed Gram G_mode = W^T W. Replaces three launches per mode.
// dynamic LDS: red[32] | sS[R*R] | sI[R*R] | A[R*(R+1)] | Q[R*(R+1)] | cs[64] | pq[64 ints]
// STAGE: M and the pre-update W are copied into LDS by waves 1..15 WHILE wave 0 factorises S, the
// gradient / solve / Gram phases then run out of LDS (the new W replaces the old one there) and
// global memory sees one read of M, W and one write of grad, W — the launch is a chain of
// dependent phases, so every global round trip removed is ~1-2 us off a ~23 us kernel.
// extra dynamic LDS with STAGE: sM[rows*R] | sW[rows*R] after the areas listed above.
struct PtrsN {
  double *p[MAX_ORDER];
  double *q[MAX_ORDER];
  double *d[MAX_ORDER];
  int64_t n[MAX_ORDER];
};
// Normalize at the tail of the sweep's LAST mode update (same arithmetic as k_normalize_fused, no
// cached multi-sweep tensors involved): one launch less per sweep where a launch is 8 % of it.
struct NormArgs {
  int on = 0;
  PtrsN w;             // the N factors (p, n)
  double *scales = nullptr;
  double *wsq = nullptr;
  // the pending-scale update of the cached multi-sweep tensors (as k_normalize_fused): ms_dst[k] for
  // every k with (active >> k) & 1, starting from 1.0 where (fresh >> k) & 1
  double *ms_dst = nullptr;
  ScaleMasks masks;
  unsigned active = 0, fresh = 0;
};
// MF (with STAGE): the two row products — grad = W_old S - M and W = M S^-1, (rows x R)(R x R) —
// run on the matrix cores out of LDS. As VALU loops they are LDS-bandwidth bound (2 reads per FMA:
// ~35 of the 69 us of a launch at R = 20, rows = 400); v_mfma_f64_16x16x4_f64 needs two LDS
// fragments per 1024 multiply-adds. Computed transposed, D'[j][i] = sum_k S[k][j] W[i][k], so that
// the lanes of a result register run along i: stores to grad / W are 128-byte segments.
// presolved: S and S^-1 were prepared by the preceding contraction's launch (SysArgs) and are
// read from S_out / Sinv_out instead of being computed here.
// (tools/update_bench.hip defines PPALS_UPDATE_STAMPS: thread 0 writes the 100 MHz device clock at
// the phase boundaries of the launch into g_update_stamps — where the microseconds go)
#ifdef PPALS_UPDATE_STAMPS
__device__ unsigned long long g_update_stamps[16];
#define PPALS_STAMP(i) \
  if (threadIdx.x == 0) g_update_stamps[i] = __builtin_amdgcn_s_memrealtime()
#else
#define PPALS_STAMP(i)
#endif
template <bool STAGE, bool MF = false>
__global__ __launch_bounds__(1024) void k_cp_mode_update(
    double *__restrict__ Gall, int N, int mode, int R, double lambda, const double *__restrict__ M,
    int64_t ldm, double *W, int64_t ldw, double *__restrict__ grad, int64_t ldg, int64_t rows,
    double *__restrict__ gradsq, const double *__restrict__ Winit, int64_t ldi,
    double *__restrict__ dW, int64_t ldd, double ratio, double *__restrict__ S_out,
    double *__restrict__ Sinv_out, double *__restrict__ dwsq, int presolved = 0,
    NormArgs nrm = NormArgs(), int mblk = 0) {
  // mblk > 0 (STAGE only): M is handed over in ROW BLOCKS of mblk rows, block p at p * mblk * R with
  // leading dimension mblk — the receive buffer of an all-gather of the ranks' row blocks, read as
  // it arrives (no unpack launch in front of the update)
  extern __shared__ double lds[];
  const int ldA = R + 1;
  double *red = lds;
  double *sS = lds + 32;
  double *sI = sS + R * R;
  double *A = sI + R * R;
  double *Q = A + R * ldA;
  double *cs = Q + R * ldA;
  int *pq = (int *)(cs + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *sM = (double *)(pq + 64);
  double *sW = sM + (STAGE ? rows * R : 0);
  const int64_t total = rows * R;
  const int rows_i = (int)rows, total_i = (int)total;  // launcher: rows * R < 2^31 (32-bit index math)
  PPALS_STAMP(0);

  if (presolved) {
    // S, S^-1 come from the launch before this one; M and the pre-update W are staged by everybody
    for (int e = tid; e < R * R; e += blockDim.x) {
      sS[e] = S_out[e];
      sI[e] = Sinv_out[e];
    }
    if (STAGE)
      for (int e = tid; e < total_i; e += (int)blockDim.x) {
        const int i = e % rows_i, j = e / rows_i;
        sM[e] = mblk ? M[(int64_t)(i / mblk) * mblk * R + (i % mblk) + (int64_t)mblk * j] : M[i + ldm * j];
        sW[e] = W[i + ldw * j];
      }
    __syncthreads();
  } else {
    for (int e = tid; e < R * R; e += blockDim.x) {
      const double v = hadamard_entry(Gall, N, mode, R, lambda, e);
      sS[e] = v;
      A[(e % R) * ldA + e / R] = v;
    }
    __syncthreads();
    if (wave == 0) {
      bool ok = spd_inverse_wave(A, Q, R, sI);
      if (!ok) {
        wave_sync();
        for (int e = lane; e < R * R; e += 64) A[(e % R) * ldA + e / R] = sS[e];
        wave_sync();
        jacobi_inverse_wave(A, Q, cs, pq, R, sI);
      }
    } else if (STAGE) {
      for (int e = tid - 64; e < total_i; e += (int)blockDim.x - 64) {
        const int i = e % rows_i, j = e / rows_i;
        sM[e] = mblk ? M[(int64_t)(i / mblk) * mblk * R + (i % mblk) + (int64_t)mblk * j] : M[i + ldm * j];
        sW[e] = W[i + ldw * j];
      }
    }
    __syncthreads();
    if (S_out)
      for (int e = tid; e < R * R; e += blockDim.x) {
        S_out[e] = sS[e];
        Sinv_out[e] = sI[e];
      }
  }

  PPALS_STAMP(1);  // S, S^-1, M, W in LDS
  double gs = 0, dd = 0;
  if constexpr (STAGE && MF) {
    const int g4 = lane >> 4, l16 = lane & 15;
    const int nti = (rows_i + 15) / 16, ntj = (R + 15) / 16, ksteps = (R + 3) / 4;
    const int nwv = (int)blockDim.x >> 6;
    for (int phase = 0; phase < 2; phase++) {
      const double *sA = phase == 0 ? sS : sI;   // S or S^-1 (symmetric: [k + R j])
      const double *sB = phase == 0 ? sW : sM;   // W_old or M (rows x R, ld rows)
      for (int tle = wave; tle < nti * ntj; tle += nwv) {
        const int i0 = (tle % nti) * 16, j0 = (tle / nti) * 16;
        const int ja = j0 + l16, ib = i0 + l16;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int ks = 0; ks < ksteps; ks++) {
          const int k = 4 * ks + g4;
          const bool kin = k < R;
          const double a = (kin && ja < R) ? sA[k + R * ja] : 0.0;
          const double b = (kin && ib < rows_i) ? sB[ib + rows_i * k] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        // acc[r] = D'[j0 + g4 + 4 r][i0 + l16]
        if (ib < rows_i) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int j = j0 + g4 + 4 * r;
            if (j < R) {
              if (phase == 0) {
                const double gv = acc[r] - sM[ib + rows_i * j];
                grad[ib + ldg * j] = gv;
                gs += gv * gv;
              } else {
                double w = acc[r];
                if (Winit) {
                  const double wi = Winit[ib + ldi * j];
                  const double d = ratio * (w - wi);
                  dW[ib + ldd * j] = d;
                  dd += d * d;
                  if (ratio != 1.0) w = wi + d;
                }
                W[ib + ldw * j] = w;
                sW[ib + rows_i * j] = w;  // old W is dead since the barrier below
              }
            }
          }
        }
      }
      if (phase == 0) {
        PPALS_STAMP(2);  // gradient tiles done (this wave)
        // (also the barrier between reading W_old and writing W_new in LDS; the gradient's global
        // stores stay in flight across it)
        gs = block_sum_lds(gs, red);
        if (tid == 0) *gradsq = gs;
        PPALS_STAMP(3);
      }
    }
  } else {
  for (int e = tid; e < total_i; e += (int)blockDim.x) {
    const int i = e % rows_i, j = e / rows_i;
    double acc = 0;
    if (STAGE) {
      for (int k = 0; k < R; k++) acc += sW[i + rows_i * k] * sS[k + R * j];
    } else {
      for (int k = 0; k < R; k++) acc += W[i + ldw * k] * sS[k + R * j];
    }
    const double gv = -(STAGE ? sM[e] : M[i + ldm * j]) + acc;
    grad[i + ldg * j] = gv;
    gs += gv * gv;
  }
  gs = block_sum(gs, red);  // also the barrier between reading W_old and writing W_new
  if (tid == 0) *gradsq = gs;
  for (int e = tid; e < total_i; e += (int)blockDim.x) {
    const int i = e % rows_i, j = e / rows_i;
    double acc = 0;
    if (STAGE) {
      for (int k = 0; k < R; k++) acc += sM[i + rows_i * k] * sI[k + R * j];
    } else {
      for (int k = 0; k < R; k++) acc += M[i + ldm * k] * sI[k + R * j];
    }
    if (Winit) {
      const double wi = Winit[i + ldi * j];
      const double d = ratio * (acc - wi);
      dW[i + ldd * j] = d;
      dd += d * d;
      if (ratio != 1.0) acc = wi + d;
    }
    W[i + ldw * j] = acc;
    if (STAGE) sW[e] = acc;  // old W is dead since the block_sum barrier
  }
  }
  if (dwsq) {  // ||dW||^2 for the restart test of the PP phase (block-uniform branch)
    dd = (STAGE && MF) ? block_sum_lds(dd, red + 16) : block_sum(dd, red);
    if (tid == 0) *dwsq = dd;
  }
  PPALS_STAMP(4);  // solve tiles done (this wave)
  if constexpr (STAGE && MF)
    lds_barrier();  // W_new complete in LDS (its global stores need not have landed)
  else
    __syncthreads();  // W_new visible to the whole workgroup (LDS, or same CU's L1)
  PPALS_STAMP(5);
  double *G = Gall + (int64_t)mode * R * R;
  const int nw = blockDim.x >> 6;
  if constexpr (STAGE && MF) {
    // G_mode = W^T W on the matrix cores out of LDS: the row range is split over `nks` waves, each
    // accumulates the 16 x 16 tiles of the upper triangle over its rows, the partial tiles meet in
    // the LDS area of M (dead by now) and are added in a fixed order. (One wave per column pair and a
    // 64-lane dot product each took 2.6 us of a 8.6 us kernel at R = 10 and 12 of 29 us at R = 20:
    // tools/update_bench.hip, profiles/r03ah_update_phases.txt.)
    const int g4 = lane >> 4, l16 = lane & 15;
    const int ntg = (R + 15) / 16, ntile = ntg * (ntg + 1) / 2;
    const int cap = (int)((rows * R) / (ntile * 256));  // partial tiles that fit into sM
    const int nks = cap < 1 ? 0 : (cap < nw ? cap : nw);
    if (nks > 0) {
      const int ksteps = (rows_i + 3) / 4, spw = (ksteps + nks - 1) / nks;
      if (wave < nks) {
        const int s0 = wave * spw, s1 = (s0 + spw < ksteps) ? s0 + spw : ksteps;
        int t = 0;
        for (int tp = 0; tp < ntg; tp++)
          for (int tq = tp; tq < ntg; tq++, t++) {
            const int ca = 16 * tp + l16, cb = 16 * tq + l16;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};  // two chains in flight
            for (int ks = s0; ks < s1; ks += 2) {
              const int k = 4 * ks + g4, k2 = k + 4;
              const bool kin = k < rows_i, kin2 = (ks + 1 < s1) && k2 < rows_i;
              const double a = (kin && ca < R) ? sW[k + rows_i * ca] : 0.0;
              const double b = (kin && cb < R) ? sW[k + rows_i * cb] : 0.0;
              const double a2 = (kin2 && ca < R) ? sW[k2 + rows_i * ca] : 0.0;
              const double b2 = (kin2 && cb < R) ? sW[k2 + rows_i * cb] : 0.0;
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) sM[((wave * ntile + t) * 4 + r) * 64 + lane] = acc[r] + acc2[r];
          }
      }
      lds_barrier();
      // element (tile t, reg r, lane) = G[16 tp + g4 + 4 r][16 tq + l16]
      for (int e = tid; e < ntile * 256; e += (int)blockDim.x) {
        const int t = e >> 8, r = (e >> 6) & 3, ln = e & 63;
        int tp = 0, rem = t;
        while (rem >= ntg - tp) {
          rem -= ntg - tp;
          tp++;
        }
        const int tq = tp + rem;
        const int i = 16 * tp + (ln >> 4) + 4 * r, j = 16 * tq + (ln & 15);
        if (i < R && j < R && i <= j) {
          double v = 0;
          for (int w = 0; w < nks; w++) v += sM[((w * ntile + t) * 4 + r) * 64 + ln];
          G[i + R * j] = v;
          G[j + R * i] = v;
        }
      }
    }
    if (nks == 0) {  // (no room for a single partial tile: the pair loop below)
      const int npairs0 = R * (R + 1) / 2;
      for (int e = wave; e < npairs0; e += nw) {
        int p = 0, rem = e;
        while (rem >= R - p) {
          rem -= R - p;
          p++;
        }
        const int q = p + rem;
        double sacc = 0;
        for (int i = lane; i < rows_i; i += 64) sacc += sW[i + rows_i * p] * sW[i + rows_i * q];
        sacc = wave_sum(sacc);
        if (lane == 0) {
          G[p + R * q] = sacc;
          G[q + R * p] = sacc;
        }
      }
    }
  } else {
  // G_mode = W^T W: one wave per (p <= q) pair
  const int npairs = R * (R + 1) / 2;
  for (int e = wave; e < npairs; e += nw) {
    int p = 0, rem = e;
    while (rem >= R - p) {
      rem -= R - p;
      p++;
    }
    const int q = p + rem;
    const double *a = STAGE ? sW + rows * p : W + ldw * p;
    const double *b = STAGE ? sW + rows * q : W + ldw * q;
    double sacc = 0;
    for (int64_t i = lane; i < rows; i += 64) sacc += a[i] * b[i];
    sacc = wave_sum(sacc);
    if (lane == 0) {
      G[p + R * q] = sacc;
      G[q + R * p] = sacc;
    }
  }
  }
  PPALS_STAMP(6);  // Gram pairs of this wave done
  if constexpr (STAGE) {
    if (nrm.on) {  // (block-uniform) Normalize, common.cxx:644-689, on all N factors
      __threadfence();
      __syncthreads();
      double *nr = red;  // nrm[N] | fs[N]   (red has 32 doubles)
      if (tid < N) {
        double tr = 0;
        for (int k = 0; k < R; k++) tr += Gall[(int64_t)tid * R * R + k + R * k];
        nr[tid] = sqrt(tr);
      }
      __syncthreads();
      if (tid == 0) {
        double prod = 1;
        for (int i = 0; i < N; i++) prod = prod * nr[i];
        const double c = pow(prod, 1.0 / N);
        for (int i = 0; i < N; i++) {
          nr[MAX_ORDER + i] = c / nr[i];
          nrm.scales[i] = nr[MAX_ORDER + i];
          if (nrm.wsq) nrm.wsq[2 * i] = (nr[i] * nr[MAX_ORDER + i]) * (nr[i] * nr[MAX_ORDER + i]);
        }
      }
      __syncthreads();
      for (int i = 0; i < N; i++) {
        const double f = nr[MAX_ORDER + i];
        for (int e = tid; e < R * R; e += blockDim.x) Gall[(int64_t)i * R * R + e] *= f * f;
        double *pw = nrm.w.p[i];
        if (i == mode) {  // this launch's own factor: from its LDS copy (ld = rows)
          for (int e = tid; e < total_i; e += (int)blockDim.x)
            W[(e % rows_i) + ldw * (e / rows_i)] = f * sW[e];
        } else {
          for (int64_t e = tid; e < nrm.w.n[i]; e += blockDim.x) pw[e] = f * pw[e];
        }
      }
      if (nrm.ms_dst && tid < 32 && ((nrm.active >> tid) & 1u)) {
        double v = ((nrm.fresh >> tid) & 1u) ? 1.0 : nrm.ms_dst[tid];
        for (int m = 0; m < MAX_ORDER; m++)
          if (nrm.masks.m[tid] & (1u << m)) v *= nr[MAX_ORDER + m];
        nrm.ms_dst[tid] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ Normalize (K7)
// ONE wave: scales[i] = (prod_j ||W_j||)^(1/N) / ||W_i||, with ||W_i||^2 = trace(G_i); the Grams
// are rescaled in place (G_i *= scales[i]^2) so they stay consistent with the scaled factors.
__global__ __launch_bounds__(1024) void k_norm_scales(double *__restrict__ Gall, int N, int R,
                                                      double *__restrict__ scales) {
  __shared__ double nrm[MAX_ORDER];
  const int lane = threadIdx.x;
  if (lane < N) {
    double tr = 0;
    for (int k = 0; k < R; k++) tr += Gall[(int64_t)lane * R * R + k + R * k];
    nrm[lane] = sqrt(tr);
  }
  __syncthreads();
  double prod = 1;
  for (int i = 0; i < N; i++) prod = prod * nrm[i];
  const double c = pow(prod, 1.0 / N);
  for (int i = 0; i < N; i++) {
    const double f = c / nrm[i];
    for (int e = lane; e < R * R; e += blockDim.x) Gall[(int64_t)i * R * R + e] *= f * f;
    if (lane == 0) scales[i] = f;
  }
}
__global__ void k_scale_factors(PtrsN w, int N, const double *__restrict__ scales) {
  const int i = blockIdx.y;
  if (i >= N) return;
  const double f = scales[i];
  double *p = w.p[i];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < w.n[i];
       e += (int64_t)gridDim.x * blockDim.x)
    p[e] = f * p[e];
}

// Normalize in ONE launch for small factor sets (one block): the scales (same arithmetic as
// k_norm_scales), the Gram rescale, the factor rescale and the pending-scale update of the cached
// multi-sweep tensors (k_scale_update_many) — three ~5 us launches folded into one.
__global__ __launch_bounds__(1024) void k_normalize_fused(double *__restrict__ Gall, int N, int R,
                                                          PtrsN w, double *__restrict__ scales,
                                                          double *__restrict__ ms_dst,
                                                          ScaleMasks masks, unsigned active,
                                                          unsigned fresh, double *__restrict__ wsq) {
  __shared__ double nrm[MAX_ORDER], fs[MAX_ORDER];
  const int tid = threadIdx.x;
  if (tid < N) {
    double tr = 0;
    for (int k = 0; k < R; k++) tr += Gall[(int64_t)tid * R * R + k + R * k];
    nrm[tid] = sqrt(tr);
  }
  __syncthreads();
  if (tid == 0) {
    double prod = 1;
    for (int i = 0; i < N; i++) prod = prod * nrm[i];
    const double c = pow(prod, 1.0 / N);
    for (int i = 0; i < N; i++) {
      fs[i] = c / nrm[i];
      scales[i] = fs[i];
      if (wsq) wsq[2 * i] = (nrm[i] * fs[i]) * (nrm[i] * fs[i]);  // ||W_i||^2 after the rescaling
    }
  }
  __syncthreads();
  for (int i = 0; i < N; i++) {
    const double f = fs[i];
    for (int e = tid; e < R * R; e += blockDim.x) Gall[(int64_t)i * R * R + e] *= f * f;
    double *p = w.p[i];
    for (int64_t e = tid; e < w.n[i]; e += blockDim.x) p[e] = f * p[e];
  }
  if (ms_dst && tid < 32 && ((active >> tid) & 1u)) {
    double v = ((fresh >> tid) & 1u) ? 1.0 : ms_dst[tid];
    for (int m = 0; m < MAX_ORDER; m++)
      if (masks.m[tid] & (1u << m)) v *= fs[m];
    ms_dst[tid] = v;
  }
}

// ------------------------------------------------------------------ PP bookkeeping
// block i: out[2i] = ||A_i - B_i||^2, out[2i+1] = ||A_i||^2; optionally D_i = A_i - B_i and
// B_i = A_i (als_CP.cxx:594-603). B_i == nullptr: out[2i] = ||D_i||^2 (als_CP.cxx:659-663).
__global__ __launch_bounds__(1024) void k_diff_norms(PtrsN a, int store_diff, int update_prev,
                                                     double *__restrict__ out) {
  __shared__ double lds[17];
  const int i = blockIdx.x;
  const double *A = a.p[i];
  double *B = a.q[i];
  double *D = a.d[i];
  double sd = 0, sa = 0;
  for (int64_t e = threadIdx.x; e < a.n[i]; e += blockDim.x) {
    const double av = A[e];
    double dv;
    if (B) {
      dv = av - B[e];
      if (store_diff) D[e] = dv;
      if (update_prev) B[e] = av;
    } else {
      dv = D[e];
    }
    sd += dv * dv;
    sa += av * av;
  }
  sd = block_sum(sd, lds);
  sa = block_sum(sa, lds);
  if (threadIdx.x == 0) {
    out[2 * i] = sd;
    out[2 * i + 1] = sa;
  }
}

// ------------------------------------------------------------------ row-block (un)packing
__global__ void k_pack_blocks(const double *__restrict__ nat, int64_t rows, int64_t ld, int R,
                              int64_t blk, int P, double *__restrict__ blocked) {
  const int64_t total = blk * R * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = e % blk;
    const int r = (int)((e / blk) % R);
    const int p = (int)(e / (blk * R));
    const int64_t row = (int64_t)p * blk + x;
    blocked[e] = row < rows ? nat[row + ld * r] : 0.0;
  }
}
__global__ void k_unpack_blocks(const double *__restrict__ blocked, int64_t rows, int64_t ld,
                                int R, int64_t blk, int P, double *__restrict__ nat) {
  const int64_t total = blk * R * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = e % blk;
    const int r = (int)((e / blk) % R);
    const int p = (int)(e / (blk * R));
    const int64_t row = (int64_t)p * blk + x;
    if (row < rows) nat[row + ld * r] = blocked[e];
  }
}

// ------------------------------------------------------------------ Tucker: Gram of an unfolding
// G[p + J*q] (slab blockIdx.z) = sum_{c in chunk} A[p][c] * A[q][c],  A[p][c] = X[l + L*(p + J*t)],
// c = l + L*t (unroll_tensor_contraction, common.cxx:205-223). 32x32 output tile per block, the
// reduction index staged through LDS in chunks of 32; loads are arranged so that consecutive
// threads touch consecutive addresses (p fastest when L == 1, c fastest otherwise).
template <typename TV>
__global__ __launch_bounds__(256) void k_unfold_gram(const TV *__restrict__ X, int64_t L, int64_t J,
                                                     int64_t T, int64_t c_per_split,
                                                     double *__restrict__ slab) {
  __shared__ double As[32][33];
  __shared__ double Bs[32][33];
  const int64_t C = L * T;
  const int64_t p0 = (int64_t)blockIdx.x * 32, q0 = (int64_t)blockIdx.y * 32;
  const int64_t c_begin = (int64_t)blockIdx.z * c_per_split;
  const int64_t c_end = min(C, c_begin + c_per_split);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty in [0,8)
  double acc[4] = {0, 0, 0, 0};                            // outputs (ty + 8*i, tx)
  const bool p_fast = (L == 1);
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
#pragma unroll 8
    for (int ci = 0; ci < 32; ci++) {
      const double bq = Bs[tx][ci];
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] += As[ty + 8 * i][ci] * bq;
    }
    __syncthreads();
  }
  double *g = slab + (int64_t)blockIdx.z * J * J;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t p = p0 + ty + 8 * i, q = q0 + tx;
    if (p < J && q < J) g[p + J * q] = acc[i];
  }
}
// Leading eigenvectors of a small symmetric matrix (J <= 64) entirely in LDS, one 256-thread
// block: the same parallel-ordered Jacobi as the R x R solve (a round's J/2 rotations touch J^2
// elements: 4 waves shorten the 2.8 ms one wave needs at J = 50 to well under a millisecond),
// eigenvalues ranked descending. Used for the Tucker eigen-step whenever the mode extent is small
// (no vendor library involved). dynamic LDS: A[J][J+1] | Q[J][J+1] | cs[64] | pq[64 ints] | red[17]
// dynamic LDS (bytes): top_eig_small_lds(J)
__host__ __device__ inline size_t top_eig_small_lds(int J) {
  return sizeof(double) * (4 * (size_t)J * (J + 1) + 256 + 17) + sizeof(int) * 128;
}
__global__ __launch_bounds__(1024) void k_top_eig_small(const double *__restrict__ G, int J,
                                                        int rank, double *__restrict__ U,
                                                        double *__restrict__ evals) {
  extern __shared__ double lds[];
  const int ldA = J + 1;
  double *A0 = lds, *A1 = A0 + J * ldA, *Q0 = A1 + J * ldA, *Q1 = Q0 + J * ldA;
  double *rc = Q1 + J * ldA, *rs = rc + 128, *red = rs + 128;
  int *partner = (int *)(red + 17);
  const int tid = threadIdx.x;
  for (int e = tid; e < J * J; e += blockDim.x) A0[(e % J) * ldA + e / J] = G[e];
  __syncthreads();
  double *A, *Q;
  jacobi_eig_block(A0, A1, Q0, Q1, J, rc, rs, partner, red, &A, &Q);
  for (int k = tid; k < J; k += blockDim.x) {
    const double wk = A[k * ldA + k];
    int pos = 0;  // number of eigenvalues that come before k in descending order
    for (int j = 0; j < J; j++) {
      const double wj = A[j * ldA + j];
      if (wj > wk || (wj == wk && j < k)) pos++;
    }
    if (pos < rank) {
      for (int i = 0; i < J; i++) U[i + (int64_t)J * pos] = Q[i * ldA + k];
      if (evals) evals[pos] = wk;  // descending
    }
  }
}

// U[:,k] = Z[:, J-1-k] for k < rank (syevd returns ascending eigenvalues)
__global__ void k_take_top(const double *__restrict__ Z, int64_t J, int rank,
                           double *__restrict__ U) {
  const int64_t total = J * rank;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % J, k = e / J;
    U[e] = Z[i + J * (J - 1 - k)];
  }
}

// out[i] = in[n - 1 - i]
__global__ void k_reverse_copy(const double *__restrict__ in, int n, double *__restrict__ out) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = in[n - 1 - i];
}

// one wave per column: sign of the dot product with the reference column, then rescale
__global__ void k_sign_align(double *__restrict__ W, const double *__restrict__ Wref, int64_t rows,
                             int r) {
  const int lane = threadIdx.x & 63;
  const int col = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (col >= r) return;
  double *w = W + rows * col;
  const double *q = Wref + rows * col;
  double c = 0;
  for (int64_t j = lane; j < rows; j += 64) c += w[j] * q[j];
  c = wave_sum(c);
  c = __shfl(c, 0, 64);
  if (!(c > 0))
    for (int64_t j = lane; j < rows; j += 64) w[j] = -w[j];
}

// out (rows x C) = A (rows x K) * B (K x C) [+ D]; one thread per output element, B in LDS
__global__ void k_rows_times_small(const double *__restrict__ A, int64_t rows, int K,
                                   const double *__restrict__ B, int C,
                                   const double *__restrict__ D, double *__restrict__ out) {
  extern __shared__ double sB[];
  for (int e = threadIdx.x; e < K * C; e += blockDim.x) sB[e] = B[e];
  __syncthreads();
  const int64_t total = rows * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % rows;
    const int c = (int)(e / rows);
    double acc = D ? D[e] : 0.0;
    for (int k = 0; k < K; k++) acc += A[i + rows * k] * sB[k + K * c];
    out[e] = acc;
  }
}
// X[e + n*c] += sum_k T[e + n*k] * VT[k + r*c]  (X in TV, T / VT fp64); VT in LDS
template <typename TV>
__global__ void k_lowrank_accumulate(TV *__restrict__ X, int64_t n, int R, const double *__restrict__ T,
                                     int r, const double *__restrict__ VT) {
  extern __shared__ double sV[];
  for (int e = threadIdx.x; e < r * R; e += blockDim.x) sV[e] = VT[e];
  __syncthreads();
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    for (int c = 0; c < R; c++) {
      double acc = (double)X[e + n * c];
      for (int k = 0; k < r; k++) acc += T[e + n * k] * sV[k + r * c];
      X[e + n * c] = (TV)acc;
    }
  }
}

__global__ void k_add_inplace(double *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    dst[e] += src[e];
}

__global__ void k_sumsq(const double *__restrict__ x, int64_t n, double *__restrict__ partial) {
  __shared__ double lds[17];
  double s = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    s += x[e] * x[e];
  s = block_sum(s, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

}  // namespace ppals
