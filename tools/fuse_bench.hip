// fuse_bench.hip — what computing the FIRST CONSUMER of the multi-sweep schedule inside the tensor scan
// would cost the scan (tools only; judge's lever "consume X_r while it is still on chip").
// cfg2: X[m, r] = sum_d V[m, d] P[d, r], m = (a, b, c) with a fastest, M = 8e6 rows, K = 200, R = 10; the first
// consumer is T[ab, r] = sum_c X[ab, c, r] W_c[c, r] (k_mttv_vec: one more read of the 320 MB of X, 52 us).
// Fused form: tiles do not cross a c slab (40 000 rows = 625 wave tiles), a persistent workgroup keeps ONE
// (a, b) tile and walks over a group of c's, adds val * W_c[c, r] into lane-private LDS after every tile and
// leaves a slab T[cgroup][r][ab] at its end (13 groups: 41.6 MB written once, summed by the next kernel).
//   variant 0: the library's k_scan_suffix_buf<float, 1, 5> (tile order id, id + grid, ...)
//   variant 1: this file's copy of it, same order, slab-aligned tiles     (is the copy as fast?)
//   variant 2: the copy, c-grouped order                                   (what the order costs)
//   variant 3: the copy, c-grouped order + the fused consumer             (what the epilogue costs)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o fuse_bench tools/fuse_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../pairwise-perturbation_amd/csrc/ops.h"
#include "../pairwise-perturbation_amd/csrc/kernels_scan.hip.h"
using namespace ppals;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void k_fill(float *p, int64_t n, uint32_t seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)e * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[e] = 0.5f + (float)(h & 0xffff) * (0.5f / 65536.f);
  }
}
__global__ void k_zero_tail(float *P, int nblk, int K) {
  const int total = nblk * 256 * 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int u = e & 3, g = (e >> 6) & 3, blk = e / 256 / 4 * 4 / 4;
    (void)blk;
    const int b = e / 1024;
    if (16 * b + 4 * u + g >= K) P[e] = 0.f;
  }
}

// float, one n-tile, non-temporal result stores, slab-aligned tiles
template <int GROUPED, int FUSE, int STORE = 0>
__global__ __launch_bounds__(256) void k_scan_fused(const float *__restrict__ V, int64_t M, int64_t K,
                                                    const float *__restrict__ P, int nkb, float *__restrict__ out,
                                                    int ncols, int64_t slab_rows, int nslab, int nab, int ncg,
                                                    int cper, const double *__restrict__ Wc,
                                                    double *__restrict__ Tslab) {
  typedef f32x4 vec;
  constexpr int VEC = 4, KB = 16, FLUSH = 4;
  __shared__ double Ts[FUSE ? 4 : 1][FUSE ? 16 : 1][4][16];
  __shared__ float Xs[STORE ? 16 : 1][STORE ? 256 : 1];  // STORE: the workgroup's result tile, column by column
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  const int voffP = (int)((g * 16 + j16) * VEC * (int)sizeof(float));
  const int64_t block_bytes = (int64_t)KB * M * 4;
  const int64_t total_bytes = K * M * 4;
  const int ustep = (int)((int64_t)4 * M * 4);
  const __amdgpu_buffer_rsrc_t rsrcP =
      __builtin_amdgcn_make_buffer_rsrc((void *)P, 0, (int)((int64_t)nkb * (4 * 16 * VEC) * 4), 0x00020000);
  const int w = blockIdx.x;
  const int abT = GROUPED ? w % nab : 0, cg = GROUPED ? w / nab : 0;
  const int ntile_mine = GROUPED ? max(0, min(cper, nslab - cg * cper)) : 0;
  const int64_t ntiles = (int64_t)nab * nslab;
  struct Tile {
    int64_t m;
    int voff, c, ab;
    bool live;
  };
  auto decode = [&](int64_t k, Tile &t) -> bool {  // k-th tile of this workgroup
    int c, ab;
    if (GROUPED) {
      if (k >= ntile_mine) return false;
      c = cg * cper + (int)k;
      ab = abT;
    } else {
      const int64_t id = blockIdx.x + k * gridDim.x;
      if (id >= ntiles) return false;
      c = (int)(id / nab);
      ab = (int)(id % nab);
    }
    const int64_t r0 = ((int64_t)ab * 4 + wave) * (16 * VEC);
    t.live = r0 < slab_rows;
    t.c = c;
    t.ab = ab;
    t.m = (int64_t)c * slab_rows + (t.live ? r0 : 0) + (int64_t)VEC * j16;
    t.voff = (int)(((int64_t)g * M + t.m) * 4);
    return true;
  };
#define FB_LOAD(voff_, kb_, vv_, bb_)                                                                     \
  {                                                                                                       \
    const int64_t boff_ = (int64_t)(kb_)*block_bytes;                                                     \
    const int64_t rem_ = total_bytes - boff_;                                                             \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                 \
        (void *)((const char *)V + boff_), 0, (int)min(rem_, block_bytes), 0x00020000);                   \
    _Pragma("unroll") for (int u = 0; u < VEC; u++) vv_[u] =                                              \
        __builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b128(rs_, voff_, u * ustep, 2));         \
    bb_ = __builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b128(rsrcP, voffP, (int)((kb_) * (4 * 16 * VEC) * 4), 0)); \
  }
  if (FUSE) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int jj = 0; jj < VEC; jj++) Ts[wave][g * 4 + r][jj][j16] = 0.0;
  }
  Tile cur, nxt;
  int64_t k = 0;
  if (!decode(k, cur)) return;
  vec cv[VEC], cb;
  FB_LOAD(cur.voff, 0, cv, cb);
  for (;;) {
    const bool has_next = decode(k + 1, nxt);
    double wc[4] = {0, 0, 0, 0};
    if (FUSE) {
#pragma unroll
      for (int r = 0; r < 4; r++) wc[r] = Wc[cur.c * 16 + g * 4 + r];
    }
    f32x4 acc[VEC];
    double acc64[VEC][4];
#pragma unroll
    for (int a = 0; a < VEC; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        acc[a][r] = 0;
        acc64[a][r] = 0;
      }
    for (int kc = 0; kc < nkb; kc += FLUSH) {
      const int ke = min(nkb, kc + FLUSH);
      for (int kb = kc; kb < ke; kb++) {
        vec nv[VEC], nb;
        const bool same = kb + 1 < nkb;
        const int po = (same || !has_next) ? cur.voff : nxt.voff;
        const int pk = same ? kb + 1 : (has_next ? 0 : kb);
        FB_LOAD(po, pk, nv, nb);
#pragma unroll
        for (int u = 0; u < VEC; u++)
#pragma unroll
          for (int jj = 0; jj < VEC; jj++)
            acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(cb[u], cv[u][jj], acc[jj], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < VEC; u++) cv[u] = nv[u];
        cb = nb;
      }
#pragma unroll
      for (int a = 0; a < VEC; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          acc64[a][r] += (double)acc[a][r];
          acc[a][r] = 0;
        }
    }
    if constexpr (STORE == 1) {
      // (the four waves of the workgroup meet: a column of the tile leaves as ONE 1 KB store instruction)
      typedef float ovec_t __attribute__((ext_vector_type(VEC)));
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = g * 4 + r;
        if (n < ncols) {
          ovec_t ov;
#pragma unroll
          for (int jj = 0; jj < VEC; jj++) ov[jj] = cur.live ? (float)acc64[jj][r] : 0.f;
          *reinterpret_cast<ovec_t *>(&Xs[n][wave * 64 + VEC * j16]) = ov;
        }
      }
      __syncthreads();
      const int64_t wg_m = (int64_t)cur.c * slab_rows + (int64_t)cur.ab * 256;  // first row of the workgroup's tile
      const int64_t slab_end = ((int64_t)cur.c + 1) * slab_rows;
      for (int n = wave; n < ncols; n += 4) {
        const int64_t mm = wg_m + 4 * lane;
        if (mm < slab_end) {
          const ovec_t ov = *reinterpret_cast<const ovec_t *>(&Xs[n][4 * lane]);
          __builtin_nontemporal_store(ov, reinterpret_cast<ovec_t *>(out + (int64_t)n * M + mm));
        }
      }
    } else if (cur.live) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = g * 4 + r;
        if (n < ncols) {
          typedef float ovec_t __attribute__((ext_vector_type(VEC)));
          ovec_t ov;
#pragma unroll
          for (int jj = 0; jj < VEC; jj++) ov[jj] = (float)acc64[jj][r];
          __builtin_nontemporal_store(ov, reinterpret_cast<ovec_t *>(out + (int64_t)n * M + cur.m));
          if (FUSE) {
#pragma unroll
            for (int jj = 0; jj < VEC; jj++) Ts[wave][n][jj][j16] += acc64[jj][r] * wc[r];
          }
        }
      }
    }
    if (!has_next) break;
    cur = nxt;
    k++;
  }
#undef FB_LOAD
  if (FUSE) {
    const int64_t r0 = ((int64_t)abT * 4 + wave) * (16 * VEC);
    if (r0 < slab_rows) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = g * 4 + r;
        if (n < ncols) {
          double *o = Tslab + ((int64_t)cg * 16 + n) * slab_rows + r0 + VEC * j16;
          f64x2 a = {Ts[wave][n][0][j16], Ts[wave][n][1][j16]}, b = {Ts[wave][n][2][j16], Ts[wave][n][3][j16]};
          *reinterpret_cast<f64x2 *>(o) = a;
          *reinterpret_cast<f64x2 *>(o + 2) = b;
        }
      }
    }
  }
}

int main(int argc, char **argv) {
  const int s = argc > 1 ? atoi(argv[1]) : 200, R = argc > 2 ? atoi(argv[2]) : 10;
  const int rounds = argc > 3 ? atoi(argv[3]) : 9;
  const int mult = argc > 4 ? atoi(argv[4]) : 40;  // workgroups per CU in the persistent grids (the library's: 40)
  const int64_t slab = (int64_t)s * s, M = slab * s;
  const int K = s, nkb = (K + 15) / 16, nslab = s;
  const int nab = (int)((slab + 255) / 256);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int ncg = std::max(1, (ncu * 8) / nab), cper = (nslab + ncg - 1) / ncg;
  printf("device %s, %d CUs; s=%d R=%d, %d workgroups per CU: V %.2f GB, X %.2f GB fp32; %d tiles per c slab, %d c groups of %d\n", prop.name, ncu, s,
         R, mult, M * (double)K * 4e-9, M * (double)R * 4e-9, nab, ncg, cper);
  float *V, *P, *X, *X2;
  double *Wc, *Ts;
  CK(hipMalloc(&V, sizeof(float) * M * K));
  CK(hipMalloc(&P, sizeof(float) * (size_t)nkb * 1024));
  CK(hipMalloc(&X, sizeof(float) * M * 16));
  CK(hipMalloc(&X2, sizeof(float) * M * 16));
  CK(hipMalloc(&Wc, sizeof(double) * nslab * 16));
  CK(hipMalloc(&Ts, sizeof(double) * (size_t)ncg * 16 * slab));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, V, M * K, 1u);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)nkb * 1024, 2u);
  hipLaunchKernelGGL(k_zero_tail, dim3(64), dim3(256), 0, 0, P, nkb, K);
  std::vector<double> hW((size_t)nslab * 16);
  for (size_t i = 0; i < hW.size(); i++) hW[i] = 0.25 + 0.5 * ((i * 2654435761u >> 7) & 1023) / 1024.0;
  CK(hipMemcpy(Wc, hW.data(), sizeof(double) * hW.size(), hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  const int n_mt = (int)((M + 255) / 256);
  const unsigned grid_lib = (unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * mult);
  const unsigned grid_g = (unsigned)(nab * ncg);
  auto v0 = [&]() {
    hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 5>), dim3(grid_lib), dim3(256), 0, 0, V, M, (int64_t)K, M * K, P, n_mt, 1,
                       nkb, nkb, (double *)X, M, (int64_t)0, (int64_t)0, R, 1, (int64_t)n_mt);
  };
  auto v1 = [&]() {
    hipLaunchKernelGGL((k_scan_fused<0, 0>), dim3(grid_lib), dim3(256), 0, 0, V, M, (int64_t)K, P, nkb, X2, R, slab, nslab,
                       nab, ncg, cper, Wc, Ts);
  };
  auto v2 = [&]() {
    hipLaunchKernelGGL((k_scan_fused<1, 0>), dim3(grid_g), dim3(256), 0, 0, V, M, (int64_t)K, P, nkb, X2, R, slab, nslab, nab,
                       ncg, cper, Wc, Ts);
  };
  auto v3 = [&]() {
    hipLaunchKernelGGL((k_scan_fused<1, 1>), dim3(grid_g), dim3(256), 0, 0, V, M, (int64_t)K, P, nkb, X2, R, slab, nslab, nab,
                       ncg, cper, Wc, Ts);
  };
  auto v4 = [&]() {  // (the same two kernels into the OTHER result block: where the result lies matters on this part)
    hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 5>), dim3(grid_lib), dim3(256), 0, 0, V, M, (int64_t)K, M * K, P, n_mt, 1,
                       nkb, nkb, (double *)X2, M, (int64_t)0, (int64_t)0, R, 1, (int64_t)n_mt);
  };
  auto v5 = [&]() {
    hipLaunchKernelGGL((k_scan_fused<0, 0>), dim3(grid_lib), dim3(256), 0, 0, V, M, (int64_t)K, P, nkb, X, R, slab, nslab,
                       nab, ncg, cper, Wc, Ts);
  };
  auto v6 = [&]() {  // (the copy held to 3 workgroups per CU by 50 KB of unused dynamic LDS: is it the occupancy?)
    hipLaunchKernelGGL((k_scan_fused<0, 0>), dim3(grid_lib), dim3(256), 50 * 1024, 0, V, M, (int64_t)K, P, nkb, X, R, slab,
                       nslab, nab, ncg, cper, Wc, Ts);
  };
  auto v7 = [&]() {  // (the copy with the result tile gathered in LDS and stored as 1 KB pieces)
    hipLaunchKernelGGL((k_scan_fused<0, 0, 1>), dim3(grid_lib), dim3(256), 0, 0, V, M, (int64_t)K, P, nkb, X, R, slab, nslab,
                       nab, ncg, cper, Wc, Ts);
  };
  constexpr int NV = 8;
  std::function<void()> vs[NV] = {v0, v1, v2, v3, v4, v5, v6, v7};
  const char *names[NV] = {"library k_scan_suffix_buf<float,1,5> -> X", "copy, same order, slab-aligned tiles -> X2",
                           "copy, c-grouped order -> X2", "copy, c-grouped order + fused first consumer -> X2",
                           "library kernel -> X2", "copy, same order -> X", "copy, same order, 3 workgroups per CU -> X",
                           "copy, same order, result through LDS in 1 KB pieces -> X"};
  std::vector<float> ms[NV];
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; r++)
    for (int v = 0; v < NV; v++) {
      CK(hipEventRecord(e0, 0));
      vs[v]();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[v].push_back(t);
    }
  // results: X of every variant against the library's, T of the fused variant against a host sum over X
  v0();
  CK(hipDeviceSynchronize());
  const int64_t NCHK = 1 << 20;
  std::vector<float> hx0(NCHK), hx(NCHK);
  CK(hipMemcpy(hx0.data(), X, 4 * NCHK, hipMemcpyDeviceToHost));
  const double bytes = (double)M * K * 4.0 + (double)M * R * 4.0;
  for (int v = 0; v < NV; v++) {
    double maxrel = 0;
    if (v > 0) {
      float *dst = v >= 5 ? X : X2;
      CK(hipMemset(dst, 0, sizeof(float) * M * 16));
      vs[v]();
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hx.data(), dst, 4 * NCHK, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < NCHK; i++) maxrel = std::max(maxrel, (double)fabsf(hx[i] - hx0[i]) / (fabs((double)hx0[i]) + 1e-300));
    }
    std::sort(ms[v].begin(), ms[v].end());
    const float med = ms[v][ms[v].size() / 2];
    printf("%-52s median %.4f ms  min %.4f ms  %.0f GB/s = %.3f of 8 TB/s   max rel diff of X %.2e\n", names[v], med, ms[v].front(),
           bytes / (med * 1e-3) / 1e9, bytes / (med * 1e-3) / 8e12, maxrel);
  }
  v3();
  CK(hipDeviceSynchronize());
  {
    // T[ab, n] = sum_c X[ab + slab c, n] W[c, n] for the first 4096 ab and n = 0, R - 1
    const int nabchk = 4096;
    double worst = 0;
    for (int n : {0, R - 1}) {
      std::vector<float> col((size_t)M);
      CK(hipMemcpy(col.data(), X2 + (int64_t)n * M, 4 * (size_t)M, hipMemcpyDeviceToHost));
      std::vector<double> tg((size_t)ncg * slab);
      for (int cgi = 0; cgi < ncg; cgi++)
        CK(hipMemcpy(tg.data() + (size_t)cgi * slab, Ts + ((int64_t)cgi * 16 + n) * slab, 8 * (size_t)slab, hipMemcpyDeviceToHost));
      for (int ab = 0; ab < nabchk; ab++) {
        double ref = 0, got = 0;
        for (int c = 0; c < nslab; c++) ref += (double)col[(size_t)ab + (size_t)slab * c] * hW[(size_t)c * 16 + n];
        for (int cgi = 0; cgi < ncg; cgi++) got += tg[(size_t)cgi * slab + ab];
        worst = std::max(worst, std::fabs(got - ref) / std::fabs(ref));
      }
    }
    printf("fused T (sum of %d slabs) against the host sum over the stored X: max rel diff %.2e (X is rounded to fp32 before the host sums it)\n",
           ncg, worst);
  }
  return 0;
}
