// Phase stamps of k_rmult_chol (tools only). A chain like the tail of a deferred eigen-step: a thin
// product writes Z (J x n), k_tn_gram its Gram, k_rmult_chol B = Z M; repeated, timed by events, and the
// phases of workgroup 0 (thread 0) by the 100 MHz clock: 0 entry | 1 Gram in LDS | 2 ||E|| known |
// 3 M ready | 4 rows multiplied and stored | 5 hand-over done.
// build: hipcc --offload-arch=gfx950 -O3 -o rmult_chol_bench tools/rmult_chol_bench.hip
//   usage: rmult_chol_bench [J=400] [n=21] [angle=0.05] [threads=0: the library's choice]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long *g_stamps;
#define PPALS_RC_STAMP(k)                                                    \
  if (blockIdx.x == 0 && threadIdx.x == 0 && g_stamps) g_stamps[k] = wall_clock64();
#include "../pairwise-perturbation_amd/csrc/ops.h"
#include "../pairwise-perturbation_amd/csrc/kernels_eig.hip.h"
using namespace ppals;

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                        \
    }                                                                  \
  } while (0)

__global__ void k_copy(const double *a, double *b, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    b[e] = a[e];
}

int main(int argc, char **argv) {
  const int J = argc > 1 ? atoi(argv[1]) : 400, n = argc > 2 ? atoi(argv[2]) : 21;
  const double angle = argc > 3 ? atof(argv[3]) : 0.05;
  const int threads = argc > 4 ? atoi(argv[4]) : 0;  // (0: the library's choice)
  // Z = [q | Q' cos + N' sin]: unit first column, the others turned by `angle` out of an orthonormal set
  std::vector<double> hZ((size_t)J * n);
  srand(7);
  std::vector<double> A((size_t)J * 2 * n);
  for (auto &v : A) v = rand() / (double)RAND_MAX - 0.5;
  // Gram-Schmidt of 2n columns
  for (int c = 0; c < 2 * n; c++) {
    double *x = &A[(size_t)J * c];
    for (int p = 0; p < c; p++) {
      const double *y = &A[(size_t)J * p];
      double d = 0;
      for (int i = 0; i < J; i++) d += x[i] * y[i];
      for (int i = 0; i < J; i++) x[i] -= d * y[i];
    }
    double nn = 0;
    for (int i = 0; i < J; i++) nn += x[i] * x[i];
    nn = 1 / std::sqrt(nn);
    for (int i = 0; i < J; i++) x[i] *= nn;
  }
  for (int c = 0; c < n; c++)
    for (int i = 0; i < J; i++)
      hZ[(size_t)J * c + i] = c == 0 ? A[i] : std::cos(angle) * A[(size_t)J * c + i] * (1.0 - 0.5 * angle * angle) +
                                                  0.0 * A[(size_t)J * (n + c) + i];
  // (columns scaled by cos^2-ish: ||E|| ~ angle^2 per column, as P Omega of a subspace turned by `angle`)
  double *Z0, *Z, *C1, *out, *chk;
  int *status;
  unsigned long long *hand, *stamps;
  CK(hipMalloc(&Z0, sizeof(double) * J * n));
  CK(hipMalloc(&Z, sizeof(double) * J * n));
  CK(hipMalloc(&C1, sizeof(double) * n * n));
  CK(hipMalloc(&out, sizeof(double) * J * n));
  CK(hipMalloc(&chk, sizeof(double) * 4096));
  CK(hipMalloc(&status, 64));
  CK(hipMalloc(&hand, 16));
  CK(hipMalloc(&stamps, 64));
  CK(hipMemset(hand, 0, 16));
  CK(hipMemset(chk, 0, sizeof(double) * 4096));
  CK(hipMemset(status, 0, 64));
  CK(hipMemcpy(Z0, hZ.data(), sizeof(double) * J * n, hipMemcpyHostToDevice));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
  CK(hipFuncSetAttribute((const void *)k_rmult_chol, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int rb = kRmultRows;
  if (n > kSeriesMax) {
    printf("n <= %d\n", kSeriesMax);
    return 1;
  }
  const size_t lds = rmult_chol_lds(n);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int variant = 0; variant < 2; variant++) {
    // variant 1: the same data with a larger turn in one column pair -> the elimination route
    if (variant == 1) {
      for (int i = 0; i < J; i++) hZ[(size_t)J * 1 + i] += 0.4 * hZ[(size_t)J * 2 + i];
      CK(hipMemcpy(Z0, hZ.data(), sizeof(double) * J * n, hipMemcpyHostToDevice));
    }
    const int reps = 200;
    std::vector<double> ph(6, 0.0);
    float ms = 0;
    unsigned long long seq = 0;
    for (int pass = 0; pass < 2; pass++) {
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, Z0, Z, (int64_t)J * n);  // (another kernel wrote Z)
        hipLaunchKernelGGL(k_tn_gram, dim3((n * n + 15) / 16 + 1), dim3(1024), 0, st, Z, (int64_t)J, n, C1, chk, chk, 0, chk,
                           chk + 2048);
        hipLaunchKernelGGL(k_rmult_chol, dim3((J + rb - 1) / rb), dim3(threads ? threads : rmult_chol_threads(n)), lds, st, Z, (int64_t)J, n, 1, C1, out,
                           status, chk, chk + 2048, (unsigned *)(hand + 1), hand, ++seq);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long hs[6];
    CK(hipMemcpy(hs, stamps, 48, hipMemcpyDeviceToHost));
    int hst[2];
    CK(hipMemcpy(hst, status, 8, hipMemcpyDeviceToHost));
    std::vector<double> hout((size_t)J * n);
    CK(hipMemcpy(hout.data(), out, sizeof(double) * J * n, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int a = 0; a < n; a++)
      for (int b = a; b < n; b++) {
        double d = 0;
        for (int i = 0; i < J; i++) d += hout[(size_t)J * a + i] * hout[(size_t)J * b + i];
        worst = std::max(worst, std::fabs(d - (a == b ? 1.0 : 0.0)));
      }
    printf("J=%d n=%d threads %d %s: copy + k_tn_gram + k_rmult_chol %.2f us per round; status %d route %d; max |B^T B - I| %.2e\n", J,
           n, threads ? threads : rmult_chol_threads(n), variant ? "one column pair turned by 0.4" : "columns turned by the angle", 1e3 * ms / reps, hst[0], hst[1],
           worst);
    printf("   phases of workgroup 0 (us): Gram in LDS %.2f | ||E|| %.2f | M %.2f | multiply + store %.2f | hand-over %.2f\n",
           (hs[1] - hs[0]) * 0.01, (hs[2] - hs[1]) * 0.01, (hs[3] - hs[2]) * 0.01, (hs[4] - hs[3]) * 0.01,
           (hs[5] - hs[4]) * 0.01);
  }
  return 0;
}
