// nsprod_bench.hip — one product of the Newton-Schulz sign iteration (J x J x J, fp64 MFMA, symmetric
// result) as a chain of dependent launches: what a product costs by the number of waves that split
// K inside a workgroup, against the non-symmetric tile kernel.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/nsprod_bench tools/nsprod_bench.hip
//   run:   tools/nsprod_bench [J=400]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../pairwise-perturbation_amd/csrc/ops.h"
#include "../pairwise-perturbation_amd/csrc/kernels_eig.hip.h"
using namespace ppals;
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <int NW>
static void chain(const char *name, double *X, double *Y, int J, int n) {
  const unsigned nt = (unsigned)((J + 15) / 16), ntri = nt * (nt + 1) / 2;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int rep = 0; rep < 7; rep++) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; i++) {
      hipLaunchKernelGGL(k_dgemm_nt_sym<NW>, dim3(ntri), dim3(64 * NW), 0, 0, X, (int64_t)J, X, (int64_t)J,
                         (const double *)nullptr, (int64_t)0, Y, (int64_t)J, J, J, 1e-3, 0.0, 0,
                         (double *)nullptr);
      std::swap(X, Y);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1e3f / n);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-34s J=%d: %.2f us per product (chain of %d, median of 7)\n", name, J, ts[3], n);
}

int main(int argc, char **argv) {
  const int J = argc > 1 ? atoi(argv[1]) : 400;
  double *X, *Y;
  CK(hipMalloc(&X, sizeof(double) * J * J));
  CK(hipMalloc(&Y, sizeof(double) * J * J));
  std::vector<double> h((size_t)J * J);
  for (int i = 0; i < J; i++)
    for (int j = 0; j < J; j++) h[i + (size_t)J * j] = (i == j ? 1.0 : 1e-3 * ((i * 7 + j * 13) % 11 - 5));
  CK(hipMemcpy(X, h.data(), sizeof(double) * J * J, hipMemcpyHostToDevice));
  chain<4>("k_dgemm_nt_sym<4>  (256 threads)", X, Y, J, 40);
  chain<8>("k_dgemm_nt_sym<8>  (512 threads)", X, Y, J, 40);
  chain<16>("k_dgemm_nt_sym<16> (1024 threads)", X, Y, J, 40);
  {  // the non-symmetric tile kernel on the full matrix, for scale
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int n = 40;
    dim3 grid((unsigned)((J + 15) / 16), (unsigned)((J + 15) / 16));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; i++) {
      hipLaunchKernelGGL(k_dgemm_nx<false>, grid, dim3(256), 0, 0, X, (int64_t)J, X, (int64_t)J,
                         (const double *)nullptr, (int64_t)0, Y, (int64_t)J, J, J, J, 1e-3, 0.0);
      std::swap(X, Y);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s J=%d: %.2f us per product\n", "k_dgemm_nx<false> (all tiles)", J, ms * 1e3f / n);
  }
  {  // an empty-ish kernel chain: the floor of a dependent launch on this box
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int n = 200;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; i++)
      hipLaunchKernelGGL(k_chk_sums, dim3(1), dim3(256), 0, 0, X, X, 1, Y);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s       %.2f us per launch\n", "trivial one-workgroup kernel chain", ms * 1e3f / n);
  }
  return 0;
}
