// scan2_bench.hip — A/B of the several-n-tile forms of the tensor scan in the single-mode regime of
// the multi-sweep schedule (M rows kept, K = one mode extent reduced, fp32 result): the current
// launcher choice (k_scan_suffix_fast), k_scan_suffix_buf and the k_scan_suffix_lean<NT,U,ACC>
// forms, interleaved in one process, with a cross-check of the results and the register counts.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/scan2_bench tools/scan2_bench.hip
//   run:   tools/scan2_bench [Mrows=4000000] [K=400] [R=20] [rounds=7]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../pairwise-perturbation_amd/csrc/ops.h"
#include "../pairwise-perturbation_amd/csrc/kernels_scan.hip.h"
#include "scan_lean_kernels.hip.h"

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
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)e * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[e] = 0.5f + (float)(h & 0xffff) * (0.5f / 65536.f);
  }
}
__global__ void k_zero_tail(float *P, int nblk, int NT, int K) {
  const int total = nblk * NT * 256 * 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int u = e & 3, g = (e >> 6) & 3, blk = e / (NT * 256);
    if (16 * blk + 4 * u + g >= K) P[e] = 0.f;
  }
}
struct Variant {
  std::string name;
  std::function<void()> launch;
  const void *fn;
  std::vector<float> ms;
};

template <int NT>
int run(int64_t M, int K, int R, int rounds) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int nblk = (K + 15) / 16;
  printf("device %s, %d CUs; M=%lld K=%d R=%d (NT=%d): V %.2f GB, result %.2f GB fp32\n", prop.name,
         ncu, (long long)M, K, R, NT, M * (double)K * 4e-9, M * (double)R * 4e-9);
  float *V, *P, *out, *ref;
  CK(hipMalloc(&V, sizeof(float) * M * K));
  CK(hipMalloc(&P, sizeof(float) * (size_t)nblk * NT * 256 * 4));
  CK(hipMalloc(&out, sizeof(float) * M * 16 * NT));
  CK(hipMalloc(&ref, sizeof(float) * M * 16 * NT));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, V, M * K, 1u);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)nblk * NT * 1024, 2u);
  hipLaunchKernelGGL(k_zero_tail, dim3(64), dim3(256), 0, 0, P, nblk, NT, K);
  CK(hipDeviceSynchronize());
  const int n_mt = (int)((M + 255) / 256);
  std::vector<Variant> vs;
  auto fast = [=](float *o) {
    return [=]() {
      hipLaunchKernelGGL((k_scan_suffix_fast<float, NT, 1>), dim3((unsigned)n_mt), dim3(256), 0, 0, V,
                         M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk, (double *)o, M, (int64_t)0,
                         (int64_t)0, R, 1);
    };
  };
#define PERSIST(KERN, mult)                                                                          \
  [=]() {                                                                                            \
    hipLaunchKernelGGL(KERN, dim3((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * (mult))),         \
                       dim3(256), 0, 0, V, M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk,             \
                       (double *)out, M, (int64_t)0, (int64_t)0, R, 1, (int64_t)n_mt);               \
  }
  vs.push_back({"fast (launcher choice r01)", fast(out), (const void *)k_scan_suffix_fast<float, NT, 1>, {}});
  // XCD-aware block order (each XCD a contiguous eighth of the tile space) and, for the same bytes,
  // the batched shape [L = M/T2 rows | K | T2 batches] the middle-mode roots scan
  vs.push_back({"fast nt+xcd order", [=]() {
                  hipLaunchKernelGGL((k_scan_suffix_fast<float, NT, 3>), dim3((unsigned)n_mt), dim3(256), 0,
                                     0, V, M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk, (double *)out, M,
                                     (int64_t)0, (int64_t)0, R, 1);
                },
                (const void *)k_scan_suffix_fast<float, NT, 3>, {}});
  {
    const int64_t T2 = K, M2 = M / T2;  // e.g. 64e6 rows = 160000 x 400 batches
    const int n_mt2 = (int)((M2 + 255) / 256);
    vs.push_back({"fast batched shape", [=]() {
                    hipLaunchKernelGGL((k_scan_suffix_fast<float, NT, 1>), dim3((unsigned)(n_mt2 * T2)),
                                       dim3(256), 0, 0, V, M2, (int64_t)K, M2 * K, P, n_mt2, 1, nblk, nblk,
                                       (double *)out, M2 * T2, (int64_t)0, M2, R, 1);
                  },
                  (const void *)k_scan_suffix_fast<float, NT, 1>, {}});
  }
  vs.push_back({"fast, MFMAs of tile 0 only (wrong result: timing)", [=]() {
                  hipLaunchKernelGGL((k_scan_suffix_fast<float, NT, 17>), dim3((unsigned)n_mt), dim3(256), 0,
                                     0, V, M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk, (double *)out, M,
                                     (int64_t)0, (int64_t)0, R, 1);
                },
                (const void *)k_scan_suffix_fast<float, NT, 17>, {}});
  vs.push_back({"fast, non-temporal stores", [=]() {
                  hipLaunchKernelGGL((k_scan_suffix_fast<float, NT, 5>), dim3((unsigned)n_mt), dim3(256), 0,
                                     0, V, M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk, (double *)out, M,
                                     (int64_t)0, (int64_t)0, R, 1);
                },
                (const void *)k_scan_suffix_fast<float, NT, 5>, {}});
  vs.push_back({"buf  U4 fp64 x40", PERSIST((k_scan_suffix_buf<float, NT, 1>), 40), (const void *)k_scan_suffix_buf<float, NT, 1>, {}});
#define LEAN(NAME, U_, ACC_, W_, mult)                                                      \
  vs.push_back({NAME, PERSIST((k_scan_suffix_lean<NT, U_, ACC_, W_>), mult),                \
                (const void *)k_scan_suffix_lean<NT, U_, ACC_, W_>, {}})
  LEAN("lean U4 fp64  w1 x40", 4, 0, 1, 40);
  LEAN("lean U2 fp64  w1 x40", 2, 0, 1, 40);
  LEAN("lean U2 fp64  w3 x40", 2, 0, 3, 40);
  LEAN("lean U2 fp64  w3 x24", 2, 0, 3, 24);
  LEAN("lean U2 fp64  w3 x12", 2, 0, 3, 12);
  LEAN("lean U4 fp64  w3 x40", 4, 0, 3, 40);
  LEAN("lean U4 2xf32 w3 x40", 4, 1, 3, 40);
  LEAN("lean U4 2xf32 w3 x12", 4, 1, 3, 12);
  LEAN("lean U2 2xf32 w1 x40", 2, 1, 1, 40);
  LEAN("lean U2 2xf32 w3 x40", 2, 1, 3, 40);
  LEAN("lean U2 2xf32 w4 x40", 2, 1, 4, 40);
  fast(ref)();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vs) {
      CK(hipEventRecord(e0, 0));
      v.launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
    }
  const double bytes = (double)M * K * 4.0 + (double)M * R * 4.0;
  const int64_t NCHK = std::min<int64_t>(M * R, 1 << 20);
  std::vector<float> hr(NCHK), ho(NCHK);
  CK(hipMemcpy(hr.data(), ref, sizeof(float) * NCHK, hipMemcpyDeviceToHost));
  printf("%-28s %6s %6s %9s %9s %10s %6s %12s\n", "variant", "vgpr", "waves", "med ms", "min ms",
         "GB/s(med)", "frac", "max rel diff");
  for (auto &v : vs) {
    CK(hipMemset(out, 0, sizeof(float) * M * 16 * NT));
    v.launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho.data(), out, sizeof(float) * NCHK, hipMemcpyDeviceToHost));
    double maxrel = 0;
    for (int64_t i = 0; i < NCHK; i++)
      maxrel = std::max(maxrel, (double)fabsf(ho[i] - hr[i]) / (fabs((double)hr[i]) + 1e-300));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, v.fn));
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, v.fn, 256, 0));
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
    printf("%-28s %6d %6d %9.4f %9.4f %10.1f %6.3f %12.3e\n", v.name.c_str(), fa.numRegs, nb, med, mn,
           bytes / (med * 1e-3) / 1e9, bytes / (med * 1e-3) / 8e12, maxrel);
  }
  return 0;
}

int main(int argc, char **argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 4000000;
  const int K = argc > 2 ? atoi(argv[2]) : 400;
  const int R = argc > 3 ? atoi(argv[3]) : 20;
  const int rounds = argc > 4 ? atoi(argv[4]) : 7;
  if (R <= 16) return run<1>(M, K, R, rounds);
  if (R <= 32) return run<2>(M, K, R, rounds);
  return run<4>(M, K, R, rounds);
}
