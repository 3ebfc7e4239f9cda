// stride_bench.hip — what a column stride that is not a multiple of 128 B costs the tensor scan.
// The hot-path kernel (k_scan_suffix_buf<float,1,1>, persistent) on batched shapes [M rows kept |
// K reduced | T batches] of the SAME byte count, with M a power of the reference scripts' mode
// extent (s = 50: 2500 / 125000 / 6250000 rows, 16-B aligned columns only) against M rounded up to
// 128 B and against a buffer shifted by 16 … 64 B.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/stride_bench tools/stride_bench.hip
//   run:   tools/stride_bench [GB=20] [R=6] [rounds=5]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
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
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)e * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[e] = 0.5f + (float)(h & 0xffff) * (0.5f / 65536.f);
  }
}

struct Case {
  const char *name;
  int64_t M;
  int K;
  int64_t T;
  int shift;  // floats added to the buffer base (4 = 16 B)
  int fast;   // 1: the global-load kernel instead of the persistent buffer-load one
};

int main(int argc, char **argv) {
  const double GB = argc > 1 ? atof(argv[1]) : 20.0;
  const int R = argc > 2 ? atoi(argv[2]) : 6;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int64_t elems = (int64_t)(GB * 1e9 / 4);
  auto batches = [&](int64_t M, int K) { return std::max<int64_t>(1, elems / (M * K)); };
  std::vector<Case> cs = {
      {"s^2 rows   2500 (16 B)", 2500, 2500, 0, 0, 0},
      {"s^2 rows   2528 (128 B)", 2528, 2500, 0, 0, 0},
      {"s^2 rows   2560 (tile)", 2560, 2500, 0, 0, 0},
      {"s^2 rows   2560 +16 B", 2560, 2500, 0, 4, 0},
      {"s^2 rows   2560 +64 B", 2560, 2500, 0, 16, 0},
      {"s^2 rows   2500 fast", 2500, 2500, 0, 0, 1},
      {"s^2 rows   2560 fast", 2560, 2500, 0, 0, 1},
      {"s^3 rows 125000 (32 B)", 125000, 2500, 0, 0, 0},
      {"s^3 rows 125024 (128 B)", 125024, 2500, 0, 0, 0},
      {"s^3 rows 125184 (tile)", 125184, 2500, 0, 0, 0},
      {"s^3 rows 125000 fast", 125000, 2500, 0, 0, 1},
      {"s^4 rows 6250000 (64 B)", 6250000, 2500, 1, 0, 0},
      {"s^4 rows 6250016 (128 B)", 6250016, 2500, 1, 0, 0},
      {"324^2 rows (64 B)", 104976, 324, 0, 0, 0},
      {"324^2 rows 104992 (128B)", 104992, 324, 0, 0, 0},
      {"324 rows (16 B)", 324, 324, 0, 0, 0},
      {"352 rows (128 B)", 352, 324, 0, 0, 0},
      {"200^2 rows (cfg2)", 40000, 200, 0, 0, 0},
  };
  size_t maxV = 0, maxO = 0;
  int maxnblk = 0;
  for (auto &c : cs) {
    if (c.T == 0) c.T = batches(c.M, c.K);
    if (c.T == 1 && c.M * (int64_t)c.K > elems) c.M = (elems / c.K) / 32 * 32 + (c.M % 32);
    maxV = std::max(maxV, (size_t)c.M * c.K * c.T + 64);
    maxO = std::max(maxO, (size_t)c.M * c.T * R);
    maxnblk = std::max(maxnblk, (c.K + 15) / 16);
  }
  float *V, *P, *out;
  CK(hipMalloc(&V, sizeof(float) * maxV));
  CK(hipMalloc(&P, sizeof(float) * (size_t)maxnblk * 1024));
  CK(hipMalloc(&out, sizeof(float) * maxO));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, V, (int64_t)maxV, 1u);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)maxnblk * 1024, 2u);
  CK(hipDeviceSynchronize());
  printf("device %s, %d CUs; ~%.1f GB per scan, R = %d, fp32 result\n", prop.name, ncu, GB, R);
  printf("%-26s %9s %6s %9s %9s %9s %6s\n", "shape", "M", "K", "T", "med ms", "GB/s", "frac");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> ms(cs.size());
  for (int r = 0; r < rounds + 1; r++)
    for (size_t ci = 0; ci < cs.size(); ci++) {
      const Case &c = cs[ci];
      const int64_t M = c.M, T = c.T;
      const int K = c.K, nblk = (K + 15) / 16;
      const int n_mt = (int)((M + 255) / 256);
      const int64_t ntiles = (int64_t)n_mt * T;
      const float *Vs = V + c.shift;
      CK(hipEventRecord(e0, 0));
      if (c.fast)
        hipLaunchKernelGGL((k_scan_suffix_fast<float, 1, 1>), dim3((unsigned)ntiles), dim3(256), 0, 0,
                           Vs, M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk, (double *)out, M * T,
                           (int64_t)0, M, R, 1);
      else
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>),
                           dim3((unsigned)std::min<int64_t>(ntiles, (int64_t)ncu * 40)), dim3(256), 0,
                           0, Vs, M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk, (double *)out, M * T,
                           (int64_t)0, M, R, 1, ntiles);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[ci].push_back(t);
    }
  for (size_t ci = 0; ci < cs.size(); ci++) {
    const Case &c = cs[ci];
    std::sort(ms[ci].begin(), ms[ci].end());
    const float med = ms[ci][ms[ci].size() / 2];
    const double bytes = (double)c.M * c.K * c.T * 4.0 + (double)c.M * c.T * R * 4.0;
    printf("%-26s %9lld %6d %9lld %9.4f %9.1f %6.3f\n", c.name, (long long)c.M, c.K, (long long)c.T,
           med, bytes / (med * 1e-3) / 1e9, bytes / (med * 1e-3) / 8e12);
  }
  return 0;
}
