// f64_bench.hip — the fp64-storage scan (the reference's precision, the drivers' default) in the
// single-mode regime of the multi-sweep schedule: persistent buffer-load kernel vs global-load
// kernel, with the fp64 result, an fp32 result and no result stores at all, against the fp32 tensor
// of the same byte count.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/f64_bench tools/f64_bench.hip
//   run:   tools/f64_bench [GB=12.8] [K=200] [R=10] [rounds=7]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
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

template <typename T>
__global__ void k_fill(T *p, int64_t n, uint32_t seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)e * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[e] = (T)(0.5 + (double)(h & 0xffff) * (0.5 / 65536.0));
  }
}

struct Var {
  std::string name;
  std::function<void()> go;
  double bytes;
  std::vector<float> ms;
};

int main(int argc, char **argv) {
  const double GB = argc > 1 ? atof(argv[1]) : 12.8;
  const int K = argc > 2 ? atoi(argv[2]) : 200;
  const int R = argc > 3 ? atoi(argv[3]) : 10;
  const int rounds = argc > 4 ? atoi(argv[4]) : 7;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int64_t Md = (int64_t)(GB * 1e9 / 8 / K) / 256 * 256;  // rows of the fp64 tensor
  const int64_t Mf = 2 * Md;                                   // fp32 tensor of the same bytes
  void *V, *P, *out;
  CK(hipMalloc(&V, (size_t)Md * K * 8));
  CK(hipMalloc(&P, (size_t)((K + 7) / 8) * 1024 * 8));
  CK(hipMalloc(&out, (size_t)Mf * 16 * 8));
  hipLaunchKernelGGL(k_fill<double>, dim3(8192), dim3(256), 0, 0, (double *)V, Md * K, 1u);
  hipLaunchKernelGGL(k_fill<double>, dim3(64), dim3(256), 0, 0, (double *)P, (int64_t)((K + 7) / 8) * 128, 2u);
  CK(hipDeviceSynchronize());
  std::vector<Var> vs;
  auto add_d = [&](const char *name, int fast, int ncols, int out32) {
    const int nblk = (K + 7) / 8, n_mt = (int)((Md + 127) / 128);
    const double bytes = (double)Md * K * 8 + (double)Md * ncols * (out32 ? 4 : 8);
    if (fast)
      vs.push_back({name, [=]() {
                      hipLaunchKernelGGL((k_scan_suffix_fast<double, 1, 1>), dim3((unsigned)n_mt), dim3(256),
                                         0, 0, (const double *)V, Md, (int64_t)K, Md * K, (const double *)P,
                                         n_mt, 1, nblk, nblk, (double *)out, Md, (int64_t)0, (int64_t)0,
                                         ncols, out32);
                    }, bytes, {}});
    else
      vs.push_back({name, [=]() {
                      hipLaunchKernelGGL((k_scan_suffix_buf<double, 1, 1>),
                                         dim3((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * 40)),
                                         dim3(256), 0, 0, (const double *)V, Md, (int64_t)K, Md * K,
                                         (const double *)P, n_mt, 1, nblk, nblk, (double *)out, Md,
                                         (int64_t)0, (int64_t)0, ncols, out32, (int64_t)n_mt);
                    }, bytes, {}});
  };
  auto add_f = [&](const char *name, int fast, int ncols, int out32) {
    const int nblk = (K + 15) / 16, n_mt = (int)((Mf + 255) / 256);
    const double bytes = (double)Mf * K * 4 + (double)Mf * ncols * (out32 ? 4 : 8);
    if (fast)
      vs.push_back({name, [=]() {
                      hipLaunchKernelGGL((k_scan_suffix_fast<float, 1, 1>), dim3((unsigned)n_mt), dim3(256),
                                         0, 0, (const float *)V, Mf, (int64_t)K, Mf * K, (const float *)P,
                                         n_mt, 1, nblk, nblk, (double *)out, Mf, (int64_t)0, (int64_t)0,
                                         ncols, out32);
                    }, bytes, {}});
    else
      vs.push_back({name, [=]() {
                      hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>),
                                         dim3((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * 40)),
                                         dim3(256), 0, 0, (const float *)V, Mf, (int64_t)K, Mf * K,
                                         (const float *)P, n_mt, 1, nblk, nblk, (double *)out, Mf,
                                         (int64_t)0, (int64_t)0, ncols, out32, (int64_t)n_mt);
                    }, bytes, {}});
  };
  add_d("f64 buf  result f64", 0, R, 0);
  add_d("f64 fast result f64", 1, R, 0);
  add_d("f64 buf  result f32", 0, R, 1);
  add_d("f64 fast result f32", 1, R, 1);
  add_d("f64 buf  no stores", 0, 0, 0);
  add_d("f64 fast no stores", 1, 0, 0);
  add_f("f32 buf  result f32", 0, R, 1);
  add_f("f32 fast result f32", 1, R, 1);
  add_f("f32 buf  result f64", 0, R, 0);
  add_f("f32 buf  no stores", 0, 0, 1);
  add_f("f32 fast no stores", 1, 0, 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vs) {
      CK(hipEventRecord(e0, 0));
      v.go();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) v.ms.push_back(t);
    }
  printf("device %s, %d CUs; %.2f GB tensor, K = %d, R = %d\n", prop.name, ncu, (double)Md * K * 8e-9, K, R);
  printf("%-24s %9s %9s %6s\n", "variant", "med ms", "GB/s", "frac");
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2];
    printf("%-24s %9.4f %9.1f %6.3f\n", v.name.c_str(), med, v.bytes / (med * 1e-3) / 1e9,
           v.bytes / (med * 1e-3) / 8e12);
  }
  return 0;
}
