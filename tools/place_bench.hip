// place_bench.hip — does the placement of the tensor relative to the result buffer change the
// speed of the single-mode scan? (Inside a sweep the two scans of the tensor's own buffer are ~8 %
// slower than the two of the second resident layout: same kernel, same shapes.) Allocates
// A, OUT1, B, OUT2 in that order and times k_scan_suffix_buf<float,1,1> for every (tensor, result)
// pair, plus results placed at offsets inside one large block.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/place_bench tools/place_bench.hip
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
__global__ void k_fill(float *p, int64_t n, uint32_t seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)e * 2654435761u ^ seed;
    h ^= h >> 15;
    p[e] = 0.5f + (float)(h & 0xffff) * (0.5f / 65536.f);
  }
}
int main(int argc, char **argv) {
  const int64_t M = 8000000;
  const int K = 200, R = 10, rounds = argc > 1 ? atoi(argv[1]) : 7;
  const int nblk = (K + 15) / 16;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float *A, *B, *P, *O1, *O2, *BIG;
  const size_t vbytes = sizeof(float) * M * K, obytes = sizeof(float) * M * 16;
  CK(hipMalloc(&A, vbytes));
  CK(hipMalloc(&O1, obytes));
  CK(hipMalloc(&B, vbytes));
  CK(hipMalloc(&O2, obytes));
  CK(hipMalloc(&BIG, obytes + (512u << 20)));
  CK(hipMalloc(&P, sizeof(float) * (size_t)nblk * 1024));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, A, M * K, 1u);
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, B, M * K, 3u);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)nblk * 1024, 2u);
  CK(hipDeviceSynchronize());
  printf("A %p O1 %p B %p O2 %p BIG %p\n", (void *)A, (void *)O1, (void *)B, (void *)O2, (void *)BIG);
  const int n_mt = (int)((M + 255) / 256);
  struct Var {
    std::string name;
    const float *v;
    float *o;
    std::vector<float> ms;
  };
  std::vector<Var> vs;
  vs.push_back({"read A -> O1", A, O1, {}});
  vs.push_back({"read A -> O2", A, O2, {}});
  vs.push_back({"read B -> O1", B, O1, {}});
  vs.push_back({"read B -> O2", B, O2, {}});
  std::vector<size_t> offs;
  for (size_t k = 0; k <= 32; k++) offs.push_back(k * (256u << 10));          // 0 .. 8 MB step 256 KB
  for (size_t mb : {12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 500}) offs.push_back(mb << 20);
  for (size_t off : offs)
    vs.push_back({"read A -> BIG+" + std::to_string(off >> 10) + "K", A, (float *)((char *)BIG + off), {}});
  for (size_t off : {size_t(0), size_t(3) << 20, size_t(5) << 20})
    vs.push_back({"read B -> BIG+" + std::to_string(off >> 10) + "K", B, (float *)((char *)BIG + off), {}});
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vs) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>),
                         dim3((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * 40)), dim3(256), 0, 0, v.v,
                         M, (int64_t)K, M * K, P, n_mt, 1, nblk, nblk, (double *)v.o, M, (int64_t)0,
                         (int64_t)0, R, 1, (int64_t)n_mt);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
    }
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    printf("%-28s med %.4f ms  min %.4f  %.0f GB/s\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.ms[0],
           (M * (double)K * 4 + M * (double)R * 4) / (v.ms[v.ms.size() / 2] * 1e-3) / 1e9);
  }
  return 0;
}
