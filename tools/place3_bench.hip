// place3_bench.hip — follow-up of place2_bench: the speed class of the single-mode scan belongs to the
// SOURCE buffer of a process (profiles/r03l_place2.txt: tensor A fast and B slow in one process, the
// reverse in the next, same virtual alignment every time). Is it a property of the allocation
// (several 6.4 GB buffers side by side) or of the region inside one large allocation (offsets inside
// a 56 GB arena)? "none" = all tiles store to one place (read side only), "flat" = the product's
// result layout into one fixed block.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/place3_bench tools/place3_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
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
  const size_t vbytes = sizeof(float) * M * K, obytes = sizeof(float) * M * 16;
  float *P, *X;
  CK(hipMalloc(&X, obytes));
  CK(hipMalloc(&P, sizeof(float) * (size_t)nblk * 1024));
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)nblk * 1024, 2u);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n_mt = (int)((M + 255) / 256);
  const int64_t Lb = 40000, Tb = 200;
  const int n_mtb = (int)((Lb + 255) / 256);
  auto time_one = [&](const float *v, int kind) {
    std::vector<float> ms;
    for (int r = 0; r < rounds + 1; r++) {
      CK(hipEventRecord(e0, 0));
      const dim3 gridf((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * 40));
      const dim3 gridb((unsigned)std::min<int64_t>((int64_t)n_mtb * Tb, (int64_t)ncu * 40));
      if (kind == 0)
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridf, dim3(256), 0, 0, v, M, (int64_t)K,
                           M * K, P, n_mt, 1, nblk, nblk, (double *)X, M, (int64_t)0, (int64_t)0, R, 1,
                           (int64_t)n_mt);
      else
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridb, dim3(256), 0, 0, v, Lb, (int64_t)K,
                           Lb * K, P, n_mtb, 1, nblk, nblk, (double *)X, Lb, (int64_t)0, (int64_t)0, R,
                           1, (int64_t)n_mtb * Tb);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
  };
  const int NB = 8;
  float *buf[NB];
  for (int i = 0; i < NB; i++) {
    CK(hipMalloc(&buf[i], vbytes));
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, buf[i], M * K, 1u + i);
  }
  CK(hipDeviceSynchronize());
  for (int i = 0; i < NB; i++) {
    const float tf = time_one(buf[i], 0), tn = time_one(buf[i], 1);
    printf("buffer %d at %p: flat %.4f ms  none %.4f ms\n", i, (void *)buf[i], tf, tn);
  }
  for (int i = 0; i < NB; i++) CK(hipFree(buf[i]));
  char *arena;
  const size_t abytes = (size_t)56 << 30;
  CK(hipMalloc(&arena, abytes));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (float *)arena, (int64_t)(abytes / 4), 7u);
  CK(hipDeviceSynchronize());
  printf("arena at %p\n", (void *)arena);
  std::vector<size_t> offs;
  for (size_t g = 0; g <= 48; g += 2) offs.push_back(g << 30);
  for (size_t mb : {512, 1024, 1536, 3072, 5120}) offs.push_back(mb << 20);
  for (size_t off : offs) {
    const float tf = time_one((const float *)(arena + off), 0), tn = time_one((const float *)(arena + off), 1);
    printf("arena +%5zu MB: flat %.4f ms  none %.4f ms\n", off >> 20, tf, tn);
  }
  return 0;
}
