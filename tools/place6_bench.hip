// place6_bench.hip — is the speed class of the single-mode scan a property of the source buffer, of
// the result buffer, or of the pair? Six 6.4 GB sources x six 384 MB result blocks (allocated
// alternately), every pair timed with the product's result layout, plus each source with no result
// stream ("none"). Medians over the rounds; rows = sources, columns = result blocks.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/place6_bench tools/place6_bench.hip
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
  const int K = 200, R = 10, rounds = argc > 1 ? atoi(argv[1]) : 5;
  const int order = argc > 2 ? atoi(argv[2]) : 0;
  const int nt = argc > 3 ? atoi(argv[3]) : 0;  // 1: non-temporal result stores, 2: read side sc1 off, 3: columns 3-5 = blocks 0-2 again with non-temporal stores  // 0: S,X alternately; 1: all S then all X
  const int nblk = (K + 15) / 16;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const size_t vbytes = sizeof(float) * M * K, obytes = sizeof(float) * M * 12;
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int NS = 6, NX = 6;
  float *S[NS], *X[NX], *P;
  CK(hipMalloc(&P, sizeof(float) * (size_t)nblk * 1024));
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)nblk * 1024, 2u);
  if (order == 0) {
    for (int i = 0; i < NS; i++) {
      CK(hipMalloc(&S[i], vbytes));
      CK(hipMalloc(&X[i], obytes));
    }
  } else {
    for (int i = 0; i < NS; i++) CK(hipMalloc(&S[i], vbytes));
    for (int i = 0; i < NX; i++) CK(hipMalloc(&X[i], obytes));
  }
  for (int i = 0; i < NS; i++) hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, S[i], M * K, 1u + i);
  CK(hipDeviceSynchronize());
  for (int i = 0; i < NS; i++) printf("S%d %p   X%d %p\n", i, (void *)S[i], i, (void *)X[i]);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n_mt = (int)((M + 255) / 256);
  const int64_t Lb = 40000, Tb = 200;
  const int n_mtb = (int)((Lb + 255) / 256);
  std::vector<float> ms[NS][NX + 1];
  for (int r = 0; r < rounds + 1; r++)
    for (int i = 0; i < NS; i++)
      for (int j = 0; j <= NX; j++) {
        CK(hipEventRecord(e0, 0));
        const dim3 gridf((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * 40));
        const dim3 gridb((unsigned)std::min<int64_t>((int64_t)n_mtb * Tb, (int64_t)ncu * 40));
        if (j < NX && nt == 3 && j >= 3)
          hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 5>), gridf, dim3(256), 0, 0, S[i], M, (int64_t)K,
                             M * K, P, n_mt, 1, nblk, nblk, (double *)X[j - 3], M, (int64_t)0, (int64_t)0,
                             R, 1, (int64_t)n_mt);
        else if (j < NX && nt == 1)
          hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 5>), gridf, dim3(256), 0, 0, S[i], M, (int64_t)K,
                             M * K, P, n_mt, 1, nblk, nblk, (double *)X[j], M, (int64_t)0, (int64_t)0, R,
                             1, (int64_t)n_mt);
        else if (j < NX && nt == 2)
          hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 0>), gridf, dim3(256), 0, 0, S[i], M, (int64_t)K,
                             M * K, P, n_mt, 1, nblk, nblk, (double *)X[j], M, (int64_t)0, (int64_t)0, R,
                             1, (int64_t)n_mt);
        else if (j < NX)
          hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridf, dim3(256), 0, 0, S[i], M, (int64_t)K,
                             M * K, P, n_mt, 1, nblk, nblk, (double *)X[j], M, (int64_t)0, (int64_t)0, R,
                             1, (int64_t)n_mt);
        else
          hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridb, dim3(256), 0, 0, S[i], Lb, (int64_t)K,
                             Lb * K, P, n_mtb, 1, nblk, nblk, (double *)X[0], Lb, (int64_t)0, (int64_t)0,
                             R, 1, (int64_t)n_mtb * Tb);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) ms[i][j].push_back(t);
      }
  printf("        ");
  for (int j = 0; j < NX; j++) printf("   X%d   ", j);
  printf("  none\n");
  for (int i = 0; i < NS; i++) {
    printf("S%d     ", i);
    for (int j = 0; j <= NX; j++) {
      std::sort(ms[i][j].begin(), ms[i][j].end());
      printf("  %.4f", ms[i][j][ms[i][j].size() / 2]);
    }
    printf("\n");
  }
  return 0;
}
