// place2_bench.hip — why is the single-mode scan of the multi-sweep schedule 1.06 ms in one process
// and 1.22 ms in the next on the same box (profiles/r03j_place_lottery.txt), whatever the offset
// of its result inside a 2 GB window? The read side alone is steady (the dt scans, which write
// almost nothing, run at 0.99 ms in every session), so this times the same kernel with the result
// laid out three ways:
//   flat     X[r][m]            R pieces of 1 KB per workgroup tile, 4*M bytes apart (the product's)
//   blocked  X[m/256][r][m%256] one contiguous 256*R*4-byte piece per workgroup tile
//   none     every tile stores to the same 10 KB (no result stream at all: the floor)
// and the batched shape (L = 40000, T = 200) as  X[r][t][l]  (product) against  X[t][r][l].
// Run it in several processes: the question is the spread BETWEEN processes per layout.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/place2_bench tools/place2_bench.hip
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
  const int K = 200, R = 10, rounds = argc > 1 ? atoi(argv[1]) : 9;
  const int nblk = (K + 15) / 16;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  float *A, *B, *P, *X;
  const size_t vbytes = sizeof(float) * M * K, obytes = sizeof(float) * M * 16;
  // the same order of allocations as a session: tensor, second layout, result block
  CK(hipMalloc(&A, vbytes));
  CK(hipMalloc(&B, vbytes));
  CK(hipMalloc(&X, obytes + (256u << 20)));
  CK(hipMalloc(&P, sizeof(float) * (size_t)nblk * 1024));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, A, M * K, 1u);
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, B, M * K, 3u);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)nblk * 1024, 2u);
  CK(hipDeviceSynchronize());
  printf("A %p B %p X %p\n", (void *)A, (void *)B, (void *)X);
  struct Var {
    std::string name;
    int kind;  // 0 flat, 1 blocked, 2 none, 3 batched [r][t][l], 4 batched [t][r][l]
    const float *v;
    float *o;
    std::vector<float> ms;
  };
  std::vector<Var> vs;
  for (const float *src : {(const float *)A, (const float *)B})
    for (size_t mb : {0, 16, 64, 192}) {
      float *o = (float *)((char *)X + (mb << 20));
      const std::string tag = std::string(src == A ? "A" : "B") + " -> X+" + std::to_string(mb) + "M";
      vs.push_back({"flat     " + tag, 0, src, o, {}});
      vs.push_back({"blocked  " + tag, 1, src, o, {}});
      vs.push_back({"[r][t][l] " + tag, 3, src, o, {}});
      vs.push_back({"[t][r][l] " + tag, 4, src, o, {}});
    }
  vs.push_back({"none     A", 2, A, X, {}});
  vs.push_back({"none     B", 2, B, X, {}});
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n_mt = (int)((M + 255) / 256);
  const int64_t Lb = 40000, Tb = 200;
  const int n_mtb = (int)((Lb + 255) / 256);
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vs) {
      CK(hipEventRecord(e0, 0));
      const dim3 gridf((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * 40));
      const dim3 gridb((unsigned)std::min<int64_t>((int64_t)n_mtb * Tb, (int64_t)ncu * 40));
      if (v.kind == 0)
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridf, dim3(256), 0, 0, v.v, M, (int64_t)K,
                           M * K, P, n_mt, 1, nblk, nblk, (double *)v.o, M, (int64_t)0, (int64_t)0, R, 1,
                           (int64_t)n_mt);
      else if (v.kind == 1)
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 3>), gridf, dim3(256), 0, 0, v.v, M, (int64_t)K,
                           M * K, P, n_mt, 1, nblk, nblk, (double *)v.o, M, (int64_t)0, (int64_t)0, R, 1,
                           (int64_t)n_mt);
      else if (v.kind == 2)  // every tile a batch of its own with stride 0: all store to one place
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridb, dim3(256), 0, 0, v.v, Lb, (int64_t)K,
                           Lb * K, P, n_mtb, 1, nblk, nblk, (double *)v.o, Lb, (int64_t)0, (int64_t)0, R,
                           1, (int64_t)n_mtb * Tb);
      else if (v.kind == 3)
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridb, dim3(256), 0, 0, v.v, Lb, (int64_t)K,
                           Lb * K, P, n_mtb, 1, nblk, nblk, (double *)v.o, Lb * Tb, (int64_t)0, Lb, R, 1,
                           (int64_t)n_mtb * Tb);
      else
        hipLaunchKernelGGL((k_scan_suffix_buf<float, 1, 1>), gridb, dim3(256), 0, 0, v.v, Lb, (int64_t)K,
                           Lb * K, P, n_mtb, 1, nblk, nblk, (double *)v.o, Lb, (int64_t)0, Lb * R, R, 1,
                           (int64_t)n_mtb * Tb);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
    }
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    printf("%-28s med %.4f ms  min %.4f\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.ms[0]);
  }
  return 0;
}
