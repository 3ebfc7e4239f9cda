// place4_bench.hip — follow-up of place2_bench: the speed class of the single-mode scan belongs to the
// SOURCE buffer of a process (profiles/r03l_place2.txt: tensor A fast and B slow in one process, the
// reverse in the next, same virtual alignment every time). Is it a property of the allocation
// (several 6.4 GB buffers side by side) or of the region inside one large allocation (offsets inside
// a 56 GB arena)? "none" = all tiles store to one place (read side only), "flat" = the product's
// result layout into one fixed block.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/place4_bench tools/place4_bench.hip
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
  auto fill = [&](void *p, size_t bytes) {
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (float *)p, (int64_t)(bytes / 4), 7u);
    CK(hipDeviceSynchronize());
  };
  auto arena_pass = [&](size_t gib, const char *tag) {
    char *arena;
    const size_t abytes = gib << 30;
    CK(hipMalloc(&arena, abytes));
    fill(arena, abytes);
    printf("%s: %zu GiB arena at %p\n", tag, gib, (void *)arena);
    const size_t last = (abytes - vbytes) & ~(size_t)((2u << 20) - 1);
    std::vector<size_t> offs = {0};
    for (size_t g = 8; (g << 30) + vbytes <= abytes; g += 8) offs.push_back(g << 30);
    if (last > 0) offs.push_back(last);
    for (size_t off : offs)
      printf("  +%5zu MB: flat %.4f ms  none %.4f ms\n", off >> 20, time_one((const float *)(arena + off), 0),
             time_one((const float *)(arena + off), 1));
    if (abytes >= 2 * vbytes + obytes) {  // the result inside the same arena, right after the tensor
      float *Xs = X;
      X = (float *)(arena + ((vbytes + (2u << 20)) & ~(size_t)((2u << 20) - 1)));
      printf("  result inside the arena: flat %.4f ms\n", time_one((const float *)arena, 0));
      X = Xs;
    }
    CK(hipFree(arena));
  };
  auto separate_pass = [&](int nb, const char *tag) {
    std::vector<float *> buf(nb);
    for (int i = 0; i < nb; i++) {
      CK(hipMalloc(&buf[i], vbytes));
      fill(buf[i], vbytes);
    }
    for (int i = 0; i < nb; i++)
      printf("%s: buffer %d at %p: flat %.4f ms  none %.4f ms\n", tag, i, (void *)buf[i],
             time_one(buf[i], 0), time_one(buf[i], 1));
    for (int i = 0; i < nb; i++) CK(hipFree(buf[i]));
  };
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  if (mode == 0) {  // the large arena FIRST, in a fresh process
    arena_pass(56, "first");
    separate_pass(4, "after the arena");
    for (size_t g : {8, 16, 32}) arena_pass(g, "ladder");
    separate_pass(4, "last");
  } else {  // small first
    separate_pass(4, "first");
    for (size_t g : {7, 8, 12, 16, 24}) arena_pass(g, "ladder");
    arena_pass(56, "large");
    separate_pass(4, "last");
  }
  return 0;
}
