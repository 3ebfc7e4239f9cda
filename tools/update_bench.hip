// update_bench.hip — latency of the fused CP mode update (k_cp_mode_update) by block size and
// staging, R and rows as in cfg2 / cfg4; 200 back-to-back launches per variant.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/update_bench tools/update_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define PPALS_UPDATE_STAMPS 1
#include "../pairwise-perturbation_amd/csrc/ops.h"
#include "../pairwise-perturbation_amd/csrc/kernels_small.hip.h"
using namespace ppals;
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)
int main() {
  CK(hipFuncSetAttribute((const void *)k_cp_mode_update<false, false>,
                         hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  CK(hipFuncSetAttribute((const void *)k_cp_mode_update<true, false>,
                         hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  CK(hipFuncSetAttribute((const void *)k_cp_mode_update<true, true>,
                         hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  for (int cfg = 0; cfg < 2; cfg++) {
    const int R = cfg ? 20 : 10, N = 4;
    const int64_t rows = cfg ? 400 : 200;
    std::vector<double> hW(rows * R), hG(N * R * R, 0.0);
    for (auto &x : hW) x = 0.1 + (rand() % 1000) * 1e-3;
    for (int m = 0; m < N; m++)
      for (int i = 0; i < R; i++)
        for (int j = 0; j < R; j++) hG[m * R * R + i + R * j] = (i == j ? 2.0 : 0.3);
    double *G, *M, *W, *grad, *gs, *S, *Si;
    CK(hipMalloc(&G, sizeof(double) * N * R * R));
    CK(hipMalloc(&M, sizeof(double) * rows * R));
    CK(hipMalloc(&W, sizeof(double) * rows * R));
    CK(hipMalloc(&grad, sizeof(double) * rows * R));
    CK(hipMalloc(&gs, 64));
    CK(hipMalloc(&S, sizeof(double) * R * R));
    CK(hipMalloc(&Si, sizeof(double) * R * R));
    CK(hipMemcpy(M, hW.data(), sizeof(double) * rows * R, hipMemcpyHostToDevice));
    const size_t lds = sizeof(double) * (32 + 2 * (size_t)R * R + 2 * (size_t)R * (R + 1) + 64) + sizeof(int) * 64;
    const size_t stage = 2 * sizeof(double) * (size_t)rows * R;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // variants: 0 = no LDS staging, 1 = staged VALU loops (round 2), 2 = staged + the two row
    // products on the matrix cores, 3 = 2 + S / S^-1 handed in (prepared by the previous launch)
    for (int var = 3; var >= 0; var--)
      for (int bs : {256, 512, 1024}) {
        // one launch at a time from pristine inputs (repeating the update on its own output
        // drives the Grams out of range and into the Jacobi fallback): median of 40
        std::vector<float> ts;
        for (int it = 0; it < 40; it++) {
          CK(hipMemcpy(G, hG.data(), sizeof(double) * N * R * R, hipMemcpyHostToDevice));
          CK(hipMemcpy(W, hW.data(), sizeof(double) * rows * R, hipMemcpyHostToDevice));
          if (var == 3) {  // a valid system in S / Si
            hipLaunchKernelGGL(k_gram_system, dim3(1), dim3(64),
                               sizeof(double) * (2 * (size_t)R * (R + 1) + 64) + sizeof(int) * 64, 0, G, N,
                               it % N, R, 0.0, S, Si, 0);
          }
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, 0));
          if (var >= 2)
            hipLaunchKernelGGL((k_cp_mode_update<true, true>), dim3(1), dim3(bs), lds + stage, 0, G, N,
                               it % N, R, 0.0, M, rows, W, rows, grad, rows, rows, gs,
                               (const double *)nullptr, rows, (double *)nullptr, rows, 1.0, S, Si,
                               (double *)nullptr, var == 3 ? 1 : 0);
          else if (var == 1)
            hipLaunchKernelGGL((k_cp_mode_update<true, false>), dim3(1), dim3(bs), lds + stage, 0, G, N,
                               it % N, R, 0.0, M, rows, W, rows, grad, rows, rows, gs,
                               (const double *)nullptr, rows, (double *)nullptr, rows, 1.0, S, Si,
                               (double *)nullptr, 0);
          else
            hipLaunchKernelGGL((k_cp_mode_update<false, false>), dim3(1), dim3(bs), lds, 0, G, N, it % N, R,
                               0.0, M, rows, W, rows, grad, rows, rows, gs, (const double *)nullptr, rows,
                               (double *)nullptr, rows, 1.0, S, Si, (double *)nullptr, 0);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          CK(hipGetLastError());
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          ts.push_back(ms);
          if (it == 39 && var == 3) {  // the phase stamps of the last launch (100 MHz clock: 10 ns ticks)
            unsigned long long st[16];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_update_stamps), sizeof(st)));
            printf("  phases of thread 0 (us since kernel entry): staged %.2f | grad tiles %.2f | block sum %.2f | "
                   "solve tiles %.2f | barrier %.2f | Gram %.2f\n",
                   (st[1] - st[0]) * 0.01, (st[2] - st[0]) * 0.01, (st[3] - st[0]) * 0.01, (st[4] - st[0]) * 0.01,
                   (st[5] - st[0]) * 0.01, (st[6] - st[0]) * 0.01);
          }
        }
        std::sort(ts.begin(), ts.end());
        printf("R=%d rows=%lld variant=%d block=%4d: %.2f us per launch (single launch between events, median)\n",
               R, (long long)rows, var, bs, ts[ts.size() / 2] * 1e3);
      }
  }
  return 0;
}
