// mfma64_rate.hip — what v_mfma_f64_16x16x4_f64 sustains on this part: W waves per SIMD each issue a
// long run of MFMAs on C independent accumulators; prints TFLOP/s (2048 flop per instruction).
// The residual kernel K10 (k_rank_mfma) spends 12 of them per 4 KB of tensor: its floor follows.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma64_rate tools/mfma64_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)
template <int C>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
  f64x4 acc[C];
#pragma unroll
  for (int c = 0; c < C; c++) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678) out[0] = s;  // never: keeps the chain alive
}
template <int C>
void run(int wg_per_cu, int ncu, double *out) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mfma<C>, dim3(ncu * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0, 1e-3);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  const double n = (double)ncu * wg_per_cu * 4 * iters * C;  // instructions
  printf("%d independent accumulators, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s, %.1f ns per instruction per SIMD\n",
         C, wg_per_cu, best, n * 2048 / (best * 1e-3) / 1e12, best * 1e6 / (iters * C * wg_per_cu));
}
int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  double *out;
  CK(hipMalloc(&out, 64));
  printf("%d CUs, clock %d MHz\n", ncu, prop.clockRate / 1000);
  run<1>(1, ncu, out);
  run<4>(1, ncu, out);
  run<4>(2, ncu, out);
  run<4>(3, ncu, out);
  return 0;
}
