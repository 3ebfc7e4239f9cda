// What a cross-stream hand-over costs the PRODUCING stream (tools only, not part of the library).
// Main stream: A -> B -> A -> B ... (dependent launches); a second stream runs C after every A.
//   variant 0: no second stream at all (the floor)
//   variant 1: hipEventRecord(ev, s1) after A, hipStreamWaitEvent(s2, ev), C on s2   (what the engine does)
//   variant 2: A stores a sequence number into a flag, hipStreamWaitValue64(s2, flag, seq, GTE), C on s2
//              — no packet on the main stream at all
// Reported: median gap end(A) -> start(B) on the main stream, median delay end(A) -> start(C), and whether
// every C saw its A's data. Flag memory kinds tried for variant 2: signal memory
// (hipExtMallocWithFlags + hipMallocSignalMemory), plain device memory, pinned host memory.
// build: hipcc --offload-arch=gfx950 -O3 -o waitvalue_bench tools/waitvalue_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                             \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void kA(double *data, int n, unsigned long long seq, unsigned long long *flag,
                   unsigned long long *t_end) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) data[i] = (double)seq;
  __threadfence_system();
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *t_end = wall_clock64();
    if (flag) {
      __threadfence_system();
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__global__ void kB(unsigned long long *t_start, double *sink) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *t_start = wall_clock64();
  if (sink && threadIdx.x == 9999) sink[0] = 1;
}
__global__ void kC(const double *data, int n, unsigned long long seq, unsigned long long *t_start, int *ok) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *t_start = wall_clock64();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && data[i] != (double)seq) atomicAdd(ok, 1);
}

static double med(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  const int n = 8400, iters = 300;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  double *data;
  CK(hipMalloc(&data, sizeof(double) * n));
  unsigned long long *tA, *tB, *tC;
  CK(hipMalloc(&tA, 8 * iters));
  CK(hipMalloc(&tB, 8 * iters));
  CK(hipMalloc(&tC, 8 * iters));
  int *bad;
  CK(hipMalloc(&bad, 4));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  unsigned long long *flags[3] = {nullptr, nullptr, nullptr};
  const char *kind[3] = {"signal memory", "device memory", "pinned host memory"};
  hipError_t e0 = hipExtMallocWithFlags((void **)&flags[0], 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e0));
  if (e0 != hipSuccess) flags[0] = nullptr;
  CK(hipMalloc(&flags[1], 8));
  CK(hipHostMalloc(&flags[2], 8, hipHostMallocDefault));
  for (int variant = 0; variant < 5; variant++) {
    unsigned long long *flag = variant >= 2 ? flags[variant - 2] : nullptr;
    if (variant >= 2 && !flag) continue;
    if (flag) {
      if (variant == 4) *flag = 0;
      else CK(hipMemset(flag, 0, 8));
    }
    CK(hipMemset(bad, 0, 4));
    CK(hipMemset(tC, 0, 8 * iters));
    CK(hipDeviceSynchronize());
    bool failed = false;
    for (int it = 0; it < iters && !failed; it++) {
      const unsigned long long seq = it + 1;
      hipLaunchKernelGGL(kA, dim3((n + 255) / 256), dim3(256), 0, s1, data, n, seq, flag, tA + it);
      if (variant == 1) {
        CK(hipEventRecord(ev, s1));
        CK(hipStreamWaitEvent(s2, ev, 0));
      } else if (variant >= 2) {
        hipError_t e = hipStreamWaitValue64(s2, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
        if (e != hipSuccess) {
          printf("variant 2 (%s): hipStreamWaitValue64 -> %s\n", kind[variant - 2], hipGetErrorString(e));
          failed = true;
          break;
        }
      }
      if (variant >= 1) hipLaunchKernelGGL(kC, dim3((n + 255) / 256), dim3(256), 0, s2, data, n, seq, tC + it, bad);
      hipLaunchKernelGGL(kB, dim3(1), dim3(64), 0, s1, tB + it, (double *)nullptr);
      if (variant >= 1) {
        // (the next A overwrites data: order it behind this C, as the engine's slots are by their own events)
        CK(hipStreamSynchronize(s2));
      }
    }
    CK(hipDeviceSynchronize());
    if (failed) continue;
    std::vector<unsigned long long> hA(iters), hB(iters), hC(iters);
    CK(hipMemcpy(hA.data(), tA, 8 * iters, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), tB, 8 * iters, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hC.data(), tC, 8 * iters, hipMemcpyDeviceToHost));
    int hbad = 0;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    std::vector<double> gAB, gAC;
    for (int it = 20; it < iters; it++) {
      gAB.push_back(((double)hB[it] - (double)hA[it]) * 0.01);  // 100 MHz -> us
      if (variant >= 1) gAC.push_back(((double)hC[it] - (double)hA[it]) * 0.01);
    }
    printf("variant %d%s%s: end(A) -> start(B) on the main stream %.2f us", variant == 0 ? 0 : (variant == 1 ? 1 : 2),
           variant >= 2 ? ", " : "", variant >= 2 ? kind[variant - 2] : (variant == 1 ? ", event record + stream wait" : ", main stream only"),
           med(gAB));
    if (variant >= 1) printf(" | end(A) -> start(C) %.2f us | stale reads in C: %d", med(gAC), hbad);
    printf("\n");
  }
  return 0;
}
