// barrier_bench.hip — what a device-wide barrier between co-resident workgroups costs on gfx950
// (agent-scope release + atomic arrive + spin + acquire), as a function of the workgroup count: the
// budget of a single-launch approximate (PP) sweep, which needs one per mode update.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/barrier_bench tools/barrier_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// monotone counter: barrier k is passed when the counter reaches k * gridDim.x. Every workgroup
// gives up after a bounded number of polls (the kernel always drains) and reports it.
__device__ inline bool grid_barrier(unsigned *cnt, unsigned target, int *err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();  // release what this workgroup wrote
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int polls = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++polls > (1 << 22)) {
        *err = 1;
        ok = false;
        break;
      }
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

__global__ void k_barriers(unsigned *cnt, int nbar, double *data, int *err) {
  for (int b = 1; b <= nbar; b++) {
    // a little cross-workgroup traffic so that the fences have something to order
    if (threadIdx.x < 16) data[blockIdx.x * 16 + threadIdx.x] += 1.0;
    if (!grid_barrier(cnt, (unsigned)b * gridDim.x, err)) return;
    if (threadIdx.x == 0) data[blockIdx.x * 16] += data[((blockIdx.x + 1) % gridDim.x) * 16 + 1];
  }
}

int main() {
  unsigned *cnt;
  double *data;
  int *err;
  CK(hipMalloc(&cnt, 4));
  CK(hipMalloc(&data, 8 * 16 * 1024));
  CK(hipMalloc(&err, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int threads : {256, 1024})
    for (int g : {8, 13, 25, 50, 100, 200}) {
      float t[2];
      for (int pass = 0; pass < 2; pass++) {
        const int nbar = pass == 0 ? 1 : 101;
        CK(hipMemset(cnt, 0, 4));
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(data, 0, 8 * 16 * 1024));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_barriers, dim3(g), dim3(threads), 0, 0, cnt, nbar, data, err);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[pass], e0, e1));
      }
      int herr = 0;
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("%4d workgroups x %4d threads: %.2f us per barrier (launch + 1 barrier %.1f us)%s\n", g,
             threads, (t[1] - t[0]) * 1e3 / 100.0, t[0] * 1e3, herr ? "  TIMED OUT" : "");
    }
  return 0;
}
