// seam_bench.hip — what would merging the PP correction (multi-workgroup: streams the pair operators
// of a mode, writes the s x R matrix M) and the mode update (ONE workgroup: needs all of M) into one
// launch buy? The hand-over inside a launch — every workgroup publishes its rows, the last arriver
// continues — against the kernel boundary it replaces, in the geometry of BASELINE config 3
// (s = 200, R = 10: 9.6 MB of operators per mode, M = 16 KB), as a chain of dependent pairs.
//   two:    k_correct<<<nwg>>> ; k_update<<<1>>>           (what the product does: 2 launches per mode)
//   merged: k_merged<<<nwg>>> — plain stores of M, release fence + agent-scope ticket per workgroup,
//           the last arriver acquires and runs the update's read of M
//   merged_sc1: the same with write-through (sc1) stores of M and sc1 loads instead of the fences
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/seam_bench tools/seam_bench.hip
//   run:   tools/seam_bench [nwg=50]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int kRows = 200, kR = 10, kM = kRows * kR;

// the correction's share of a workgroup: stream its slice of the operators, reduce, write its rows of M
__device__ inline void correct_body(const double *__restrict__ T, size_t n_per_wg, double *__restrict__ M,
                                    bool sc1) {
  const double *t = T + (size_t)blockIdx.x * n_per_wg;
  double acc = 0;
  for (size_t i = threadIdx.x; i < n_per_wg; i += blockDim.x) acc += __builtin_nontemporal_load(t + i);
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const int per = (kM + gridDim.x - 1) / gridDim.x;
  const int e = blockIdx.x * per + threadIdx.x;
  if ((int)threadIdx.x < per && e < kM) {
    const double v = red[0] * 1e-9 + e;
    if (sc1)
      __hip_atomic_store(M + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      M[e] = v;
  }
}
// the update's read of M by one workgroup (stand-in for S^-1, gradient, W = M S^-1, Gram)
__device__ inline void update_body(const double *__restrict__ M, double *__restrict__ out, bool sc1) {
  double acc = 0;
  for (int e = threadIdx.x; e < kM; e += blockDim.x)
    acc += sc1 ? __hip_atomic_load(M + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : M[e];
  __shared__ double red2[256];
  red2[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red2[threadIdx.x] += red2[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red2[0];
}
__global__ __launch_bounds__(256) void k_correct(const double *T, size_t n_per_wg, double *M) {
  correct_body(T, n_per_wg, M, false);
}
__global__ __launch_bounds__(256) void k_update(const double *M, double *out) { update_body(M, out, false); }
template <bool SC1>
__global__ __launch_bounds__(256) void k_merged(const double *T, size_t n_per_wg, double *M, double *out,
                                                unsigned *ticket, unsigned round) {
  correct_body(T, n_per_wg, M, SC1);
  __shared__ int last;
  if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!SC1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t + 1 == gridDim.x * round);
    if (last && !SC1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (last) update_body(M, out, SC1);
}

int main(int argc, char **argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 50;
  const size_t nT = (size_t)3 * kRows * kRows * kR;  // three pair operators of a mode: 9.6 MB
  const size_t per = nT / nwg;
  double *T, *M, *out;
  unsigned *ticket;
  CK(hipMalloc(&T, sizeof(double) * nT));
  CK(hipMemset(T, 0, sizeof(double) * nT));
  CK(hipMalloc(&M, sizeof(double) * kM));
  CK(hipMalloc(&out, sizeof(double) * 8));
  CK(hipMalloc(&ticket, sizeof(unsigned) * 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int pairs = 200;
  auto run = [&](int form) {
    std::vector<float> ts;
    double check = 0;
    for (int rep = 0; rep < 7; rep++) {
      CK(hipMemset(ticket, 0, sizeof(unsigned) * 64));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < pairs; i++) {
        if (form == 0) {
          hipLaunchKernelGGL(k_correct, dim3(nwg), dim3(256), 0, 0, T, per, M);
          hipLaunchKernelGGL(k_update, dim3(1), dim3(256), 0, 0, M, out);
        } else if (form == 1) {
          hipLaunchKernelGGL(k_merged<false>, dim3(nwg), dim3(256), 0, 0, T, per, M, out, ticket, (unsigned)(i + 1));
        } else {
          hipLaunchKernelGGL(k_merged<true>, dim3(nwg), dim3(256), 0, 0, T, per, M, out, ticket, (unsigned)(i + 1));
        }
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms * 1e3f / pairs);
      CK(hipMemcpy(&check, out, sizeof(double), hipMemcpyDeviceToHost));
    }
    std::sort(ts.begin(), ts.end());
    const double want = (double)kM * (kM - 1) / 2.0;
    printf("%-44s %3d workgroups: %.2f us per (correction + update), median of 7 chains of %d; sum(M) %s\n",
           form == 0 ? "two launches (product)" : form == 1 ? "one launch, fences + ticket, last arriver"
                                                             : "one launch, sc1 stores/loads + ticket",
           nwg, ts[3], pairs, check == want ? "ok" : "WRONG");
  };
  run(0);
  run(1);
  run(2);
  run(0);
  return 0;
}
