// stream_bench — what this box's HBM does with plain streams, independent of the scan kernel:
// read-only, write-only, copy, and a 20:1 read:write mix (the scan kernel's ratio at cfg2), with
// plain / nontemporal / write-through (sc1) stores. usage: stream_bench [GB=6.4] [rounds=7]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((vector_size(16)));

// every workgroup sweeps its own contiguous chunks; 4 independent 16-byte loads in flight per lane
template <int NT>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ src, int64_t n, float *sink) {
  f4 acc = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x); i + 3 * 256LL * gridDim.x < n; i += stride) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const f4 *p = src + i + (int64_t)u * 256 * gridDim.x;
      f4 v = NT ? __builtin_nontemporal_load(p) : *p;
      acc += v;
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}
template <int POL>  // 0 plain, 1 nontemporal, 2 sc1 (write-through)
__device__ inline void store16(f4 *p, f4 v) {
  if (POL == 0) {
    *p = v;
  } else if (POL == 1) {
    __builtin_nontemporal_store(v, p);
  } else {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 16, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, 0, 0, 16);
  }
}
template <int POL>
__global__ __launch_bounds__(256) void k_write(f4 *__restrict__ dst, int64_t n) {
  const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (POL == 2)
      dst[i] = v;  // per-lane descriptors would be divergent: sc1 measured in k_mix only
    else
      store16<POL>(dst + i, v);
  }
}
template <int POL>
__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ src, f4 *__restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    f4 v = __builtin_nontemporal_load(src + i);
    if (POL == 1) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
  }
}
// read `ratio` 16-byte vectors for every one written (sum of the reads is what gets stored)
template <int POL>
__global__ __launch_bounds__(256) void k_mix(const f4 *__restrict__ src, f4 *__restrict__ dst,
                                             int64_t nw, int ratio) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nw; i += stride) {
    f4 acc = {0, 0, 0, 0};
    for (int r = 0; r < ratio; r += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) acc += __builtin_nontemporal_load(src + i + (int64_t)(r + u) * nw);
    }
    if (POL == 2) {
      // wave-uniform descriptor on the wave's 1 KiB span, per-lane 16-byte offset
      const int64_t w0 = i - (threadIdx.x & 63);
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void *)(dst + w0), 0, 1024, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc), rs,
                                             (int)((threadIdx.x & 63) * 16), 0, 16);
    } else {
      store16<POL>(dst + i, acc);
    }
  }
}

// the same 20:1 mix, but results wait in an LDS queue and every wave of the chip stores them in the
// same short window of the device-wide 100 MHz clock (period / window in ticks of 10 ns): does
// separating the writes from the reads IN TIME remove the mixed-stream penalty?
template <int QN>
__global__ __launch_bounds__(256) void k_mix_phased(const f4 *__restrict__ src, f4 *__restrict__ dst,
                                                    int64_t nw, int ratio, int period, int window) {
  __shared__ f4 qv[QN][256];
  __shared__ int64_t qi[QN][256];
  const int64_t stride = (int64_t)gridDim.x * 256;
  int qn = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nw; i += stride) {
    f4 acc = {0, 0, 0, 0};
    for (int r = 0; r < ratio; r += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) acc += __builtin_nontemporal_load(src + i + (int64_t)(r + u) * nw);
    }
    qv[qn][threadIdx.x] = acc;
    qi[qn][threadIdx.x] = i;
    qn++;
    const unsigned long long t = wall_clock64();
    if (qn == QN || (int)(t % (unsigned)period) < window) {
      for (int q = 0; q < qn; q++) dst[qi[q][threadIdx.x]] = qv[q][threadIdx.x];
      qn = 0;
    }
  }
  for (int q = 0; q < qn; q++) dst[qi[q][threadIdx.x]] = qv[q][threadIdx.x];
}

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 6.4;
  const int rounds = argc > 2 ? atoi(argv[2]) : 7;
  const int ratio = 20;
  int64_t nw = (int64_t)(gb * 1e9 / 16 / ratio) / 256 * 256;  // written vectors in the mix
  const int64_t n = nw * ratio;                                // vectors of the big buffer
  f4 *A, *B;
  float *sink;
  CK(hipMalloc(&A, sizeof(f4) * n));
  CK(hipMalloc(&B, sizeof(f4) * n));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0, sizeof(f4) * n));
  CK(hipMemset(B, 0, sizeof(f4) * n));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("%d CUs; big buffer %.2f GB, mix writes %.3f GB\n", ncu, n * 16.0 / 1e9, nw * 16.0 / 1e9);
  struct Var {
    std::string name;
    std::function<void()> launch;
    double bytes;
    std::vector<float> ms;
  };
  std::vector<Var> vs;
  const dim3 grid(ncu * 32), blk(256);
  const double big = n * 16.0, small = nw * 16.0;
  vs.push_back({"read  (plain loads)", [=]() { hipLaunchKernelGGL(k_read<0>, grid, blk, 0, 0, A, n, sink); }, big, {}});
  vs.push_back({"read  (nt loads)", [=]() { hipLaunchKernelGGL(k_read<1>, grid, blk, 0, 0, A, n, sink); }, big, {}});
  vs.push_back({"write (plain stores)", [=]() { hipLaunchKernelGGL(k_write<0>, grid, blk, 0, 0, B, n); }, big, {}});
  vs.push_back({"write (nt stores)", [=]() { hipLaunchKernelGGL(k_write<1>, grid, blk, 0, 0, B, n); }, big, {}});
  vs.push_back({"copy  (nt load, plain store)", [=]() { hipLaunchKernelGGL(k_copy<0>, grid, blk, 0, 0, A, B, n); }, 2 * big, {}});
  vs.push_back({"copy  (nt load, nt store)", [=]() { hipLaunchKernelGGL(k_copy<1>, grid, blk, 0, 0, A, B, n); }, 2 * big, {}});
  vs.push_back({"mix 20:1 (plain stores)", [=]() { hipLaunchKernelGGL(k_mix<0>, grid, blk, 0, 0, A, B, nw, ratio); }, big + small, {}});
  vs.push_back({"mix 20:1 (nt stores)", [=]() { hipLaunchKernelGGL(k_mix<1>, grid, blk, 0, 0, A, B, nw, ratio); }, big + small, {}});
  vs.push_back({"mix 20:1 (sc1 stores)", [=]() { hipLaunchKernelGGL(k_mix<2>, grid, blk, 0, 0, A, B, nw, ratio); }, big + small, {}});
  for (int period : {1000, 2000, 4000, 8000})
    for (int window : {period / 16, period / 8}) {
      char nm[64];
      snprintf(nm, sizeof nm, "mix 20:1 phased q8 P=%dus W=%.1fus", period / 100, window / 100.0);
      vs.push_back({nm, [=]() { hipLaunchKernelGGL(k_mix_phased<8>, grid, blk, 0, 0, A, B, nw, ratio, period, window); }, big + small, {}});
    }
  vs.push_back({"mix 20:1 phased q8, window never (flush when full)", [=]() { hipLaunchKernelGGL(k_mix_phased<8>, grid, blk, 0, 0, A, B, nw, ratio, 1000, 0); }, big + small, {}});
  vs.push_back({"mix 20:1 reads only (ratio 20, no store)", [=]() { hipLaunchKernelGGL(k_read<1>, grid, blk, 0, 0, A, n, sink); }, big, {}});
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; r++)
    for (auto &v : vs) {
      CK(hipEventRecord(e0, 0));
      v.launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
    }
  printf("%-42s %9s %9s %12s\n", "stream", "med ms", "min ms", "GB/s (med)");
  for (auto &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2];
    printf("%-42s %9.4f %9.4f %12.1f\n", v.name.c_str(), med, v.ms.front(), v.bytes / (med * 1e-3) / 1e9);
  }
  return 0;
}
