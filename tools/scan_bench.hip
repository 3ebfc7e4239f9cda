// scan_bench.hip — A/B micro-benchmark of the tensor-scan kernel variants (development tool, not
// part of the product). All variants run interleaved in ONE process on the same buffers
// (cdna_hip_programming.md §5.4 rule 24); prints median / min ms and algorithmic GB/s.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/scan_bench tools/scan_bench.hip
//   run:   tools/scan_bench [s=200] [R=10] [rounds=7]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../pairwise-perturbation_amd/csrc/ops.h"
#include "../pairwise-perturbation_amd/csrc/kernels_scan.hip.h"

using namespace ppals;

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__global__ void k_fill(float *p, int64_t n, uint32_t seed) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)e * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[e] = 0.5f + (float)(h & 0xffff) * (0.5f / 65536.f);
  }
}

struct Variant {
  std::string name;
  std::function<void()> launch;
  std::vector<float> ms;
  int kind = 0, nsplit = 1;  // filled from the name below
};

// packed[((((blk*NT+nt)*4+g)*16+n)*4+u] holds k = 16*blk + 4*u + g (suffix packing, VEC = 4)
__global__ void k_zero_tail(float *P, int nblk, int NT, int K) {
  const int total = nblk * NT * 256 * 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int u = e & 3, g = (e >> 6) & 3, blk = e / (NT * 256);
    if (16 * blk + 4 * u + g >= K) P[e] = 0.f;
  }
}

int main(int argc, char **argv) {
  const int s = argc > 1 ? atoi(argv[1]) : 200;
  const int R = argc > 2 ? atoi(argv[2]) : 10;
  const int rounds = argc > 3 ? atoi(argv[3]) : 7;
  const int64_t M = (int64_t)s * s, K = (int64_t)s * s;
  const int NT = R <= 16 ? 1 : 2;
  constexpr int VEC = 4;
  const int nblk = (int)((K + 15) / 16);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs; s=%d R=%d: V = %lld x %lld fp32 = %.2f GB\n", prop.name, ncu, s, R,
         (long long)M, (long long)K, M * K * 4.0 / 1e9);

  float *V, *P;
  double *slab, *out;
  CK(hipMalloc(&V, sizeof(float) * M * K));
  CK(hipMalloc(&P, sizeof(float) * (size_t)nblk * NT * 256 * VEC));
  CK(hipMalloc(&slab, std::max<size_t>(sizeof(double) * 64 * 16 * NT * M, sizeof(float) * 16 * NT * (size_t)M * (size_t)s + 4096)));  // also holds the s^3 x 16 fp32 result of the ttm probes
  CK(hipMalloc(&out, sizeof(double) * 16 * NT * M));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, V, M * K, 1u);
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, P, (int64_t)nblk * NT * 256 * VEC, 2u);
  CK(hipDeviceSynchronize());

  const int n_mtiles = (int)((M + 255) / 256);
  auto suffix_cfg = [&](int target_mult, int &nsplit, int &per) {
    nsplit = std::max(1, std::min((ncu * target_mult + n_mtiles - 1) / n_mtiles, nblk / 32));
    per = (nblk + nsplit - 1) / nsplit;
    nsplit = (nblk + per - 1) / per;
  };
  const int64_t ncolgrp = (K + 63) / 64;

  std::vector<Variant> vs;
  int ns8, per8, ns4, per4, ns16, per16, ns12, per12;
  suffix_cfg(12, ns12, per12);
  suffix_cfg(8, ns8, per8);
  suffix_cfg(4, ns4, per4);
  suffix_cfg(16, ns16, per16);
#define SUFFIX(KERN, ns, per)                                                                     \
  [=]() {                                                                                         \
    hipLaunchKernelGGL(KERN, dim3((unsigned)(n_mtiles * ns)), dim3(256), 0, 0, V, M, K, M * K, P, \
                       n_mtiles, ns, per, nblk, slab, M, (int64_t)16 * NT * M, (int64_t)0, R, 0); \
  }
#define SUFFIXP(KERN, ns, per) /* persistent buffer-load kernel: takes the tile count */           \
  [=]() {                                                                                         \
    hipLaunchKernelGGL(KERN, dim3((unsigned)std::min<int64_t>((int64_t)n_mtiles * ns, ncu * 40)), \
                       dim3(256), 0, 0, V, M, K, M * K, P, n_mtiles, ns, per, nblk, slab, M,      \
                       (int64_t)16 * NT * M, (int64_t)0, R, 0, (int64_t)n_mtiles * ns);           \
  }
#define PREFIXI(KERN, nsp)                                                                        \
  [=]() {                                                                                         \
    int per_ = (nblk + (nsp)-1) / (nsp);                                                          \
    hipLaunchKernelGGL(KERN, dim3((unsigned)((K + 15) / 16), (unsigned)(nsp)), dim3(256), 0, 0, V, \
                       M, K, P, per_, nblk, slab, (int64_t)1, K, (int64_t)16 * NT * K, R, 0);     \
  }
#define PREFIX(KERN, nsp)                                                                        \
  [=]() {                                                                                        \
    int per_ = (nblk + (nsp)-1) / (nsp);                                                         \
    hipLaunchKernelGGL(KERN, dim3((unsigned)ncolgrp, (unsigned)(nsp)), dim3(256), 0, 0, V, M, K, \
                       P, per_, nblk, slab, (int64_t)1, K, (int64_t)16 * NT * K, R, 0);          \
  }
  if (NT == 1) {
    vs.push_back({"suffix generic          x8", SUFFIX((k_scan_suffix<float, 1, true>), ns8, per8), {}});
    vs.push_back({"suffix fast             x8", SUFFIX((k_scan_suffix_fast<float, 1, 0>), ns8, per8), {}});
    vs.push_back({"suffix fast nt          x8", SUFFIX((k_scan_suffix_fast<float, 1, 1>), ns8, per8), {}});
    vs.push_back({"suffix buf nt           x8", SUFFIXP((k_scan_suffix_buf<float, 1, 1>), ns8, per8), {}});
    vs.push_back({"suffix buf nt          x12", SUFFIXP((k_scan_suffix_buf<float, 1, 1>), ns12, per12), {}});
    vs.push_back({"suffix fast nt+xcd      x8", SUFFIX((k_scan_suffix_fast<float, 1, 3>), ns8, per8), {}});
    vs.push_back({"suffix fast nt          x4", SUFFIX((k_scan_suffix_fast<float, 1, 1>), ns4, per4), {}});
    vs.push_back({"suffix fast nt         x16", SUFFIX((k_scan_suffix_fast<float, 1, 1>), ns16, per16), {}});
    vs.push_back({"prefix generic      split1", PREFIX((k_scan_prefix<float, 1, true, 4>), 1), {}});
    vs.push_back({"prefix fast         split1", PREFIX((k_scan_prefix_fast<float, 1, 0>), 1), {}});
    vs.push_back({"prefix fast nt      split1", PREFIX((k_scan_prefix_fast<float, 1, 1>), 1), {}});
    vs.push_back({"prefix fast         split2", PREFIX((k_scan_prefix_fast<float, 1, 0>), 2), {}});
    vs.push_back({"prefix fast perm    split1", PREFIX((k_scan_prefix_fast<float, 1, 4>), 1), {}});
    vs.push_back({"prefix fast perm    split2", PREFIX((k_scan_prefix_fast<float, 1, 4>), 2), {}});
    vs.push_back({"prefix fast perm+nt split2", PREFIX((k_scan_prefix_fast<float, 1, 5>), 2), {}});
    vs.push_back({"prefix fast il      split1", PREFIXI((k_scan_prefix_fast<float, 1, 8>), 1), {}});
    vs.push_back({"prefix fast il+perm split1", PREFIXI((k_scan_prefix_fast<float, 1, 12>), 1), {}});
    vs.push_back({"prefix fast il+p+nt split1", PREFIXI((k_scan_prefix_fast<float, 1, 13>), 1), {}});
    vs.push_back({"prefix fast il+perm split2", PREFIXI((k_scan_prefix_fast<float, 1, 12>), 2), {}});
  } else {
    vs.push_back({"suffix generic          x8", SUFFIX((k_scan_suffix<float, 2, true>), ns8, per8), {}});
    vs.push_back({"suffix fast             x8", SUFFIX((k_scan_suffix_fast<float, 2, 0>), ns8, per8), {}});
    vs.push_back({"suffix fast nt          x8", SUFFIX((k_scan_suffix_fast<float, 2, 1>), ns8, per8), {}});
    vs.push_back({"suffix buf nt           x8", SUFFIXP((k_scan_suffix_buf<float, 2, 1>), ns8, per8), {}});
    vs.push_back({"prefix generic      split1", PREFIX((k_scan_prefix<float, 2, true, 4>), 1), {}});
    vs.push_back({"prefix fast         split1", PREFIX((k_scan_prefix_fast<float, 2, 0>), 1), {}});
    vs.push_back({"prefix fast perm    split1", PREFIX((k_scan_prefix_fast<float, 2, 4>), 1), {}});
    vs.push_back({"prefix fast perm    split2", PREFIX((k_scan_prefix_fast<float, 2, 4>), 2), {}});
    vs.push_back({"prefix fast il+perm split1", PREFIXI((k_scan_prefix_fast<float, 2, 12>), 1), {}});
    vs.push_back({"prefix fast il+p+nt split1", PREFIXI((k_scan_prefix_fast<float, 2, 13>), 1), {}});
  }
  // single-mode TTM regime of the multi-sweep schedule: M = s^3 rows, K = s, fp32 result.
  // (Earlier probes of this regime — nontemporal stores, a rank-fastest result layout, padded row
  // counts, result placement offsets — changed nothing or lost; their numbers are kept in
  // profiles/r01h_scan_bench_ttm_regime_s200_r10.txt.)
  if (argc > 4) {
    vs.clear();
    const int64_t M3 = (int64_t)s * s * s, K1 = s;
    const int nblk1 = (int)((K1 + 15) / 16);
    const int n_mt = (int)((M3 + 255) / 256);
#define TTM_DISPATCH(BODY)  \
  if (NT == 1) {           \
    constexpr int NTC = 1; \
    BODY                   \
  } else {                 \
    constexpr int NTC = 2; \
    BODY                   \
  }
    auto ttm_fast = [=]() {
      return [=]() {
        TTM_DISPATCH(hipLaunchKernelGGL((k_scan_suffix_fast<float, NTC, 1>), dim3((unsigned)n_mt),
                                        dim3(256), 0, 0, V, M3, K1, M3 * K1, P, n_mt, 1, nblk1, nblk1,
                                        slab, M3, (int64_t)0, (int64_t)0, R, 1);)
      };
    };
    auto ttm = [=](int mult, int ncols_) {  // ncols_ = 0: no stores at all
      return [=]() {
        TTM_DISPATCH(hipLaunchKernelGGL(
            (k_scan_suffix_buf<float, NTC, 1>),
            dim3((unsigned)std::min<int64_t>(n_mt, (int64_t)ncu * mult)), dim3(256), 0, 0, V, M3, K1,
            M3 * K1, P, n_mt, 1, nblk1, nblk1, slab, M3, (int64_t)0, (int64_t)0, ncols_, 1,
            (int64_t)n_mt);)
      };
    };
    auto ttm_b = [=](int mult) {  // the same bytes as s batches of [s^2 x s] (root N-2 / root 0)
      const int64_t M2 = (int64_t)s * s;
      const int n_mt2 = (int)((M2 + 255) / 256);
      const int64_t nt2 = (int64_t)n_mt2 * s;
      return [=]() {
        TTM_DISPATCH(hipLaunchKernelGGL(
            (k_scan_suffix_buf<float, NTC, 1>),
            dim3((unsigned)std::min<int64_t>(nt2, (int64_t)ncu * mult)), dim3(256), 0, 0, V, M2, K1,
            M2 * K1, P, n_mt2, 1, nblk1, nblk1, slab, M2 * s, (int64_t)0, M2, R, 1, nt2);)
      };
    };
    // the product's k_krp_pack zero-fills the packed operand beyond K; do the same here so that the
    // clamping (fast) and the descriptor-bounded (buf) kernels agree on the partial last block
    hipLaunchKernelGGL(k_zero_tail, dim3(64), dim3(256), 0, 0, P, (int)nblk1, NT, (int)K1);
    CK(hipDeviceSynchronize());
    vs.push_back({"ttm fast (1 tile/WG)    x1", ttm_fast(), {}});
    vs.push_back({"ttm buf persistent      x3", ttm(3, R), {}});
    vs.push_back({"ttm buf persistent     x20", ttm(20, R), {}});
    vs.push_back({"ttm buf persistent     x40", ttm(40, R), {}});
    vs.push_back({"ttmB batched s x [s2 x s] x40", ttm_b(40), {}});
    vs.push_back({"ttmP no stores         x40", ttm(40, 0), {}});
  }
  // a plain streaming read of V as the practical ceiling on this device
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
  const double bytes = (double)M * K * 4.0;
  // correctness cross-check: every variant must reproduce the first variant of its kind on the
  // first 2048 outputs of column n=0..1 (summed over its split slabs)
  const int NCHK = 2048;
  std::vector<double> ref[2];
  printf("%-28s %9s %9s %10s %12s\n", "variant", "med ms", "min ms", "GB/s(med)", "max rel diff");
  for (auto &v : vs) {
    v.kind = v.name.rfind("prefix", 0) == 0 ? 1 : 0;
    if (v.name.rfind("ttm", 0) == 0) {
      v.nsplit = 1;
    } else if (v.kind == 0) {
      int mult = atoi(v.name.c_str() + v.name.rfind('x') + 1);
      int ns, per;
      suffix_cfg(mult, ns, per);
      v.nsplit = ns;
    } else {
      v.nsplit = atoi(v.name.c_str() + v.name.size() - 1);
    }
    CK(hipMemset(slab, 0, sizeof(double) * 64 * 16 * NT * M));
    v.launch();
    CK(hipDeviceSynchronize());
    std::vector<double> acc(NCHK, 0.0), tmp(NCHK);
    const int64_t stride = (int64_t)16 * NT * (v.kind == 0 ? M : K);
    if (v.name.rfind("ttm", 0) == 0) {  // fp32 result, one slab
      std::vector<float> tf(NCHK);
      CK(hipMemcpy(tf.data(), slab, sizeof(float) * NCHK, hipMemcpyDeviceToHost));
      for (int i = 0; i < NCHK; i++) acc[i] = tf[i];
    } else
      for (int sp = 0; sp < v.nsplit; sp++) {
        CK(hipMemcpy(tmp.data(), slab + sp * stride, sizeof(double) * NCHK, hipMemcpyDeviceToHost));
        for (int i = 0; i < NCHK; i++) acc[i] += tmp[i];
      }
    double maxrel = 0;
    if (v.name.rfind("ttmB", 0) == 0 || v.name.rfind("ttmP", 0) == 0)
      ;  // different problem shape: timing only
    else if (ref[v.kind].empty())
      ref[v.kind] = acc;
    else
      for (int i = 0; i < NCHK; i++)
        maxrel = std::max(maxrel, fabs(acc[i] - ref[v.kind][i]) / (fabs(ref[v.kind][i]) + 1e-300));
    std::sort(v.ms.begin(), v.ms.end());
    float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
    printf("%-28s %9.4f %9.4f %10.1f %12.3e\n", v.name.c_str(), med, mn,
           bytes / (med * 1e-3) / 1e9, maxrel);
  }
  return 0;
}
