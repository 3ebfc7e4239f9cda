// Can a latency chain of small launches run BESIDE an HBM-streaming kernel without being starved,
// when the two streams own disjoint sets of CUs (hipExtStreamCreateWithCUMask)? Tools only.
//
//   scan  : a persistent streaming read of `GB` gigabytes (16-byte non-temporal loads, one result word
//           per workgroup) — the shape of k_scan_suffix_buf's read side
//   chain : `NCH` dependent launches of one of three kinds
//             tiny  one workgroup, 256 threads, a few hundred flops (mode update / Normalize class)
//             mid   104 workgroups reading 40 MB (a pass over a P = 8 shard's X_r)
//             pass  1024 workgroups reading 320 MB (the 53 us pass over cfg2's X_r)
// Reported per arrangement: chain time alone, scan time alone, both started together (wall time of the
// pair, the chain's own time from its events, the scan's own time), and which XCCs / CUs the chain's
// workgroups ran on (HW_ID / XCC_ID), so the mask's bit order can be read off.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/cumask_bench tools/cumask_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                        \
    }                                                                  \
  } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_stream(const float4v *__restrict__ src, size_t n4, float *sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4v a = __builtin_nontemporal_load(src + i);
    float4v b = __builtin_nontemporal_load(src + i + stride);
    float4v c = __builtin_nontemporal_load(src + i + 2 * stride);
    float4v d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
  }
  for (; i < n4; i += stride) {
    float4v a = __builtin_nontemporal_load(src + i);
    acc += a.x + a.y + a.z + a.w;
  }
  if (acc == 12345.678f) sink[blockIdx.x] = acc;
}

// a dependent link: reads what the previous link wrote
__global__ __launch_bounds__(256) void k_tiny(double *buf, unsigned *where) {
  __shared__ double s[256];
  double v = buf[threadIdx.x];
  for (int k = 0; k < 16; k++) v = v * 1.0000001 + 1e-9;
  s[threadIdx.x] = v;
  __syncthreads();
  buf[threadIdx.x] = s[(threadIdx.x + 1) & 255];
  if (threadIdx.x == 0 && where) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 15;
    where[blockIdx.x] = (xcc << 28) | (hw & 0x0fffffff);
  }
}

__global__ __launch_bounds__(256) void k_read(const float4v *__restrict__ src, size_t n4, double *buf,
                                              unsigned *where) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = (float)buf[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4v a = src[i];
    acc += a.x + a.y + a.z + a.w;
  }
  if (acc == 12345.678f) buf[1] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) buf[0] += 1.0;
  if (threadIdx.x == 0 && where) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 15;
    where[blockIdx.x] = (xcc << 28) | (hw & 0x0fffffff);
  }
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Chain {
  int kind;  // 0 tiny, 1 mid (40 MB), 2 pass (320 MB)
  int n;
};

int main(int argc, char **argv) {
  const double GB = argc > 1 ? atof(argv[1]) : 6.4;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.name, ncu);
  const size_t n4 = (size_t)(GB * 1e9 / 16);
  float4v *big;
  CK(hipMalloc(&big, n4 * 16));
  CK(hipMemset(big, 0, n4 * 16));
  const size_t small4 = (size_t)(320e6 / 16);
  float4v *xr;
  CK(hipMalloc(&xr, small4 * 16));
  CK(hipMemset(xr, 0, small4 * 16));
  float *sink;
  CK(hipMalloc(&sink, 4 * 65536));
  double *buf;
  CK(hipMalloc(&buf, 8 * 256));
  CK(hipMemset(buf, 0, 8 * 256));
  unsigned *where;
  CK(hipMalloc(&where, 4 * 4096));
  CK(hipDeviceSynchronize());

  // ---- streams: the arrangement table
  struct Arr {
    const char *name;
    int side_cus;      // 0: no mask on the side stream
    bool main_compl;   // main stream masked to the complement
    int layout;        // 0: the first `side_cus` mask bits, 1: bits side_cus.. spread with stride (ncu / side_cus)
  };
  const Arr arrs[] = {
      {"no masks", 0, false, 0},
      {"side 8 CUs (bits 0-7), main unmasked", 8, false, 0},
      {"side 8 CUs (bits 0-7), main complement", 8, true, 0},
      {"side 16 CUs (bits 0-15), main unmasked", 16, false, 0},
      {"side 16 CUs (bits 0-15), main complement", 16, true, 0},
      {"side 32 CUs (bits 0-31), main complement", 32, true, 0},
      {"side 16 CUs (every 16th bit), main complement", 16, true, 1},
      {"side 32 CUs (bits 0-31 = one XCC if bits are XCC-major), main unmasked", 32, false, 0},
  };
  const Chain chains[] = {{0, 30}, {1, 12}, {2, 4}};
  const int words = (ncu + 31) / 32;
  const int scan_grid = ncu * 8;

  for (const Arr &a : arrs) {
    hipStream_t sm, ss;
    std::vector<uint32_t> mside(words, 0), mmain(words, 0);
    if (a.side_cus > 0) {
      for (int k = 0; k < a.side_cus; k++) {
        const int bit = a.layout == 0 ? k : k * (ncu / a.side_cus);
        mside[bit / 32] |= 1u << (bit % 32);
      }
      for (int b = 0; b < ncu; b++)
        if (!((mside[b / 32] >> (b % 32)) & 1u)) mmain[b / 32] |= 1u << (b % 32);
      hipError_t e = hipExtStreamCreateWithCUMask(&ss, words, mside.data());
      if (e != hipSuccess) {
        printf("[%s] hipExtStreamCreateWithCUMask(side) -> %s\n", a.name, hipGetErrorString(e));
        continue;
      }
      if (a.main_compl) {
        e = hipExtStreamCreateWithCUMask(&sm, words, mmain.data());
        if (e != hipSuccess) {
          printf("[%s] hipExtStreamCreateWithCUMask(main) -> %s\n", a.name, hipGetErrorString(e));
          continue;
        }
      } else {
        CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking));
      }
    } else {
      CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking));
      CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    }
    printf("\n=== %s\n", a.name);
    hipEvent_t e0, e1, c0, c1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&c0));
    CK(hipEventCreate(&c1));
    auto launch_chain = [&](const Chain &c, bool rec_where) {
      for (int k = 0; k < c.n; k++) {
        unsigned *w = (rec_where && k == c.n - 1) ? where : nullptr;
        if (c.kind == 0)
          k_tiny<<<1, 256, 0, ss>>>(buf, w);
        else if (c.kind == 1)
          k_read<<<104, 256, 0, ss>>>(xr, small4 / 8, buf, w);
        else
          k_read<<<1024, 256, 0, ss>>>(xr, small4, buf, w);
      }
    };
    // warm-up
    k_stream<<<scan_grid, 256, 0, sm>>>(big, n4, sink);
    launch_chain(chains[0], false);
    CK(hipDeviceSynchronize());
    // scan alone
    float scan_alone = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, sm));
      k_stream<<<scan_grid, 256, 0, sm>>>(big, n4, sink);
      CK(hipEventRecord(e1, sm));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      scan_alone = std::min(scan_alone, ms);
    }
    printf("scan alone: %.3f ms = %.2f TB/s (grid %d x 256)\n", scan_alone, GB / scan_alone, scan_grid);
    for (const Chain &c : chains) {
      const char *kn = c.kind == 0 ? "tiny x30" : c.kind == 1 ? "mid(40MB,104wg) x12" : "pass(320MB,1024wg) x4";
      float alone = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(c0, ss));
        launch_chain(c, rep == 2);
        CK(hipEventRecord(c1, ss));
        CK(hipEventSynchronize(c1));
        float ms;
        CK(hipEventElapsedTime(&ms, c0, c1));
        alone = std::min(alone, ms);
      }
      // where the last link ran
      {
        const int nw = c.kind == 0 ? 1 : c.kind == 1 ? 104 : 1024;
        std::vector<unsigned> h(nw);
        CK(hipMemcpy(h.data(), where, 4 * nw, hipMemcpyDeviceToHost));
        unsigned xccs = 0;
        std::vector<unsigned> ids;
        for (unsigned v : h) {
          xccs |= 1u << (v >> 28);
          // HW_ID (gfx9): cu_id bits 8-11, sh_id bit 12, se_id bits 13-15 (3 bits on gfx94x/95x)
          const unsigned cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7;
          ids.push_back(((v >> 28) << 12) | (se << 8) | (sh << 4) | cu);
        }
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        printf("  chain %-22s alone %.1f us/link; ran on XCC mask 0x%02x, %zu distinct (xcc,se,sh,cu)", kn,
               1e3 * alone / c.n, xccs, ids.size());
        if (ids.size() <= 32) {
          printf(":");
          for (unsigned id : ids) printf(" %x.%x.%x.%x", id >> 12, (id >> 8) & 15, (id >> 4) & 15, id & 15);
        }
        printf("\n");
      }
      // together: the scan first, the chain immediately behind it on the other stream
      float best_pair = 1e30f, ch_t = 0, sc_t = 0;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now_ms();
        CK(hipEventRecord(e0, sm));
        k_stream<<<scan_grid, 256, 0, sm>>>(big, n4, sink);
        CK(hipEventRecord(e1, sm));
        CK(hipEventRecord(c0, ss));
        launch_chain(c, false);
        CK(hipEventRecord(c1, ss));
        CK(hipDeviceSynchronize());
        const float pair = (float)(now_ms() - t0);
        float a_ms, b_ms;
        CK(hipEventElapsedTime(&a_ms, e0, e1));
        CK(hipEventElapsedTime(&b_ms, c0, c1));
        if (pair < best_pair) {
          best_pair = pair;
          ch_t = b_ms;
          sc_t = a_ms;
        }
      }
      printf("        together: pair %.3f ms (serial would be %.3f) | scan %.3f ms (+%.1f %%) | chain %.1f us/link (x%.1f)\n",
             best_pair, scan_alone + alone, sc_t, 100.0 * (sc_t / scan_alone - 1.0), 1e3 * ch_t / c.n,
             ch_t / alone);
    }
    CK(hipStreamDestroy(sm));
    CK(hipStreamDestroy(ss));
  }
  return 0;
}
