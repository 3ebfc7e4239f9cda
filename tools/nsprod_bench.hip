// nsprod_bench.hip — one product of the Newton-Schulz sign iteration (J x J x J, fp64 MFMA, symmetric
// result) as a chain of dependent launches: what a product costs by the number of waves that split
// K inside a workgroup, against the non-symmetric tile kernel.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/nsprod_bench tools/nsprod_bench.hip
//   run:   tools/nsprod_bench [J=400]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../pairwise-perturbation_amd/csrc/ops.h"
#include "../pairwise-perturbation_amd/csrc/kernels_eig.hip.h"
using namespace ppals;
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <int NW>
static void chain(const char *name, double *X, double *Y, int J, int n) {
  const unsigned nt = (unsigned)((J + 15) / 16), ntri = nt * (nt + 1) / 2;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int rep = 0; rep < 7; rep++) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; i++) {
      hipLaunchKernelGGL(k_dgemm_nt_sym<NW>, dim3(ntri), dim3(64 * NW), 0, 0, X, (int64_t)J, X, (int64_t)J,
                         (const double *)nullptr, (int64_t)0, Y, (int64_t)J, J, J, 1e-3, 0.0, 0,
                         (double *)nullptr);
      std::swap(X, Y);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1e3f / n);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-34s J=%d: %.2f us per product (chain of %d, median of 7)\n", name, J, ts[3], n);
}

// ---- where does a product's time go, and does it matter which XCD computes which tile? A copy of
// k_dgemm_nt_sym<8> with (a) the tile of a workgroup taken from a table (XCD-aware orders: workgroups
// b and b + 8 share an XCD, so a table can give every XCD a compact patch of the tile triangle and
// with it fewer operand row-blocks to pull into its L2 after the kernel boundary emptied it) and
// (b) optional phase stamps (s_memrealtime, 10 ns) by lane 0 of wave 0: start | operands in
// registers | MFMAs done | partial tiles combined | stores issued.
template <int NW, bool STAMP, bool SB = false>
__global__ __launch_bounds__(64 * NW) void k_sym_probe(const double *__restrict__ A, const double *__restrict__ Bt,
                                                       double *__restrict__ C, int M, double alpha,
                                                       const ushort2 *__restrict__ tile_map,
                                                       unsigned long long *__restrict__ stamps) {
  constexpr int UN = 13;
  __shared__ double part[NW - 1][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  if (STAMP && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
  const ushort2 tt = tile_map[blockIdx.x];
  const int i0 = tt.x * 16, j0 = tt.y * 16;
  const int ia = min(i0 + l16, M - 1), jb = min(j0 + l16, M - 1);
  const double *__restrict__ ap = A + ia;
  const double *__restrict__ bp = Bt + jb;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int ksteps = (M + 3) / 4;
  const int spw = (ksteps + NW - 1) / NW;
  const int s_begin = wave * spw, s_end = min(ksteps, s_begin + spw);
  for (int s0 = s_begin; s0 < s_end; s0 += UN) {
    double av[UN], bv[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int k = (s0 + u) * 4 + g;
      const bool ok = (s0 + u) < s_end && k < M;
      const int kc = ok ? k : 0;
      const double a = ap[(int64_t)M * kc], b = bp[(int64_t)M * kc];
      av[u] = ok ? a : 0.0;
      bv[u] = ok ? b : 0.0;
    }
    if (STAMP) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) t1 = __builtin_amdgcn_s_memrealtime();
    }
    if (SB) __builtin_amdgcn_sched_barrier(0);  // every load of the round issued before the first MFMA
#pragma unroll
    for (int u = 0; u < UN; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  if (STAMP && threadIdx.x == 0) {
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    t2 = __builtin_amdgcn_s_memrealtime();
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) part[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (STAMP && threadIdx.x == 0) t3 = __builtin_amdgcn_s_memrealtime();
  if (wave > 0) return;
  const int j = j0 + l16;
  if (j < M) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = i0 + g + 4 * r;
      if (i < M && i <= j) {
        double v = acc[r];
#pragma unroll
        for (int w = 0; w < NW - 1; w++) v += part[w][r][lane];
        v *= alpha;
        C[i + (int64_t)M * j] = v;
        if (i != j) C[j + (int64_t)M * i] = v;
      }
    }
  }
  if (STAMP && threadIdx.x == 0) {
    unsigned long long *st = stamps + 8 * (size_t)blockIdx.x;
    st[0] = t0;
    st[1] = t1;
    st[2] = t2;
    st[3] = t3;
    st[4] = __builtin_amdgcn_s_memrealtime();
    st[5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7;
  }
}
static unsigned morton2(unsigned x, unsigned y) {
  unsigned r = 0;
  for (int b = 0; b < 8; b++) r |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
  return r;
}
// tile tables: 0 = row-major upper triangle (what the product does), 1 = Morton order cut into 8
// consecutive patches, patch = blockIdx % 8, 2 = block rows dealt to XCDs (tile rows ti % 8 == xcd)
static std::vector<ushort2> tile_table(int J, int kind) {
  const int nt = (J + 15) / 16;
  std::vector<ushort2> all;
  for (int ti = 0; ti < nt; ti++)
    for (int tj = ti; tj < nt; tj++) all.push_back(ushort2{(unsigned short)ti, (unsigned short)tj});
  if (kind == 0) return all;
  std::vector<ushort2> ord = all;
  if (kind == 1)
    std::sort(ord.begin(), ord.end(), [](ushort2 a, ushort2 b) { return morton2(a.x, a.y) < morton2(b.x, b.y); });
  else
    std::stable_sort(ord.begin(), ord.end(), [](ushort2 a, ushort2 b) { return (a.x % 8) < (b.x % 8); });
  const int n = (int)ord.size();
  std::vector<int> start(9, 0);
  for (int c = 0; c < 8; c++) start[c + 1] = start[c] + n / 8 + (c < n % 8 ? 1 : 0);
  std::vector<ushort2> out(n);
  for (int b = 0; b < n; b++) out[b] = ord[start[b % 8] + b / 8];
  return out;
}

// ---- the same chain of products as ONE persistent launch (review of round 3: "measure it"): one
// 16 x 16 tile per workgroup exactly as k_dgemm_nt_sym<8>, the iterate ping-pongs between X and Y,
// and between two products every workgroup passes a grid barrier. Two barrier forms:
//   flat: one monotonic counter; lane 0 of every workgroup: release fence -> atomic add -> sc1-load
//         poll -> acquire fence (the guide's "barrier-counter");
//   xcd:  per-XCD arrival counters (the last arriver of an XCD makes the release and arrives at the
//         top counter; the last XCD publishes the generation), everybody polls the generation and
//         acquires (the guide's "barrier-xcd").
// Spins are bounded: a barrier that does not complete sets *abort and every workgroup leaves.
struct ChainBar {
  unsigned cnt;            // flat: arrivals (monotonic)
  unsigned pad0[31];
  unsigned top;            // xcd: XCDs that have arrived (monotonic)
  unsigned pad1[31];
  unsigned gen;            // xcd: completed barriers
  unsigned pad2[31];
  unsigned abort;
  unsigned pad3[31];
  unsigned xcnt[8][32];    // xcd: arrivals per XCD (monotonic), a 128-byte line each
  unsigned census[8][32];  // workgroups per XCD (filled before the first product)
};
__device__ inline unsigned ld_sc1(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool bar_flat(ChainBar *b, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&b->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (ld_sc1(&b->cnt) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22) || ld_sc1(&b->abort)) {
        __hip_atomic_store(&b->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __shared__ int okk;
  if (threadIdx.x == 0) okk = ok;
  __syncthreads();
  return okk;
}
__device__ inline bool bar_xcd(ChainBar *b, unsigned round /* 1, 2, ... */, unsigned xcc, unsigned nx_here,
                               unsigned nxcd) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    // (stores of this workgroup are in the XCD's L2; the last arriver of the XCD writes the L2 back)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned a = __hip_atomic_fetch_add(&b->xcnt[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == nx_here * round) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1 == nxcd * round)
        __hip_atomic_store(&b->gen, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int spins = 0;
    while (ld_sc1(&b->gen) < round) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22) || ld_sc1(&b->abort)) {
        __hip_atomic_store(&b->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __shared__ int okk;
  if (threadIdx.x == 0) okk = ok;
  __syncthreads();
  return okk;
}
template <int NW, bool XCD>
__global__ __launch_bounds__(64 * NW) void k_ns_chain(double *X, double *Y, int J, int nprod, double alpha,
                                                      ChainBar *bar) {
  constexpr int UN = 13;
  __shared__ double part[NW - 1][4][64];
  __shared__ unsigned s_info[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int nt = (J + 15) / 16;
  int ti = 0, rem = blockIdx.x;
  while (rem >= nt - ti) {
    rem -= nt - ti;
    ti++;
  }
  const int tj = ti + rem;
  const int i0 = ti * 16, j0 = tj * 16;
  const int ia = min(i0 + l16, J - 1), jb = min(j0 + l16, J - 1);
  unsigned xcc = 0, nx_here = 0, nxcd = 0;
  unsigned flat_target = 0, round = 0;
  if (XCD) {
    // census: who shares my XCD (placement is read, never assumed), then one flat barrier
    if (threadIdx.x == 0) {
      xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7;
      __hip_atomic_fetch_add(&bar->census[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    flat_target += gridDim.x;
    if (!bar_flat(bar, flat_target)) return;
    if (threadIdx.x == 0) {
      unsigned n = 0;
      for (int x = 0; x < 8; x++) n += ld_sc1(&bar->census[x][0]) > 0;
      s_info[0] = ld_sc1(&bar->census[xcc][0]);
      s_info[1] = n;
    }
    __syncthreads();
    nx_here = s_info[0];
    nxcd = s_info[1];
  }
  for (int it = 0; it < nprod; it++) {
    const double *__restrict__ ap = X + ia;
    const double *__restrict__ bp = X + jb;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int ksteps = (J + 3) / 4;
    const int spw = (ksteps + NW - 1) / NW;
    const int s_begin = wave * spw, s_end = min(ksteps, s_begin + spw);
    for (int s0 = s_begin; s0 < s_end; s0 += UN) {
      double av[UN], bv[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int k = (s0 + u) * 4 + g;
        const bool ok = (s0 + u) < s_end && k < J;
        const int kc = ok ? k : 0;
        const double a = ap[(int64_t)J * kc], b = bp[(int64_t)J * kc];
        av[u] = ok ? a : 0.0;
        bv[u] = ok ? b : 0.0;
      }
#pragma unroll
      for (int u = 0; u < UN; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) part[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
      const int j = j0 + l16;
      if (j < J) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = i0 + g + 4 * r;
          if (i < J && i <= j) {
            double v = acc[r];
#pragma unroll
            for (int w = 0; w < NW - 1; w++) v += part[w][r][lane];
            v *= alpha;
            Y[i + (int64_t)J * j] = v;
            if (i != j) Y[j + (int64_t)J * i] = v;
          }
        }
      }
    }
    double *t = X;
    X = Y;
    Y = t;
    if (it + 1 < nprod) {
      if (XCD) {
        if (!bar_xcd(bar, ++round, xcc, nx_here, nxcd)) return;
      } else {
        flat_target += gridDim.x;
        if (!bar_flat(bar, flat_target)) return;
      }
    }
  }
}
int main(int argc, char **argv) {
  const int J = argc > 1 ? atoi(argv[1]) : 400;
  double *X, *Y;
  CK(hipMalloc(&X, sizeof(double) * J * J));
  CK(hipMalloc(&Y, sizeof(double) * J * J));
  std::vector<double> h((size_t)J * J);
  for (int i = 0; i < J; i++)
    for (int j = 0; j < J; j++) h[i + (size_t)J * j] = (i == j ? 1.0 : 1e-3 * ((i * 7 + j * 13) % 11 - 5));
  CK(hipMemcpy(X, h.data(), sizeof(double) * J * J, hipMemcpyHostToDevice));
  chain<4>("k_dgemm_nt_sym<4>  (256 threads)", X, Y, J, 40);
  chain<8>("k_dgemm_nt_sym<8>  (512 threads)", X, Y, J, 40);
  chain<16>("k_dgemm_nt_sym<16> (1024 threads)", X, Y, J, 40);
  if (J <= 512) {  // tile -> XCD orders and the phases of one product
    const unsigned nt = (unsigned)((J + 15) / 16), ntri = nt * (nt + 1) / 2;
    ushort2 *dmap;
    unsigned long long *dst;
    CK(hipMalloc(&dmap, sizeof(ushort2) * ntri));
    CK(hipMalloc(&dst, sizeof(unsigned long long) * 8 * ntri));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 3; kind++) {
      std::vector<ushort2> tab = tile_table(J, kind);
      CK(hipMemcpy(dmap, tab.data(), sizeof(ushort2) * ntri, hipMemcpyHostToDevice));
      std::vector<float> tsv;
      const int n = 40;
      for (int rep = 0; rep < 7; rep++) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; i++) {
          hipLaunchKernelGGL((k_sym_probe<8, false>), dim3(ntri), dim3(512), 0, 0, X, X, Y, J, 1e-3, dmap, dst);
          std::swap(X, Y);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tsv.push_back(ms * 1e3f / n);
      }
      std::sort(tsv.begin(), tsv.end());
      // operand row-blocks an XCD needs under this table (blockIdx % 8 = XCD, as observed)
      double need = 0;
      for (int c = 0; c < 8; c++) {
        std::vector<char> seen(nt, 0);
        for (unsigned b = c; b < ntri; b += 8) seen[tab[b].x] = seen[tab[b].y] = 1;
        for (char v : seen) need += v;
      }
      printf("tile order %-28s J=%d: %.2f us per product; an XCD touches %.1f of %u operand row-blocks\n",
             kind == 0 ? "row-major (product)" : kind == 1 ? "Morton patches per XCD" : "tile rows dealt to XCDs", J,
             tsv[3], need / 8, nt);
    }
    {  // all 26 loads of a lane in flight before the first MFMA (the compiler otherwise sinks loads to save registers)
      std::vector<ushort2> tab = tile_table(J, 0);
      CK(hipMemcpy(dmap, tab.data(), sizeof(ushort2) * ntri, hipMemcpyHostToDevice));
      std::vector<float> tsv;
      const int n = 40;
      for (int rep = 0; rep < 7; rep++) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; i++) {
          hipLaunchKernelGGL((k_sym_probe<8, false, true>), dim3(ntri), dim3(512), 0, 0, X, X, Y, J, 1e-3, dmap, dst);
          std::swap(X, Y);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tsv.push_back(ms * 1e3f / n);
      }
      std::sort(tsv.begin(), tsv.end());
      printf("row-major, scheduling barrier between the loads and the MFMAs  J=%d: %.2f us per product\n", J, tsv[3]);
    }
    {  // phases (stamps cost a wait for all loads before the first MFMA: an upper bound of the load phase)
      std::vector<ushort2> tab = tile_table(J, 0);
      CK(hipMemcpy(dmap, tab.data(), sizeof(ushort2) * ntri, hipMemcpyHostToDevice));
      for (int i = 0; i < 6; i++) {  // a chain, so that the stamped launch starts behind a kernel boundary
        hipLaunchKernelGGL((k_sym_probe<8, true>), dim3(ntri), dim3(512), 0, 0, X, X, Y, J, 1e-3, dmap, dst);
        std::swap(X, Y);
      }
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> st(8 * (size_t)ntri);
      CK(hipMemcpy(st.data(), dst, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
      unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
      double ph[4] = {0, 0, 0, 0};
      for (unsigned b = 0; b < ntri; b++) {
        const unsigned long long *q = &st[8 * (size_t)b];
        tmin = std::min(tmin, q[0]);
        smax = std::max(smax, q[0]);
        tmax = std::max(tmax, q[4]);
        for (int k = 0; k < 4; k++) ph[k] += (double)(q[k + 1] - q[k]) * 0.01 / ntri;
      }
      printf("phases of one product (lane 0 of wave 0, mean over %u workgroups, us): operands %.2f | MFMA %.2f | "
             "combine %.2f | stores %.2f;  first -> last workgroup start %.2f us, first start -> last stamp %.2f us\n",
             ntri, ph[0], ph[1], ph[2], ph[3], (double)(smax - tmin) * 0.01, (double)(tmax - tmin) * 0.01);
    }
    CK(hipMemcpy(X, h.data(), sizeof(double) * J * J, hipMemcpyHostToDevice));
    CK(hipFree(dmap));
    CK(hipFree(dst));
  }
  // the LDS-tiled kernel (large matrices)
  for (int ts : {32, 64}) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned nt = (unsigned)((J + ts - 1) / ts), ntri = nt * (nt + 1) / 2;
    std::vector<float> tsv;
    const int n = 40;
    for (int rep = 0; rep < 7; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < n; i++) {
        if (ts == 32)
          hipLaunchKernelGGL(k_dgemm_nt_sym_lds<32>, dim3(ntri), dim3(256), 0, 0, X, (int64_t)J, X, (int64_t)J,
                             (const double *)nullptr, (int64_t)0, Y, (int64_t)J, J, J, 1e-3, 0.0, 0, (double *)nullptr);
        else
          hipLaunchKernelGGL(k_dgemm_nt_sym_lds<64>, dim3(ntri), dim3(256), 0, 0, X, (int64_t)J, X, (int64_t)J,
                             (const double *)nullptr, (int64_t)0, Y, (int64_t)J, J, J, 1e-3, 0.0, 0, (double *)nullptr);
        std::swap(X, Y);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tsv.push_back(ms * 1e3f / n);
    }
    std::sort(tsv.begin(), tsv.end());
    printf("k_dgemm_nt_sym_lds<%d> (%u tiles)       J=%d: %.2f us per product = %.1f TFLOP/s nominal (2 J^3)\n", ts, ntri, J,
           tsv[3], 2.0 * J * J * J / tsv[3] * 1e-6);
  }
  {  // correctness of the LDS kernel against the latency kernel: one product of the start matrix
    CK(hipMemcpy(X, h.data(), sizeof(double) * J * J, hipMemcpyHostToDevice));
    double *R1, *R2;
    CK(hipMalloc(&R1, sizeof(double) * J * J));
    CK(hipMalloc(&R2, sizeof(double) * J * J));
    const unsigned nt16 = (unsigned)((J + 15) / 16), nt64 = (unsigned)((J + 63) / 64), nt32 = (unsigned)((J + 31) / 32);
    hipLaunchKernelGGL(k_dgemm_nt_sym<8>, dim3(nt16 * (nt16 + 1) / 2), dim3(512), 0, 0, X, (int64_t)J, X, (int64_t)J,
                       (const double *)X, (int64_t)J, R1, (int64_t)J, J, J, 0.7, 0.3, 0, (double *)nullptr);
    std::vector<double> a((size_t)J * J), b((size_t)J * J);
    CK(hipMemcpy(a.data(), R1, sizeof(double) * J * J, hipMemcpyDeviceToHost));
    for (int ts : {32, 64}) {
      if (ts == 32)
        hipLaunchKernelGGL(k_dgemm_nt_sym_lds<32>, dim3(nt32 * (nt32 + 1) / 2), dim3(256), 0, 0, X, (int64_t)J, X,
                           (int64_t)J, (const double *)X, (int64_t)J, R2, (int64_t)J, J, J, 0.7, 0.3, 0, (double *)nullptr);
      else
        hipLaunchKernelGGL(k_dgemm_nt_sym_lds<64>, dim3(nt64 * (nt64 + 1) / 2), dim3(256), 0, 0, X, (int64_t)J, X,
                           (int64_t)J, (const double *)X, (int64_t)J, R2, (int64_t)J, J, J, 0.7, 0.3, 0, (double *)nullptr);
      CK(hipMemcpy(b.data(), R2, sizeof(double) * J * J, hipMemcpyDeviceToHost));
      double md = 0, mx = 0;
      for (size_t e = 0; e < a.size(); e++) {
        md = std::max(md, std::fabs(a[e] - b[e]));
        mx = std::max(mx, std::fabs(a[e]));
      }
      printf("   lds<%d> vs latency kernel: max |diff| %.3e of max %.3e\n", ts, md, mx);
    }
    CK(hipFree(R1));
    CK(hipFree(R2));
  }
  // the chain as ONE persistent launch with a grid barrier between the products
  for (int form = 0; form < 2; form++) {
    if (((J + 15) / 16) * (((J + 15) / 16) + 1) / 2 > 512) {
      printf("persistent chain: skipped at J=%d (more workgroups than can be resident at once)\n", J);
      break;
    }
    ChainBar *bar;
    CK(hipMalloc(&bar, sizeof(ChainBar)));
    const unsigned nt = (unsigned)((J + 15) / 16), ntri = nt * (nt + 1) / 2;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int n = 40;
    std::vector<float> tsv;
    unsigned aborted = 0;
    std::vector<double> ref((size_t)J * J), got((size_t)J * J);
    for (int rep = 0; rep < 8; rep++) {
      CK(hipMemcpy(X, h.data(), sizeof(double) * J * J, hipMemcpyHostToDevice));
      if (rep == 0) {  // reference: three products launch by launch, and the same three in one launch
        double *a = X, *b = Y;
        for (int i = 0; i < 3; i++) {
          hipLaunchKernelGGL(k_dgemm_nt_sym<8>, dim3(ntri), dim3(512), 0, 0, a, (int64_t)J, a, (int64_t)J,
                             (const double *)nullptr, (int64_t)0, b, (int64_t)J, J, J, 1.0, 0.0, 0, (double *)nullptr);
          std::swap(a, b);
        }
        CK(hipMemcpy(ref.data(), a, sizeof(double) * J * J, hipMemcpyDeviceToHost));
        CK(hipMemcpy(X, h.data(), sizeof(double) * J * J, hipMemcpyHostToDevice));
        CK(hipMemset(bar, 0, sizeof(ChainBar)));
        if (form == 0)
          hipLaunchKernelGGL((k_ns_chain<8, false>), dim3(ntri), dim3(512), 0, 0, X, Y, J, 3, 1.0, bar);
        else
          hipLaunchKernelGGL((k_ns_chain<8, true>), dim3(ntri), dim3(512), 0, 0, X, Y, J, 3, 1.0, bar);
        CK(hipMemcpy(got.data(), Y, sizeof(double) * J * J, hipMemcpyDeviceToHost));  // (3 products: the result is in Y)
        continue;
      }
      CK(hipMemset(bar, 0, sizeof(ChainBar)));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      if (form == 0)
        hipLaunchKernelGGL((k_ns_chain<8, false>), dim3(ntri), dim3(512), 0, 0, X, Y, J, n, 1.0, bar);
      else
        hipLaunchKernelGGL((k_ns_chain<8, true>), dim3(ntri), dim3(512), 0, 0, X, Y, J, n, 1.0, bar);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tsv.push_back(ms * 1e3f / n);
      ChainBar hb;
      CK(hipMemcpy(&hb, bar, sizeof(ChainBar), hipMemcpyDeviceToHost));
      aborted |= hb.abort;
    }
    double md = 0, mx = 0;
    for (size_t e = 0; e < ref.size(); e++) {
      md = std::max(md, std::fabs(ref[e] - got[e]));
      mx = std::max(mx, std::fabs(ref[e]));
    }
    std::sort(tsv.begin(), tsv.end());
    printf("persistent chain, %s grid barrier (%u workgroups x 512)  J=%d: %.2f us per product (one launch of %d, median of 7)%s; "
           "max |diff| to the launch-by-launch chain %.3e of %.3e\n",
           form == 0 ? "flat" : "XCD-hierarchical", ntri, J, tsv[3], n, aborted ? "  [A BARRIER TIMED OUT]" : "", md, mx);
    CK(hipFree(bar));
  }
  CK(hipMemcpy(X, h.data(), sizeof(double) * J * J, hipMemcpyHostToDevice));
  {  // the non-symmetric tile kernel on the full matrix, for scale
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int n = 40;
    dim3 grid((unsigned)((J + 15) / 16), (unsigned)((J + 15) / 16));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; i++) {
      hipLaunchKernelGGL(k_dgemm_nx<false>, grid, dim3(256), 0, 0, X, (int64_t)J, X, (int64_t)J,
                         (const double *)nullptr, (int64_t)0, Y, (int64_t)J, J, J, J, 1e-3, 0.0);
      std::swap(X, Y);
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s J=%d: %.2f us per product\n", "k_dgemm_nx<false> (all tiles)", J, ms * 1e3f / n);
  }
  {  // an empty-ish kernel chain: the floor of a dependent launch on this box
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int n = 200;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; i++)
      hipLaunchKernelGGL(k_chk_sums, dim3(1), dim3(256), 0, 0, X, X, 1, Y);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s       %.2f us per launch\n", "trivial one-workgroup kernel chain", ms * 1e3f / n);
  }
  return 0;
}
