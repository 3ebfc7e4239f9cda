// hip_ops.hip — the product implementation of ppals::Ops: hand-written HIP kernels for gfx950,
// launched on one engine-owned stream. No CPU fallback: construction throws without a HIP device.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <map>
#include <vector>

#include "hip_ops.h"
#include "roctx_ranges.h"
#include "preload_policy.h"
#include "kernels_scan.hip.h"
#include "kernels_small.hip.h"
#include "kernels_eig.hip.h"

namespace ppals {
#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + \
                               __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");  \
  } while (0)

// workgroup size of k_chol_m for an n x 2n elimination (a barrier per pivot: fewer waves, cheaper barrier)
static inline unsigned k_chol_threads(int n) {
  (void)n;  // measured: 256 threads at n = 20 make the launch 17.8 -> 22.3 us (the per-pivot update,
            // not the barrier, is the longer part); 1024 with LDS-only barriers: see below
  return 1024u;
}
static inline int grid_for(int64_t n, int block, int cap = 4096) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// The vendor eigensolver libraries, process-wide. See hip_preload_eigensolver (hip_ops.h).
struct EigLibs {
  void *blas = nullptr, *solver = nullptr;
};
static bool g_hip_runtime_up = false;  // set by the first HipOps (this library's own use of the runtime)
static EigLibs &eig_libs() {
  static EigLibs l;
  if (l.blas && l.solver) return l;
  switch (late_preload_policy(false, g_hip_runtime_up, getenv("PPALS_STRICT_PRELOAD"))) {
    case kPreloadRefuse: throw Unsupported(late_preload_message());
    case kPreloadWarn: fprintf(stderr, "%s\n", late_preload_message()); fflush(stderr); break;
    default: break;
  }
  l.blas = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
  if (!l.blas) l.blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
  l.solver = dlopen("librocsolver.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!l.solver) l.solver = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
  if (!l.blas || !l.solver)
    throw std::runtime_error(std::string("ppals: cannot load rocSOLVER: ") + dlerror());
  return l;
}

class HipOps : public Ops {
 public:
  explicit HipOps(int device) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      throw std::runtime_error(
          "ppals: no HIP device available (the engine has no CPU fallback; hipGetDeviceCount: " +
          std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
    if (device < 0 || device >= ndev)
      throw std::runtime_error("ppals: device index out of range");
    dev_ = device;
    g_hip_runtime_up = true;
    HIP_CHECK(hipSetDevice(dev_));
    HIP_CHECK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev_));
    ncu_ = prop.multiProcessorCount;
    HIP_CHECK(hipFuncSetAttribute((const void *)k_gram_system,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_cp_update,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_top_eig_small,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_cp_mode_update<false, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_cp_mode_update<true, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    if (const char *v = getenv("PPALS_FORCE_JACOBI")) force_jacobi_ = atoi(v);
    if (const char *v = getenv("PPALS_EIG_FAST")) {
      eig_fast_ = atoi(v);  // 0: always the full solver; 2: warm steps as usual, cold starts on the full solver
      eig_cold_ = eig_fast_ != 2;
    }
    if (const char *v = getenv("PPALS_EIG_DEBUG")) eig_debug_ = atoi(v);
    if (const char *v = getenv("PPALS_EIG_SIGMA_SCALE")) eig_sigma_scale_ = atof(v);
    if (const char *v = getenv("PPALS_FORCE_EIGINV")) force_eiginv_ = atoi(v);
    HIP_CHECK(hipFuncSetAttribute((const void *)k_chol_rinv,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_chol_m,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_rr_apply,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    if (const char *v = getenv("PPALS_SYM_LDS_MIN")) sym_lds_min_ = std::max(32, atoi(v));
    if (const char *v = getenv("PPALS_EIG_DEFER")) {
      eig_defer_ok_ = atoi(v);
      handover_ok_ = eig_defer_ok_ != 2;  // (2: deferred, the second stream handed over by events)
    }
    if (const char *v = getenv("PPALS_EIG_DEFER_FAIL")) eig_defer_fail_ = atoi(v);
    if (const char *v = getenv("PPALS_SCAN_TAIL")) scan_tail_on_ = atoi(v) != 0;
    if (const char *v = getenv("PPALS_SCAN_WIDE")) wide_enabled_ = atoi(v) != 0;
    if (const char *v = getenv("PPALS_COLD_SUBSPACE_FROM")) cold_subspace_from_ = std::max(65, atoi(v));
    if (const char *v = getenv("PPALS_PERSIST_MULT")) persist_mult_ = std::max(1, atoi(v));  // (probe: tools/runs/r05_m.sh)
    HIP_CHECK(hipFuncSetAttribute((const void *)k_rmult_chol,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_jacobi_onesided,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_rr_small,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_gram_system_lds,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_gram_system_mfma,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    if (const char *v = getenv("PPALS_GJ_SCALAR")) gj_scalar_ = atoi(v);
  }
  ~HipOps() override {
    hipSetDevice(dev_);
    hipStreamSynchronize(st_);
    if (st2_) hipStreamSynchronize(st2_);  // (checks of deferred steps still read the slots' buffers)
    for (auto &ev : events_) {
      hipEventDestroy(ev.a);
      hipEventDestroy(ev.b);
    }
    for (auto &t : timers_) {
      hipEventDestroy(t.a);
      hipEventDestroy(t.b);
    }
    if (ws_mttv_) hipFree(ws_mttv_);
    if (ws_pack_) hipFree(ws_pack_);
    if (ws_slab_) hipFree(ws_slab_);
    if (ws_krp_) hipFree(ws_krp_);
    if (ws_part_) hipFree(ws_part_);
    if (ws_small_) hipFree(ws_small_);
    if (ws_big_) hipFree(ws_big_);
    if (ws_big2_) hipFree(ws_big2_);
    if (ws_eig_) hipFree(ws_eig_);
    if (ws_orth_) hipFree(ws_orth_);
    if (ws_pow_) hipFree(ws_pow_);
    if (ws_cold_) hipFree(ws_cold_);
    if (ws_cold2_) hipFree(ws_cold2_);
    if (ws_cholm_) hipFree(ws_cholm_);
    if (ws_gsfb_) hipFree(ws_gsfb_);
    if (ws_jac_) hipFree(ws_jac_);
    if (eig_host_) hipHostFree(eig_host_);
    if (ws_part2_) hipFree(ws_part2_);
    for (auto &kv : eig_state_) {
      if (kv.second.Q) hipFree(kv.second.Q);
      if (kv.second.Qn) hipFree(kv.second.Qn);
      free_lazy(kv.second);
    }
    for (auto &kv : eig_small_)
      if (kv.second.Q) hipFree(kv.second.Q);
    if (st2_) hipStreamDestroy(st2_);
    if (handover_) hipFree(handover_);
    hipStreamDestroy(st_);
  }

  void *alloc(size_t bytes) override {
    void *p = nullptr;
    HIP_CHECK(hipSetDevice(dev_));
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
    return p;
  }
  void free(void *p) override {
    if (p) {
      hipStreamSynchronize(st_);
      hipFree(p);
    }
  }
  void h2d(void *dst, const void *src, size_t bytes) override {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
  }
  void d2h(void *dst, const void *src, size_t bytes) override {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
  }
  void d2d(void *dst, const void *src, size_t bytes) override {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st_));
  }
  void zero(void *p, size_t bytes) override { HIP_CHECK(hipMemsetAsync(p, 0, bytes, st_)); }
  void sync() override { HIP_CHECK(hipStreamSynchronize(st_)); }
  void *stream() override { return (void *)st_; }
  void bind() override { HIP_CHECK(hipSetDevice(dev_)); }

  // ------------------------------------------------------------------ generation / norms
  void fill_uniform(void *V, int dt, int64_t l0, int64_t g0, int64_t row0, int64_t rest,
                    uint64_t seed, double lo, double hi) override {
    int64_t n = l0 * rest;
    int g = grid_for(n, 256, 16384);
    if (dt == F32)
      hipLaunchKernelGGL(k_fill_uniform<float>, dim3(g), dim3(256), 0, st_, (float *)V, l0, g0,
                         row0, rest, seed, lo, hi);
    else
      hipLaunchKernelGGL(k_fill_uniform<double>, dim3(g), dim3(256), 0, st_, (double *)V, l0, g0,
                         row0, rest, seed, lo, hi);
    HIP_CHECK(hipGetLastError());
  }

  void fill_laplacian(void *V, int dt, int64_t l0, int64_t g0, int64_t row0, int64_t rest,
                      int ndigits, int s) override {
    int g = grid_for(l0 * rest, 256, 16384);
    if (dt == F32)
      hipLaunchKernelGGL(k_fill_laplacian<float>, dim3(g), dim3(256), 0, st_, (float *)V, l0, g0,
                         row0, rest, ndigits, s);
    else
      hipLaunchKernelGGL(k_fill_laplacian<double>, dim3(g), dim3(256), 0, st_, (double *)V, l0, g0,
                         row0, rest, ndigits, s);
    HIP_CHECK(hipGetLastError());
  }
  void add_uniform_noise(void *V, int dt, int64_t l0, int64_t g0, int64_t row0, int64_t rest,
                         uint64_t seed, double lo, double hi, double alpha) override {
    int g = grid_for(l0 * rest, 256, 16384);
    if (dt == F32)
      hipLaunchKernelGGL((k_uniform_noise<float, 0>), dim3(g), dim3(256), 0, st_, (float *)V, l0,
                         g0, row0, rest, seed, lo, hi, alpha, (double *)nullptr);
    else
      hipLaunchKernelGGL((k_uniform_noise<double, 0>), dim3(g), dim3(256), 0, st_, (double *)V, l0,
                         g0, row0, rest, seed, lo, hi, alpha, (double *)nullptr);
    HIP_CHECK(hipGetLastError());
  }
  void uniform_sumsq(int64_t l0, int64_t g0, int64_t row0, int64_t rest, uint64_t seed, double lo,
                     double hi, double *out) override {
    int g = grid_for(l0 * rest, 256, 4096);
    double *part = (double *)ensure(ws_part_, ws_part_sz_, sizeof(double) * g);
    hipLaunchKernelGGL((k_uniform_noise<double, 1>), dim3(g), dim3(256), 0, st_, (double *)nullptr,
                       l0, g0, row0, rest, seed, lo, hi, 0.0, part);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, g, out);
    HIP_CHECK(hipGetLastError());
  }

  // K10 on the matrix cores (kernels_scan.hip.h: k_rank_mfma); false: shape not covered
  template <typename TV, int MODE>
  bool rank_mfma(void *V, int64_t M, int64_t K, const double *Q, const double *P, int R,
                 double *out) {
    constexpr int VEC = ScanTraits<TV>::VEC;
    if (R > 32 || M % VEC != 0 || M < VEC || (((uintptr_t)V) & 15) != 0 || K < 1)
      return false;
    const int64_t nkb64 = (K + 15) / 16;
    if (nkb64 > 0x7fffffff) return false;
    const int nkb = (int)nkb64;
    const int64_t n_mtiles = (M + 64 * VEC - 1) / (64 * VEC);
    // enough workgroups to fill the chip several times, chunks of >= 8 column blocks
    // (workgroups per CU x 4 .. 128 instead of 16: 1.60-1.73 ms for the whole [diffV] call at cfg2,
    // 1.61 at 16 — tools/k10_probe.py, profiles/r03z_k10_geometry.txt: the geometry is not the lever)
    int nchunk = (int)std::min<int64_t>(nkb, std::max<int64_t>(1, ((int64_t)ncu_ * 16 + n_mtiles - 1) / n_mtiles));
    int per = (nkb + nchunk - 1) / nchunk;
    per = std::max(per, std::min(nkb, 8));
    nchunk = (nkb + per - 1) / per;
    if (n_mtiles > 0x7fffffff || nchunk > 65535) return false;
    // residual: a remainder of 1 or 2 ranks goes to the vector pipe instead of a padded matrix-core
    // step (kernels_scan.hip.h, REM), its P columns in LDS — if a chunk's share of them fits
    // (measured, tools/k10_probe.py at cfg2, profiles/r04I_k10_probe.txt: R = 10 1.28 -> 1.26 ms, R = 9
    // 1.17 ms, R = 13 1.38 ms against ~1.5 ms with a padded fourth step; two ranks beside three or four
    // steps — R = 14: 1.63 ms, R = 18: 1.78 ms — lose to padding: 2 waves per SIMD)
    int rem = (MODE == 1 && R > 4 && R <= 17 && (R % 4 == 1 || (R % 4 == 2 && R <= 10))) ? R % 4 : 0;
    if (sizeof(double) * (size_t)per * 16 * rem > 48 * 1024) rem = 0;
    const int RB = MODE == 2 ? 1 : (R - rem + 3) / 4;
    double *Ppk = nullptr;
    if (MODE != 2) {
      Ppk = (double *)ensure(ws_pack_, ws_pack_sz_, sizeof(double) * (size_t)nkb * RB * 64);
      hipLaunchKernelGGL(k_rank_pack, dim3(grid_for((int64_t)nkb * RB * 64, 256)), dim3(256), 0, st_,
                         P, K, R - rem, RB, nkb, Ppk);
    }
    dim3 grid((unsigned)n_mtiles, (unsigned)nchunk);
    double *part = nullptr;
    const int64_t npart = (int64_t)n_mtiles * nchunk;
    if (MODE != 0) part = (double *)ensure(ws_part_, ws_part_sz_, npart * sizeof(double));
    prof_begin(1, (double)M * K * sizeof(TV));
    const size_t lds_rem = sizeof(double) * (size_t)per * 16 * rem;
    bool done = false;
    if constexpr (MODE == 1) {
      if (rem) {
#define PPALS_RANK_REM(MRB, RM)                                                                        \
  hipLaunchKernelGGL((k_rank_mfma<TV, 1, MRB, RM>), grid, dim3(256), lds_rem, st_, (TV *)V, M, K, Q, Ppk, R, \
                     RB, per, nkb, part, P)
        if (RB <= 2) {
          if (rem == 1) PPALS_RANK_REM(2, 1);
          else PPALS_RANK_REM(2, 2);
        } else {
          PPALS_RANK_REM(4, 1);
        }
#undef PPALS_RANK_REM
        done = true;
      }
    }
    if (done) {
    } else if (RB <= 3)
      hipLaunchKernelGGL((k_rank_mfma<TV, MODE, 3>), grid, dim3(256), 0, st_, (TV *)V, M, K, Q, Ppk, R,
                         RB, per, nkb, part);
    else if (RB <= 4)
      hipLaunchKernelGGL((k_rank_mfma<TV, MODE, 4>), grid, dim3(256), 0, st_, (TV *)V, M, K, Q, Ppk, R,
                         RB, per, nkb, part);
    else
      hipLaunchKernelGGL((k_rank_mfma<TV, MODE, 8>), grid, dim3(256), 0, st_, (TV *)V, M, K, Q, Ppk, R,
                         RB, per, nkb, part);
    prof_end();
    HIP_CHECK(hipGetLastError());
    if (MODE != 0) {
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, (int)npart, out);
      HIP_CHECK(hipGetLastError());
    }
    return true;
  }
  // generation / residual for 32 < R <= 256 on the matrix cores (k_rank_split); false: not covered
  template <typename TV, int MODE>
  bool rank_split(void *V, int64_t M, int64_t K, const double *Q, const double *P, int R,
                  double *out) {
    constexpr int VEC = ScanTraits<TV>::VEC;
    if (R <= 32 || R > 256 || M % VEC != 0 || M < VEC ||
        (((uintptr_t)V) & 15) != 0 || K < 1)
      return false;
    const int RB = (R + 3) / 4;
    const int64_t nkb64 = (K + 15) / 16;
    if (nkb64 > 0x7fffffff) return false;
    const int nkb = (int)nkb64;
    double *Ppk = (double *)ensure(ws_pack_, ws_pack_sz_, sizeof(double) * (size_t)nkb * RB * 64);
    hipLaunchKernelGGL(k_rank_pack, dim3(grid_for((int64_t)nkb * RB * 64, 256)), dim3(256), 0, st_, P,
                       K, R, RB, nkb, Ppk);
    const int64_t n_mtiles = (M + 16 * VEC - 1) / (16 * VEC);
    int nchunk = (int)std::min<int64_t>(nkb, std::max<int64_t>(1, ((int64_t)ncu_ * 8 + n_mtiles - 1) / n_mtiles));
    int per = (nkb + nchunk - 1) / nchunk;
    per = std::max(per, std::min(nkb, 8));
    nchunk = (nkb + per - 1) / per;
    if (n_mtiles > 0x7fffffff || nchunk > 65535) return false;
    dim3 grid((unsigned)n_mtiles, (unsigned)nchunk);
    const int64_t npart = (int64_t)n_mtiles * nchunk;
    double *part = MODE == 1 ? (double *)ensure(ws_part_, ws_part_sz_, npart * sizeof(double)) : nullptr;
    prof_begin(1, (double)M * K * sizeof(TV));
    const int rbw = (RB + 3) / 4;
    if (rbw <= 4)
      hipLaunchKernelGGL((k_rank_split<TV, MODE, 4>), grid, dim3(256), 0, st_, (TV *)V, M, K, Q, Ppk, R,
                         RB, per, nkb, part);
    else if (rbw <= 8)
      hipLaunchKernelGGL((k_rank_split<TV, MODE, 8>), grid, dim3(256), 0, st_, (TV *)V, M, K, Q, Ppk, R,
                         RB, per, nkb, part);
    else
      hipLaunchKernelGGL((k_rank_split<TV, MODE, 16>), grid, dim3(256), 0, st_, (TV *)V, M, K, Q, Ppk,
                         R, RB, per, nkb, part);
    prof_end();
    HIP_CHECK(hipGetLastError());
    if (MODE == 1) {
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, (int)npart, out);
      HIP_CHECK(hipGetLastError());
    }
    return true;
  }
  template <typename TV, int MODE>
  void rank_stream(void *V, int64_t M, int64_t K, const double *Q, const double *P, int R,
                   double *out) {
    if (rank_mfma<TV, MODE>(V, M, K, Q, P, R, out)) return;
    if constexpr (MODE != 2)
      if (rank_split<TV, MODE>(V, M, K, Q, P, R, out)) return;
    int kch = 32;
    while ((K + kch - 1) / kch > 65535) kch *= 2;
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)((K + kch - 1) / kch));
    double *part = nullptr;
    int64_t npart = (int64_t)grid.x * grid.y;
    if (MODE != 0) part = (double *)ensure(ws_part_, ws_part_sz_, npart * sizeof(double));
    TV *v = (TV *)V;
    if (R > 64)
      hipLaunchKernelGGL((k_rank_stream_any<TV, MODE>), grid, dim3(256), 0, st_, v, M, K, Q, P, R,
                         kch, part);
    else if (R <= 16)
      hipLaunchKernelGGL((k_rank_stream<TV, 16, MODE>), grid, dim3(256), 0, st_, v, M, K, Q, P, R,
                         kch, part);
    else if (R <= 32)
      hipLaunchKernelGGL((k_rank_stream<TV, 32, MODE>), grid, dim3(256), 0, st_, v, M, K, Q, P, R,
                         kch, part);
    else
      hipLaunchKernelGGL((k_rank_stream<TV, 64, MODE>), grid, dim3(256), 0, st_, v, M, K, Q, P, R,
                         kch, part);
    HIP_CHECK(hipGetLastError());
    if (MODE != 0) {
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, (int)npart, out);
      HIP_CHECK(hipGetLastError());
    }
  }
  void fill_rank(void *V, int dt, int64_t M, int64_t K, const double *Q, const double *P,
                 int R) override {
    if (dt == F32)
      rank_stream<float, 0>(V, M, K, Q, P, R, nullptr);
    else
      rank_stream<double, 0>(V, M, K, Q, P, R, nullptr);
  }
  void residual_sq(const void *V, int dt, int64_t M, int64_t K, const double *Q, const double *P,
                   int R, double *out) override {
    RoctxRange roctx_("K10 residual / norm stream");
    void *v = const_cast<void *>(V);
    if (Q == nullptr) {
      if (dt == F32)
        rank_stream<float, 2>(v, M, K, nullptr, nullptr, 0, out);
      else
        rank_stream<double, 2>(v, M, K, nullptr, nullptr, 0, out);
    } else {
      if (dt == F32)
        rank_stream<float, 1>(v, M, K, Q, P, R, out);
      else
        rank_stream<double, 1>(v, M, K, Q, P, R, out);
    }
  }
  void upload_shard(void *V, int dt, const double *host_full, int64_t l0, int64_t g0,
                    int64_t row0, int64_t rest) override {
    // stage `chunk` columns (each l0 contiguous rows of the shard) at a time through a device
    // fp64 buffer, converting to the tensor's storage type on the device
    const int64_t chunk_cols = std::max<int64_t>(1, (int64_t)(64 << 20) / (8 * l0));
    std::vector<double> host((size_t)(chunk_cols * l0));
    double *stage = (double *)alloc(sizeof(double) * chunk_cols * l0);
    for (int64_t c0 = 0; c0 < rest; c0 += chunk_cols) {
      int64_t nc = std::min(chunk_cols, rest - c0);
      for (int64_t c = 0; c < nc; c++)
        std::copy(host_full + (c0 + c) * g0 + row0, host_full + (c0 + c) * g0 + row0 + l0,
                  host.begin() + c * l0);
      h2d(stage, host.data(), sizeof(double) * nc * l0);
      int64_t n = nc * l0;
      if (dt == F32)
        hipLaunchKernelGGL(k_convert_rows<float>, dim3(grid_for(n, 256)), dim3(256), 0, st_,
                           (float *)V + c0 * l0, stage, n);
      else
        hipLaunchKernelGGL(k_convert_rows<double>, dim3(grid_for(n, 256)), dim3(256), 0, st_,
                           (double *)V + c0 * l0, stage, n);
      HIP_CHECK(hipGetLastError());
      sync();
    }
    free(stage);
  }

  void download_shard(const void *V, int dt, double *host_full, int64_t l0, int64_t g0,
                      int64_t row0, int64_t rest) override {
    const int64_t chunk_cols = std::max<int64_t>(1, (int64_t)(64 << 20) / (8 * l0));
    std::vector<double> host((size_t)(chunk_cols * l0));
    double *stage = (double *)alloc(sizeof(double) * chunk_cols * l0);
    for (int64_t c0 = 0; c0 < rest; c0 += chunk_cols) {
      const int64_t nc = std::min(chunk_cols, rest - c0), n = nc * l0;
      if (dt == F32)
        hipLaunchKernelGGL(k_widen_rows<float>, dim3(grid_for(n, 256)), dim3(256), 0, st_, stage,
                           (const float *)V + c0 * l0, n);
      else
        hipLaunchKernelGGL(k_widen_rows<double>, dim3(grid_for(n, 256)), dim3(256), 0, st_, stage,
                           (const double *)V + c0 * l0, n);
      HIP_CHECK(hipGetLastError());
      d2h(host.data(), stage, sizeof(double) * n);
      for (int64_t c = 0; c < nc; c++)
        std::copy(host.begin() + c * l0, host.begin() + (c + 1) * l0,
                  host_full + (c0 + c) * g0 + row0);
    }
    free(stage);
  }

  void *try_alloc(size_t bytes) override {
    void *p = nullptr;
    hipSetDevice(dev_);
    if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    return p;
  }
  void scan_store_mode(int mode) override { scan_nt_mode_ = mode; }
  int scan_nt_mode_ = -1;
  size_t mem_available() override {
    size_t fr = 0, tot = 0;
    hipSetDevice(dev_);
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
      (void)hipGetLastError();
      return (size_t)-1;
    }
    return fr;
  }
  void unpack_shards(const void *stage, int dt, int64_t s0, int64_t rest, int64_t blk, int P,
                     int64_t chunk_bytes, void *full) override {
    const int g = grid_for(s0 * rest, 256, 16384);
    if (dt == F32)
      hipLaunchKernelGGL(k_unpack_shards<float>, dim3(g), dim3(256), 0, st_, (const char *)stage,
                         s0, rest, blk, P, chunk_bytes, (float *)full);
    else
      hipLaunchKernelGGL(k_unpack_shards<double>, dim3(g), dim3(256), 0, st_, (const char *)stage,
                         s0, rest, blk, P, chunk_bytes, (double *)full);
    HIP_CHECK(hipGetLastError());
  }
  void transpose2d(const void *src, int dt, int64_t rows, int64_t cols, void *dst) override {
    transpose_batched(src, dt, rows, cols, 1, dst);
  }
  void transpose_batched(const void *src, int dt, int64_t rows, int64_t cols, int64_t batch,
                         void *dst) override {
    const int64_t nb = ((rows + 63) / 64) * ((cols + 63) / 64);
    if (nb > 0x7fffffff) throw std::runtime_error("ppals: transpose grid too large");
    const size_t step = (size_t)rows * cols * dtype_size(dt);
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {  // grid.y limit
      const unsigned nby = (unsigned)std::min<int64_t>(65535, batch - b0);
      const char *s = (const char *)src + b0 * step;
      char *d = (char *)dst + b0 * step;
      if (dt == F32)
        hipLaunchKernelGGL(k_transpose<float>, dim3((unsigned)nb, nby), dim3(256), 0, st_,
                           (const float *)s, rows, cols, (float *)d);
      else
        hipLaunchKernelGGL(k_transpose<double>, dim3((unsigned)nb, nby), dim3(256), 0, st_,
                           (const double *)s, rows, cols, (double *)d);
    }
    HIP_CHECK(hipGetLastError());
  }

  // ------------------------------------------------------------------ KRP
  static KrpArgs krp_args(const FactorRef *f, int nf, int64_t *J) {
    KrpArgs a;
    int64_t j = 1;
    if (nf > MAX_ORDER) throw std::runtime_error("ppals: too many Khatri-Rao factors");
    for (int i = 0; i < nf; i++) {
      a.ptr[i] = f[i].ptr;
      a.rows[i] = f[i].rows;
      a.ld[i] = f[i].ld;
      j *= f[i].rows;
    }
    a.nf = nf;
    *J = j;
    return a;
  }
  void krp(double *out, const FactorRef *f, int nf, int col0, int ncols) override {
    int64_t J;
    KrpArgs a = krp_args(f, nf, &J);
    hipLaunchKernelGGL(k_krp, dim3(grid_for(J * ncols, 256)), dim3(256), 0, st_, out, a, J, col0,
                       ncols);
    HIP_CHECK(hipGetLastError());
  }

  // ------------------------------------------------------------------ scans
  template <typename TV>
  void scan_t(const TV *V, int64_t L, int64_t J, int64_t T, const FactorRef *f, int nf, int R,
              double *out, int64_t out_tstride, int64_t out_rstride, int out32, RowPad pad) {
    constexpr int VEC = ScanTraits<TV>::VEC;
    int64_t Jc;
    KrpArgs a = krp_args(f, nf, &Jc);
    if (Jc != J) throw std::runtime_error("ppals: scan_contract factor extents do not match J");
    if (pad.ld && (L % pad.ld || pad.ld % VEC || pad.valid > pad.ld || pad.valid <= 0 || L == 1))
      throw std::runtime_error("ppals: scan_contract padded rows inconsistent");
    const bool aligned_base = (((uintptr_t)V) & 15) == 0;
    // 65..128 columns of an fp32 tensor in ONE pass (k_scan_wide: MFMA-bound regime, the tensor tile
    // staged through LDS); whatever is left after whole passes of 128 goes the narrow way
    // (buffer loads: a k-block of 16 columns and the packed operand must be addressable with 32-bit offsets)
    const bool wide_ok = sizeof(TV) == 4 && wide_enabled_ && L > 1 && aligned_base && (L % 4 == 0) &&
                         L >= 4 && (double)L * J * T >= 1.0e6 && 16.0 * (double)L * 4.0 < 2.0e9 &&
                         ((double)J / 16.0 + 1.0) * 8.0 * 1024.0 < 2.0e9;
    int col_step = 64;
    for (int col0 = 0; col0 < R; col0 += col_step) {
      if (wide_ok && R - col0 > 64) {
        col_step = std::min(128, R - col0);
        if constexpr (sizeof(TV) == 4)
          scan_wide((const float *)V, L, J, T, a, col0, col_step, out, out_tstride, out_rstride, out32, pad);
        continue;
      }
      col_step = 64;
      const int ncols = std::min(64, R - col0);
      // n-tiles of 16 result columns: 3 for 33..48 columns (ranks such as 40, or the second
      // launch of R = 100 = 64 + 36) instead of padding them to 64 — these scans are MFMA-bound
      const int NT = ncols <= 16 ? 1 : (ncols <= 32 ? 2 : (ncols <= 48 ? 3 : 4));
      const int64_t nblk64 = (J + 4 * VEC - 1) / (4 * VEC);
      if (nblk64 > 0x7fffffff) throw std::runtime_error("ppals: contraction extent too large");
      const int nblk = (int)nblk64;
      const size_t pack_bytes = (size_t)nblk * NT * 4 * 16 * VEC * sizeof(TV);
      TV *P = (TV *)ensure(ws_pack_, ws_pack_sz_, pack_bytes);
      const bool prefix = (L == 1);
      hipLaunchKernelGGL(k_krp_pack<TV>, dim3(grid_for((int64_t)nblk * NT * 64 * VEC, 256)),
                         dim3(256), 0, st_, P, nblk, NT, prefix ? 1 : 0, a, J, col0, ncols);
      HIP_CHECK(hipGetLastError());
      double *o = out32 ? (double *)((float *)out + (int64_t)col0 * out_rstride)
                        : out + (int64_t)col0 * out_rstride;
      // algorithmic bytes of the launch: one read of the scanned tensor + the result it must write
      const double bytes = (double)L * (double)J * (double)T * sizeof(TV) +
                           (double)L * (double)T * ncols * (out32 ? 4.0 : 8.0);
      if (prefix) {
        // out[t + rs*n] = sum_j V[j + J*t] * B[j,n]      (M = J rows reduced, K = T columns)
        const int64_t M = J, K = T;
        const bool al = aligned_base && (M % VEC == 0);
        const int64_t ncolgrp = (K + 63) / 64;
        int nsplit = 1;
        const int target = ncu_ * 4;
        if (ncolgrp < target) nsplit = (int)std::min<int64_t>((target + ncolgrp - 1) / ncolgrp,
                                                              std::max(1, nblk / 64));
        if (nsplit < 1) nsplit = 1;
        const int per = (nblk + nsplit - 1) / nsplit;
        nsplit = (nblk + per - 1) / per;
        double *dst = o;
        int dst32 = out32;
        int64_t dst_ks = out_tstride, dst_ns = out_rstride, dst_ss = 0;
        if (nsplit > 1) {
          dst32 = 0;
          dst = (double *)ensure(ws_slab_, ws_slab_sz_, sizeof(double) * nsplit * ncols * K);
          dst_ks = 1;
          dst_ns = K;
          dst_ss = (int64_t)ncols * K;
        }
        dim3 grid((unsigned)ncolgrp, (unsigned)nsplit);
        dim3 grid_il((unsigned)((K + 15) / 16), (unsigned)nsplit);  // interleaved-waves variant
        prof_begin(0, bytes);
#define LAUNCH_PREFIX(NTv, ALv)                                                               \
  hipLaunchKernelGGL((k_scan_prefix<TV, NTv, ALv, 4>), grid, dim3(256), 0, st_, V, M, K, P, per, \
                     nblk, dst, dst_ks, dst_ns, dst_ss, ncols, dst32)
#define LAUNCH_PREFIX_FAST(NTv)                                                               \
  hipLaunchKernelGGL((k_scan_prefix_fast<TV, NTv, 12>), grid_il, dim3(256), 0, st_, V, M, K, P, per, \
                     nblk, dst, dst_ks, dst_ns, dst_ss, ncols, dst32)
        if (al && M >= VEC) {
          if (NT == 1) LAUNCH_PREFIX_FAST(1);
          else if (NT == 2) LAUNCH_PREFIX_FAST(2);
          else if (NT == 3) LAUNCH_PREFIX_FAST(3);
          else LAUNCH_PREFIX_FAST(4);
        } else if (al) {
          if (NT == 1) LAUNCH_PREFIX(1, true);
          else if (NT == 2) LAUNCH_PREFIX(2, true);
          else if (NT == 3) LAUNCH_PREFIX(3, true);
          else LAUNCH_PREFIX(4, true);
        } else {
          if (NT == 1) LAUNCH_PREFIX(1, false);
          else if (NT == 2) LAUNCH_PREFIX(2, false);
          else if (NT == 3) LAUNCH_PREFIX(3, false);
          else LAUNCH_PREFIX(4, false);
        }
#undef LAUNCH_PREFIX
#undef LAUNCH_PREFIX_FAST
        prof_end();
        HIP_CHECK(hipGetLastError());
        if (nsplit > 1) {
          hipLaunchKernelGGL(k_slab_reduce, dim3(grid_for(K * ncols, 256)), dim3(256), 0, st_, dst,
                             nsplit, dst_ss, K, ncols, o, out_tstride, out_rstride, out32,
                             (int64_t)0, (int64_t)0);
          HIP_CHECK(hipGetLastError());
        }
      } else {
        // out[l + L*t + rs*n] = sum_j V[l + L*(j + J*t)] * B[j,n]   (M = L rows kept, K = J)
        const int64_t M = L, K = J;
        const bool al = aligned_base && (M % VEC == 0);
        const int64_t n_mtiles64 = (M + 64 * VEC - 1) / (64 * VEC);
        const int n_mtiles = (int)n_mtiles64;
        int nsplit = 1;
        const int target = ncu_ * 8;
        if (T == 1 && n_mtiles < target)
          nsplit = (int)std::min<int64_t>((target + n_mtiles - 1) / n_mtiles,
                                          std::max(1, nblk / 32));
        // a batched scan of a small tensor (the second-level mode products of a Tucker sweep:
        // 400 x 400 x 20 -> 2 row tiles x 20 batches = 40 workgroups, 29 us for 25.6 MB): K-split
        // down to 4 column blocks per workgroup until there are two workgroups per CU
        else if (T > 1 && n_mtiles64 * T < ncu_ && (double)M * T * ncols * nblk / 4 * 8.0 < 64e6)
          nsplit = (int)std::min<int64_t>(((int64_t)ncu_ * 2 + n_mtiles64 * T - 1) / (n_mtiles64 * T),
                                          std::max(1, nblk / 4));
        if (nsplit < 1) nsplit = 1;
        const int per = (nblk + nsplit - 1) / nsplit;
        nsplit = (nblk + per - 1) / per;
        double *dst = o;
        int dst32 = out32;
        int64_t dst_ns = out_rstride, dst_ss = 0, dst_bs = out_tstride;
        if (nsplit > 1) {
          dst32 = 0;
          dst = (double *)ensure(ws_slab_, ws_slab_sz_, sizeof(double) * nsplit * ncols * M * T);
          dst_ns = M;
          dst_ss = (int64_t)ncols * M * T;
          dst_bs = (int64_t)ncols * M;
        }
        const int64_t nblocks = (int64_t)n_mtiles * nsplit * T;
        if (nblocks > 0x7fffffff) throw std::runtime_error("ppals: scan grid too large");
        dim3 grid((unsigned)nblocks);
        prof_begin(0, bytes);
        // padded layout: the kernel compacts the rows as it stores them — unless the partial sums
        // of several k-splits go through the slab, which k_slab_reduce compacts
        const int64_t k_ld = nsplit > 1 ? 0 : pad.ld, k_valid = nsplit > 1 ? 0 : pad.valid;
#define LAUNCH_SUFFIX(NTv, ALv)                                                                  \
  hipLaunchKernelGGL((k_scan_suffix<TV, NTv, ALv>), grid, dim3(256), 0, st_, V, M, K, M * K, P, \
                     n_mtiles, nsplit, per, nblk, dst, dst_ns, dst_ss, dst_bs, ncols, dst32, k_ld, \
                     k_valid)
#define LAUNCH_SUFFIX_FAST_O(NTv, OPTv)                                                             \
  hipLaunchKernelGGL((k_scan_suffix_fast<TV, NTv, OPTv>), grid, dim3(256), 0, st_, V, M, K, M * K, P, \
                     n_mtiles, nsplit, per, nblk, dst, dst_ns, dst_ss, dst_bs, ncols, dst32, k_ld,    \
                     k_valid)
#define LAUNCH_SUFFIX_FAST(NTv)   \
  if (nt_store) {                 \
    LAUNCH_SUFFIX_FAST_O(NTv, 5); \
  } else {                        \
    LAUNCH_SUFFIX_FAST_O(NTv, 1); \
  }
        // persistent launch: ncu*40 workgroups (measured best of 3..40 per CU), each walks over its tiles
        // (fewer tiles than that: every workgroup would take exactly ONE tile and the launch loses what the
        // persistent form is for — the P = 8 shard of cfg2, 3906 tiles: 6 workgroups per CU walking 2.5
        // tiles each scan it in 156.3 us instead of 159.6-161.0, profiles/r05m_persist_mult_shard.txt)
        const int64_t pgrid = nblocks < (int64_t)ncu_ * persist_mult_ ? (int64_t)ncu_ * std::min(persist_mult_, 6)
                                                                       : (int64_t)ncu_ * persist_mult_;
        dim3 grid_p((unsigned)std::min<int64_t>(nblocks, pgrid));
#define LAUNCH_SUFFIX_BUF_O(NTv, OPTv)                                                                \
  hipLaunchKernelGGL((k_scan_suffix_buf<TV, NTv, OPTv>), grid_p, dim3(256), 0, st_, V, M, K, M * K, P, \
                     n_mtiles, nsplit, per, nblk, dst, dst_ns, dst_ss, dst_bs, ncols, dst32, nblocks,  \
                     k_ld, k_valid)
        // A result too large for the 256 MB Infinity Cache is stored non-temporally: the stream of
        // ordinary stores costs the scan 0.2 ms per 320 MB (cfg2) next to its 6.4 GB of reads, the
        // non-temporal one 0.1 ms (tools/place6_bench, profiles/r03q_place6_nt.txt: 1.09-1.10 ms
        // against 1.19-1.20 ms per launch for the same source and result buffers), and what reads
        // the result next streams it from HBM either way. Not on every pair of buffers, though: in one
        // process (r03q_place6_nt_same_process.txt) 1.21 -> 1.06 ms for some, +1 % for others — so
        // the engine's online placement choice tries both kinds for the first-level intermediate
        // (scan_store_mode); this rule is for every other large result.
        constexpr double nt_min_bytes = 192.0 * 1048576.0;
        bool nt_store =
            nsplit == 1 && (scan_nt_mode_ < 0 ? (double)M * T * ncols * (dst32 ? 4.0 : 8.0) >= nt_min_bytes
                                              : scan_nt_mode_ == 1);
        // (Measured and rejected, tools/runs/r03_u.sh: second-level sums in fp32 for short
        // reductions — 16 NT registers fewer, the two-tile global-load kernel 174 -> 130 registers and
        // 2 -> 3 waves per SIMD, the one-tile buffer kernel 3 -> 4 waves: cfg4 36.0 / 36.7 -> 33.0 /
        // 33.6 sweeps/s, cfg2 no gain. More waves in flight make the mixed read/write stream worse.
        // Fewer do not help either: 64 KB of dynamic LDS per workgroup (2 instead of 3 workgroups per
        // CU) x 6..40 workgroups per CU in the persistent grid: profiles/r03v_occupancy_sweep.txt.)
#define LAUNCH_SUFFIX_BUF(NTv)           \
  if (nt_store) {                        \
    LAUNCH_SUFFIX_BUF_O(NTv, 5);         \
  } else {                               \
    LAUNCH_SUFFIX_BUF_O(NTv, 1);         \
  }
        // buffer-load variant: needs 32-bit byte offsets inside one 16-column block
        const bool buf_ok = (16.0 * (double)M * sizeof(TV) < 2.0e9) && (pack_bytes < 2000000000ull);
        // Which KERNEL runs must not depend on the store kind: the engine's placement exploration tries
        // both kinds and keeps what the stopwatch prefers, and the tail-mode kernel (which has no
        // non-temporal instantiation) sums k in another order than the others — results would then vary
        // from run to run with timing noise. So the tail mode is decided first, and where it applies
        // the stores are ordinary whatever was asked for.
        const bool takes_buf = al && M >= VEC && buf_ok && (NT == 1 || sizeof(TV) == 8) &&
                               !(sizeof(TV) == 8 && nsplit > 1);
        const bool takes_tail = !takes_buf && al && M >= VEC && T == 1 && nsplit == 1 && NT <= 2 && k_ld == 0 &&
                                scan_tail_split<TV>(NT, n_mtiles) >= 0;
        if (takes_tail) nt_store = false;
        // (measured: with NT >= 2 the fp32 build of the buffer variant drops to 2 waves/SIMD and
        // loses to the global-load kernel, so it is used for one n-tile / fp64 storage only)
        // (and: K-split scans of an fp64 tensor — few, long work items of half-size blocks — run
        // at 0.80 of peak on the global-load kernel against 0.70 here: tools/r02_f64.sh)
        // (A two-tile launch of 2-3 workgroups per CU — the first mode product of a Tucker sweep at
        // cfg5, 625 tiles, 73 us — is a round and a quarter at 2 waves per SIMD; handing it to the
        // register-lean persistent form with 3 waves, k_scan_suffix_lean<2,4,0,3>, was measured:
        // 40 HOOI sweeps 0.0446 / 0.0443 s against 0.0442 / 0.0441 s. Not kept.)
        if (takes_buf) {
          if (NT == 1) {
            LAUNCH_SUFFIX_BUF(1)
          } else if (NT == 2) {
            LAUNCH_SUFFIX_BUF(2)
          } else if (NT == 3) {
            LAUNCH_SUFFIX_BUF(3)
          } else {
            LAUNCH_SUFFIX_BUF(4)
          }
        } else if (takes_tail) {
          // a round and a bit of resident workgroups (cfg5's 625 tiles on 512 slots: 0.49 of the HBM
          // peak): the tiles of the last, partial round as four quarter-length work items each
          // (kernels_scan.hip.h, TAIL MODE)
          const int tail_from = scan_tail_split<TV>(NT, n_mtiles);
          const int64_t strips = ((M + 16 * VEC - 1) / (16 * VEC)) - (int64_t)tail_from * 4;
          dim3 grid_t((unsigned)(tail_from + strips));
          const size_t lds_t = sizeof(double) * 3 * (size_t)(VEC * NT * 4) * 64;
#define LAUNCH_SUFFIX_TAIL(NTv)                                                                            \
  hipLaunchKernelGGL((k_scan_suffix_fast<TV, NTv, 9>), grid_t, dim3(256), lds_t, st_, V, M, K, M * K, P,   \
                     n_mtiles, nsplit, per, nblk, dst, dst_ns, dst_ss, dst_bs, ncols, dst32, (int64_t)0,   \
                     (int64_t)0, tail_from)
          if (NT == 1) {
            LAUNCH_SUFFIX_TAIL(1);
          } else {
            LAUNCH_SUFFIX_TAIL(2);
          }
#undef LAUNCH_SUFFIX_TAIL
        } else if (al && M >= VEC) {
          if (NT == 1) {
            LAUNCH_SUFFIX_FAST(1)
          } else if (NT == 2) {
            LAUNCH_SUFFIX_FAST(2)
          } else if (NT == 3) {
            LAUNCH_SUFFIX_FAST(3)
          } else {
            LAUNCH_SUFFIX_FAST(4)
          }
        } else if (al) {
          if (NT == 1) LAUNCH_SUFFIX(1, true);
          else if (NT == 2) LAUNCH_SUFFIX(2, true);
          else if (NT == 3) LAUNCH_SUFFIX(3, true);
          else LAUNCH_SUFFIX(4, true);
        } else {
          if (NT == 1) LAUNCH_SUFFIX(1, false);
          else if (NT == 2) LAUNCH_SUFFIX(2, false);
          else if (NT == 3) LAUNCH_SUFFIX(3, false);
          else LAUNCH_SUFFIX(4, false);
        }
#undef LAUNCH_SUFFIX
#undef LAUNCH_SUFFIX_FAST
#undef LAUNCH_SUFFIX_FAST_O
#undef LAUNCH_SUFFIX_BUF
#undef LAUNCH_SUFFIX_BUF_O
        prof_end();
        HIP_CHECK(hipGetLastError());
        if (nsplit > 1) {
          hipLaunchKernelGGL(k_slab_reduce, dim3(grid_for(M * ncols, 256), (unsigned)T), dim3(256), 0,
                             st_, dst, nsplit, dst_ss, M, ncols, o, (int64_t)1, out_rstride, out32,
                             pad.ld, pad.valid, dst_bs, out_tstride);
          HIP_CHECK(hipGetLastError());
        }
      }
    }
  }
  // One pass of the wide kernel over columns [col0, col0 + ncols), 64 < ncols <= 128 (fp32 tensor,
  // suffix / batched form): pack, launch, combine the k-splits.
  void scan_wide(const float *V, int64_t L, int64_t J, int64_t T, const KrpArgs &a, int col0, int ncols,
                 double *out, int64_t out_tstride, int64_t out_rstride, int out32, RowPad pad) {
    const int NT = (ncols + 15) / 16;
    const int64_t nblk64 = (J + 15) / 16;
    if (nblk64 > 0x7fffffff) throw std::runtime_error("ppals: contraction extent too large");
    const int nblk = (int)nblk64;
    const size_t pack_bytes = (size_t)nblk * NT * 256 * sizeof(float);
    float *P = (float *)ensure(ws_pack_, ws_pack_sz_, pack_bytes);
    hipLaunchKernelGGL(k_krp_pack<float>, dim3(grid_for((int64_t)nblk * NT * 256, 256)), dim3(256), 0, st_,
                       P, nblk, NT, 0, a, J, col0, ncols);
    HIP_CHECK(hipGetLastError());
    double *o = out32 ? (double *)((float *)out + (int64_t)col0 * out_rstride)
                      : out + (int64_t)col0 * out_rstride;
    const int64_t M = L, K = J;
    const int64_t n_mtiles64 = (M + 63) / 64;
    if (n_mtiles64 * T > 0x7fffffff) throw std::runtime_error("ppals: scan grid too large");
    const int n_mtiles = (int)n_mtiles64;
    // k-split: the launch is MFMA-bound, so what counts is how evenly its workgroups fill the resident
    // slots (a partial last round idles the matrix cores): take the split whose number of rounds is
    // closest below a whole number, a little in favour of fewer splits (slab traffic, the combine)
    int &occ = wide_occ_[NT - 1];
    if (occ == 0) {
      int nb = 0;
      hipError_t e = hipErrorUnknown;
      switch (NT) {
        case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_scan_wide<5, 1>, 512, 0); break;
        case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_scan_wide<6, 1>, 512, 0); break;
        case 7: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_scan_wide<7, 1>, 512, 0); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_scan_wide<8, 1>, 512, 0); break;
      }
      if (e != hipSuccess) (void)hipGetLastError();
      occ = (e == hipSuccess && nb > 0) ? nb : 2;
    }
    const double slots = (double)ncu_ * occ;
    const int64_t tiles = n_mtiles64 * T;
    int nsplit = 1;
    if ((double)tiles < 8.0 * slots) {
      double best = -1.0;
      const int ns_max = (int)std::min<int64_t>(24, std::max(1, nblk / 16));
      for (int ns = 1; ns <= ns_max; ns++) {
        const double rounds = (double)tiles * ns / slots;
        const double score = rounds / std::ceil(rounds) - 0.004 * ns;
        if (score > best) {
          best = score;
          nsplit = ns;
        }
      }
    }
    const int per = (nblk + nsplit - 1) / nsplit;
    nsplit = (nblk + per - 1) / per;
    double *dst = o;
    int dst32 = out32;
    int64_t dst_ns = out_rstride, dst_ss = 0, dst_bs = out_tstride;
    if (nsplit > 1) {
      dst32 = 0;
      dst = (double *)ensure(ws_slab_, ws_slab_sz_, sizeof(double) * nsplit * ncols * M * T);
      dst_ns = M;
      dst_ss = (int64_t)ncols * M * T;
      dst_bs = (int64_t)ncols * M;
    }
    const int64_t nblocks = (int64_t)n_mtiles * nsplit * T;
    if (nblocks > 0x7fffffff) throw std::runtime_error("ppals: scan grid too large");
    const int64_t k_ld = nsplit > 1 ? 0 : pad.ld, k_valid = nsplit > 1 ? 0 : pad.valid;
    const double bytes = (double)L * (double)J * (double)T * 4.0 + (double)L * (double)T * ncols * (out32 ? 4.0 : 8.0);
    prof_begin(0, bytes);
#define LAUNCH_WIDE(NTv)                                                                                  \
  hipLaunchKernelGGL((k_scan_wide<NTv, 1>), dim3((unsigned)nblocks), dim3(512), 0, st_, V, M, K, M * K, P, \
                     n_mtiles, nsplit, per, nblk, dst, dst_ns, dst_ss, dst_bs, ncols, dst32, k_ld, k_valid)
    switch (NT) {
      case 5: LAUNCH_WIDE(5); break;
      case 6: LAUNCH_WIDE(6); break;
      case 7: LAUNCH_WIDE(7); break;
      default: LAUNCH_WIDE(8); break;
    }
#undef LAUNCH_WIDE
    prof_end();
    HIP_CHECK(hipGetLastError());
    if (nsplit > 1) {
      hipLaunchKernelGGL(k_slab_reduce, dim3(grid_for(M * ncols, 256), (unsigned)T), dim3(256), 0, st_, dst,
                         nsplit, dst_ss, M, ncols, o, (int64_t)1, out_rstride, out32, pad.ld, pad.valid,
                         dst_bs, out_tstride);
      HIP_CHECK(hipGetLastError());
    }
  }
  int gj_scalar_ = 0;  // PPALS_GJ_SCALAR=1: the scalar in-LDS sweeps for 64 < R <= 128 (A/B, tests)
  bool wide_enabled_ = true;  // PPALS_SCAN_WIDE=0: chunks of 64 columns (A/B, tests)
  int wide_occ_[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // resident workgroups per CU of k_scan_wide<1..8>
  // Where the tail mode of k_scan_suffix_fast starts (first tile of the last, partial round of resident
  // workgroups), or -1: the launch fits one round, has many rounds (the tail is a small share), or its
  // last round is nearly full anyway. Resident workgroups per CU from the runtime's occupancy query
  // of the tail-mode instantiation (its LDS included).
  template <typename TV>
  int scan_tail_split(int NT, int n_mtiles) {
    constexpr int VEC = ScanTraits<TV>::VEC;
    int &per_cu = scan_tail_occ_[(sizeof(TV) == 8 ? 2 : 0) + (NT - 1)];
    if (!scan_tail_on_) return -1;
    if (per_cu == 0) {
      const size_t lds_t = sizeof(double) * 3 * (size_t)(VEC * NT * 4) * 64;
      int nb = 0;
      hipError_t e = NT == 1
          ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_scan_suffix_fast<TV, 1, 9>, 256, lds_t)
          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_scan_suffix_fast<TV, 2, 9>, 256, lds_t);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        nb = 0;
      }
      per_cu = nb > 0 ? nb : -1;
    }
    if (per_cu < 0) return -1;
    const int slots = ncu_ * per_cu;
    if (n_mtiles <= slots || n_mtiles > 4 * slots) return -1;
    const int rem = n_mtiles % slots;
    if (rem == 0 || rem * 4 > slots * 3) return -1;   // (a last round that is 3/4 full is left alone)
    return n_mtiles - rem;
  }
  int scan_tail_occ_[4] = {0, 0, 0, 0};
  bool scan_tail_on_ = true;  // PPALS_SCAN_TAIL=0: no tail mode (A/B)
  using Ops::scan_contract;
  void scan_contract(const void *V, int dt, int64_t L, int64_t J, int64_t T, const FactorRef *f,
                     int nf, int R, void *out, int out_dt, int64_t out_tstride,
                     int64_t out_rstride, RowPad pad) override {
    RoctxRange roctx_("K1/K2/K8/K11 tensor scan");
    const int out32 = out_dt == F32 ? 1 : 0;
    if (dt == F32)
      scan_t<float>((const float *)V, L, J, T, f, nf, R, (double *)out, out_tstride, out_rstride,
                    out32, pad);
    else
      scan_t<double>((const double *)V, L, J, T, f, nf, R, (double *)out, out_tstride,
                     out_rstride, out32, pad);
  }
  // Tucker mode product keeping the mode in place. Small inputs (the second-level products of a
  // chain: tens of MB with at least a tile of contiguous rows) run as ONE batched thin GEMM on the
  // fp64 matrix cores instead of the streaming scan's pack + scan + slab-combine launches; everything
  // else is the scan.
  void ttm_keep(const void *X, int dt, int64_t L, int64_t J, int64_t T, const double *W, int64_t ldw,
                int Kc, double *out) override {
    const double bytes = (double)L * J * T * dtype_size(dt);
    if (L >= 16 && Kc <= 64 && bytes <= 64e6 && (double)L * J < 2e9 && T <= 65535 && J >= 16 &&
        (double)ldw * Kc < 4e9) {
      RoctxRange roctx_("K11 mode product (batched thin GEMM)");
      const int nt = Kc <= 16 ? 1 : (Kc <= 32 ? 2 : 4);
      dim3 grid((unsigned)((L + 15) / 16), (unsigned)((Kc + 16 * nt - 1) / (16 * nt)), (unsigned)T);
      prof_begin(1, bytes);
#define PPALS_MPS(TA_, NT_)                                                                               \
  hipLaunchKernelGGL((k_mode_product_small<TA_, NT_>), grid, dim3(512), 0, st_, (const TA_ *)X, (int)L, (int)J, W, \
                     ldw, Kc, out)
      if (dt == F32) {
        if (nt == 1) PPALS_MPS(float, 1); else if (nt == 2) PPALS_MPS(float, 2); else PPALS_MPS(float, 4);
      } else {
        if (nt == 1) PPALS_MPS(double, 1); else if (nt == 2) PPALS_MPS(double, 2); else PPALS_MPS(double, 4);
      }
#undef PPALS_MPS
      prof_end();
      HIP_CHECK(hipGetLastError());
      return;
    }
    Ops::ttm_keep(X, dt, L, J, T, W, ldw, Kc, out);
  }
  bool ttm_lead_front(const void *X, int dt, int64_t J, int64_t S, int64_t T, const double *W, int64_t ldw,
                      int Kc, double *out) override {
    const double bytes = (double)J * S * T * dtype_size(dt);
    if (!(S >= 16 && J >= 16 && Kc <= 64 && bytes <= 64e6 && (double)J * S < 2e9 && T <= 65535 &&
          (double)ldw * Kc < 4e9))
      return false;
    RoctxRange roctx_("K11 leading-mode product (batched thin GEMM)");
    const int nt = Kc <= 16 ? 1 : (Kc <= 32 ? 2 : 4);
    dim3 grid((unsigned)((S + 15) / 16), (unsigned)((Kc + 16 * nt - 1) / (16 * nt)), (unsigned)T);
    prof_begin(1, bytes);
#define PPALS_MPL(TA_, NT_)                                                                               \
  hipLaunchKernelGGL((k_mode_product_lead<TA_, NT_>), grid, dim3(512), 0, st_, (const TA_ *)X, (int)J, (int)S, W,  \
                     ldw, Kc, out)
    if (dt == F32) {
      if (nt == 1) PPALS_MPL(float, 1); else if (nt == 2) PPALS_MPL(float, 2); else PPALS_MPL(float, 4);
    } else {
      if (nt == 1) PPALS_MPL(double, 1); else if (nt == 2) PPALS_MPL(double, 2); else PPALS_MPL(double, 4);
    }
#undef PPALS_MPL
    prof_end();
    HIP_CHECK(hipGetLastError());
    return true;
  }
  // padded resident layout, see ops.h: a pitched copy (rows == 1) or a transposition whose
  // destination rows are padded
  void pad_layout(const void *src, int dt, int64_t rows, int64_t cols, int64_t blk, int64_t ld,
                  void *dst) override {
    if (cols % blk || ld < blk) throw std::runtime_error("ppals: pad_layout block inconsistent");
    const size_t esz = dtype_size(dt);
    HIP_CHECK(hipMemsetAsync(dst, 0, (size_t)ld * (cols / blk) * rows * esz, st_));
    if (rows == 1) {
      HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)ld * esz, src, (size_t)blk * esz, (size_t)blk * esz,
                                 (size_t)(cols / blk), hipMemcpyDeviceToDevice, st_));
      return;
    }
    const int64_t nb = ((rows + 63) / 64) * ((cols + 63) / 64);
    if (nb > 0x7fffffff) throw std::runtime_error("ppals: transpose grid too large");
    if (dt == F32)
      hipLaunchKernelGGL(k_transpose_pad<float>, dim3((unsigned)nb), dim3(256), 0, st_,
                         (const float *)src, rows, cols, blk, ld, (float *)dst);
    else
      hipLaunchKernelGGL(k_transpose_pad<double>, dim3((unsigned)nb), dim3(256), 0, st_,
                         (const double *)src, rows, cols, blk, ld, (double *)dst);
    HIP_CHECK(hipGetLastError());
  }

  // ------------------------------------------------------------------ mttv
  template <typename TX>
  void mttv_t(const TX *X, int64_t L, int64_t J, int64_t T, int R, const double *B, int64_t ldb,
              double *out, int64_t rs, int accumulate, const double *scale) {
    constexpr int VL = 16 / sizeof(TX);
    // one more workgroup prepares S, S^-1 of the mode update this contraction feeds (armed by the
    // engine: arm_gram_system) — they depend on the other modes' Grams only
    SysArgs sys;
    size_t lds = 0;
    int extra = 0;
    if (sys_armed_ && sys_R_ == R) {
      sys = sys_;
      extra = 1;
      lds = sizeof(double) * (2 * (size_t)R * (R + 1) + 64) + sizeof(int) * 64;
    }
    prof_begin(1, (double)L * J * T * R * sizeof(TX));
    if (L == 1) {
      int64_t nw = ((T + 3) / 4) * R;
      int g = grid_for(nw * 64, 256, 16384);
      hipLaunchKernelGGL(k_mttv_1<TX>, dim3(g + extra), dim3(256), lds, st_, X, J, T, R, B, ldb, out, rs,
                         accumulate, scale, sys);
    } else if (L % VL == 0 && L >= 16 * VL && ((uintptr_t)X & 15) == 0 &&
               ((L + 64 * VL - 1) / (64 * VL)) * T * R >= 1024) {
      // streaming regime: enough (l-tile, t, r) waves to fill the chip on their own
      int64_t nb = ((L + 64 * VL - 1) / (64 * VL)) * T * R;
      int g = (int)std::min<int64_t>(nb, 1 << 20);
      hipLaunchKernelGGL(k_mttv_vec<TX>, dim3(g + extra), dim3(64), lds, st_, X, L, J, T, R, B, ldb, out,
                         rs, accumulate, scale, sys);
    } else if (L < 64 && L * J <= 2048 && T * R >= 2048) {
      // a short plane per (t, r) and many of them: one wave each, no workgroup barrier (k_mttv_s)
      const int64_t nwaves = T * R;
      int g = (int)std::min<int64_t>((nwaves + 3) / 4, 16384);
      hipLaunchKernelGGL(k_mttv_s<TX>, dim3(g + extra), dim3(256), lds, st_, X, (int)L, J, T, R, B, ldb, out,
                         rs, accumulate, scale, sys);
    } else {
      int64_t nb = ((L + 63) / 64) * T * R;
      // a long reduction over few row tiles (128 x 7200 x R: 20 workgroups): the j range is cut into
      // pieces, one workgroup each, and a second launch adds the partial sums in a fixed order
      int jsplit = 1;
      if (nb < 512 && J >= 1024) jsplit = (int)std::min<int64_t>((1024 + nb - 1) / nb, J / 64);
      if (jsplit > 1) {
        const int64_t jchunk = (J + jsplit - 1) / jsplit;
        jsplit = (int)((J + jchunk - 1) / jchunk);
        double *partial = (double *)ensure(ws_mttv_, ws_mttv_sz_, sizeof(double) * (size_t)jsplit * L * T * R);
        const int64_t nbs = nb * jsplit;
        int g = (int)std::min<int64_t>(nbs, 32768);
        hipLaunchKernelGGL((k_mttv_l<TX, 4>), dim3(g + extra), dim3(256), lds, st_, X, L, J, T, R, B, ldb,
                           out, rs, accumulate, scale, sys, jsplit, jchunk, partial);
        hipLaunchKernelGGL(k_mttv_combine, dim3(grid_for(L * T * R, 256, 1024)), dim3(256), 0, st_, partial,
                           jsplit, L * T, R, out, rs, accumulate, scale);
      } else {
        int g = (int)std::min<int64_t>(nb, 32768);
        if (nb * 4 < 1024)  // few blocks: split the j loop 16 ways instead of 4
          hipLaunchKernelGGL((k_mttv_l<TX, 16>), dim3(g + extra), dim3(1024), lds, st_, X, L, J, T, R, B,
                             ldb, out, rs, accumulate, scale, sys);
        else
          hipLaunchKernelGGL((k_mttv_l<TX, 4>), dim3(g + extra), dim3(256), lds, st_, X, L, J, T, R, B, ldb,
                             out, rs, accumulate, scale, sys);
      }
    }
    if (extra) {
      sys_ready_ = true;
      sys_armed_ = false;
    }
    prof_end();
    HIP_CHECK(hipGetLastError());
  }
  void mttv(const void *X, int xdt, int64_t L, int64_t J, int64_t T, const FactorRef *f, int nf,
            int R, double *out, int64_t out_rstride, int accumulate,
            const double *out_scale) override {
    RoctxRange roctx_("K3/K9 mttv (cached intermediate)");
    const double *B;
    int64_t ldb;
    if (nf == 1) {
      if (f[0].rows != J) throw std::runtime_error("ppals: mttv factor extent mismatch");
      B = f[0].ptr;
      ldb = f[0].ld;
    } else {
      double *kb = (double *)ensure(ws_krp_, ws_krp_sz_, sizeof(double) * J * R);
      krp(kb, f, nf, 0, R);
      B = kb;
      ldb = J;
    }
    if (xdt == F32)
      mttv_t<float>((const float *)X, L, J, T, R, B, ldb, out, out_rstride, accumulate, out_scale);
    else
      mttv_t<double>((const double *)X, L, J, T, R, B, ldb, out, out_rstride, accumulate,
                     out_scale);
  }
  void scale_update(double *dst, const double *scales, unsigned mask, int set_one) override {
    hipLaunchKernelGGL(k_scale_update, dim3(1), dim3(64), 0, st_, dst, scales, mask, set_one);
    HIP_CHECK(hipGetLastError());
  }
  void scale_update_many(double *dst, const double *scales, const unsigned *masks, unsigned active,
                         unsigned fresh) override {
    ScaleMasks sm;
    for (int k = 0; k < 32; k++) sm.m[k] = masks[k];
    hipLaunchKernelGGL(k_scale_update_many, dim3(1), dim3(64), 0, st_, dst, scales, sm, active,
                       fresh);
    HIP_CHECK(hipGetLastError());
  }
  const double *normalize_scales() override { return small(MAX_ORDER); }

  // ------------------------------------------------------------------ R x R side
  void gram(const double *W, int64_t rows, int64_t ld, int R, double *G) override {
    RoctxRange roctx_("K4 gram");
    int npairs = R * (R + 1) / 2;
    int g = std::min(64, (npairs + 15) / 16);
    hipLaunchKernelGGL(k_gram, dim3(g), dim3(1024), 0, st_, W, rows, ld, R, G);
    HIP_CHECK(hipGetLastError());
  }
  void gram_system(const double *Gall, int N, int mode, int R, double lambda, double *S,
                   double *Sinv) override {
    if (sys_ready_ && sys_.Gall == Gall && sys_.mode == mode && sys_.S == S && sys_.Sinv == Sinv &&
        sys_.lambda == lambda && sys_R_ == R) {  // prepared on the side of the last contraction
      sys_ready_ = false;
      return;
    }
    sys_ready_ = false;
    if (R > 64) {  // S, S^-1 out of global memory (kernels_small.hip.h, "rank above 64")
      double *work = (double *)ensure(ws_big_, ws_big_sz_, sizeof(double) * (size_t)R * R + 64);
      int *status = (int *)(work + (size_t)R * R);
      const size_t lds_gj = sizeof(double) * ((size_t)R * R + 2 * (size_t)R);
      const size_t RP = ((size_t)R + 15) & ~(size_t)15;
      const size_t LDm = RP + ((34 - (RP & 31)) & 31);
      const size_t lds_mf = sizeof(double) * (LDm * RP + 8 * RP);
      if (lds_mf <= 150 * 1024 && !gj_scalar_)   // block sweeps, trailing update on the fp64 matrix cores
        hipLaunchKernelGGL(k_gram_system_mfma, dim3(1), dim3(1024), lds_mf, st_, Gall, N, mode, R,
                           lambda, S, Sinv, status);
      else if (lds_gj <= 150 * 1024)
        hipLaunchKernelGGL(k_gram_system_lds, dim3(1), dim3(1024), lds_gj, st_, Gall, N, mode, R,
                           lambda, S, Sinv, status);
      else
        hipLaunchKernelGGL(k_gram_system_big, dim3(1), dim3(1024), 0, st_, Gall, N, mode, R, lambda,
                           S, Sinv, work, status);
      HIP_CHECK(hipGetLastError());
      // a non-positive pivot (S not positive definite: a rank above a mode extent, collinear
      // columns) leaves NaNs in Sinv: take the reference's route instead, the untruncated inverse
      // through a full eigen-decomposition.
      // Up to 128 columns that route is two CONDITIONAL launches on the stream — the one-workgroup
      // one-sided Jacobi of S and Z diag(1/w) Z^T, both returning at once unless the elimination set
      // its status word — so a mode update above 64 columns holds no host synchronisation any more
      // (round 6: four read-backs per sweep were 40 us of idle stream each, and milliseconds each
      // whenever the host's cores were busy). PPALS_FORCE_EIGINV=2 runs the two launches ungated (tests).
      if (R <= kJacobiBigMax && force_eiginv_ != 1) {
        double *fb = (double *)ensure(ws_gsfb_, ws_gsfb_sz_, sizeof(double) * (2 * (size_t)R * R + R));
        double *Vt = fb, *Yz = Vt + (size_t)R * R, *wv = Yz + (size_t)R * R;
        const int *gate = force_eiginv_ == 2 ? nullptr : status;
        const size_t lds_j = sizeof(double) * ((size_t)R * (R + 1) + 256) + sizeof(int) * 256;
        hipLaunchKernelGGL(k_jacobi_onesided, dim3(1), dim3(1024), lds_j, st_, (const double *)S, R, Vt, Yz, wv,
                           (double *)nullptr, gate);
        hipLaunchKernelGGL(k_eig_inverse, dim3(grid_for((int64_t)R * R, 256)), dim3(256), 0, st_,
                           (const double *)Yz, (const double *)wv, R, Sinv, gate);
        HIP_CHECK(hipGetLastError());
        return;
      }
      int bad = 0;
      HIP_CHECK(hipMemcpyAsync(&bad, status, sizeof(int), hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipStreamSynchronize(st_));
      if (bad || force_eiginv_) {
        RocSolver &rs = rocsolver();
        double *D = (double *)ensure(ws_krp_, ws_krp_sz_, sizeof(double) * (2 * (size_t)R + 2));
        double *E = D + R;
        int *info = (int *)(E + R);
        HIP_CHECK(hipMemcpyAsync(work, S, sizeof(double) * (size_t)R * R, hipMemcpyDeviceToDevice, st_));
        if (rs.dsyevd(rs.handle, 211 /*evect_original*/, 121 /*fill_upper*/, R, work, R, D, E, info) != 0)
          throw std::runtime_error("ppals: rocsolver_dsyevd failed (R x R normal equations)");
        hipLaunchKernelGGL(k_eig_inverse, dim3(grid_for((int64_t)R * R, 256)), dim3(256), 0, st_, work,
                           D, R, Sinv);
        HIP_CHECK(hipGetLastError());
      }
      return;
    }
    size_t lds = sizeof(double) * (2 * (size_t)R * (R + 1) + 64) + sizeof(int) * 64;
    hipLaunchKernelGGL(k_gram_system, dim3(1), dim3(64), lds, st_, Gall, N, mode, R, lambda, S,
                       Sinv, force_jacobi_);
    HIP_CHECK(hipGetLastError());
  }
  void cp_mode_update_blocked(double *Gall, int N, int mode, int R, double lambda, const double *Mblk,
                              int64_t blk, int P, double *scratch, double *W, int64_t ldw, double *grad,
                              int64_t ldg, int64_t rows, double *gradsq, const double *Winit, int64_t ldi,
                              double *dW, int64_t ldd, double ratio, double *S, double *Sinv) override {
    // the staged fused launch reads the gathered blocks as they lie (one launch less per update of the
    // partitioned mode); anything else re-assembles them first
    const size_t lds = sizeof(double) * (32 + 2 * (size_t)R * R + 2 * (size_t)R * (R + 1) + 64) + sizeof(int) * 64;
    const size_t stage = 2 * sizeof(double) * (size_t)rows * R;
    if (R > 64 || force_jacobi_ || norm_armed_ || lds + stage > 150 * 1024 || blk * P != rows ||
        blk > 0x7fffffff) {
      Ops::cp_mode_update_blocked(Gall, N, mode, R, lambda, Mblk, blk, P, scratch, W, ldw, grad, ldg, rows,
                                  gradsq, Winit, ldi, dW, ldd, ratio, S, Sinv);
      return;
    }
    update_mblk_ = (int)blk;
    cp_mode_update(Gall, N, mode, R, lambda, Mblk, rows, W, ldw, grad, ldg, rows, gradsq, Winit, ldi, dW,
                   ldd, ratio, S, Sinv, nullptr);
    update_mblk_ = 0;
  }
  int update_mblk_ = 0;
  // rows x R from which a mode update that does not fit the staged launch is row-parallel: the unstaged
  // one-workgroup launch costs 4 us + 4.5 us per 1000 entries (53 us at 1344 x 10, 333 us at 7200 x 10),
  // the row-parallel route five short launches (~30 us)
  static constexpr int64_t kUpdateRowParallelFrom = 6144;
  void cp_mode_update(double *Gall, int N, int mode, int R, double lambda, const double *M,
                      int64_t ldm, double *W, int64_t ldw, double *grad, int64_t ldg, int64_t rows,
                      double *gradsq, const double *Winit, int64_t ldi, double *dW, int64_t ldd,
                      double ratio, double *S, double *Sinv, double *dwsq) override {
    RoctxRange roctx_("K4-K6 mode update");
    if (R > 64) {  // unfused route: S / S^-1, row-parallel update, Gram refresh
      Ops::cp_mode_update(Gall, N, mode, R, lambda, M, ldm, W, ldw, grad, ldg, rows, gradsq, Winit,
                          ldi, dW, ldd, ratio, S, Sinv, dwsq);
      return;
    }
    if (force_jacobi_) {  // A/B path: the three separate kernels with the Jacobi inverse
      Ops::cp_mode_update(Gall, N, mode, R, lambda, M, ldm, W, ldw, grad, ldg, rows, gradsq, Winit,
                          ldi, dW, ldd, ratio, S, Sinv, dwsq);
      return;
    }
    if (!Winit) dwsq = nullptr;
    size_t lds = sizeof(double) * (32 + 2 * (size_t)R * R + 2 * (size_t)R * (R + 1) + 64) +
                 sizeof(int) * 64;
    const size_t stage = 2 * sizeof(double) * (size_t)rows * R;
    // A long mode (7200 rows at R = 10: coil-100's last mode) does not fit the staged launch, and ONE
    // workgroup walking rows x R entries out of global memory takes 333 us there: the row-parallel
    // unfused route (S / S^-1, rows over many workgroups, Gram) takes a few short launches instead
    if (lds + stage > 150 * 1024 && (int64_t)rows * R > kUpdateRowParallelFrom && !norm_armed_ && !update_mblk_) {
      Ops::cp_mode_update(Gall, N, mode, R, lambda, M, ldm, W, ldw, grad, ldg, rows, gradsq, Winit, ldi, dW,
                          ldd, ratio, S, Sinv, dwsq);
      return;
    }
    // S / S^-1 prepared by the contraction launched before (arm_gram_system + pp_correct)?
    const int presolved = (sys_ready_ && sys_.Gall == Gall && sys_.mode == mode && sys_.S == S &&
                           sys_.Sinv == Sinv && S && sys_.lambda == lambda) ? 1 : 0;
    sys_ready_ = false;
    sys_armed_ = false;
    NormArgs nrm;
    if (norm_armed_) {
      if (!(norm_.on && norm_mode_ == mode && norm_G_ == Gall && ldw == rows))
        throw std::logic_error("ppals: armed Normalize does not match the mode update that followed");
      nrm = norm_;
      nrm.scales = small(MAX_ORDER);
      norm_armed_ = false;
    }
    if (lds + stage <= 150 * 1024) {
      hipLaunchKernelGGL((k_cp_mode_update<true, true>), dim3(1), dim3(1024), lds + stage, st_, Gall, N, mode,
                         R, lambda, M, ldm, W, ldw, grad, ldg, rows, gradsq, Winit, ldi, dW, ldd, ratio, S,
                         Sinv, dwsq, presolved, nrm, update_mblk_);
    } else {
      if (update_mblk_) throw std::logic_error("ppals: blocked mode update needs the staged launch");
      hipLaunchKernelGGL((k_cp_mode_update<false, false>), dim3(1), dim3(1024), lds, st_, Gall, N, mode,
                         R, lambda, M, ldm, W, ldw, grad, ldg, rows, gradsq, Winit, ldi, dW, ldd, ratio,
                         S, Sinv, dwsq, presolved);
    }
    HIP_CHECK(hipGetLastError());
  }
  bool arm_normalize(double *const *W, const int64_t *rows, int N, int R, double *Gall, int mode,
                     double *wsq, double *ms_dst, const unsigned *masks, unsigned active,
                     unsigned fresh) override {
    norm_armed_ = false;
    if (R > 64 || force_jacobi_ || N > MAX_ORDER) return false;
    int64_t tot = 0;
    for (int i = 0; i < N; i++) tot += rows[i] * R;
    const size_t lds = sizeof(double) * (32 + 2 * (size_t)R * R + 2 * (size_t)R * (R + 1) + 64) +
                       sizeof(int) * 64;
    if (tot > 65536 || lds + 2 * sizeof(double) * (size_t)rows[mode] * R > 150 * 1024) return false;
    norm_ = NormArgs();
    norm_.on = 1;
    for (int i = 0; i < N; i++) {
      norm_.w.p[i] = W[i];
      norm_.w.n[i] = rows[i] * R;
    }
    norm_.wsq = wsq;
    if (active && ms_dst && masks) {
      norm_.ms_dst = ms_dst;
      for (int k = 0; k < 32; k++) norm_.masks.m[k] = masks[k];
      norm_.active = active;
      norm_.fresh = fresh;
    }
    norm_mode_ = mode;
    norm_G_ = Gall;
    (void)small(MAX_ORDER);  // (allocated now: no workspace growth inside the armed launch)
    norm_armed_ = true;
    return true;
  }
  void arm_gram_system(const double *Gall, int N, int mode, int R, double lambda, double *S,
                       double *Sinv) override {
    sys_ready_ = false;
    sys_armed_ = false;
    if (force_jacobi_ || R > 32 || !S || !Sinv) return;
    sys_.Gall = Gall;
    sys_.N = N;
    sys_.mode = mode;
    sys_.lambda = lambda;
    sys_.S = S;
    sys_.Sinv = Sinv;
    sys_.force_jacobi = 0;
    sys_R_ = R;
    sys_armed_ = true;
  }
  void pp_correct(const double *M0, int64_t rows, int R, const PPTerm *terms, int nterms,
                  double *M) override {
    RoctxRange roctx_("K9 pp_correct");
    if (nterms > MAX_ORDER) throw std::runtime_error("ppals: too many PP correction terms");
    PPTerms tm;
    tm.n = nterms;
    for (int t = 0; t < nterms; t++) {
      tm.T[t] = terms[t].T;
      tm.dW[t] = terms[t].dW;
      tm.ny[t] = terms[t].ny;
      tm.lddw[t] = terms[t].lddw;
      tm.keep_first[t] = terms[t].keep_first;
    }
    dim3 grid((unsigned)((rows + 15) / 16), (unsigned)R);
    SysArgs sys;
    size_t lds = 0;
    if (sys_armed_ && sys_R_ == R) {  // one more column of blocks: its first block prepares S, S^-1
      sys = sys_;
      grid.x += 1;
      lds = sizeof(double) * (2 * (size_t)R * (R + 1) + 64) + sizeof(int) * 64;
    }
    prof_begin(1, 0.0);
    hipLaunchKernelGGL(k_pp_correct, grid, dim3(256), lds, st_, M0, rows, R, tm, M, sys);
    prof_end();
    HIP_CHECK(hipGetLastError());
    if (sys.Gall) sys_ready_ = true;
    sys_armed_ = false;
  }
  void cp_update(const double *M, int64_t ldm, const double *Wold, int64_t ldw, double *Wnew,
                 int64_t ldn, double *grad, int64_t ldg, int64_t rows, int R, const double *S,
                 const double *Sinv, double *gradsq, const double *Winit, int64_t ldi, double *dW,
                 int64_t ldd, double ratio) override {
    if (rows <= 0) {  // a rank that owns no rows of this mode (row-block plan): nothing to launch
      HIP_CHECK(hipMemsetAsync(gradsq, 0, sizeof(double), st_));
      return;
    }
    if (R > 64 && ldg == rows && ldn == rows && (!Winit || ratio == 1.0) && rows * (int64_t)R < (1 << 30)) {
      // both products on the matrix cores (S and S^-1 are symmetric: they ARE their own
      // transposed operand): grad = W_old S - M, then W = M S^-1 (may alias W_old: stream order)
      gemm_nt(Wold, ldw, S, R, M, ldm, grad, ldg, (int)rows, R, R, 1.0, -1.0);
      sumsq(grad, rows * R, gradsq);
      gemm_nt(M, ldm, Sinv, R, nullptr, 0, Wnew, ldn, (int)rows, R, R, 1.0, 0.0);
      if (Winit) {  // SVD_solve_mod tail with ratio 1: dW = W - W_init
        if (ldi != rows || ldd != rows) throw std::runtime_error("ppals: cp_update expects packed factors");
        double *A[1] = {Wnew}, *B[1] = {const_cast<double *>(Winit)}, *D[1] = {dW};
        int64_t n[1] = {rows * R};
        diff_norms(A, B, n, 1, 1, D, 0, (double *)ensure(ws_big2_, ws_big2_sz_, 4 * sizeof(double)));
      }
      HIP_CHECK(hipGetLastError());
      return;
    }
    if (R > 64 || (int64_t)rows * R > kUpdateRowParallelFrom) {
      // several blocks update the rows: W_old is read from a scratch copy so that Wnew may alias it
      const int nb = (int)((rows + 63) / 64);
      double *wcopy = (double *)ensure(ws_big2_, ws_big2_sz_,
                                       sizeof(double) * ((size_t)rows * R + (size_t)nb));
      double *part = wcopy + (size_t)rows * R;
      HIP_CHECK(hipMemcpy2DAsync(wcopy, sizeof(double) * rows, Wold, sizeof(double) * ldw,
                                 sizeof(double) * rows, R, hipMemcpyDeviceToDevice, st_));
      hipLaunchKernelGGL(k_cp_update_big, dim3(nb), dim3(256), 0, st_, M, ldm, wcopy, rows, Wnew, ldn,
                         grad, ldg, rows, R, S, Sinv, part, Winit, ldi, dW, ldd, ratio);
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, nb, gradsq);
      HIP_CHECK(hipGetLastError());
      return;
    }
    size_t lds = sizeof(double) * (32 + 2 * (size_t)R * R);
    hipLaunchKernelGGL(k_cp_update, dim3(1), dim3(1024), lds, st_, M, ldm, Wold, ldw, Wnew, ldn,
                       grad, ldg, rows, R, S, Sinv, gradsq, Winit, ldi, dW, ldd, ratio);
    HIP_CHECK(hipGetLastError());
  }
  void normalize(double *const *W, const int64_t *rows, int N, int R, double *Gall) override {
    double *scales = small(MAX_ORDER);
    hipLaunchKernelGGL(k_norm_scales, dim3(1), dim3(1024), 0, st_, Gall, N, R, scales);
    HIP_CHECK(hipGetLastError());
    PtrsN w;
    int64_t mx = 1;
    for (int i = 0; i < N; i++) {
      w.p[i] = W[i];
      w.n[i] = rows[i] * R;
      mx = std::max(mx, w.n[i]);
    }
    hipLaunchKernelGGL(k_scale_factors, dim3(grid_for(mx, 256, 64), N), dim3(256), 0, st_, w, N,
                       scales);
    HIP_CHECK(hipGetLastError());
  }
  void normalize_ms(double *const *W, const int64_t *rows, int N, int R, double *Gall,
                    double *ms_dst, const unsigned *masks, unsigned active, unsigned fresh,
                    double *wsq) override {
    RoctxRange roctx_("K7 normalize");
    int64_t tot = 0;
    for (int i = 0; i < N; i++) tot += rows[i] * R;
    if (tot > 65536) {  // big factors: the grid-wide scaling kernel pays off
      Ops::normalize_ms(W, rows, N, R, Gall, ms_dst, masks, active, fresh, wsq);
      return;
    }
    PtrsN w;
    for (int i = 0; i < N; i++) {
      w.p[i] = W[i];
      w.n[i] = rows[i] * R;
    }
    ScaleMasks sm;
    for (int k = 0; k < 32; k++) sm.m[k] = masks ? masks[k] : 0u;
    hipLaunchKernelGGL(k_normalize_fused, dim3(1), dim3(1024), 0, st_, Gall, N, R, w,
                       small(MAX_ORDER), active ? ms_dst : nullptr, sm, active, fresh, wsq);
    HIP_CHECK(hipGetLastError());
  }
  void diff_norms(double *const *A, double *const *B, const int64_t *n, int N, int store_diff,
                  double *const *D, int update_prev, double *out) override {
    PtrsN a;
    for (int i = 0; i < N; i++) {
      a.p[i] = A[i];
      a.q[i] = B ? B[i] : nullptr;
      a.d[i] = D ? D[i] : nullptr;
      a.n[i] = n[i];
    }
    hipLaunchKernelGGL(k_diff_norms, dim3(N), dim3(1024), 0, st_, a, store_diff, update_prev, out);
    HIP_CHECK(hipGetLastError());
  }
  void pack_blocks(const double *nat, int64_t rows, int64_t ld, int R, int64_t blk, int P,
                   double *blocked) override {
    hipLaunchKernelGGL(k_pack_blocks, dim3(grid_for(blk * R * P, 256)), dim3(256), 0, st_, nat,
                       rows, ld, R, blk, P, blocked);
    HIP_CHECK(hipGetLastError());
  }
  void unpack_blocks(const double *blocked, int64_t rows, int64_t ld, int R, int64_t blk, int P,
                     double *nat) override {
    hipLaunchKernelGGL(k_unpack_blocks, dim3(grid_for(blk * R * P, 256)), dim3(256), 0, st_,
                       blocked, rows, ld, R, blk, P, nat);
    HIP_CHECK(hipGetLastError());
  }
  // K12/K13: Gram of the mode unfolding (fp64 accumulation), reduction split over grid.z slabs
  void unfold_gram(const void *X, int dt, int64_t L, int64_t J, int64_t T, double *G) override {
    RoctxRange roctx_("K12/K13 unfold gram");
    const int64_t C = L * T;
    if (dt == F64 && J >= 64 && J <= 8192 && C >= 16 && C <= 4096 &&
        (double)J * (double)C * 8.0 <= 64e6) {
      // the Gram of a HOOI leaf (a few hundred rows and columns): the latency-bound symmetric
      // product of the eigen-step (one 16 x 16 tile per workgroup, 8 waves split the columns) on
      // the unfolding with the mode in front — 8 us instead of 31 at cfg5
      const double *A = (const double *)X;
      if (L > 1) {
        double *Ym = (double *)ensure(ws_big2_, ws_big2_sz_, sizeof(double) * (size_t)J * C);
        transpose_batched(X, F64, L, J, T, Ym);
        A = Ym;
      }
      prof_begin(1, (double)C * J * 8.0);
      sym_product(A, J, A, J, nullptr, 0, G, J, (int)J, (int)C, 1.0, 0.0);
      prof_end();
      HIP_CHECK(hipGetLastError());
      return;
    }
    if (dt == F32 && J >= 64 && C >= 4096 &&
        ((L == 1 && J % 4 == 0) || (L > 1 && L % 4 == 0))) {
      // K13 at size (the HOSVD Grams of the full tensor): the tiled SYRK, upper triangle only
      const int nt = (int)((J + 63) / 64);
      const int ntri = nt * (nt + 1) / 2;
      // (three workgroups per CU are resident — 134 registers, 40 KB of LDS —: the split is rounded DOWN
      // so that tiles x splits fit one round; rounded up, cfg5's 28 x 28 = 784 workgroups ran as 768 +
      // 16, i.e. two rounds: 1.10 ms per Gram)
      int nsplit = (int)std::min<int64_t>(std::max<int64_t>(1, ((int64_t)ncu_ * 3) / ntri),
                                          std::max<int64_t>(1, C / 1024));
      nsplit = std::min(nsplit, 512);
      int64_t per = ((C + nsplit - 1) / nsplit + 31) / 32 * 32;
      nsplit = (int)((C + per - 1) / per);
      double *dst = nsplit > 1 ? (double *)ensure(ws_slab_, ws_slab_sz_, sizeof(double) * nsplit * J * J) : G;
      prof_begin(1, (double)C * J * dtype_size(dt));
      hipLaunchKernelGGL(k_unfold_syrk_f32, dim3((unsigned)ntri, 1, (unsigned)nsplit), dim3(256), 0, st_,
                         (const float *)X, L, J, T, per, dst);
      prof_end();
      HIP_CHECK(hipGetLastError());
      if (nsplit > 1) {
        hipLaunchKernelGGL(k_slab_reduce, dim3(grid_for(J * J, 256)), dim3(256), 0, st_, dst, nsplit,
                           J * J, J * J, 1, G, (int64_t)1, (int64_t)0, 0, (int64_t)0, (int64_t)0);
        HIP_CHECK(hipGetLastError());
      }
      return;
    }
    const int tiles = (int)((J + 31) / 32);
    int nsplit = 1;
    const int64_t want = (int64_t)ncu_ * 4;
    if ((int64_t)tiles * tiles < want) nsplit = (int)std::min<int64_t>(
        (want + (int64_t)tiles * tiles - 1) / ((int64_t)tiles * tiles), std::max<int64_t>(1, C / 256));
    nsplit = std::max(1, std::min(nsplit, 1024));
    int64_t per = (C + nsplit - 1) / nsplit;
    per = (per + 31) / 32 * 32;
    nsplit = (int)((C + per - 1) / per);
    double *dst = G;
    if (nsplit > 1) dst = (double *)ensure(ws_slab_, ws_slab_sz_, sizeof(double) * nsplit * J * J);
    dim3 grid(tiles, tiles, nsplit);
    prof_begin(1, (double)C * J * dtype_size(dt));
    if (J >= 16) {  // matrix cores (K13); tiny modes: the VALU tile kernel
      if (dt == F32)
        hipLaunchKernelGGL(k_unfold_gram_mfma<float>, grid, dim3(256), 0, st_, (const float *)X, L, J,
                           T, per, dst);
      else
        hipLaunchKernelGGL(k_unfold_gram_mfma<double>, grid, dim3(256), 0, st_, (const double *)X, L,
                           J, T, per, dst);
    } else if (dt == F32)
      hipLaunchKernelGGL(k_unfold_gram<float>, grid, dim3(256), 0, st_, (const float *)X, L, J, T,
                         per, dst);
    else
      hipLaunchKernelGGL(k_unfold_gram<double>, grid, dim3(256), 0, st_, (const double *)X, L, J,
                         T, per, dst);
    prof_end();
    HIP_CHECK(hipGetLastError());
    if (nsplit > 1) {
      hipLaunchKernelGGL(k_slab_reduce, dim3(grid_for(J * J, 256)), dim3(256), 0, st_, dst, nsplit,
                         J * J, J * J, 1, G, (int64_t)1, (int64_t)0, 0, (int64_t)0, (int64_t)0);
      HIP_CHECK(hipGetLastError());
    }
  }
  // Leading eigenvectors of the s x s Gram (the reference's MTM.svd(U,S,VT,rank),
  // als_Tucker.cxx:20,402). SURVEY.md §2.1 K12 allows the vendor symmetric eigensolver here:
  // rocSOLVER dsyevd, resolved with dlopen on first use (only Tucker sessions ever load it).
  void top_eigvecs(double *G, int64_t J, int rank, double *U) override {
    RoctxRange roctx_("K12 eig (full solver)");
    if (J <= 64) {  // small modes: in-LDS Jacobi, one block
      launch_top_eig_small(G, (int)J, rank, U, nullptr);
      HIP_CHECK(hipGetLastError());
      return;
    }
    if (J <= kJacobiBigMax) {
      // 64 < J <= 128 (the small side of a tall unfolding at core ranks 70-100, a short mode): the
      // whole eigen-decomposition by the one-workgroup one-sided Jacobi, no vendor library
      double *w = (double *)ensure(ws_jac_, ws_jac_sz_, sizeof(double) * (2 * (size_t)J * J + 2 * J));
      double *Vt = w, *Y = Vt + J * J, *evd = Y + J * J, *D = evd + J;
      const size_t lds = sizeof(double) * ((size_t)J * (J + 1) + 256) + sizeof(int) * 256;
      hipLaunchKernelGGL(k_jacobi_onesided, dim3(1), dim3(1024), lds, st_, G, (int)J, Vt, Y, evd, (double *)nullptr);
      HIP_CHECK(hipMemcpyAsync(U, Y, sizeof(double) * J * rank, hipMemcpyDeviceToDevice, st_));
      // (eig_bootstrap reads the eigenvalues ASCENDING from the start of ws_krp_, as dsyevd leaves them)
      double *Dk = (double *)ensure(ws_krp_, ws_krp_sz_, sizeof(double) * (2 * J + 2));
      hipLaunchKernelGGL(k_reverse_copy, dim3(1), dim3(256), 0, st_, evd, (int)J, Dk);
      (void)D;
      HIP_CHECK(hipGetLastError());
      return;
    }
    RocSolver &rs = rocsolver();
    double *D = (double *)ensure(ws_krp_, ws_krp_sz_, sizeof(double) * (2 * J + 2));
    double *E = D + J;
    int *info = (int *)(E + J);
    int rc = rs.dsyevd(rs.handle, 211 /*evect_original*/, 121 /*fill_upper*/, (int)J, G, (int)J, D,
                       E, info);
    if (rc != 0) throw std::runtime_error("ppals: rocsolver_dsyevd failed");
    hipLaunchKernelGGL(k_take_top, dim3(grid_for(J * rank, 256)), dim3(256), 0, st_, G, J, rank, U);
    HIP_CHECK(hipGetLastError());
  }
  static constexpr int kTailBlocks = 64;
  // the one read-back of a projector step (layout: kernels_eig.hip.h, kEigOff*)
  static_assert(kTailBlocks == 64, "residual shares: 64 words");
  static constexpr size_t kEE = (size_t)kEigEvMax * kEigEvMax;  // a (rank + wide)^2 scratch matrix
  static constexpr size_t kEigReadback = sizeof(double) * kEigChkDoubles;
  // ---- K12 inside a HOOI iteration: spectral projector by the scaled Newton-Schulz sign iteration
  // (kernels_eig.hip.h), warm-started per slot, verified by trace(P) == rank, else the full solver.
  struct EigState {
    int64_t J = 0;
    int rank = 0;
    bool valid = false;
    double lamR = 0, lamR1 = 0, rho = 0;  // rank-th / next eigenvalue, largest eigenvalue
    double rho_frob = 0;  // 1.0001 ||G - deflation - sigma I||_F of the last projector step (0: none)
    double head = 4.0;    // head room of the spectral-bound scale (projector_step, fused form)
    double dom_move = -1;  // sine between the dominant eigenvector of the last two steps (< 0: not measured)
    // lazy Rayleigh-Ritz (eig_lazy): the Jacobi of H runs on the second stream
    bool lazy = false;          // the session allows it for this slot
    bool lazy_pending = false;  // eigenvalues of the last step still to be read (evd)
    bool rot_pending = false;   // the caller has not applied Y yet
    int lazy_m = 0;
    bool lazy_strict = false;
    bool jacobi_launched = false;
    double *Hd = nullptr, *Yd = nullptr, *evd = nullptr;  // 64*64, 64*64, 64 (device)
    double *ev_pinned = nullptr;  // 64 doubles of pinned host memory the Jacobi writes directly
    hipEvent_t ev_h = nullptr, ev_done = nullptr;
    double *Q = nullptr;                  // previous basis (J x rank)
    double *Qn = nullptr;                 // spare buffer of the same size (the step's result lands here)
    double evh[kEigEvMax] = {0};          // eigenvalues of the last accepted step (descending)
    int fast = 0, full = 0;
    // deferred acceptance (eig_defer / eig_verify): the step's read-back lands in the slot's own
    // pinned block behind ev_chk; what the checks need from the time of the call is kept in `dp`
    bool defer = false;     // the session allows it for this slot
    int stable = 0;         // consecutive steps accepted at the first attempt, as scheduled
    bool deferred = false;  // a step is waiting for eig_verify
    double *chk_pinned = nullptr;
    hipEvent_t ev_chk = nullptr;
    double *GBd = nullptr, *chkd = nullptr;  // G*B (J x rank) and the checks' device block, the slot's own
    double *Gd = nullptr;                    // the slot's own Gram (eig_gram): the second stream reads it
    size_t GBd_elems = 0, Gd_elems = 0;
    struct {
      double sigma = 0, rho = 0, ell0 = 0;
      int m = 0, iters = 0, pow = 0;
    } dp;
    int n_deferred = 0, n_defer_failed = 0;
  };
  // The symmetric product C = alpha A Bt^T + beta D of the sign iteration (and of the leaf Grams),
  // by size: a few hundred rows are a latency problem — one 16 x 16 tile per workgroup, eight waves
  // splitting K — from sym_lds_min_ rows on (PPALS_SYM_LDS_MIN, 768: tools/nsprod_bench.hip,
  // profiles/r04d_nsprod_*.txt: 640 rows 18.6 vs 21.3 us, 896 rows 35.6 vs 30.8, 1344 rows 88.6 vs
  // 68.1) it is 2 J^3 flops — 32 x 32 tiles staged through LDS. sym_tiles() = workgroups = check-sum
  // partials per product.
  // (the GEMM kernels address their operands with 32-bit element offsets)
  static void gemm_offsets_fit(int64_t lda, int64_t ca, int64_t ldb, int64_t cb) {
    if ((double)lda * (double)ca >= 4.0e9 || (double)ldb * (double)cb >= 4.0e9)
      throw std::runtime_error("ppals: dense operand of the eigen-step beyond 2^32 elements");
  }
  unsigned sym_tiles(int M) const {
    const unsigned nt = (unsigned)((M + (M >= sym_lds_min_ ? 31 : 15)) / (M >= sym_lds_min_ ? 32 : 16));
    return nt * (nt + 1) / 2;
  }
  void sym_product(const double *A, int64_t lda, const double *Bt, int64_t ldb, const double *D, int64_t ldd,
                   double *C, int64_t ldc, int M, int K, double alpha, double beta, int chk_mode = 0,
                   double *chk_part = nullptr) {
    gemm_offsets_fit(lda, K, ldb, K);
    if (M >= sym_lds_min_)
      hipLaunchKernelGGL(k_dgemm_nt_sym_lds<32>, dim3(sym_tiles(M)), dim3(256), 0, st_, A, lda, Bt, ldb, D, ldd, C,
                         ldc, M, K, alpha, beta, chk_mode, chk_part);
    else
      hipLaunchKernelGGL(k_dgemm_nt_sym<8>, dim3(sym_tiles(M)), dim3(512), 0, st_, A, lda, Bt, ldb, D, ldd, C, ldc,
                         M, K, alpha, beta, chk_mode, chk_part);
  }
  void gemm_nt(const double *A, int64_t lda, const double *Bt, int64_t ldb, const double *D,
               int64_t ldd, double *C, int64_t ldc, int M, int N, int K, double alpha, double beta) {
    gemm_offsets_fit(lda, K, ldb, K);
    dim3 grid((unsigned)((M + 15) / 16), (unsigned)((N + 15) / 16));
    hipLaunchKernelGGL(k_dgemm_nx<false>, grid, dim3(256), 0, st_, A, lda, Bt, ldb, D, ldd, C, ldc, M,
                       N, K, alpha, beta);
  }
  // in-LDS eigen-decomposition of a small symmetric matrix (J <= 64) by one workgroup: a thread
  // per element when there are that many (two barriers per Jacobi round whatever the size)
  void launch_top_eig_small(const double *G, int J, int rank, double *U, double *evals) {
    // (a thread per element up to 15 waves, plus the wave that prepares the next round's angles)
    const int nthr = std::min(1024, std::max(192, (J * J + 63) / 64 * 64) + 64);
    hipLaunchKernelGGL(k_top_eig_small, dim3(1), dim3(nthr), top_eig_small_lds(J), st_, G, J, rank, U,
                       evals);
  }
  // the same with the second operand as it is (K x N, column-major): thin tails, no transposition
  void gemm_nn(const double *A, int64_t lda, const double *B, int64_t ldb, const double *D,
               int64_t ldd, double *C, int64_t ldc, int M, int N, int K, double alpha, double beta,
               const double *Qov = nullptr, int mov = 0, hipStream_t st = nullptr) {
    gemm_offsets_fit(lda, K, ldb, N);
    dim3 grid((unsigned)((M + 15) / 16), (unsigned)((N + 15) / 16));
    hipLaunchKernelGGL(k_dgemm_nx<true>, grid, dim3(256), 0, st ? st : st_, A, lda, B, ldb, D, ldd, C, ldc,
                       M, N, K, alpha, beta, Qov, mov);
  }
  // full solver + what the next call of the slot needs (rank-th and next eigenvalue, the basis)
  void eig_bootstrap(EigState &es, double *G, int64_t J, int rank, double *U) {
    // ||G||_F^2 = sum of squared eigenvalues: taken before dsyevd overwrites G
    top_eigvecs(G, J, rank, U);
    es.full++;
    if (J <= 64 || rank >= J || rank + 16 > kEigEvMax) {
      es.valid = false;
      return;
    }
    // dsyevd left the ascending eigenvalues in ws_krp_ (D)
    double lam[kEigEvMax + 1];
    const double *D = (const double *)ws_krp_;
    HIP_CHECK(hipMemcpyAsync(lam, D + (J - rank - 1), (rank + 1) * sizeof(double),
                             hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    es.lamR1 = lam[0];
    es.lamR = lam[1];
    for (int d = 0; d < rank; d++) es.evh[d] = lam[rank - d];  // descending
    es.rho = lam[rank];
    if (!es.Q || es.J != J || es.rank != rank) {
      if (es.Q) hipFree(es.Q);
      if (es.Qn) hipFree(es.Qn);
      HIP_CHECK(hipMalloc(&es.Q, sizeof(double) * J * rank));
      HIP_CHECK(hipMalloc(&es.Qn, sizeof(double) * J * rank));
    }
    es.J = J;
    es.rank = rank;
    HIP_CHECK(hipMemcpyAsync(es.Q, U, sizeof(double) * J * rank, hipMemcpyDeviceToDevice, st_));
    es.rho_frob = 0;  // the next projector step reads its norm once
    es.head = 4.0;
    es.valid = es.lamR > 0 && es.lamR1 >= 0 && es.lamR > es.lamR1 * (1 + 1e-9);
  }
  // Z (J x r) -> orthonormal columns, Cholesky QR twice (status 2 of a pass = "far from
  // orthonormal" is no failure here: the second pass repairs it); returns the buffer holding the
  // result. npass = 1: for a Z that is close to orthonormal already — the caller must treat ANY
  // non-zero status as a failure then.
  double *chol_qr2(double *cur, double *nxt, int64_t J, int r, double *C, int *status,
                   int npass = 2) {
    if (r > 64) {
      // more columns than the one-wave Cholesky holds (core ranks above 48, test_ALS.cxx:366-379):
      // block Gram-Schmidt, 64 columns at a time, in place — its own read-back says whether it held
      const int st[2] = {orthonormalize(cur, J, r) ? 0 : 1, 0};
      HIP_CHECK(hipMemcpyAsync(status, st, sizeof(int) * std::min(npass, 2), hipMemcpyHostToDevice, st_));
      HIP_CHECK(hipStreamSynchronize(st_));
      return cur;
    }
    // R^-1 by ONE elimination of [C | I] on a whole workgroup (k_chol_m: two pivots per barrier, all
    // 2 r^2 elements of a step in parallel) instead of the one-wave Cholesky + column-wise substitution
    // (k_chol_rinv: 175 us a call at 64 columns, 1.8 ms of a time-lapse HOOI sweep; round 6)
    const size_t lds_m = sizeof(double) * (4 * (size_t)r * r + r + 8);
    double *Minv = (double *)ensure(ws_cholm_, ws_cholm_sz_, sizeof(double) * 64 * 64);
    for (int pass = 0; pass < npass; pass++) {
      hipLaunchKernelGGL(k_tn_small, dim3((r * r + 15) / 16), dim3(1024), 0, st_, cur, cur, J, r, C);
      hipLaunchKernelGGL(k_chol_m, dim3(1), dim3(k_chol_threads(r)), lds_m, st_, cur, (const double *)nullptr, J, r, 0,
                         (const double *)C, Minv, status + pass);
      hipLaunchKernelGGL(k_right_mult, dim3(grid_for(J * r, 256)), dim3(256),
                         sizeof(double) * r * r, st_, cur, J, r, Minv, nxt);
      std::swap(cur, nxt);
    }
    return cur;
  }
  // block Gram-Schmidt with re-orthogonalisation: 64 columns at a time are cleared of the columns
  // before them (twice) and made orthonormal by Cholesky QR (twice)
  bool orthonormalize(double *U, int64_t rows, int r) override {
    if (r <= 0) return true;
    if (r > 2048) throw std::runtime_error("ppals: orthonormalize supports at most 2048 columns");
    const int nblk = (r + 63) / 64;
    double *w = (double *)ensure(ws_orth_, ws_orth_sz_,
                                 sizeof(double) * ((size_t)rows * 64 + 64 * 64 + (size_t)r * 64) +
                                     sizeof(int) * 2 * 64);
    double *tmp = w, *C = tmp + (size_t)rows * 64, *T = C + 64 * 64;
    int *status = (int *)(T + (size_t)r * 64);
    HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int) * 2 * nblk, st_));
    for (int b = 0; b < nblk; b++) {
      const int b0 = b * 64, rb = std::min(64, r - b0);
      double *Zb = U + (size_t)rows * b0;
      for (int pass = 0; pass < 2 && b0 > 0; pass++) {
        hipLaunchKernelGGL(k_tn_rect, dim3((b0 * rb + 15) / 16), dim3(1024), 0, st_, U, b0, Zb, rb,
                           rows, T);
        hipLaunchKernelGGL(k_sub_mult, dim3(grid_for(rows * rb, 256)), dim3(256), 0, st_, Zb, rows,
                           rb, U, b0, T);
      }
      double *res = chol_qr2(Zb, tmp, rows, rb, C, status + 2 * b);  // two passes: ends in Zb
      if (res != Zb) throw std::logic_error("ppals: orthonormalize buffer parity");
    }
    int hs[2 * 32];
    HIP_CHECK(hipMemcpyAsync(hs, status, sizeof(int) * 2 * nblk, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    for (int b = 0; b < nblk; b++)  // first pass: 2 = "far from orthonormal" is what the second is for
      if (hs[2 * b] == 1 || hs[2 * b + 1] != 0) return false;
    return true;
  }
  // Rayleigh-Ritz of G on the orthonormal basis B (J x r): U = eigenvectors sorted descending,
  // ev (device) = eigenvalues; GB/H/Yr/Bt scratch. Leaves G*U in GU when GU != nullptr.
  void rayleigh_ritz(const double *G, const double *B, int64_t J, int r, double *Bt, double *GB,
                     double *H, double *Yr, double *U, double *ev, double *GU) {
    transpose2d(B, F64, J, r, Bt);
    gemm_nt(G, J, Bt, r, nullptr, 0, GB, J, (int)J, r, (int)J, 1.0, 0.0);
    if (r > 64) {
      // above the in-LDS two-sided Jacobi: H = B^T (G B) and the products with Y on the matrix
      // cores, the eigen-decomposition of H by the one-workgroup one-sided Jacobi (r <= 128)
      if (r > kJacobiBigMax) throw std::runtime_error("ppals: Rayleigh-Ritz supports at most 128 columns");
      gemm_nn(Bt, r, GB, J, nullptr, 0, H, r, r, r, (int)J, 1.0, 0.0);
      double *Vt = (double *)ensure(ws_jac_, ws_jac_sz_, sizeof(double) * (size_t)r * r);
      const size_t lds = sizeof(double) * ((size_t)r * (r + 1) + 256) + sizeof(int) * 256;
      hipLaunchKernelGGL(k_jacobi_onesided, dim3(1), dim3(1024), lds, st_, H, r, Vt, Yr, ev, (double *)nullptr);
      gemm_nn(B, J, Yr, r, nullptr, 0, U, J, (int)J, r, r, 1.0, 0.0);
      if (GU) gemm_nn(GB, J, Yr, r, nullptr, 0, GU, J, (int)J, r, r, 1.0, 0.0);
      return;
    }
    hipLaunchKernelGGL(k_tn_small, dim3((r * r + 15) / 16), dim3(1024), 0, st_, B, GB, J, r, H);
    launch_top_eig_small(H, r, r, Yr, ev);
    hipLaunchKernelGGL(k_right_mult, dim3(grid_for(J * r, 256)), dim3(256),
                       sizeof(double) * r * r, st_, B, J, r, Yr, U);
    if (GU)
      hipLaunchKernelGGL(k_right_mult, dim3(grid_for(J * r, 256)), dim3(256),
                         sizeof(double) * r * r, st_, GB, J, r, Yr, GU);
  }
  // Small modes (16 <= J <= 64: the in-LDS Jacobi, 1 ms at J = 50 from a cold start — 15 % of an
  // order-6 s = 50 HOOI sweep): the Jacobi runs on H = Q^T G Q with Q the slot's previous
  // eigenvector matrix. G changes little from sweep to sweep, so H is nearly diagonal and the
  // cyclic Jacobi (quadratically convergent) needs 2-3 sweeps instead of ~8; the result Q Y is
  // the same eigen-decomposition, with no assumption about gaps. Every 64th call starts cold
  // (the product Q Y accumulates rounding in its orthogonality).
  struct SmallEig {
    int64_t J = 0;
    double *Q = nullptr;  // J x J, ranked descending
    bool valid = false;
    int age = 0;
  };
  std::map<int, SmallEig> eig_small_;  // by slot (session base + mode, see eig_session_new)
  void top_eigvecs_small_warm(double *G, int64_t J, int rank, double *U, int slot) {
    SmallEig &sm = eig_small_[slot];
    const int Ji = (int)J;
    const size_t nJJ = (size_t)J * J;
    if (sm.J != J) {
      if (sm.Q) hipFree(sm.Q);
      HIP_CHECK(hipMalloc(&sm.Q, sizeof(double) * nJJ));
      sm.J = J;
      sm.valid = false;
    }
    if (!sm.valid || ++sm.age >= 64) {
      launch_top_eig_small(G, Ji, Ji, sm.Q, nullptr);
      sm.valid = true;
      sm.age = 0;
    } else {
      double *w = (double *)ensure(ws_eig_, ws_eig_sz_, sizeof(double) * 5 * nJJ);
      double *Qt = w, *C1 = Qt + nJJ, *H = C1 + nJJ, *Y = H + nJJ, *Qn = Y + nJJ;
      transpose2d(sm.Q, F64, J, J, Qt);
      gemm_nt(G, J, Qt, J, nullptr, 0, C1, J, Ji, Ji, Ji, 1.0, 0.0);  // G Q
      hipLaunchKernelGGL(k_tn_small, dim3((Ji * Ji + 15) / 16), dim3(1024), 0, st_, sm.Q, C1, J, Ji, H);
      launch_top_eig_small(H, Ji, Ji, Y, nullptr);
      hipLaunchKernelGGL(k_right_mult, dim3(grid_for(J * J, 256)), dim3(256), sizeof(double) * nJJ, st_,
                         sm.Q, J, Ji, Y, Qn);
      HIP_CHECK(hipMemcpyAsync(sm.Q, Qn, sizeof(double) * nJJ, hipMemcpyDeviceToDevice, st_));
    }
    HIP_CHECK(hipMemcpyAsync(U, sm.Q, sizeof(double) * J * rank, hipMemcpyDeviceToDevice, st_));
    HIP_CHECK(hipGetLastError());
  }
  void top_eigvecs_warm(double *G, int64_t J, int rank, double *U, int slot) override {
    RoctxRange roctx_("K12 eig (projector route)");
    if (J >= 16 && J <= 64 && rank <= J && slot >= 0 && eig_fast_) {
      top_eigvecs_small_warm(G, J, rank, U, slot);
      return;
    }
    // (up to core rank 112: rank + 16 columns of a wide tail / cold start must fit kEigEvMax)
    if (J <= 64 || rank + 16 > kEigEvMax || rank + 16 >= J || slot < 0 || !eig_fast_) {
      top_eigvecs(G, J, rank, U);
      return;
    }
    EigState &es = eig_state_[slot];
    resolve_lazy(es);
    es.rot_pending = false;  // (whatever rotation was owed belonged to the factor this call replaces)
    if (es.valid && (es.J != J || es.rank != rank)) es.valid = false;
    if (es.valid && projector_step(es, G, J, rank, U, slot, false)) return;
    if (es.valid) {  // the spectral bound may have been outrun: once more on the measured norm
      eig_frob_once_ = true;
      const bool ok = projector_step(es, G, J, rank, U, slot, false);
      eig_frob_once_ = false;
      if (ok) {
        es.head = 8.0;
        return;
      }
    }
    // nothing known about this matrix (first call of the slot, a shift that no longer separates
    // the wanted eigenvalues): a few steps of block subspace iteration place the shift, the same
    // projector step — checked against machine precision, not against an estimated gap — delivers
    // the eigenpairs; the full solver remains the fallback of the fallback
    if (eig_cold_) {
      if (cold_ritz_state(es, G, J, rank, slot)) {
        // (long modes: thin products instead of ~46 products of J^3; kColdSubspaceFrom: where a J^3
        // product, 7.4 us at J = 400, stops being cheap against a step of thin products + Rayleigh-Ritz)
        if (J >= cold_subspace_from_ && cold_subspace(es, G, J, rank, U, slot)) return;
        if (projector_step(es, G, J, rank, U, slot, true)) return;
      }
      // a flat spectrum around the cut (the Gram of a noise tensor: the HOSVD initialisation), where
      // Ritz values cannot place the shift: place it by COUNTING eigenvalues with the sign iteration
      if (cold_ok_ && cold_bisect(es, G, J, rank, U, slot)) return;
    }
    es.valid = false;
    if (eig_debug_)
      fprintf(stderr, "[ppals eig] slot %d J %lld rank %d: full solver (%s)\n", slot, (long long)J, rank,
              J <= kJacobiBigMax ? "one-sided Jacobi, one workgroup" : "dsyevd");
    eig_bootstrap(es, G, J, rank, U);
  }
  // Warm-start state belongs to the SESSION that made it: a session draws a block of 64 slots
  // here and gives it back when it ends, so sessions that alternate on one context (or the thin
  // c x c route and its s x s fallback inside one factor update, which use different slots of the
  // block) never see each other's bases, gaps or eigenvalue scales.
  void free_lazy(EigState &es) {
    if (st2_) hipStreamSynchronize(st2_);  // (nothing on the second stream reads the slot's buffers any more)
    if (es.Hd) hipFree(es.Hd);
    if (es.Yd) hipFree(es.Yd);
    if (es.evd) hipFree(es.evd);
    if (es.ev_pinned) hipHostFree(es.ev_pinned);
    es.ev_pinned = nullptr;
    if (es.chk_pinned) hipHostFree(es.chk_pinned);
    es.chk_pinned = nullptr;
    if (es.ev_chk) hipEventDestroy(es.ev_chk);
    es.ev_chk = nullptr;
    if (es.GBd) hipFree(es.GBd);
    if (es.chkd) hipFree(es.chkd);
    if (es.Gd) hipFree(es.Gd);
    es.GBd = es.chkd = es.Gd = nullptr;
    es.GBd_elems = es.Gd_elems = 0;
    if (es.ev_h) hipEventDestroy(es.ev_h);
    if (es.ev_done) hipEventDestroy(es.ev_done);
    es.Hd = es.Yd = es.evd = nullptr;
    es.ev_h = es.ev_done = nullptr;
  }
  // second stream, events and the slot's buffers: set up when a session ASKS for lazy steps (its
  // set-up), not inside the first timed sweep
  void lazy_prepare(EigState &es) {
    if (!st2_) HIP_CHECK(hipStreamCreateWithFlags(&st2_, hipStreamNonBlocking));
    if (!handover_) {
      // (zero BEFORE any stream can wait on it: a recycled allocation may hold a larger sequence number)
      HIP_CHECK(hipMalloc(&handover_, 16));
      HIP_CHECK(hipMemset(handover_, 0, 16));
      HIP_CHECK(hipDeviceSynchronize());
      // A runtime without stream memory operations, or one on which a value stored by a KERNEL is not
      // seen by the waiting stream, gets the event hand-over instead. The probe is the real thing once:
      // a one-workgroup kernel on the sweep's stream publishes 1, the second stream waits for it and
      // records an event, which must arrive within a deadline. If it does not, the second stream is
      // released from the host and the flag is never waited on again.
      if (handover_ok_) {
        hipEvent_t probe = nullptr;
        bool ok = hipEventCreateWithFlags(&probe, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipStreamWaitValue64(st2_, handover_, 1, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull) == hipSuccess;
        if (ok) {
          ok = hipEventRecord(probe, st2_) == hipSuccess;
          hipLaunchKernelGGL(k_handover_probe, dim3(1), dim3(64), 0, st_, handover_, 1ull);
          ok = ok && hipGetLastError() == hipSuccess;
          const auto t0 = std::chrono::steady_clock::now();
          bool arrived = false;
          while (ok && !arrived) {
            const hipError_t q = hipEventQuery(probe);
            if (q == hipSuccess) arrived = true;
            else if (q != hipErrorNotReady) ok = false;
            else if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) break;
          }
          if (!arrived) {
            // release the waiting stream whatever went wrong (the wait is queued already)
            const unsigned long long one = 1;
            (void)hipMemcpy(handover_, &one, sizeof one, hipMemcpyHostToDevice);
            (void)hipStreamSynchronize(st2_);
            ok = false;
          }
        }
        if (probe) hipEventDestroy(probe);
        (void)hipGetLastError();
        if (!ok) handover_ok_ = false;
        handover_seq_ = 1;  // (sequence numbers go on from the probe's)
      }
    }
    if (!es.Hd) {
      HIP_CHECK(hipMalloc(&es.Hd, sizeof(double) * 64 * 64));
      HIP_CHECK(hipMalloc(&es.Yd, sizeof(double) * 64 * 64));
      HIP_CHECK(hipMalloc(&es.evd, sizeof(double) * 64));
      HIP_CHECK(hipHostMalloc(&es.ev_pinned, sizeof(double) * 64, hipHostMallocDefault));
      HIP_CHECK(hipEventCreateWithFlags(&es.ev_h, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&es.ev_done, hipEventDisableTiming));
      HIP_CHECK(hipHostMalloc(&es.chk_pinned, kEigReadback, hipHostMallocDefault));
      HIP_CHECK(hipEventCreateWithFlags(&es.ev_chk, hipEventDisableTiming));
      HIP_CHECK(hipMalloc(&es.chkd, kEigReadback));
    }
  }
  void eig_lazy(int slot, bool on) override {
    if (slot < 0) return;
    EigState &es = eig_state_[slot];
    es.lazy = on;
    if (es.lazy) lazy_prepare(es);
  }
  void eig_defer(int slot, bool on) override {
    if (slot < 0) return;
    eig_state_[slot].defer = on && eig_defer_ok_;
  }
  // The slot's own Gram buffer. A deferred step leaves G B, H = B^T G B and the residual to the
  // second stream, which reads G long after the sweep's stream has moved on to the next mode — so
  // the Gram of a deferring slot must not live in the caller's shared workspace. Whatever the
  // second stream still owes this slot is waited for here (the wait resolve_lazy() would do a
  // moment later), before the caller overwrites the buffer.
  double *eig_gram(int slot, int64_t J) override {
    if (slot < 0 || J > 4096) return nullptr;
    auto it = eig_state_.find(slot);
    if (it == eig_state_.end() || !it->second.defer || !it->second.lazy) return nullptr;
    EigState &es = it->second;
    if (es.jacobi_launched) HIP_CHECK(hipEventSynchronize(es.ev_done));
    if (es.Gd_elems < (size_t)(J * J)) {
      if (es.Gd) HIP_CHECK(hipFree(es.Gd));
      HIP_CHECK(hipMalloc(&es.Gd, sizeof(double) * (size_t)(J * J)));
      es.Gd_elems = (size_t)(J * J);
    }
    return es.Gd;
  }
  // What decides whether a projector step's result is accepted, from its one read-back `hc`
  // (chk[16] | evW[64] | status[8 ints] | lamD[64] | per-workgroup residual shares, already summed
  // into hc[4]) and the quantities the step was scheduled with.
  struct StepCheck {
    bool converged = false, good = false;
    double cnt = 0, res = 0, rho_now = 0, gap_now = 0, res_tol = 0;
  };
  StepCheck check_step(const double *hc, int64_t J, int rank, double sigma, double rho, bool strict,
                       bool fused_tail, bool fused_scale, bool lazy_used) const {
    StepCheck c;
    const double *evn = hc + 16;
    const int *hs = (const int *)(hc + kEigOffStatus);
    c.cnt = 0.5 * (hc[1] + (double)J);
    c.res = std::sqrt(hc[4]);
    c.rho_now = fused_scale ? rho : 1.0001 * std::sqrt(hc[8]);
    // (fused tail: hc[0] is ||E||_F^2 of the iterate BEFORE the last step, E = X^2 - I; the step
    // squares it — E_new = (3 E^2 + E^3) / 4 — so ||E_new||_F <= ||E||_F^2 = hc[0])
    c.converged = fused_tail ? hc[0] <= 1e-10 * std::sqrt((double)J) : hc[0] <= 1e-20 * (double)J;
    // accepted when the eigenpair residual is at the level the full solver reaches: 1e-9 of the
    // gap (projector error <= 1e-9), or the rounding floor eps * lambda_1 of any method
    // (from THIS step's quantities only: trace(P) == rank says that exactly `rank` eigenvalues
    // lie above this step's shift, so the gap below the rank-th one is at least its distance to
    // the shift; the slot's previous gap / eigenvalue scale only scheduled the iteration)
    // (lazy tail: the Jacobi has not run yet — Gershgorin bounds of H's spectrum stand in for
    // the smallest / largest eigenvalue of the subspace, and the residual is the subspace's)
    const double ev_lo = lazy_used ? hc[5] : evn[rank - 1], ev_hi = lazy_used ? hc[6] : evn[0];
    c.gap_now = std::max(0.0, ev_lo - sigma);
    c.res_tol = (strict ? 1e-13 * ev_hi : std::max(1e-9 * c.gap_now, 1e-14 * ev_hi)) * std::sqrt((double)rank);
    const bool chol_ok = hs[0] == 0;
    c.good = c.converged && std::fabs(c.cnt - rank) < 1e-6 && chol_ok && hs[2] != 1 && hs[3] == 0 &&
             c.res <= c.res_tol && std::isfinite(c.rho_now);
    return c;
  }
  bool eig_deferred(int slot) override {
    auto it = eig_state_.find(slot);
    return it != eig_state_.end() && it->second.deferred;
  }
  int eig_verify(int slot, bool discard) override {
    auto it = eig_state_.find(slot);
    if (it == eig_state_.end() || !it->second.deferred) return -1;
    EigState &es = it->second;
    es.deferred = false;
    HIP_CHECK(hipEventSynchronize(es.ev_chk));
    double *hc = es.chk_pinned;
    {
      const double *rp = hc + kEigOffResp;
      double r2 = 0;
      for (int b2 = 0; b2 < kTailBlocks; b2++) r2 += rp[b2];
      hc[4] = r2;
    }
    const StepCheck c = check_step(hc, es.J, es.rank, es.dp.sigma, es.dp.rho, false, true, true, true);
    bool ok = c.good && !discard;
    if (ok && eig_defer_fail_ > 0 && ++eig_defer_count_ % eig_defer_fail_ == 0) ok = false;  // (tests)
    if (eig_debug_) {
      const int *hs = (const int *)(hc + kEigOffStatus);
      fprintf(stderr, "[ppals eig] slot %d J %lld rank %d: lamR %.6e lamR1 %.6e deflated %d rho %.3e "
                      "(now %.3e) ell0 %.2e iters %d | ||X^2-I||^2 %.3e count %.6f residual %.3e "
                      "(gap %.3e) dominant vector moved %.1e after %d power steps chol %d%d%d%d -> %s (fast %d full %d)\n",
              slot, (long long)es.J, es.rank, es.lamR, es.lamR1, es.dp.m, es.dp.rho, c.rho_now, es.dp.ell0,
              es.dp.iters, hc[0], c.cnt, c.res, c.gap_now, es.dp.m == 1 ? std::sqrt(std::max(0.0, hc[9])) : 0.0, es.dp.pow,
              hs[0], hs[1], hs[2], hs[3],
              ok ? "accepted (deferred check)" : (discard ? "dropped (its input was withdrawn)"
                                                          : "NOT accepted (deferred check): the caller repeats the step"),
              es.fast, es.full);
    }
    if (ok) {
      // the eigenvalues arrive with the second stream's Jacobi: resolve_lazy() turns them into
      // the slot's state when the slot is used next; the caller holds the basis, Y is owed to it
      es.lazy_pending = true;
      es.rot_pending = true;
      es.lazy_m = es.dp.m;
      es.lazy_strict = false;
      es.rho_frob = c.rho_now;
      es.dom_move = es.dp.m == 1 ? std::sqrt(std::max(0.0, hc[9])) : -1.0;
      es.fast++;
      es.stable++;
      return 0;
    }
    std::swap(es.Q, es.Qn);  // (the step had put its basis in front)
    es.stable = 0;
    es.dom_move = -1;
    es.n_defer_failed++;
    return 1;
  }
  const double *eig_pending_rotation(int slot) override {
    auto it = eig_state_.find(slot);
    if (it == eig_state_.end() || !it->second.rot_pending) return nullptr;
    HIP_CHECK(hipEventSynchronize(it->second.ev_done));
    // (the rotation is read by launches on the main stream: order them behind the Jacobi)
    HIP_CHECK(hipStreamWaitEvent(st_, it->second.ev_done, 0));
    return it->second.Yd;
  }
  void eig_rotation_done(int slot) override {
    auto it = eig_state_.find(slot);
    if (it != eig_state_.end()) it->second.rot_pending = false;
  }
  // the eigenvalues a lazy step left on the second stream become the slot's state (what the
  // accepted branch of projector_step does at once when the Jacobi runs inside the step)
  void resolve_lazy(EigState &es) {
    if (!es.lazy_pending) return;
    es.lazy_pending = false;
    double evn[kEigEvMax];
    // (the Jacobi wrote them into pinned host memory itself: waiting for ITS event is all it takes —
    // a copy would queue behind the other slots' Jacobis on the second stream, or drain the main one)
    HIP_CHECK(hipEventSynchronize(es.ev_done));
    es.jacobi_launched = false;  // (nothing of this slot is in flight on the second stream any more)
    for (int d = 0; d < es.rank; d++) evn[d] = es.ev_pinned[d];
    const int rank = es.rank, m = es.lazy_m;
    bool sane = true;
    for (int d = 0; d < rank; d++) sane = sane && std::isfinite(evn[d]);
    if (!sane) {
      es.valid = false;
      return;
    }
    const int mi = std::min(m, rank - 1);
    const double growth = es.evh[mi] > 0 ? evn[mi] / es.evh[mi] : 2.0;
    es.head = std::min(8.0, std::max(1.25, 1.25 * growth * growth));
    for (int d = 0; d < rank; d++) es.evh[d] = evn[d];
    const double shift = evn[rank - 1] - es.lamR;
    es.lamR = evn[rank - 1];
    es.lamR1 = std::min(std::max(0.0, es.lamR1 + shift), es.lamR);
    if (es.lazy_strict) es.lamR1 = std::min(es.lamR1, es.lamR * (1 - 1e-6));
    if (!(es.lamR > es.lamR1 * (1 + 1e-9))) es.valid = false;
  }
  int eig_session_new() override {
    eig_next_base_ += 64;
    return eig_next_base_;
  }
  void eig_session_free(int base) override {
    hipStreamSynchronize(st_);
    for (auto it = eig_state_.begin(); it != eig_state_.end();)
      if (it->first >= base && it->first < base + 64) {
        if (it->second.Q) hipFree(it->second.Q);
        if (it->second.Qn) hipFree(it->second.Qn);
        free_lazy(it->second);
        it = eig_state_.erase(it);
      } else {
        ++it;
      }
    for (auto it = eig_small_.begin(); it != eig_small_.end();)
      if (it->first >= base && it->first < base + 64) {
        if (it->second.Q) hipFree(it->second.Q);
        it = eig_small_.erase(it);
      } else {
        ++it;
      }
  }

  // The tail of a projector step in six launches, none of them a lone workgroup with the rows in
  // The tail of a projector step in seven launches, everything that touches the J rows multi-
  // workgroup: Z = (Omega + X Omega) / 2 | C = Z^T Z and t = Q_D^T Z | M with [Q_D | Z] M
  // orthonormal (one workgroup, cols x cols) | B = [Q_D | Z] M | G B | H = B^T G B | every
  // workgroup diagonalises H itself (block Jacobi in LDS) and forms its rows of U = B Y and of the
  // residual. `rounds` = 2 repeats the Gram / Cholesky / product on B (a basis far from
  // orthonormal: cold starts, wide tails).
  void fused_tail_launches(const double *G, const double *X, int64_t J, int cols, int rank,
                           const double *Omega, const double *QD, int m, double *Z, double *Z2,
                           double *GZ, double *C, double *H, double *Uout, double *evW, double *chk,
                           int *status, const double *pe2, const double *ptr_, int np, int rounds = 1,
                           double *Uout2 = nullptr, EigState *lazy = nullptr, double *host_chk = nullptr) {
    const int Ji = (int)J;
    const int nblk = (int)std::min<int64_t>(kTailBlocks, (J + 31) / 32);
    const int rows_per = (int)((J + nblk - 1) / nblk);
    const bool deferred = lazy && host_chk && rounds == 1 && m <= 1 && cols <= kSeriesMax;
    if (deferred) {
      // The tail of a DEFERRED step (kernels_eig.hip.h, k_tn_gram / k_rmult_chol). On the sweep's
      // stream only what the next mode waits for: Z' = [q_D | P Omega_rest], its Gram, B = Z' R^-1
      // straight into the caller's buffer — three launches. Everything the CHECKS read (G B,
      // H = B^T G B, the subspace residual, the Jacobi of H) is formed from B on the second stream,
      // in the slot's own buffers, from the slot's own copy of the Gram (eig_gram); a Gram in the
      // caller's workspace is multiplied on the sweep's stream first.
      EigState &es = *lazy;
      lazy_prepare(es);
      if (es.GBd_elems < (size_t)J * cols) {
        if (es.GBd) HIP_CHECK(hipFree(es.GBd));
        HIP_CHECK(hipMalloc(&es.GBd, sizeof(double) * (size_t)J * cols));
        es.GBd_elems = (size_t)J * cols;
      }
      double *C1 = C;
      const bool own_gram = G == es.Gd;
      gemm_nn(X, J, Omega, J, Omega, J, Z, J, Ji, cols, Ji, 0.5, 0.5, QD, m);  // Z' = [q_D | P Omega_rest]
      // (a Jacobi of this slot that is still in flight — a step that was not accepted — must be through
      // with the slot's buffers before they are written again)
      if (es.jacobi_launched) HIP_CHECK(hipStreamWaitEvent(st_, es.ev_done, 0));
      hipLaunchKernelGGL(k_tn_gram, dim3((cols * cols + 15) / 16 + 1), dim3(1024), 0, st_, Z, J, cols, C1, pe2, ptr_,
                         np, chk, es.chkd);
      // The second stream takes over from the kernel itself when it reads the slot's own Gram: the
      // last workgroup of k_rmult_chol publishes a sequence number, the second stream waits for the
      // value — no marker packet on the sweep's stream (an event record + wait costs it 6-8 us per
      // step: profiles/r04c_cfg5_timeline_deferred_checks.txt, tools/waitvalue_bench.hip).
      const bool by_flag = own_gram && handover_ok_ && handover_;
      const unsigned long long seq = by_flag ? ++handover_seq_ : 0;
      const int rb = kRmultRows;
      hipLaunchKernelGGL(k_rmult_chol, dim3((unsigned)((J + rb - 1) / rb)), dim3(rmult_chol_threads(cols)),
                         rmult_chol_lds(cols), st_, Z, J, cols, m,
                         C1, Uout, status, chk,
                         es.chkd, by_flag ? (unsigned *)(handover_ + 1) : (unsigned *)nullptr,
                         by_flag ? handover_ : (unsigned long long *)nullptr, seq);
      // (a launch that failed never publishes its number: nothing may be queued behind it)
      if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
        if (by_flag) --handover_seq_;
        throw std::runtime_error(std::string("ppals: k_rmult_chol launch failed: ") + hipGetErrorString(le));
      }
      if (!own_gram) gemm_nn(G, J, Uout, J, nullptr, 0, es.GBd, J, Ji, cols, Ji, 1.0, 0.0);
      es.jacobi_launched = true;
      if (by_flag) {
        HIP_CHECK(hipStreamWaitValue64(st2_, handover_, seq, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
      } else {
        // (measured, round 4: making this event the completion signal of the launch above —
        // hipExtLaunchKernelGGL's stop event — instead of a marker packet leaves 30 us of idle main
        // stream behind it instead of 6-8: profiles/r04v_ext_launch_event.txt)
        HIP_CHECK(hipEventRecord(es.ev_h, st_));
        HIP_CHECK(hipStreamWaitEvent(st2_, es.ev_h, 0));
      }
      if (own_gram) gemm_nn(G, J, Uout, J, nullptr, 0, es.GBd, J, Ji, cols, Ji, 1.0, 0.0, nullptr, 0, st2_);
      hipLaunchKernelGGL(k_tn_small, dim3((cols * cols + 15) / 16), dim3(1024), 0, st2_, Uout, es.GBd, J, cols,
                         es.Hd, (double *)nullptr);
      hipLaunchKernelGGL(k_sub_residual, dim3(nblk), dim3(256), sizeof(double) * ((size_t)cols * cols + 17 + 128),
                         st2_, Uout, es.GBd, J, cols, es.Hd, rows_per, (const double *)nullptr,
                         (const double *)nullptr, -1, (double *)nullptr, Uout2, es.chkd, es.chkd + kEigOffResp,
                         host_chk);
      HIP_CHECK(hipEventRecord(es.ev_chk, st2_));
      const int nthr_j = std::min(1024, std::max(192, (cols * cols / 2 + 63) / 64 * 64) + 64);
      hipLaunchKernelGGL(k_rr_small, dim3(1), dim3(nthr_j), top_eig_small_lds(cols) + sizeof(int) * 64, st2_,
                         es.Hd, cols, es.Yd, es.evd, es.ev_pinned);
      HIP_CHECK(hipEventRecord(es.ev_done, st2_));
      return;
    }
    gemm_nn(X, J, Omega, J, Omega, J, Z, J, Ji, cols, Ji, 0.5, 0.5);
    double *src = Z, *dst = Z2;
    for (int rd = 0; rd < rounds; rd++) {
      const int mm = rd == 0 ? m : 0;  // (the deflated columns are in place after the first round)
      hipLaunchKernelGGL(k_tn_two, dim3((cols * cols + mm * cols + 15) / 16), dim3(1024), 0, st_, src, QD,
                         J, cols, mm, C);
      const int nn = cols - mm;
      const size_t lds_c = sizeof(double) * (4 * (size_t)nn * nn + (size_t)std::max(mm, 1) * cols + 8);
      // (M goes into H's area, which is free until the Gram of G B)
      hipLaunchKernelGGL(k_chol_m, dim3(1), dim3(k_chol_threads(nn)), lds_c, st_, src, QD, J, cols, mm, C, H,
                         status + rd);
      gemm_nn(src, J, H, cols, nullptr, 0, dst, J, Ji, cols, cols, 1.0, 0.0);
      std::swap(src, dst);
    }
    double *B = src;  // (rounds == 1: Z2, rounds == 2: Z)
    double *GB = GZ;
    gemm_nn(G, J, B, J, nullptr, 0, GB, J, Ji, cols, Ji, 1.0, 0.0);
    double *Hd2 = nullptr;
    if (lazy) {
      // (the Jacobi on the second stream works on its own copy of H — the workspace copy is
      // overwritten by the next step — which the Gram kernel writes beside H; a Jacobi of this slot
      // that is still in flight — a step that was not accepted — must be through with the buffers
      // first)
      EigState &es = *lazy;
      lazy_prepare(es);
      if (es.jacobi_launched) HIP_CHECK(hipStreamWaitEvent(st_, es.ev_done, 0));
      Hd2 = es.Hd;
    }
    hipLaunchKernelGGL(k_tn_small, dim3((cols * cols + 15) / 16), dim3(1024), 0, st_, B, GB, J, cols, H, Hd2);
    double *resp = chk + kEigOffResp;
    if (lazy) {
      // the basis itself goes out, checked as a subspace; the Jacobi of H moves to the second stream
      EigState &es = *lazy;
      es.jacobi_launched = true;
      HIP_CHECK(hipEventRecord(es.ev_h, st_));
      HIP_CHECK(hipStreamWaitEvent(st2_, es.ev_h, 0));
      const int nthr_j = std::min(1024, std::max(192, (cols * cols / 2 + 63) / 64 * 64) + 64);
      hipLaunchKernelGGL(k_rr_small, dim3(1), dim3(nthr_j), top_eig_small_lds(cols) + sizeof(int) * 64, st2_,
                         es.Hd, cols, es.Yd, es.evd, es.ev_pinned);
      HIP_CHECK(hipEventRecord(es.ev_done, st2_));
      hipLaunchKernelGGL(k_sub_residual, dim3(nblk), dim3(256),
                         sizeof(double) * ((size_t)cols * cols + 17 + 128), st_, B, GB, J, cols, H, rows_per,
                         pe2, ptr_, np, Uout, Uout2, chk, resp);
      return;
    }
    // (two elements of H per thread + the wave that prepares the next Jacobi round's angles)
    const int nthr_rr = std::min(1024, std::max(192, (cols * cols / 2 + 63) / 64 * 64) + 64);
    hipLaunchKernelGGL(k_rr_apply, dim3(nblk), dim3(nthr_rr), top_eig_small_lds(cols) + sizeof(int) * 64, st_, B,
                       GB, J, cols, rank, H, rows_per, pe2, ptr_, np, Uout, Uout2, evW, chk, resp);
  }
  // The tail of a projector step by plain launches, for any number of columns up to kEigEvMax
  // (PPALS_EIG_FUSED=0, block deflation, and every step of a core rank above 48: rank + 16 columns
  // no longer fit the one-workgroup kernels of the fused tail): Z = (Omega + X Omega) / 2 | the
  // deflated directions projected out and put in front | orthonormal basis (Cholesky QR, block
  // Gram-Schmidt above 64 columns) | Rayleigh-Ritz | residual of the leading `rank` pairs -> chk[4].
  void plain_tail(const double *G, const double *X, int64_t J, int cols, int rank, const double *Omega,
                  const double *QD, int m, double *Ot, double *Z, double *Z2, double *GZ, double *Ut, double *GU,
                  double *C, double *H, double *Yr, double *Uout, double *evW, double *chk, int *status,
                  int npass) {
    const int Ji = (int)J;
    transpose2d(Omega, F64, J, cols, Ot);  // Omega^T (cols x J): coalesced B operand
    gemm_nt(X, J, Ot, cols, Omega, J, Z, J, Ji, cols, Ji, 0.5, 0.5);
    if (m > 0) {
      double *Zr = Z + (size_t)J * m;
      const int nz = cols - m;
      for (int pass = 0; pass < 2; pass++) {
        hipLaunchKernelGGL(k_tn_rect, dim3((m * nz + 15) / 16), dim3(1024), 0, st_, QD, m, Zr, nz, J, H);
        hipLaunchKernelGGL(k_sub_mult, dim3(grid_for((int64_t)J * nz, 256)), dim3(256), 0, st_, Zr, J, nz, QD,
                           m, H);
      }
      HIP_CHECK(hipMemcpyAsync(Z, QD, sizeof(double) * J * m, hipMemcpyDeviceToDevice, st_));
    }
    // (P Omega is within a HOOI sweep's change of orthonormal: one pass, verified by status;
    // the unit vectors of the wide tail are not: two)
    double *B = chol_qr2(Z, Z2, J, cols, C, status, npass);
    rayleigh_ritz(G, B, J, cols, Ut, GZ, H, Yr, Uout, evW, GU);
    hipLaunchKernelGGL(k_eig_residual, dim3(1), dim3(1024), 0, st_, GU, Uout, evW, J, rank, chk + 4);
  }
  // One projector step from the state of the slot; false: not accepted (the caller falls back).
  // strict: the state is a rough estimate (cold start) — accept only residuals at the rounding
  // floor eps * lambda_1, whatever the estimated gap says.
  bool projector_step(EigState &es, double *G, int64_t J, int rank, double *U, int slot, bool strict) {
    // The whole step is enqueued without a read-back: what the host needs in order to schedule it
    // (where the gap lies, the scale of the shifted matrix) it knows from the previous call of the
    // slot, and everything that decides whether the result is accepted (||X^2 - I||, trace(P),
    // Cholesky pivots, eigenpair residual) is read once, at the end, together with the new
    // eigenvalues and the norm the next call will schedule with.
    const int Ji = (int)J;
    const size_t nJJ = (size_t)J * J, nJR = (size_t)J * rank;
    // (column buffers with room for kWide extra columns: the "wide tail" below)
    constexpr int kWide = 16;
    const size_t nJW = (size_t)J * (rank + kWide);
    double *w = (double *)ensure(ws_eig_, ws_eig_sz_,
                                 sizeof(double) * (3 * nJJ + 10 * nJW + 4 * kEE + 512));
    double *X = w, *Xn = X + nJJ, *Y = Xn + nJJ;
    double *Ot = Y + nJJ, *Z = Ot + nJW, *Z2 = Z + nJW, *GZ = Z2 + nJW, *Ut = GZ + nJW,
           *GU = Ut + nJW, *QD = GU + nJW, *QD2 = QD + nJW, *Om = QD2 + nJW, *Uw = Om + nJW;
    const size_t nJR_end = 10 * nJW;  // (columns of the workspace behind Y)
    // the tail of the workspace is what the one read-back fetches in a single copy:
    //   chk[16] (0,1: sign check, 4: residual, 8: ||.||_F^2) | evW[64] | status[8 ints] | lamD[64]
    double *C = Ot + nJR_end, *H = C + kEE, *Yr = H + kEE, *chk = Yr + kEE,
           *evW = chk + 16;
    int *status = (int *)(chk + kEigOffStatus);
    double *lamD = chk + kEigOffLamD;
    constexpr size_t kReadback = kEigReadback;
    double sigma = 0.5 * (es.lamR + es.lamR1);
    if (eig_sigma_scale_ > 0) sigma = eig_sigma_scale_ * es.lamR1;  // tests: a shift that is too low
    // ---- dominant eigenpairs (a relative gap >= 20 above the rest of the wanted ones): refined
    // to machine precision by a few block power steps from the previous basis, then deflated —
    // the sign iteration amplifies rounding by (spectral radius / gap), and a tensor with a mean
    // component has lambda_1 ~ 1e6 x the rest
    int m = 0;
    double ratio = 1.0;
    for (int d = 1; d < rank; d++)
      if (es.evh[d] > 0 && es.evh[d - 1] / es.evh[d] >= 20.0) {
        m = d;
        ratio = es.evh[d] / es.evh[d - 1];
      }
    double tau = 0;
    // the step's status words: cleared by the kernel that sets up the iteration (k_ns_prepare /
    // k_deflate_shift) unless block deflation writes some of them before that (m > 1)
    int *zero8 = m <= 1 ? status : nullptr;
    if (!zero8) HIP_CHECK(hipMemsetAsync(status, 0, 8 * sizeof(int), st_));
    // Fused form of a warm step (PPALS_EIG_FUSED=0: the round-2 launch sequence, kept for A/B and as
    // the route of block deflation / cold starts): the scale of the iteration comes from what the
    // slot knows about the spectrum instead of a Frobenius norm measured on the device (G is a Gram
    // matrix: every eigenvalue of G - sigma I lies in [-sigma, lambda_max - sigma], so
    // max(sigma, lambda_next - sigma) bounds the spectral radius of the deflated, shifted matrix —
    // an order of magnitude below its Frobenius norm for a flat bulk of a few hundred eigenvalues,
    // i.e. 2-3 iterations fewer). Head room (EigState::head): 25 % in steady state, the square of
    // the growth the slot's leading eigenvalue showed over its last step otherwise, 4 x right after a
    // cold start (the first HOOI sweeps move the spectrum by factors); beyond it the iteration still
    // converges up to sqrt(3) x, after which an eigenvalue is folded to the wrong side, the count
    // comes out short and the step is repeated on the measured Frobenius norm (top_eigvecs_warm).
    const bool fused_scale = !strict && m <= 1 && eig_sigma_scale_ <= 0 && !eig_frob_once_;
    const double *pow_y = nullptr, *pow_p = nullptr;
    int pow_n = 0, pow_steps = 0;
    if (m == 1) {
      // one dominant eigenpair (a tensor with a mean component): power steps on one vector. The
      // previous vector is off by at most ~1e-2 (one HOOI sweep); each step gains `ratio`. The
      // Rayleigh quotient left by the last step is that of its input: second order in its error.
      // (k steps leave an error of 1e-2 * ratio^k in the vector and its square in the quotient: the
      // target 1e-14 needs ratio^k <= 1e-12 — two steps at the 1e-6 of a tensor with a mean component,
      // where counting from 1e-14 itself took three; a vector that was further off shows in the
      // residual of the step, which is then repeated cold)
      // (round 4: the 1e-2 is an assumption only until the slot has MEASURED how far its dominant vector
      // moved from one step to the next — k_ns_prepare leaves the sine in the step's check block —: the
      // next step starts from 8 x that, so a converging HOOI run drops from three launches to two)
      double start = 1e-2;
      if (es.dom_move >= 0) start = std::min(1e-2, std::max(1e-8, 8.0 * es.dom_move));
      const int nsteps = std::min(14, std::max(2, (int)std::ceil(std::log(1e-14 / start) / std::log(ratio))));
      pow_steps = nsteps;
      const int nb = (int)((J + 7) / 8);
      double *pw = (double *)ensure(ws_pow_, ws_pow_sz_, sizeof(double) * 4 * (size_t)nb);
      double *pbuf[2] = {pw, pw + 2 * (size_t)nb};
      double *ybuf[2] = {QD2, Z2};  // (Z2 is free until the Cholesky QR of the projected basis)
      const double *src = es.Q;
      const double *pin = nullptr;
      for (int it = 0; it < nsteps; it++) {
        hipLaunchKernelGGL(k_power_mv, dim3(nb), dim3(256), 0, st_, G, J, src, pin, nb, ybuf[it & 1],
                           pbuf[it & 1]);
        src = ybuf[it & 1];
        pin = pbuf[it & 1];
      }
      pow_y = src;
      pow_p = pin;
      pow_n = nb;
      tau = es.lamR * 1.5 + 1e-300;
      if (!fused_scale)
        hipLaunchKernelGGL(k_power_finish, dim3(1), dim3(256), 0, st_, src, J, pin, nb, QD, lamD);
    } else if (m > 0) {
      const int nsteps = std::min(14, std::max(3, (int)std::ceil(std::log(1e-18) / std::log(ratio))));
      HIP_CHECK(hipMemcpyAsync(QD, es.Q, sizeof(double) * J * m, hipMemcpyDeviceToDevice, st_));
      double *cur = QD;
      for (int it = 0; it < nsteps; it++) {
        double *other = (cur == QD) ? QD2 : QD;
        transpose2d(cur, F64, J, m, Ot);
        gemm_nt(G, J, Ot, m, nullptr, 0, Z, J, Ji, m, Ji, 1.0, 0.0);
        double *res = chol_qr2(Z, other, J, m, C, status + 2);  // ends in Z (two passes)
        if (res != cur) HIP_CHECK(hipMemcpyAsync(cur, res, sizeof(double) * J * m,
                                                 hipMemcpyDeviceToDevice, st_));
      }
      rayleigh_ritz(G, cur, J, m, Ot, Z, H, Yr, Z2, lamD, nullptr);
      HIP_CHECK(hipMemcpyAsync(QD, Z2, sizeof(double) * J * m, hipMemcpyDeviceToDevice, st_));
      tau = es.lamR * 1.5 + 1e-300;  // still above sigma: they keep counting as "wanted"
    }
    // X = (G - deflation - sigma I) / rho, rho = its Frobenius norm >= every |eigenvalue| (safe:
    // an eigenvalue of the scaled matrix beyond 1 would be folded back by the scaled iteration).
    // The norm stays on the device; the host schedules the iteration with the norm of the slot's
    // previous call (the matrices of consecutive HOOI sweeps differ by a few per cent at most, and
    // an under-estimated lower bound `ell` only costs a fraction of an iteration).
    const int gdef = grid_for(nJJ, 256, 1024);
    double *part = (double *)ensure(ws_part_, ws_part_sz_, sizeof(double) * (gdef + 1));
    double *fro2_d = chk + 8;
    double rho = es.rho_frob;
    if (fused_scale) {
      const double next = m < rank ? es.evh[m] : 0.0;  // largest eigenvalue that is not deflated
      rho = es.head * std::max(sigma, next - sigma);
      if (!(rho > 0) || !std::isfinite(rho)) return false;
      hipLaunchKernelGGL(k_ns_prepare, dim3(gdef), dim3(256), 0, st_, G, J, pow_y, pow_p, pow_n, m, tau,
                         sigma, 1.0 / rho, X, QD, lamD, zero8, m == 1 ? (const double *)es.Q : (const double *)nullptr,
                         m == 1 ? chk + 9 : (double *)nullptr);
    } else {
      hipLaunchKernelGGL(k_deflate_shift, dim3(gdef), dim3(256), 0, st_, G, J, QD, m, lamD, tau, sigma,
                         X, part, zero8);
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, gdef, fro2_d);
      if (!(rho > 0)) {  // first projector step of the slot: read the norm
        double fro2 = 0;
        HIP_CHECK(hipMemcpyAsync(&fro2, fro2_d, sizeof(double), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
        rho = 1.0001 * std::sqrt(fro2);
        if (!(rho > 0) || !std::isfinite(rho)) return false;
      }
      hipLaunchKernelGGL(k_scale_by_frob, dim3(grid_for(nJJ, 256)), dim3(256), 0, st_, X, (int64_t)nJJ,
                         fro2_d);
    }
    // (strict = cold start: the distance of the shift to the nearest eigenvalue is unknown — the
    // Ritz values around it are rough — so a small lower bound is assumed: a few iterations more)
    double ell0 = 0.9 * 0.5 * std::min(es.lamR - sigma, sigma - es.lamR1) / rho;
    if (strict) ell0 = std::min(ell0, 2e-5);
    double ell = std::max(ell0, eig_sigma_scale_ > 0 ? 1e-4 : 1e-14);
    int iters = 0;
    const unsigned ntri = sym_tiles(Ji);
    // fused tail: the check sums ride on the LAST step's two products (per-tile partial sums of
    // ||X_prev^2 - I||_F^2 and of trace(X_new), added up by the tail's last kernel)
    const bool fused_tail = m <= 1 && rank + kWide <= 64;
    double *pe2 = (double *)ensure(ws_part2_, ws_part2_sz_, sizeof(double) * (2 * (size_t)ntri + 2 * 256 + 2));
    double *ptr_ = pe2 + ntri;
    auto ns_step = [&](double mu, bool last = false) {
      // Y = X^2, X <- X (a I + b X^2): symmetric products, stored symmetric bit for bit
      sym_product(X, J, X, J, nullptr, 0, Y, J, Ji, Ji, 1.0, 0.0, last ? 1 : 0, pe2);
      sym_product(X, J, Y, J, X, J, Xn, J, Ji, Ji, -0.5 * mu * mu * mu, 1.5 * mu, last ? 2 : 0, ptr_);
      std::swap(X, Xn);
      iters++;
    };
    while (1.0 - ell > 1e-3 && iters < 80) {
      const double mu = std::sqrt(3.0 / (1.0 + ell + ell * ell));
      ns_step(mu);
      ell = 0.5 * mu * ell * (3.0 - mu * mu * ell * ell);
    }
    ns_step(1.0);
    ns_step(1.0, fused_tail);
    for (int attempt = 0;; attempt++) {
      // ---- check of the sign iteration: ||X^2 - I||_F^2 and trace(X) (-> chk[0], chk[1])
      if (!fused_tail) {
        sym_product(X, J, X, J, nullptr, 0, Y, J, Ji, Ji, 1.0, 0.0);
        const int gc = 256;
        double *pc = ptr_ + ntri;
        hipLaunchKernelGGL(k_sign_check, dim3(gc), dim3(256), 0, st_, Y, X, J, pc);
        hipLaunchKernelGGL(k_sum_pairs, dim3(1), dim3(256), 0, st_, pc, gc, chk);
      }
      // ---- Z = P * Omega = (Omega + X Omega) / 2 with the previous basis, Cholesky-QR twice, then
      // Rayleigh-Ritz inside the subspace: the eigenvectors one by one, sorted descending
      // (the deflated directions come from the power iteration, accurate to machine precision: the
      // projector of the DEFLATED matrix carries them only to eps * lambda_1 / gap. So the basis is
      // [Q_D | (I - Q_D Q_D^T) P Omega_rest].)
      // `tail(cols, Omega, Uout)`: basis of `cols` columns = P applied to Omega (J x cols), Cholesky
      // QR, Rayleigh-Ritz on cols x cols; the leading `rank` eigenvectors land in Uout (ld J), their
      // residual in chk[4], all `cols` eigenvalues in evW; then the one read-back of the step
      bool lazy_used = false;
      // deferred acceptance: a slot whose last steps went through as scheduled hands out the basis
      // without waiting for its checks (Ops::eig_defer / eig_verify)
      const bool defer_now = attempt == 0 && es.lazy && es.defer && es.stable >= 2 && fused_tail &&
                             fused_scale && !strict && !eig_frob_once_;
      auto tail = [&](int cols, const double *Omega, double *Uout, int npass) {
        if (fused_tail) {
          // (the leading eigenvectors also go to the slot's spare basis buffer: accepted = a swap)
          const bool lazy_now = es.lazy && cols == rank && Uout == U && npass == 1;
          lazy_used = lazy_now;
          fused_tail_launches(G, X, J, cols, rank, Omega, QD, m, Z, Z2, GZ, C, H, Uout, evW, chk, status,
                              pe2, ptr_, (int)ntri, npass, Uout == U ? es.Qn : nullptr,
                              lazy_now ? &es : nullptr, (defer_now && lazy_now) ? es.chk_pinned : nullptr);
          if (defer_now && lazy_now) {
            // no wait and no copy: the checks are finished on the second stream, whose kernel writes
            // them into the slot's pinned block (ev_chk); eig_verify() reads them when the caller
            // comes back to the slot
            HIP_CHECK(hipGetLastError());
            return;
          }
          if (!eig_host_) HIP_CHECK(hipHostMalloc(&eig_host_, kReadback, hipHostMallocDefault));
          HIP_CHECK(hipMemcpyAsync(eig_host_, chk, kReadback, hipMemcpyDeviceToHost, st_));
          HIP_CHECK(hipStreamSynchronize(st_));
          HIP_CHECK(hipGetLastError());
          {  // the residual's per-workgroup shares are added up here
            double *hcw = (double *)eig_host_;
            const double *rp = hcw + kEigOffResp;
            double r2 = 0;
            for (int b2 = 0; b2 < kTailBlocks; b2++) r2 += rp[b2];
            hcw[4] = r2;
          }
          return;
        }
        plain_tail(G, X, J, cols, rank, Omega, QD, m, Ot, Z, Z2, GZ, Ut, GU, C, H, Yr, Uout, evW, chk, status,
                   npass);
        if (!eig_host_) HIP_CHECK(hipHostMalloc(&eig_host_, kReadback, hipHostMallocDefault));
        HIP_CHECK(hipMemcpyAsync(eig_host_, chk, kReadback, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipGetLastError());
      };
      tail(rank, es.Q, U, 1);
      if (defer_now && lazy_used) {
        es.deferred = true;
        es.n_deferred++;
        es.dp.sigma = sigma;
        es.dp.rho = rho;
        es.dp.ell0 = ell0;
        es.dp.m = m;
        es.dp.iters = iters;
        es.dp.pow = pow_steps;
        std::swap(es.Q, es.Qn);  // (eig_verify swaps back when the step is not accepted)
        return true;
      }
      const double *hc = (const double *)eig_host_, *evn = hc + 16;
      const int *hs = (const int *)(hc + kEigOffStatus);
      const StepCheck sc = check_step(hc, J, rank, sigma, rho, strict, fused_tail, fused_scale, lazy_used);
      const double cnt = sc.cnt, res = sc.res, rho_now = sc.rho_now, gap_now = sc.gap_now, res_tol = sc.res_tol;
      const bool converged = sc.converged, good = sc.good;
      if (eig_debug_)
        fprintf(stderr, "[ppals eig] slot %d J %lld rank %d: lamR %.6e lamR1 %.6e deflated %d rho %.3e "
                        "(now %.3e) ell0 %.2e iters %d | ||X^2-I||^2 %.3e count %.6f residual %.3e "
                        "(gap %.3e) chol %d%d%d%d -> %s (fast %d full %d)\n",
                slot, (long long)J, rank, es.lamR, es.lamR1, m, rho, rho_now, ell0, iters, hc[0], cnt,
                res, gap_now, hs[0], hs[1], hs[2], hs[3],
                good ? "accepted" : (converged ? "not accepted as it is" : "more steps"), es.fast, es.full);
      es.stable = (good && attempt == 0) ? es.stable + 1 : 0;
      if (good && lazy_used) {
        // the eigenvalues arrive with the second stream's Jacobi: resolve_lazy() turns them into
        // the slot's state when the slot is used next; U holds the basis, Y is owed to the caller
        es.lazy_pending = true;
        es.rot_pending = true;
        es.lazy_m = m;
        es.lazy_strict = strict;
        es.rho_frob = rho_now;
        es.dom_move = (m == 1 && fused_scale) ? std::sqrt(std::max(0.0, hc[9])) : -1.0;
        std::swap(es.Q, es.Qn);
        es.fast++;
        return true;
      }
      if (lazy_used) {
        // not accepted: the Jacobi on the second stream is of no use (its buffers are the slot's own)
        lazy_used = false;
      }
      if (!good) es.dom_move = -1;
      if (good) {
        {
          const int mi = std::min(m, rank - 1);
          const double growth = es.evh[mi] > 0 ? evn[mi] / es.evh[mi] : 2.0;
          es.head = std::min(8.0, std::max(1.25, 1.25 * growth * growth));
        }
        for (int d = 0; d < rank; d++) es.evh[d] = evn[d];
        const double shift = evn[rank - 1] - es.lamR;
        es.lamR = evn[rank - 1];
        es.lamR1 = std::min(std::max(0.0, es.lamR1 + shift), es.lamR);
        es.rho_frob = rho_now;
        es.dom_move = (m == 1 && fused_scale) ? std::sqrt(std::max(0.0, hc[9])) : -1.0;
        if (strict) es.lamR1 = std::min(es.lamR1, es.lamR * (1 - 1e-6));  // (a rough estimate)
        if (!(es.lamR > es.lamR1 * (1 + 1e-9))) es.valid = false;  // next call: full solver
        if (fused_tail)
          std::swap(es.Q, es.Qn);
        else
          HIP_CHECK(hipMemcpyAsync(es.Q, U, sizeof(double) * nJR, hipMemcpyDeviceToDevice, st_));
        es.fast++;
        return true;
      }
      const int cwide = (int)std::lround(cnt);
      if (converged && std::fabs(cnt - cwide) < 1e-6 && cwide > rank && cwide <= rank + kWide &&
          cwide <= kEigEvMax && cwide < Ji && hs[2] != 1 && hs[3] == 0 && std::isfinite(rho_now)) {
        // The shift fell a few eigenvalues too low: P projects onto an invariant subspace of
        // cwide > rank dimensions — still exact. Rayleigh-Ritz on a basis of THAT many columns
        // (the previous basis plus P applied to a few unit vectors) yields its eigenpairs one by
        // one, the leading `rank` of them are the answer, and the next one is the new estimate of
        // the eigenvalue below the cut. A 0.2 ms tail instead of the 9 ms full solver.
        HIP_CHECK(hipMemcpyAsync(Om, es.Q, sizeof(double) * nJR, hipMemcpyDeviceToDevice, st_));
        HIP_CHECK(hipMemsetAsync(Om + nJR, 0, sizeof(double) * J * (cwide - rank), st_));
        hipLaunchKernelGGL(k_set_unit_cols, dim3(1), dim3(64), 0, st_, Om + nJR, J, cwide - rank);
        HIP_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int), st_));
        tail(cwide, Om, Uw, 2);
        const double resw = std::sqrt(hc[4]);
        const bool goodw = hs[0] != 1 && hs[1] == 0 && resw <= res_tol &&
                           evn[rank - 1] > evn[rank] * (1 + 1e-9);
        if (eig_debug_)
          fprintf(stderr, "[ppals eig]   wide tail with %d columns: residual %.3e chol %d%d -> %s\n",
                  cwide, resw, hs[0], hs[1], goodw ? "accepted" : "full solver");
        if (goodw) {
          HIP_CHECK(hipMemcpyAsync(U, Uw, sizeof(double) * nJR, hipMemcpyDeviceToDevice, st_));
          es.head = 4.0;  // (the estimates were off by several eigenvalues: the spectrum is moving)
          for (int d = 0; d < rank; d++) es.evh[d] = evn[d];
          es.lamR = evn[rank - 1];
          es.lamR1 = std::max(0.0, evn[rank]);
          es.rho_frob = rho_now;
          HIP_CHECK(hipMemcpyAsync(es.Q, U, sizeof(double) * nJR, hipMemcpyDeviceToDevice, st_));
          es.fast++;
          return true;
        }
        break;
      }
      if (converged || attempt >= (strict ? 4 : 2)) break;
      // the gap was narrower than estimated: a few more steps, then the tail again
      if (strict) {
        ns_step(1.0);
        ns_step(1.0);
      }
      ns_step(1.0);
      ns_step(1.0, fused_tail);
      HIP_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int), st_));  // (the Cholesky QR of the tail)
    }
    // sigma does not separate `rank` eigenvalues, a lost direction, no convergence
    return false;
  }
  // Cold start of a slot: q steps of block subspace iteration from a pseudo-random block of
  // rank + 16 columns, Rayleigh-Ritz — rough Ritz pairs (lower bounds of the eigenvalues). They
  // become the slot's state with a deliberately LOW estimate of the eigenvalue below the cut, so
  // that the shift of the projector step has at least `rank` eigenvalues above it; a few more are
  // what its wide tail is for. false: the Ritz values do not resolve the neighbourhood of the cut
  // (a flat spectrum) or the block lost rank — the caller goes to the full solver.
  bool cold_ritz_state(EigState &es, double *G, int64_t J, int rank, int slot) {
    cold_ok_ = false;
    const int b = std::min(kEigEvMax, rank + 16);
    if (b >= J || rank + 4 >= b) return false;
    const int Ji = (int)J;
    const size_t nJB = (size_t)J * b;
    double *w = (double *)ensure(ws_cold_, ws_cold_sz_, sizeof(double) * (5 * nJB + 3 * kEE + kEigEvMax + 64));
    double *Z = w, *Z2 = Z + nJB, *Zt = Z2 + nJB, *GB = Zt + nJB, *Uo = GB + nJB;
    double *C = Uo + nJB, *H = C + kEE, *Yr = H + kEE, *ev = Yr + kEE;
    int *status = (int *)(ev + kEigEvMax);
    HIP_CHECK(hipMemsetAsync(status, 0, 4 * sizeof(int), st_));
    hipLaunchKernelGGL(k_fill_hash, dim3(grid_for((int64_t)nJB, 256)), dim3(256), 0, st_, Z, (int64_t)nJB,
                       (uint64_t)(0x5eed + slot));
    double *cur = Z, *oth = Z2;
    for (int it = 0; it < 4; it++) {
      gemm_nn(G, J, cur, J, nullptr, 0, oth, J, Ji, b, Ji, 1.0, 0.0);  // G * block
      if (b <= 64) {
        // Cholesky QR twice: Gram (one wave per entry) | M = R^-1 by one workgroup | thin product
        double *src = oth, *dst = cur;
        for (int pass = 0; pass < 2; pass++) {
          hipLaunchKernelGGL(k_tn_two, dim3((b * b + 15) / 16), dim3(1024), 0, st_, src, (const double *)nullptr,
                             J, b, 0, C);
          hipLaunchKernelGGL(k_chol_m, dim3(1), dim3(k_chol_threads(b)), sizeof(double) * (4 * (size_t)b * b + b + 8), st_,
                             src, (const double *)nullptr, J, b, 0, C, H, status + pass);
          gemm_nn(src, J, H, b, nullptr, 0, dst, J, Ji, b, b, 1.0, 0.0);
          std::swap(src, dst);
        }
        // (two passes: the result is back in `oth`)
      } else {
        double *res = chol_qr2(oth, cur, J, b, C, status);  // two passes: ends in `oth`
        if (res != oth) return false;
      }
      std::swap(cur, oth);
    }
    rayleigh_ritz(G, cur, J, b, Zt, GB, H, Yr, Uo, ev, nullptr);
    double th[kEigEvMax];
    int hs[4];
    HIP_CHECK(hipMemcpyAsync(th, ev, sizeof(double) * b, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipMemcpyAsync(hs, status, sizeof(int) * 4, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    HIP_CHECK(hipGetLastError());
    const int below = std::min(b - 1, rank + 4);  // index of the estimate placed under the cut
    const bool sane = hs[0] != 1 && hs[1] == 0 && th[rank - 1] > 0 && th[below] >= 0;
    const bool ok = sane && th[rank - 1] > th[below] * 1.02;
    // (kept for cold_bisect: the Ritz block is a generic basis, its Ritz values are lower bounds)
    cold_ok_ = sane;
    cold_b_ = b;
    cold_Uo_ = Uo;
    for (int d = 0; d < b; d++) cold_th_[d] = th[d];
    if (eig_debug_)
      fprintf(stderr, "[ppals eig] slot %d J %lld rank %d: cold start, Ritz values %.4e .. %.4e | %.4e "
                      "(index %d) -> %s\n", slot, (long long)J, rank, th[0], th[rank - 1], th[below],
              below, ok ? "projector step" : "full solver");
    if (!ok) return false;
    if (!es.Q || es.J != J || es.rank != rank) {
      if (es.Q) hipFree(es.Q);
      if (es.Qn) hipFree(es.Qn);
      HIP_CHECK(hipMalloc(&es.Q, sizeof(double) * J * rank));
      HIP_CHECK(hipMalloc(&es.Qn, sizeof(double) * J * rank));
    }
    es.J = J;
    es.rank = rank;
    HIP_CHECK(hipMemcpyAsync(es.Q, Uo, sizeof(double) * J * rank, hipMemcpyDeviceToDevice, st_));
    for (int d = 0; d < rank; d++) es.evh[d] = th[d];
    es.lamR = th[rank - 1];
    es.lamR1 = th[below];
    es.rho = th[0];
    es.rho_frob = 0;
    es.head = 4.0;
    es.valid = true;
    return true;
  }

  // Cold start of a LONG mode by block subspace iteration with Rayleigh-Ritz in every step (round 6).
  // The projector route costs ~46 products of J^3 from cold — 8.9 ms each at J = 7200, the image mode of
  // coil-100 (test_ALS.cxx:296-299): 408 of the 516 ms of that HOSVD — to deliver `rank` vectors;
  // a step here costs two THIN products (G times J x (rank + 16)) and a Rayleigh-Ritz on rank + 16
  // columns. From the Ritz pairs (U, theta) of cold_ritz_state: B = orth(G U theta^-1) — scaled column
  // by column, so the block stays near orthonormal whatever the spread of the spectrum (a mean
  // component 1e6 above the rest included) —, Rayleigh-Ritz on B, residual of the leading `rank` pairs;
  // accepted at the warm steps' bar (residual <= 1e-9 x the Ritz gap below the rank-th value, or the
  // rounding floor). The iteration converges like (lambda_{rank+17} / lambda_rank)^k: where the residual
  // shows that this will not get there within the budget, false — the caller goes on to the projector.
  bool cold_subspace(EigState &es, double *G, int64_t J, int rank, double *U, int slot) {
    const int b = cold_b_, Ji = (int)J;
    if (b <= rank || !cold_Uo_) return false;
    const size_t nJB = (size_t)J * b;
    double *w = (double *)ensure(ws_cold_, ws_cold_sz_, sizeof(double) * (5 * nJB + 3 * kEE + kEigEvMax + 64));
    double *Z = w, *Z2 = Z + nJB, *Zt = Z2 + nJB, *GB = Zt + nJB, *Uo = GB + nJB;
    double *C = Uo + nJB, *H = C + kEE, *Yr = H + kEE, *ev = Yr + kEE;
    int *status = (int *)(ev + kEigEvMax);
    if (Uo != cold_Uo_) return false;  // (the workspace moved: nothing to continue from)
    double *w2 = (double *)ensure(ws_cold2_, ws_cold2_sz_, sizeof(double) * (nJB + 16));
    double *GU = w2, *res_d = GU + nJB;
    constexpr int kMaxIt = 24;
    double th[kEigEvMax], res_prev = 0;
    for (int it = 0; it < kMaxIt; it++) {
      gemm_nn(G, J, Uo, J, nullptr, 0, Z, J, Ji, b, Ji, 1.0, 0.0);  // G * (Ritz vectors)
      hipLaunchKernelGGL(k_scale_cols_inv, dim3(grid_for((int64_t)nJB, 256)), dim3(256), 0, st_, Z, J, b,
                         (const double *)ev);
      HIP_CHECK(hipMemsetAsync(status, 0, 4 * sizeof(int), st_));
      double *B = chol_qr2(Z, Z2, J, b, C, status);
      rayleigh_ritz(G, B, J, b, Zt, GB, H, Yr, Uo, ev, GU);
      hipLaunchKernelGGL(k_eig_residual, dim3(1), dim3(1024), 0, st_, GU, Uo, ev, J, rank, res_d);
      double res2 = 0;
      int hs[4];
      HIP_CHECK(hipMemcpyAsync(th, ev, sizeof(double) * b, hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipMemcpyAsync(&res2, res_d, sizeof(double), hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipMemcpyAsync(hs, status, sizeof(int) * 4, hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipStreamSynchronize(st_));
      HIP_CHECK(hipGetLastError());
      const double res = std::sqrt(res2);
      const bool sane = hs[0] != 1 && hs[1] == 0 && std::isfinite(res) && th[rank - 1] > 0 && th[rank] >= 0;
      const double gap = sane ? th[rank - 1] - th[rank] : 0.0;
      const double tol = std::max(1e-9 * gap, 1e-13 * th[0]) * std::sqrt((double)rank);
      if (eig_debug_)
        fprintf(stderr, "[ppals eig] slot %d J %lld rank %d: cold subspace step %d, Ritz %.4e .. %.4e | %.4e, "
                        "residual %.3e (bar %.3e)\n", slot, (long long)J, rank, it, th[0], th[rank - 1], th[rank],
                res, tol);
      if (!sane) return false;
      if (gap > 0 && res <= tol) {
        const int below = std::min(b - 1, rank + 4);
        if (!es.Q || es.J != J || es.rank != rank) {
          if (es.Q) hipFree(es.Q);
          if (es.Qn) hipFree(es.Qn);
          HIP_CHECK(hipMalloc(&es.Q, sizeof(double) * J * rank));
          HIP_CHECK(hipMalloc(&es.Qn, sizeof(double) * J * rank));
        }
        es.J = J;
        es.rank = rank;
        HIP_CHECK(hipMemcpyAsync(es.Q, Uo, sizeof(double) * J * rank, hipMemcpyDeviceToDevice, st_));
        HIP_CHECK(hipMemcpyAsync(U, Uo, sizeof(double) * J * rank, hipMemcpyDeviceToDevice, st_));
        for (int d = 0; d < rank; d++) es.evh[d] = th[d];
        es.lamR = th[rank - 1];
        es.lamR1 = th[below];
        es.rho = th[0];
        es.rho_frob = 0;
        es.head = 4.0;
        es.valid = es.lamR > es.lamR1 * (1 + 1e-9);
        es.fast++;
        n_cold_subspace_++;
        return true;
      }
      // the residual of a linearly converging iteration: where will it be when the budget ends?
      if (it >= 2 && res_prev > 0) {
        const double f = res / res_prev;
        if (!(f < 0.97) || res * std::pow(f, kMaxIt - 1 - it) > tol) return false;
      }
      res_prev = res;
    }
    return false;
  }
  int64_t n_cold_subspace_ = 0;
  void *ws_cold2_ = nullptr;
  size_t ws_cold2_sz_ = 0;
  void *ws_cholm_ = nullptr;
  size_t ws_cholm_sz_ = 0;
  void *ws_gsfb_ = nullptr;
  size_t ws_gsfb_sz_ = 0;

  // Cold start by eigenvalue counting. trace(sign(G - sigma I)) says how many eigenvalues lie above
  // sigma — exactly, whatever the spectrum looks like — so the shift is placed by a few trials of
  // the sign iteration instead of by Ritz values: a bracket [lo, hi] with count(lo) >= rank (the
  // rank-th Ritz value is a lower bound of the rank-th eigenvalue) and count(hi) < rank, then
  // interpolation on count^(2/3) (the edge law of a bulk spectrum) until rank <= count <=
  // rank + kWide. The projector of that trial is exact for an invariant subspace of `count`
  // dimensions; Rayleigh-Ritz on P applied to the Ritz block delivers its eigenpairs one by one and
  // the leading `rank` are the answer (the same tail as a warm step's "wide tail"). A trial costs
  // ~0.25 ms (28 products of J^3), the full solver 9 ms at J = 400. One dominant eigenpair (a mean
  // component) is refined by power steps and deflated; anything this cannot handle returns false.
  bool cold_bisect(EigState &es, double *G, int64_t J, int rank, double *U, int slot) {
    constexpr int kWide = 16;
    const int Ji = (int)J, b = cold_b_;
    const size_t nJJ = (size_t)J * J, nJR = (size_t)J * rank, nJW = (size_t)J * (rank + kWide);
    if (rank + kWide > b || rank + kWide >= Ji) return false;
    const bool big = rank + kWide > 64;  // (the counting is the same; the tail is made of plain launches)
    const double *th = cold_th_;
    int m = 0;
    for (int d = 1; d < rank; d++)
      if (th[d] > 0 && th[d - 1] / th[d] >= 20.0) m = d;
    if (m > 1) return false;
    const unsigned ntri = sym_tiles(Ji);
    double *w = (double *)ensure(ws_eig_, ws_eig_sz_,
                                 sizeof(double) * (3 * nJJ + 10 * nJW + 4 * kEE + 512));
    double *X = w, *Xn = X + nJJ, *Y = Xn + nJJ;
    double *Z = Y + nJJ, *Z2 = Z + nJW, *GZ = Z2 + nJW, *QD = GZ + nJW, *y0 = QD + nJW, *y1 = y0 + nJW,
           *Uw = y1 + nJW, *Ot = Uw + nJW, *Ut = Ot + nJW, *GU = Ut + nJW;
    double *Cw = Y + nJJ + 10 * nJW, *Hw = Cw + kEE;
    double *chk = Y + nJJ + 10 * nJW + 3 * kEE, *evW = chk + 16;
    int *status = (int *)(chk + kEigOffStatus);
    double *lamD = chk + kEigOffLamD;
    constexpr size_t kReadback = kEigReadback;
    double *pe2 = (double *)ensure(ws_part2_, ws_part2_sz_, sizeof(double) * (2 * (size_t)ntri + 2 * 256 + 2));
    double *ptr_ = pe2 + ntri;
    if (!eig_host_) HIP_CHECK(hipHostMalloc(&eig_host_, kReadback, hipHostMallocDefault));
    // the Ritz block lives in ws_cold_, which nothing below touches
    const double *Om = cold_Uo_;
    const double *pow_y = nullptr, *pow_p = nullptr;
    int pow_n = 0;
    if (m == 1) {
      const int nb = (int)((J + 7) / 8);
      double *pw = (double *)ensure(ws_pow_, ws_pow_sz_, sizeof(double) * 4 * (size_t)nb);
      double *pbuf[2] = {pw, pw + 2 * (size_t)nb};
      double *ybuf[2] = {y0, y1};
      const double *src = Om, *pin = nullptr;
      for (int it = 0; it < 4; it++) {
        hipLaunchKernelGGL(k_power_mv, dim3(nb), dim3(256), 0, st_, G, J, src, pin, nb, ybuf[it & 1],
                           pbuf[it & 1]);
        src = ybuf[it & 1];
        pin = pbuf[it & 1];
      }
      pow_y = src;
      pow_p = pin;
      pow_n = nb;
    }
    const double next_ub = 1.5 * th[m];  // generous bound of the largest eigenvalue left in place
    int trials = 0, products = 0;
    // one trial: X = sign((G - deflation - sigma I) / rho), returns count and convergence
    auto trial = [&](double sigma, double *count, bool *conv) {
      const double rho = 1.25 * std::max(sigma, next_ub - sigma);
      const int gdef = grid_for(nJJ, 256, 1024);
      hipLaunchKernelGGL(k_ns_prepare, dim3(gdef), dim3(256), 0, st_, G, J, pow_y, pow_p, pow_n, m,
                         2.0 * sigma, sigma, 1.0 / rho, X, QD, lamD);
      auto ns_step = [&](double mu, bool last) {
        sym_product(X, J, X, J, nullptr, 0, Y, J, Ji, Ji, 1.0, 0.0, last ? 1 : 0, pe2);
        sym_product(X, J, Y, J, X, J, Xn, J, Ji, Ji, -0.5 * mu * mu * mu, 1.5 * mu, last ? 2 : 0, ptr_);
        std::swap(X, Xn);
        products += 2;
      };
      double ell = 1e-5;  // assumed distance of the shift to the nearest eigenvalue (scaled)
      int it = 0;
      while (1.0 - ell > 1e-3 && it++ < 80) {
        const double mu = std::sqrt(3.0 / (1.0 + ell + ell * ell));
        ns_step(mu, false);
        ell = 0.5 * mu * ell * (3.0 - mu * mu * ell * ell);
      }
      ns_step(1.0, false);
      for (int attempt = 0;; attempt++) {
        ns_step(1.0, true);
        hipLaunchKernelGGL(k_chk_sums, dim3(1), dim3(256), 0, st_, pe2, ptr_, (int)ntri, chk);
        HIP_CHECK(hipMemcpyAsync(eig_host_, chk, 2 * sizeof(double), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
        const double *hc = (const double *)eig_host_;
        *conv = hc[0] <= 1e-10 * std::sqrt((double)J);
        *count = 0.5 * (hc[1] + (double)J);
        if (*conv || attempt >= 3) break;
        ns_step(1.0, false);  // the shift sits closer to an eigenvalue than assumed: more steps
        ns_step(1.0, false);
        ns_step(1.0, false);
      }
      trials++;
      if (eig_debug_)
        fprintf(stderr, "[ppals eig] slot %d   count trial %d: sigma %.8e -> count %.4f (%s)\n", slot,
                trials, sigma, *count, *conv ? "converged" : "NOT converged");
    };
    double lo = th[rank - 1], hi = 1.08 * th[m], c_lo = -1, c_hi = -1;
    double cnt = 0;
    bool conv = false;
    // upper end of the bracket: fewer than `rank` eigenvalues above it
    for (int k = 0; k < 6; k++) {
      trial(hi, &cnt, &conv);
      if (conv && std::lround(cnt) < rank) {
        c_hi = (double)std::lround(cnt);
        break;
      }
      if (conv && std::lround(cnt) <= rank + kWide) break;  // landed inside the window already
      if (conv) lo = hi, c_lo = (double)std::lround(cnt);
      hi *= conv ? 1.08 : 1.0003;
    }
    double sigma = hi;
    auto in_window = [&]() { return conv && std::lround(cnt) >= rank && std::lround(cnt) <= rank + kWide; };
    if (!in_window()) {
      if (c_hi < 0) return false;
      if (c_lo < 0) {  // count at the Ritz lower bound
        trial(lo, &cnt, &conv);
        if (!conv) {
          lo *= 0.9997;
          trial(lo, &cnt, &conv);
        }
        if (!conv) return false;
        c_lo = (double)std::lround(cnt);
        sigma = lo;
      }
      const double target = rank + kWide / 2;
      for (int k = 0; k < 12 && !in_window(); k++) {
        if (c_lo < rank) return false;  // the Ritz value was no lower bound: give up
        // N(sigma)^(2/3) is close to linear below a spectral edge; plain bisection every third trial
        const double tl = std::pow(c_lo, 2.0 / 3.0), thh = std::pow(std::max(c_hi, 0.0), 2.0 / 3.0),
                     tt = std::pow(target, 2.0 / 3.0);
        double f = (k % 3 == 2) ? 0.5 : (tl - tt) / std::max(tl - thh, 1e-300);
        f = std::min(0.95, std::max(0.05, f));
        sigma = lo + f * (hi - lo);
        trial(sigma, &cnt, &conv);
        if (!conv) {  // on top of an eigenvalue: nudge
          sigma = lo + std::min(0.97, f + 0.02) * (hi - lo);
          trial(sigma, &cnt, &conv);
          if (!conv) return false;
        }
        const long c = std::lround(cnt);
        if (c > rank + kWide) {
          lo = sigma;
          c_lo = (double)c;
        } else if (c < rank) {
          hi = sigma;
          c_hi = (double)c;
        }
      }
      if (!in_window()) return false;
    }
    const int cols = (int)std::lround(cnt);
    if (std::fabs(cnt - cols) > 1e-6) return false;
    // ---- the tail on `cols` columns: P applied to the Ritz block (a generic basis), two Cholesky-QR
    // passes, Rayleigh-Ritz; the deflated eigenvector takes the place of the first column
    HIP_CHECK(hipMemsetAsync(status, 0, 8 * sizeof(int), st_));
    if (big) {
      HIP_CHECK(hipMemsetAsync(chk + kEigOffResp, 0, sizeof(double) * kTailBlocks, st_));
      plain_tail(G, X, J, cols, rank, Om, QD, m, Ot, Z, Z2, GZ, Ut, GU, Cw, Hw, Hw + kEE, Uw, evW, chk, status, 2);
    } else {
      fused_tail_launches(G, X, J, cols, rank, Om, QD, m, Z, Z2, GZ, Cw, Hw, Uw, evW, chk, status, pe2, ptr_,
                          (int)ntri, 2);
    }
    HIP_CHECK(hipMemcpyAsync(eig_host_, chk, kReadback, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    HIP_CHECK(hipGetLastError());
    if (!big) {  // (the fused tail leaves per-workgroup shares of the residual; the plain one the sum)
      double *hcw = (double *)eig_host_;
      const double *rp = hcw + kEigOffResp;
      double r2 = 0;
      for (int b2 = 0; b2 < kTailBlocks; b2++) r2 += rp[b2];
      hcw[4] = r2;
    }
    const double *hc = (const double *)eig_host_, *evn = hc + 16;
    const int *hs = (const int *)(hc + kEigOffStatus);
    const double res = std::sqrt(hc[4]);
    const double res_tol = 1e-13 * evn[0] * std::sqrt((double)rank);
    const bool good = hs[0] != 1 && hs[1] == 0 && res <= res_tol && std::isfinite(res) &&
                      (cols == rank || evn[rank - 1] > evn[rank] * (1 + 1e-12));
    if (eig_debug_)
      fprintf(stderr, "[ppals eig] slot %d J %lld rank %d: cold start by counting, %d trials (%d products), "
                      "sigma %.8e count %d, residual %.3e (tol %.3e) chol %d%d -> %s\n",
              slot, (long long)J, rank, trials, products, sigma, cols, res, res_tol, hs[0], hs[1],
              good ? "accepted" : "full solver");
    if (!good) return false;
    HIP_CHECK(hipMemcpyAsync(U, Uw, sizeof(double) * nJR, hipMemcpyDeviceToDevice, st_));
    if (!es.Q || es.J != J || es.rank != rank) {
      if (es.Q) hipFree(es.Q);
      if (es.Qn) hipFree(es.Qn);
      HIP_CHECK(hipMalloc(&es.Q, sizeof(double) * J * rank));
      HIP_CHECK(hipMalloc(&es.Qn, sizeof(double) * J * rank));
    }
    es.J = J;
    es.rank = rank;
    HIP_CHECK(hipMemcpyAsync(es.Q, Uw, sizeof(double) * nJR, hipMemcpyDeviceToDevice, st_));
    for (int d = 0; d < rank; d++) es.evh[d] = evn[d];
    es.lamR = evn[rank - 1];
    es.lamR1 = cols > rank ? std::max(0.0, evn[rank]) : std::min(sigma, es.lamR * (1 - 1e-6));
    es.rho = evn[0];
    es.rho_frob = 0;
    es.head = 4.0;
    es.valid = es.lamR > es.lamR1 * (1 + 1e-9);
    es.fast++;
    return true;
  }
  bool cold_ok_ = false;
  int64_t cold_subspace_from_ = 2048;  // PPALS_COLD_SUBSPACE_FROM (tests: lower it to exercise the route)
  int cold_b_ = 0;
  const double *cold_Uo_ = nullptr;
  double cold_th_[kEigEvMax] = {0};
  double frob_shifted(const double *G, int64_t J, double sigma) {
    const int g = grid_for(J * J, 256, 1024);
    double *part = (double *)ensure(ws_part_, ws_part_sz_, sizeof(double) * (g + 1));
    hipLaunchKernelGGL(k_frob_shifted, dim3(g), dim3(256), 0, st_, G, J, sigma, part);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, g, part + g);
    double h = 0;
    HIP_CHECK(hipMemcpyAsync(&h, part + g, sizeof(double), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    return std::sqrt(h);
  }
  void sign_align(double *W, const double *Wref, int64_t rows, int r) override {
    hipLaunchKernelGGL(k_sign_align, dim3((r + 3) / 4), dim3(256), 0, st_, W, Wref, rows, r);
    HIP_CHECK(hipGetLastError());
  }
  void add_inplace(double *dst, const double *src, int64_t n) override {
    hipLaunchKernelGGL(k_add_inplace, dim3(grid_for(n, 256)), dim3(256), 0, st_, dst, src, n);
    HIP_CHECK(hipGetLastError());
  }
  void rows_times_small(const double *A, int64_t rows, int K, const double *B, int C,
                        const double *D, double *out) override {
    if ((size_t)K * C * sizeof(double) > 60 * 1024)
      throw std::runtime_error("ppals: rows_times_small: the small operand does not fit LDS");
    double *dst = out;
    if (out == A) dst = (double *)ensure(ws_big2_, ws_big2_sz_, sizeof(double) * (size_t)rows * C);
    hipLaunchKernelGGL(k_rows_times_small, dim3(grid_for(rows * C, 256)), dim3(256),
                       sizeof(double) * (size_t)K * C, st_, A, rows, K, B, C, D, dst);
    HIP_CHECK(hipGetLastError());
    if (dst != out)
      HIP_CHECK(hipMemcpyAsync(out, dst, sizeof(double) * rows * C, hipMemcpyDeviceToDevice, st_));
  }
  void lowrank_accumulate(void *X, int xdt, int64_t n, int R, const double *T, int r,
                          const double *VT) override {
    const size_t lds = sizeof(double) * (size_t)r * R;
    if (xdt == F32)
      hipLaunchKernelGGL(k_lowrank_accumulate<float>, dim3(grid_for(n, 256)), dim3(256), lds, st_,
                         (float *)X, n, R, T, r, VT);
    else
      hipLaunchKernelGGL(k_lowrank_accumulate<double>, dim3(grid_for(n, 256)), dim3(256), lds, st_,
                         (double *)X, n, R, T, r, VT);
    HIP_CHECK(hipGetLastError());
  }
  void sumsq(const double *x, int64_t n, double *out) override {
    int g = grid_for(n, 256, 1024);
    double *part = (double *)ensure(ws_part_, ws_part_sz_, sizeof(double) * g);
    hipLaunchKernelGGL(k_sumsq, dim3(g), dim3(256), 0, st_, x, n, part);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st_, part, g, out);
    HIP_CHECK(hipGetLastError());
  }

  // ------------------------------------------------------------------ stopwatch (ops.h)
  int timer_begin() override {
    int h = -1;
    for (size_t q = 0; q < timers_.size(); q++)
      if (!timers_[q].busy) h = (int)q;
    if (h < 0) {
      if (timers_.size() >= 64) return -1;
      Timer t;
      if (hipEventCreate(&t.a) != hipSuccess) return -1;
      if (hipEventCreate(&t.b) != hipSuccess) {
        hipEventDestroy(t.a);
        return -1;
      }
      timers_.push_back(t);
      h = (int)timers_.size() - 1;
    }
    timers_[h].busy = true;
    timers_[h].ended = false;
    if (hipEventRecord(timers_[h].a, st_) != hipSuccess) {
      timers_[h].busy = false;
      return -1;
    }
    return h;
  }
  void timer_end(int h) override {
    if (h < 0 || h >= (int)timers_.size() || !timers_[h].busy) return;
    timers_[h].ended = hipEventRecord(timers_[h].b, st_) == hipSuccess;
  }
  double timer_read(int h) override {
    if (h < 0 || h >= (int)timers_.size() || !timers_[h].busy) return -1.0;
    Timer &t = timers_[h];
    t.busy = false;
    float ms = 0;
    if (!t.ended || hipEventSynchronize(t.b) != hipSuccess ||
        hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess)
      return -1.0;
    return 1e-3 * (double)ms;
  }

  // ------------------------------------------------------------------ profiling
  void profile_enable(int level) override { profiling_ = level; }
  void profile_collect() override {
    HIP_CHECK(hipStreamSynchronize(st_));
    for (size_t i = 0; i < nev_; i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, events_[i].a, events_[i].b) == hipSuccess) {
        prof[events_[i].slot].launches += 1;
        prof[events_[i].slot].ms += ms;
        prof[events_[i].slot].bytes += events_[i].bytes;
      }
    }
    nev_ = 0;
  }

 private:
  struct RocSolver {
    void *handle = nullptr;
    int (*dsyevd)(void *, int, int, int, double *, int, double *, double *, int *) = nullptr;
  };
  RocSolver &rocsolver() {
    if (rs_.handle) return rs_;
    EigLibs &l = eig_libs();  // loads on first use unless hip_preload_eigensolver() ran earlier
    auto create = (int (*)(void **))dlsym(l.blas, "rocblas_create_handle");
    auto set_stream = (int (*)(void *, hipStream_t))dlsym(l.blas, "rocblas_set_stream");
    *(void **)(&rs_.dsyevd) = dlsym(l.solver, "rocsolver_dsyevd");
    if (!create || !set_stream || !rs_.dsyevd)
      throw std::runtime_error("ppals: rocSOLVER symbols missing");
    if (create(&rs_.handle) != 0 || set_stream(rs_.handle, st_) != 0)
      throw std::runtime_error("ppals: rocblas handle creation failed");
    return rs_;
  }
  RocSolver rs_;
  struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    bool busy = false, ended = false;
  };
  std::vector<Timer> timers_;
  struct Ev {
    hipEvent_t a, b;
    int slot;
    double bytes;
  };
  void prof_begin(int slot, double bytes) {
    cur_ = -1;
    if (profiling_ < 1 + slot) return;  // level 1: tensor scans only, level 2: every timed slot
    if (nev_ == events_.size()) {
      if (events_.size() >= 16384) return;
      Ev e;
      if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
      events_.push_back(e);
    }
    cur_ = (int)nev_++;
    events_[cur_].slot = slot;
    events_[cur_].bytes = bytes;
    hipEventRecord(events_[cur_].a, st_);
  }
  void prof_end() {
    if (cur_ >= 0) hipEventRecord(events_[cur_].b, st_);
    cur_ = -1;
  }
  // grow-only workspace; regrowth synchronises (only happens while shapes are first seen)
  void *ensure(void *&p, size_t &sz, size_t need) {
    if (need > sz) {
      if (p) {
        HIP_CHECK(hipStreamSynchronize(st_));
        HIP_CHECK(hipFree(p));
      }
      size_t n = need + need / 4 + 256;
      HIP_CHECK(hipMalloc(&p, n));
      sz = n;
    }
    return p;
  }
  double *small(int n) { return (double *)ensure(ws_small_, ws_small_sz_, sizeof(double) * n); }

  int dev_ = 0, ncu_ = 256, force_jacobi_ = 0, persist_mult_ = 40;
  int eig_debug_ = 0;
  SysArgs sys_;             // arm_gram_system: the system the next contraction prepares on the side
  int sys_R_ = 0;
  bool sys_armed_ = false, sys_ready_ = false;
  NormArgs norm_;           // arm_normalize: folded into the next cp_mode_update
  int norm_mode_ = -1;
  const double *norm_G_ = nullptr;
  bool norm_armed_ = false;
  bool eig_frob_once_ = false;
  void *ws_jac_ = nullptr;  // rotation product of the one-sided Jacobi (r x r)
  size_t ws_jac_sz_ = 0;
  int sym_lds_min_ = 768;   // PPALS_SYM_LDS_MIN: rows from which the symmetric product is LDS-tiled
  int eig_defer_ok_ = 1;    // PPALS_EIG_DEFER=0: every projector step waits for its own checks
  bool handover_ok_ = true;                 // second stream released by a value the kernel stores (else: events)
  unsigned long long *handover_ = nullptr;  // [0] the sequence number published last, [1] workgroup counter
  unsigned long long handover_seq_ = 0;
  int eig_defer_fail_ = 0;  // PPALS_EIG_DEFER_FAIL=n (tests): every n-th deferred check is reported as failed
  int eig_defer_count_ = 0;
  int eig_cold_ = 1;  // PPALS_EIG_FAST=2: cold starts go straight to the full solver (A/B, tests)
  void *ws_cold_ = nullptr;
  size_t ws_cold_sz_ = 0;
  double eig_sigma_scale_ = 0;  // PPALS_EIG_SIGMA_SCALE=f: shift = f * (estimate of the next eigenvalue) (tests)
  int force_eiginv_ = 0;  // PPALS_FORCE_EIGINV=1: R > 64 always inverts S through dsyevd (tests)
  int eig_fast_ = 1;  // PPALS_EIG_FAST=0: always the full eigensolver (A/B, tests)
  std::map<int, EigState> eig_state_;  // by slot
  int eig_next_base_ = 0;
  void *eig_host_ = nullptr;  // pinned: the read-back of a projector step
  void *ws_eig_ = nullptr, *ws_orth_ = nullptr, *ws_pow_ = nullptr, *ws_part2_ = nullptr;
  size_t ws_eig_sz_ = 0, ws_orth_sz_ = 0, ws_pow_sz_ = 0, ws_part2_sz_ = 0;
  hipStream_t st_ = nullptr;
  hipStream_t st2_ = nullptr;  // the Jacobi of a lazy eigen-step (created on first use)
  void *ws_mttv_ = nullptr;  // partial sums of a j-split k_mttv_l
  size_t ws_mttv_sz_ = 0;
  void *ws_pack_ = nullptr, *ws_slab_ = nullptr, *ws_krp_ = nullptr, *ws_part_ = nullptr,
       *ws_small_ = nullptr, *ws_big_ = nullptr, *ws_big2_ = nullptr;
  size_t ws_pack_sz_ = 0, ws_slab_sz_ = 0, ws_krp_sz_ = 0, ws_part_sz_ = 0, ws_small_sz_ = 0,
         ws_big_sz_ = 0, ws_big2_sz_ = 0;
  int profiling_ = 0;
  std::vector<Ev> events_;
  size_t nev_ = 0;
  int cur_ = -1;
};

Ops *make_hip_ops(int device) { return new HipOps(device); }
void hip_preload_eigensolver() { (void)eig_libs(); }

}  // namespace ppals
