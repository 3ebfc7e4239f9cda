// scan_lean_kernels.hip.h — MEASUREMENT TOOLING (tools/scan2_bench.hip only; not part of libppals.so):
// the register-lean forms of the several-n-tile tensor scan that round 2 measured against the
// launcher's choice and rejected (profiles/r02c_scan2_bench_*, r02x_*). Moved out of the product's
// kernels_scan.hip.h in round 4; needs that header's helpers, so include it first.
#pragma once
namespace ppals {

// ---------------------------------------------------------------------------------------------
// Register-lean forms of the persistent buffer-load suffix scan for SEVERAL n-tiles (16 < R <= 64,
// fp32 tensor). k_scan_suffix_buf keeps, per lane, 16*NT fp32 accumulators, 16*NT fp64 flush
// registers and a full-block register double buffer (32 + 8*NT dwords): at NT = 2 that is ~190
// VGPRs = 2 waves per SIMD, and the MFMA pipe is ~60 % busy at HBM speed, so the scan needs the
// third wave. Two independent levers, both compile-time:
//   U    k-quads per step (4 = a whole 16-column block as in k_scan_suffix_buf, 2 = half a block):
//        the double buffer shrinks to U*(4 + NT) + U*... dwords, a step is 16*U*NT/4 MFMAs
//   ACC  0: fp32 chains of <= 64 terms flushed into fp64 registers (any K)
//        1: two-level fp32 — chains of <= 64 terms summed into a second set of fp32 registers;
//           the launcher uses it only when a tile reduces <= 1024 terms (single-mode contractions
//           of the multi-sweep schedule, level-1 PP operators, Tucker mode products), where the
//           second level adds <= 16 partial sums: the rounding is that of a pairwise fp32 sum of
//           depth 2, below the fp32 rounding of the stored result (out32) times a small constant
// Same tiling, packed-operand layout, persistence and epilogue as k_scan_suffix_buf.
template <int NT, int U, int ACC, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void k_scan_suffix_lean(
    const float *__restrict__ V, int64_t M, int64_t K, int64_t batch_stride,
    const float *__restrict__ P, int n_mtiles, int nsplit, int kb_per_split, int nkb,
    double *__restrict__ out, int64_t out_nstride, int64_t out_split_stride,
    int64_t out_batch_stride, int ncols, int out32, int64_t ntiles) {
  typedef ScanTraits<float> TR;
  typedef f32x4 vec;
  typedef f32x4 acc_t;
  constexpr int VEC = 4, KB = 16;
  constexpr int SPB = 4 / U;      // steps per 16-column block
  constexpr int FLUSH = 4 * SPB;  // steps per fp32 chain (64 terms)
  typedef float pvec __attribute__((ext_vector_type(U)));

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  const int voffP = (int)((g * 16 + j16) * VEC * (int)sizeof(float));
  const int64_t block_bytes = (int64_t)KB * M * (int64_t)sizeof(float);
  const int64_t total_bytes = K * M * (int64_t)sizeof(float);
  const int ustep = (int)((int64_t)4 * M * (int64_t)sizeof(float));
  const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(
      (void *)P, 0, (int)((int64_t)nkb * NT * (4 * 16 * VEC) * (int64_t)sizeof(float)), 0x00020000);

  struct Tile {
    int64_t m, obase;
    const float *vbase;
    int voff, st0, st1;  // step range [st0, st1)
    bool live;
  };
  auto decode = [&](int64_t id, Tile &t) {
    int64_t b = id;
    const int mtile = (int)(b % n_mtiles);
    b /= n_mtiles;
    const int split = (int)(b % nsplit);
    const int64_t batch = b / nsplit;
    const int64_t m0 = ((int64_t)mtile * 4 + wave) * (16 * VEC);
    t.live = m0 < M;  // wave-uniform
    t.m = m0 + (int64_t)VEC * j16;
    const int64_t m_ld = t.live ? min(t.m, M - VEC) : 0;
    t.voff = (int)(((int64_t)g * M + m_ld) * (int64_t)sizeof(float));
    const int kb0 = split * kb_per_split;
    t.st0 = kb0 * SPB;
    t.st1 = min(nkb, kb0 + kb_per_split) * SPB;
    t.vbase = V + batch * batch_stride;
    t.obase = split * out_split_stride + batch * out_batch_stride;
  };
  // step st = block st / SPB, half h = st % SPB: k-quads u = U*h .. U*h + U - 1 of that block
#define PPALS_LEAN_LOAD(vbase_, voff_, st_, vv_, bb_)                                          \
  {                                                                                            \
    const int kb_ = (st_) / SPB, h_ = (st_) % SPB;                                             \
    const int64_t boff_ = (int64_t)kb_ * block_bytes;                                          \
    const int64_t rem_ = total_bytes - boff_;                                                  \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                      \
        (void *)((const char *)(vbase_) + boff_), 0, (int)min(rem_, block_bytes), 0x00020000); \
    _Pragma("unroll") for (int u = 0; u < U; u++) vv_[u] = __builtin_bit_cast(                 \
        vec, __builtin_amdgcn_raw_buffer_load_b128(rs_, voff_, (h_ * U + u) * ustep, 2));      \
    _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {                                        \
      const int so_ = (int)((kb_ * NT + nt) * (4 * 16 * VEC) * (int)sizeof(float)) +           \
                      h_ * U * (int)sizeof(float);                                             \
      if constexpr (U == 4)                                                                    \
        bb_[nt] = __builtin_bit_cast(pvec, __builtin_amdgcn_raw_buffer_load_b128(rsrcP, voffP, so_, 0)); \
      else                                                                                     \
        bb_[nt] = __builtin_bit_cast(pvec, __builtin_amdgcn_raw_buffer_load_b64(rsrcP, voffP, so_, 0));  \
    }                                                                                          \
  }

  Tile cur, nxt;
  int64_t id = blockIdx.x;
  for (; id < ntiles; id += gridDim.x) {
    decode(id, cur);
    if (cur.live && cur.st0 < cur.st1) break;
  }
  if (id >= ntiles) return;
  vec cv[U];
  pvec cb[NT];
  PPALS_LEAN_LOAD(cur.vbase, cur.voff, cur.st0, cv, cb);

  for (;;) {
    bool has_next = false;
    int64_t nid = id + gridDim.x;
    for (; nid < ntiles; nid += gridDim.x) {
      decode(nid, nxt);
      if (nxt.live && nxt.st0 < nxt.st1) {
        has_next = true;
        break;
      }
    }
    acc_t acc[VEC][NT];
    acc_t acc2[ACC == 1 ? VEC : 1][ACC == 1 ? NT : 1];
    double acc64[ACC == 0 ? VEC : 1][ACC == 0 ? NT : 1][4];
#pragma unroll
    for (int a = 0; a < VEC; a++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          acc[a][nt][r] = 0;
          if constexpr (ACC == 1) acc2[a][nt][r] = 0;
          if constexpr (ACC == 0) acc64[a][nt][r] = 0.0;
        }
      }
    for (int sc = cur.st0; sc < cur.st1; sc += FLUSH) {
      const int se = min(cur.st1, sc + FLUSH);
      for (int st = sc; st < se; st++) {
        vec nv[U];
        pvec nb[NT];
        const bool same = st + 1 < cur.st1;
        const float *pv = (same || !has_next) ? cur.vbase : nxt.vbase;
        const int po = (same || !has_next) ? cur.voff : nxt.voff;
        const int ps = same ? st + 1 : (has_next ? nxt.st0 : st);
        PPALS_LEAN_LOAD(pv, po, ps, nv, nb);
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int jj = 0; jj < VEC; jj++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
              acc[jj][nt] = TR::mfma(cb[nt][u], cv[u][jj], acc[jj][nt]);
#pragma unroll
        for (int u = 0; u < U; u++) cv[u] = nv[u];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) cb[nt] = nb[nt];
      }
#pragma unroll
      for (int a = 0; a < VEC; a++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            if constexpr (ACC == 0)
              acc64[a][nt][r] += (double)acc[a][nt][r];
            else
              acc2[a][nt][r] += acc[a][nt][r];
            acc[a][nt][r] = 0;
          }
    }
    const bool vec_ok = (((cur.obase | out_nstride) & (VEC - 1)) == 0);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = 16 * nt + TR::row(lane, r);
        if (n < ncols && cur.m < M) {
          double val[VEC];
#pragma unroll
          for (int jj = 0; jj < VEC; jj++) {
            if constexpr (ACC == 0)
              val[jj] = acc64[jj][nt][r];
            else
              val[jj] = (double)acc2[jj][nt][r];
          }
          const int64_t idx = cur.obase + (int64_t)n * out_nstride + cur.m;
          if (vec_ok && out32) {
            f32x4 ov;
#pragma unroll
            for (int jj = 0; jj < VEC; jj++) ov[jj] = (float)val[jj];
            *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(out) + idx) = ov;
          } else if (vec_ok) {
#pragma unroll
            for (int jj = 0; jj < VEC; jj += 2) {
              f64x2 ov = {val[jj], val[jj + 1]};
              *reinterpret_cast<f64x2 *>(out + idx + jj) = ov;
            }
          } else {
#pragma unroll
            for (int jj = 0; jj < VEC; jj++) scan_store(out, idx + jj, val[jj], out32);
          }
        }
      }
    if (!has_next) break;
    cur = nxt;
    id = nid;
  }
#undef PPALS_LEAN_LOAD
}

}  // namespace ppals
