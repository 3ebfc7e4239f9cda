// kernels_scan.hip.h — the tensor-scan kernels: the only code that touches the s^N tensor.
//
// Both kernels compute  out[.., n] = sum_j V[.., j, ..] * B[j, n]  for a skinny B (n <= 64 columns,
// the CP rank / Tucker core size) with the tensor streamed exactly once from HBM straight into
// registers (every element of V is used once, so there is nothing to stage in LDS) and the
// reduction done on the matrix cores:
//
//   f32 tensor: v_mfma_f32_16x16x4_f32, partial sums flushed into fp64 registers every FLUSH
//               k-blocks, so no fp32 accumulation chain is longer than 16*VEC*FLUSH terms
//   f64 tensor: v_mfma_f64_16x16x4_f64
//
// MFMA operand roles (16x16x4, one value per lane each, cdna_hip_programming.md §3):
//   A[i = lane&15][kk = lane>>4]  <- packed Khatri-Rao value   (i = output column n)
//   B[kk = lane>>4][j = lane&15]  <- tensor value              (j = a tensor row / column)
//   D[i][j]: lane holds column j = lane&15 and rows i = rowmap(lane, reg)
// so each lane ends up owning one tensor row (suffix kernel) or column (prefix kernel) and 4
// output columns n — consecutive lanes write consecutive addresses.
//
// The Khatri-Rao operand is pre-packed by k_krp_pack so that ONE 16-byte load per lane fetches the
// A-operand values of VEC consecutive MFMAs:
//   packed[((((blk*NT + nt)*4 + g)*16 + n)*VEC + u] = KRP[blk*4*VEC + idx(g,u)][16*nt + n]
//   suffix kernel: idx(g,u) = 4*u + g     (u-th k-quad of the block, kk = g)
//   prefix kernel: idx(g,u) = VEC*g + u   (the lane's own VEC consecutive reduction rows)
// Restates: common.cxx:56,83 (V_temp[seq_f] = V_front[seq] * W[..]) fused over the sibling modes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppals {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename TV>
struct ScanTraits;
template <>
struct ScanTraits<float> {
  static constexpr int VEC = 4;
  typedef f32x4 vec;
  typedef f32x4 acc;
  static constexpr bool NEEDS_FLUSH = true;
  __device__ static inline acc mfma(float a, float b, acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // D-matrix row held in register `reg` of `lane` (C/D map of the f32 16x16 shapes)
  __device__ static inline int row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};
template <>
struct ScanTraits<double> {
  static constexpr int VEC = 2;
  typedef f64x2 vec;
  typedef f64x4 acc;
  static constexpr bool NEEDS_FLUSH = false;
  __device__ static inline acc mfma(double a, double b, acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // f64 16x16x4 uses a different C/D map: row = (lane>>4) + 4*reg
  __device__ static inline int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};

// typed result store: the scans accumulate in fp64 and write fp64 (tree nodes, slabs) or fp32
// (the first-level intermediate of the multi-sweep schedule, stored in the tensor's own precision)
__device__ inline void scan_store(double *base, int64_t idx, double v, int out32) {
  if (out32)
    reinterpret_cast<float *>(base)[idx] = (float)v;
  else
    base[idx] = v;
}

constexpr int SCAN_FLUSH = 4;  // k-blocks between fp32 -> fp64 flushes (chains of <= 64 terms)

template <typename TV, bool ALIGNED>
__device__ inline typename ScanTraits<TV>::vec load_vec(const TV *__restrict__ base, int64_t off,
                                                        bool col_ok, int64_t row, int64_t nrows) {
  typedef typename ScanTraits<TV>::vec vec;
  constexpr int VEC = ScanTraits<TV>::VEC;
  vec v;
#pragma unroll
  for (int e = 0; e < VEC; e++) v[e] = (TV)0;
  if (ALIGNED) {
    if (col_ok && row < nrows) v = *reinterpret_cast<const vec *>(base + off);
  } else {
    if (col_ok) {
#pragma unroll
      for (int e = 0; e < VEC; e++)
        if (row + e < nrows) v[e] = base[off + e];
    }
  }
  return v;
}

// Rows of a padded resident layout (Ops::pad_layout): stored row m is row m % row_ld of block
// m / row_ld, and only the first row_valid rows of a block exist. The scans read stored rows and
// write the COMPACT result: `mo` = where the lane's first row lands, `nvalid` = how many of its
// VEC consecutive rows exist (a lane's rows never straddle a block: row_ld % VEC == 0).
struct ScanRowMap {
  int64_t mo;
  int nvalid;
};
template <int VEC>
__device__ inline ScanRowMap scan_row_map(int64_t m, int64_t M, int64_t row_ld, int64_t row_valid) {
  ScanRowMap r;
  if (row_ld == 0) {
    r.mo = m;
    r.nvalid = (int)max((int64_t)0, min((int64_t)VEC, M - m));
  } else {
    const int64_t q = m / row_ld, rem = m - q * row_ld;
    r.mo = q * row_valid + rem;
    r.nvalid = m < M ? (int)max((int64_t)0, min((int64_t)VEC, row_valid - rem)) : 0;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// suffix scan (K1 / batched single-mode TTM):
//   out[m + n*out_nstride (+ split/batch offsets)] = sum_{k in split} V[m + M*k (+batch)] * B[k,n]
// grid: 1-D, block id = mtile + n_mtiles*(split + nsplit*batch); 256 threads = 4 waves, each wave
// owns 16*VEC consecutive rows m (lane: VEC consecutive rows), all 4 waves share the k range, so
// one k column of V is read as 4 x 16*VEC consecutive elements (1 KiB) per step.
template <typename TV, int NT, bool ALIGNED>
__global__ __launch_bounds__(256) void k_scan_suffix(
    const TV *__restrict__ V, int64_t M, int64_t K, int64_t batch_stride,
    const TV *__restrict__ P, int n_mtiles, int nsplit, int kb_per_split, int nkb,
    double *__restrict__ out, int64_t out_nstride, int64_t out_split_stride,
    int64_t out_batch_stride, int ncols, int out32, int64_t row_ld = 0, int64_t row_valid = 0) {
  typedef ScanTraits<TV> TR;
  typedef typename TR::vec vec;
  typedef typename TR::acc acc_t;
  constexpr int VEC = TR::VEC;
  constexpr int KB = 4 * VEC;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  int64_t bid = blockIdx.x;
  const int mtile = (int)(bid % n_mtiles);
  bid /= n_mtiles;
  const int split = (int)(bid % nsplit);
  const int64_t batch = bid / nsplit;

  const int64_t m0 = ((int64_t)mtile * 4 + wave) * (16 * VEC);
  if (m0 >= M) return;  // wave-uniform
  const int64_t m = m0 + (int64_t)VEC * j16;
  const TV *__restrict__ Vb = V + batch * batch_stride;
  const int kb0 = split * kb_per_split;
  const int kb1 = min(nkb, kb0 + kb_per_split);

  acc_t acc[VEC][NT];
  double acc64[TR::NEEDS_FLUSH ? VEC : 1][TR::NEEDS_FLUSH ? NT : 1][4];
#pragma unroll
  for (int a = 0; a < VEC; a++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
#pragma unroll
      for (int r = 0; r < 4; r++) acc[a][nt][r] = 0;
      if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc64[a][nt][r] = 0.0;
      }
    }

  int since_flush = 0;
  for (int kb = kb0; kb < kb1; kb++) {
    vec bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
      bv[nt] = *reinterpret_cast<const vec *>(
          P + ((((int64_t)kb * NT + nt) * 4 + g) * 16 + j16) * VEC);
    vec vv[VEC];
#pragma unroll
    for (int u = 0; u < VEC; u++) {
      const int64_t k = (int64_t)kb * KB + 4 * u + g;
      vv[u] = load_vec<TV, ALIGNED>(Vb, k * M + m, k < K, m, M);
    }
#pragma unroll
    for (int u = 0; u < VEC; u++)
#pragma unroll
      for (int jj = 0; jj < VEC; jj++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[jj][nt] = TR::mfma(bv[nt][u], vv[u][jj], acc[jj][nt]);
    if constexpr (TR::NEEDS_FLUSH) {
      if (++since_flush == SCAN_FLUSH) {
        since_flush = 0;
#pragma unroll
        for (int a = 0; a < VEC; a++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              acc64[a][nt][r] += (double)acc[a][nt][r];
              acc[a][nt][r] = 0;
            }
      }
    }
  }

  const int64_t obase = split * out_split_stride + batch * out_batch_stride;
  const ScanRowMap rm = scan_row_map<VEC>(m, M, row_ld, row_valid);
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int n = 16 * nt + TR::row(lane, r);
      if (n < ncols) {
#pragma unroll
        for (int jj = 0; jj < VEC; jj++) {
          double val = (double)acc[jj][nt][r];
          if constexpr (TR::NEEDS_FLUSH) val += acc64[jj][nt][r];
          if (jj < rm.nvalid)
            scan_store(out, obase + (int64_t)n * out_nstride + rm.mo + jj, val, out32);
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// prefix scan (K2 / leading-mode TTM):
//   out[k*out_kstride + n*out_nstride (+ split offset)] = sum_{m in split} V[m + M*k] * B[m,n]
// grid: (ceil(K/64), nsplit); each wave owns 16 consecutive columns k (lane&15), the 4 lane groups
// g take VEC consecutive reduction rows each, so one step reads 16 columns x 4*VEC rows; UNROLL
// steps are issued back to back to keep >= UNROLL KiB per wave in flight.
template <typename TV, int NT, bool ALIGNED, int UNROLL>
__global__ __launch_bounds__(256) void k_scan_prefix(
    const TV *__restrict__ V, int64_t M, int64_t K, const TV *__restrict__ P, int mb_per_split,
    int nmb, double *__restrict__ out, int64_t out_kstride, int64_t out_nstride,
    int64_t out_split_stride, int ncols, int out32) {
  typedef ScanTraits<TV> TR;
  typedef typename TR::vec vec;
  typedef typename TR::acc acc_t;
  constexpr int VEC = TR::VEC;
  constexpr int MB = 4 * VEC;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  const int64_t k0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (k0 >= K) return;  // wave-uniform
  const int64_t k = k0 + j16;
  const bool k_ok = k < K;
  const int split = blockIdx.y;
  const int mb0 = split * mb_per_split;
  const int mb1 = min(nmb, mb0 + mb_per_split);
  const TV *__restrict__ Vc = V + (k_ok ? k : 0) * M;

  acc_t acc[2][NT];
  double acc64[TR::NEEDS_FLUSH ? NT : 1][4];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      acc[0][nt][r] = 0;
      acc[1][nt][r] = 0;
    }
    if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
      for (int r = 0; r < 4; r++) acc64[nt][r] = 0.0;
    }
  }

  int since_flush = 0;
  for (int mbb = mb0; mbb < mb1; mbb += UNROLL) {
    vec vv[UNROLL];
    vec bv[UNROLL][NT];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int mb = mbb + u;
      const bool ok = mb < mb1;  // wave-uniform
      const int64_t m = (int64_t)mb * MB + (int64_t)VEC * g;
      vv[u] = load_vec<TV, ALIGNED>(Vc, m, ok && k_ok, m, M);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        if (ok)
          bv[u][nt] = *reinterpret_cast<const vec *>(
              P + ((((int64_t)mb * NT + nt) * 4 + g) * 16 + j16) * VEC);
        else {
#pragma unroll
          for (int e = 0; e < VEC; e++) bv[u][nt][e] = (TV)0;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
#pragma unroll
      for (int jj = 0; jj < VEC; jj++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
          acc[jj & 1][nt] = TR::mfma(bv[u][nt][jj], vv[u][jj], acc[jj & 1][nt]);
    if constexpr (TR::NEEDS_FLUSH) {
      since_flush += UNROLL;
      if (since_flush >= SCAN_FLUSH) {
        since_flush = 0;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            acc64[nt][r] += (double)acc[0][nt][r] + (double)acc[1][nt][r];
            acc[0][nt][r] = 0;
            acc[1][nt][r] = 0;
          }
      }
    }
  }

  const int64_t obase = split * out_split_stride;
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int n = 16 * nt + TR::row(lane, r);
      if (n < ncols && k_ok) {
        double val = (double)acc[0][nt][r] + (double)acc[1][nt][r];
        if constexpr (TR::NEEDS_FLUSH) val += acc64[nt][r];
        scan_store(out, obase + (int64_t)n * out_nstride + k * out_kstride, val, out32);
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast variants (the ones the hot path runs): shapes whose leading extent is a multiple of VEC, so
// every lane access is one aligned 16-byte load. Differences from the generic kernels above:
//  * no predication: out-of-range rows/columns are CLAMPED to a valid address and their lanes'
//    results are simply not stored; out-of-range reduction indices are harmless because the
//    packed Khatri-Rao operand is zero there;
//  * the loads of block i+1 are issued before the MFMAs of block i (register double buffer), so a
//    wave overlaps its own HBM latency with its own matrix work instead of relying on occupancy;
//  * fp32 chains are flushed to fp64 every FLUSH blocks in a fixed-trip inner loop.
// OPT bit 0: non-temporal loads of the tensor (streamed once: do not displace the packed operand
// from L2). OPT bit 1: XCD-aware block remap (blocks of one k-split share an XCD's L2).
template <typename vec, int OPT>
__device__ inline vec scan_ld(const vec *p) {
  if constexpr (OPT & 1)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}

template <typename TV, int NT, int OPT = 0>
__global__ __launch_bounds__(256) void k_scan_suffix_fast(
    const TV *__restrict__ V, int64_t M, int64_t K, int64_t batch_stride,
    const TV *__restrict__ P, int n_mtiles, int nsplit, int kb_per_split, int nkb,
    double *__restrict__ out, int64_t out_nstride, int64_t out_split_stride,
    int64_t out_batch_stride, int ncols, int out32, int64_t row_ld = 0, int64_t row_valid = 0,
    int tail_from = -1) {
  // TAIL MODE (OPT bit 3; one batch, no k-split, dynamic LDS): a launch of a little more than a whole
  // number of rounds of resident workgroups (cfg5: 625 tiles on 512 slots) leaves its last tiles to a
  // quarter-empty chip. Tiles from `tail_from` on are therefore handed out as FOUR workgroups each:
  // a workgroup takes ONE 16*VEC-row strip and its four waves split the k range, the partial sums
  // meet in LDS in a fixed order. The same bytes are read once, nothing extra is written, no second
  // launch — the tail simply has four times as many, four times shorter work items.
  typedef ScanTraits<TV> TR;
  typedef typename TR::vec vec;
  typedef typename TR::acc acc_t;
  constexpr int VEC = TR::VEC;
  constexpr int KB = 4 * VEC;
  constexpr int FLUSH = 4;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  int64_t bid = blockIdx.x;
  bool tail = false;
  if constexpr ((OPT & 8) != 0) tail = tail_from >= 0 && bid >= tail_from;
  if constexpr (OPT & 2) {
    // blocks are dealt round-robin over the 8 XCDs (speed only): give each XCD a contiguous
    // eighth of the (mtile-fastest) id space so that it works on few k-splits at a time
    const int64_t nb = gridDim.x;
    const int64_t per = nb / 8;
    if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
  }
  const int mtile = (int)(bid % n_mtiles);
  bid /= n_mtiles;
  const int split = tail ? 0 : (int)(bid % nsplit);       // (tail mode: one batch, no k-split)
  const int64_t batch = tail ? 0 : bid / nsplit;

  int64_t m0 = ((int64_t)mtile * 4 + wave) * (16 * VEC);
  int kb0 = split * kb_per_split;
  int kb1 = min(nkb, kb0 + kb_per_split);
  if constexpr ((OPT & 8) != 0) {
    if (tail) {  // (workgroup-uniform: all four waves share the strip)
      m0 = ((int64_t)tail_from * 4 + (blockIdx.x - tail_from)) * (16 * VEC);
      const int per = (nkb + 3) / 4;
      kb0 = min(nkb, wave * per);
      kb1 = min(nkb, kb0 + per);
    }
  }
  if (m0 >= M) return;  // wave-uniform (tail mode: workgroup-uniform)
  const int64_t m = m0 + (int64_t)VEC * j16;
  const int64_t m_ld = min(m, M - VEC);  // clamped: lanes past the edge re-read the last rows
  const TV *__restrict__ vp = V + batch * batch_stride + m_ld;
  const TV *__restrict__ pp = P + ((int64_t)g * 16 + j16) * VEC;

  acc_t acc[VEC][NT];
  double acc64[TR::NEEDS_FLUSH ? VEC : 1][TR::NEEDS_FLUSH ? NT : 1][4];
#pragma unroll
  for (int a = 0; a < VEC; a++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
#pragma unroll
      for (int r = 0; r < 4; r++) acc[a][nt][r] = 0;
      if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc64[a][nt][r] = 0;
      }
    }

  vec cv[VEC], cb[NT];
  // block kb: k-quad u holds k = kb*KB + 4u + g. Blocks entirely below K ("full") are addressed
  // as lane-constant pointer + wave-uniform offset; the last, partial block clamps k to K-1 (the
  // packed operand is zero there, so the re-read values do not contribute).
  const int kfull = (int)min((int64_t)kb1, K / KB);
  const TV *__restrict__ vg = vp + (int64_t)g * M;
#define PPALS_LOAD_BLOCK(kb_, vv_, bb_)                                                  \
  {                                                                                      \
    _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bb_[nt] =                          \
        *reinterpret_cast<const vec *>(pp + ((int64_t)(kb_)*NT + nt) * (4 * 16 * VEC));  \
    if ((kb_) < kfull) {                                                                 \
      const TV *__restrict__ src_ = vg + (int64_t)(kb_) * ((int64_t)KB * M);             \
      _Pragma("unroll") for (int u = 0; u < VEC; u++) vv_[u] =                           \
          scan_ld<vec, OPT>(reinterpret_cast<const vec *>(src_ + (int64_t)(4 * u) * M)); \
    } else {                                                                             \
      _Pragma("unroll") for (int u = 0; u < VEC; u++) {                                  \
        const int64_t k_ = min((int64_t)(kb_)*KB + 4 * u + g, K - 1);                    \
        vv_[u] = scan_ld<vec, OPT>(reinterpret_cast<const vec *>(vp + k_ * M));          \
      }                                                                                  \
    }                                                                                    \
  }
  if (kb0 < kb1) PPALS_LOAD_BLOCK(kb0, cv, cb);
  for (int kc = kb0; kc < kb1; kc += FLUSH) {
    const int ke = min(kb1, kc + FLUSH);
    for (int kb = kc; kb < ke; kb++) {
      vec nv[VEC], nb[NT];
      const int kn = min(kb + 1, kb1 - 1);  // the last block is simply loaded twice
      PPALS_LOAD_BLOCK(kn, nv, nb);
#pragma unroll
      for (int u = 0; u < VEC; u++)
#pragma unroll
        for (int jj = 0; jj < VEC; jj++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            acc[jj][nt] = TR::mfma(cb[nt][u], cv[u][jj], acc[jj][nt]);
          }
#pragma unroll
      for (int u = 0; u < VEC; u++) cv[u] = nv[u];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) cb[nt] = nb[nt];
    }
    if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
      for (int a = 0; a < VEC; a++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            acc64[a][nt][r] += (double)acc[a][nt][r];
            acc[a][nt][r] = 0;
          }
    }
  }
#undef PPALS_LOAD_BLOCK

  if constexpr ((OPT & 8) != 0) {
    if (tail) {
      // waves 1..3 hand their sums to wave 0 through LDS: [wave - 1][slot][lane], fixed order
      extern __shared__ double tail_lds[];
      constexpr int NS = VEC * NT * 4;
      if (wave > 0) {
#pragma unroll
        for (int a = 0; a < VEC; a++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              double v;
              if constexpr (TR::NEEDS_FLUSH)
                v = acc64[a][nt][r];
              else
                v = (double)acc[a][nt][r];
              tail_lds[((wave - 1) * NS + (a * NT + nt) * 4 + r) * 64 + lane] = v;
            }
      }
      __syncthreads();
      if (wave > 0) return;
#pragma unroll
      for (int w = 0; w < 3; w++)
#pragma unroll
        for (int a = 0; a < VEC; a++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const double v = tail_lds[(w * NS + (a * NT + nt) * 4 + r) * 64 + lane];
              if constexpr (TR::NEEDS_FLUSH)
                acc64[a][nt][r] += v;
              else
                acc[a][nt][r] += v;
            }
    }
  }
  // epilogue: a lane owns VEC consecutive rows of 4 output columns -> one vector store per column
  // (16 lanes x VEC rows = 16*VEC contiguous elements); scalar stores only for unaligned strides
  const int64_t obase = split * out_split_stride + batch * out_batch_stride;
  const ScanRowMap rm = scan_row_map<VEC>(m, M, row_ld, row_valid);
  const bool vec_ok =
      (((obase | out_nstride | rm.mo) & (VEC - 1)) == 0) && rm.nvalid == VEC;
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int n = 16 * nt + TR::row(lane, r);
      if (n < ncols && rm.nvalid > 0) {
        double val[VEC];
#pragma unroll
        for (int jj = 0; jj < VEC; jj++) {
          if constexpr (TR::NEEDS_FLUSH)
            val[jj] = acc64[jj][nt][r];
          else
            val[jj] = (double)acc[jj][nt][r];
        }
        const int64_t idx = obase + (int64_t)n * out_nstride + rm.mo;
        if (vec_ok && out32) {
          typedef float ovec_t __attribute__((ext_vector_type(VEC)));
          ovec_t ov;
#pragma unroll
          for (int jj = 0; jj < VEC; jj++) ov[jj] = (float)val[jj];
          if constexpr (OPT & 4)  // non-temporal result stores (see the launcher)
            __builtin_nontemporal_store(ov, reinterpret_cast<ovec_t *>(reinterpret_cast<float *>(out) + idx));
          else
            *reinterpret_cast<ovec_t *>(reinterpret_cast<float *>(out) + idx) = ov;
        } else if (vec_ok) {
#pragma unroll
          for (int jj = 0; jj < VEC; jj += 2) {
            f64x2 ov = {val[jj], val[jj + 1]};
            if constexpr (OPT & 4)
              __builtin_nontemporal_store(ov, reinterpret_cast<f64x2 *>(out + idx + jj));
            else
              *reinterpret_cast<f64x2 *>(out + idx + jj) = ov;
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < VEC; jj++)
            if (jj < rm.nvalid) scan_store(out, idx + jj, val[jj], out32);
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Suffix scan with BUFFER loads (the variant the hot path runs). Same tiling and MFMA mapping as
// k_scan_suffix_fast, but every address is  wave-uniform descriptor (SGPRs) + loop-invariant
// per-lane byte offset (one VGPR) + wave-uniform SGPR offset:
//   * no per-iteration VALU address arithmetic and no address temporaries for the register
//     allocator to recycle under in-flight loads, so the compiler can place COUNTED vmcnt waits and
//     the register double buffer really overlaps HBM latency with the MFMAs of the same wave;
//   * the descriptor's num_records ends at the last valid k column of the block, so the partial
//     tail block needs no code at all: out-of-range lanes read 0 in hardware.
// Preconditions (checked by the launcher): 16*M*sizeof(TV) < 2^31 and the packed operand < 2^31 B.
typedef unsigned int scan_u32x4 __attribute__((vector_size(16)));

// Persistent form: a workgroup walks over tiles id = blockIdx.x, blockIdx.x + gridDim.x, ... and the
// first block of the NEXT tile is requested while the last block of the current one is multiplied
// and its results are stored, so the load pipeline never drains at a tile boundary. This matters
// for the single-mode contractions of the multi-sweep schedule, where a tile has only K/16 = 13
// blocks (cfg2) and pipeline fill + drain + workgroup launch were ~10 % of its lifetime.
// Measured and rejected (profiles/r01j_scan_bench_*, same-box A/B of bench.py): a two-cursor form
// with the load cursor 2 blocks ahead: -9 % at one n-tile (registers), and at two n-tiles (2
// waves/SIMD) 4.6 TB/s, still behind the global-load kernel k_scan_suffix_fast (5.6 TB/s) that the
// launcher keeps for that case; the same form with 1 block ahead: -2.5 % against this one; one
// column tile per WAVE (4/ct row groups x ct column tiles per workgroup, one-tile register budget,
// the tensor bytes shared through the L2): 3.8 TB/s at two tiles — the doubled L2 -> CU traffic
// costs more than the lost occupancy.
template <typename TV, int NT, int OPT = 1>
__global__ __launch_bounds__(256) void k_scan_suffix_buf(
    const TV *__restrict__ V, int64_t M, int64_t K, int64_t batch_stride,
    const TV *__restrict__ P, int n_mtiles, int nsplit, int kb_per_split, int nkb,
    double *__restrict__ out, int64_t out_nstride, int64_t out_split_stride,
    int64_t out_batch_stride, int ncols, int out32, int64_t ntiles, int64_t row_ld = 0,
    int64_t row_valid = 0) {
  typedef ScanTraits<TV> TR;
  typedef typename TR::vec vec;
  typedef typename TR::acc acc_t;
  constexpr int VEC = TR::VEC;
  constexpr int KB = 4 * VEC;
  constexpr int FLUSH = 4;
  constexpr int AUXV = (OPT & 1) ? 2 : 0;  // nt: streamed once, keep the packed operand in L2

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  const int voffP = (int)((g * 16 + j16) * VEC * (int)sizeof(TV));
  const int64_t block_bytes = (int64_t)KB * M * (int64_t)sizeof(TV);
  const int64_t total_bytes = K * M * (int64_t)sizeof(TV);
  const int ustep = (int)((int64_t)4 * M * (int64_t)sizeof(TV));
  const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(
      (void *)P, 0, (int)((int64_t)nkb * NT * (4 * 16 * VEC) * (int64_t)sizeof(TV)), 0x00020000);

  // per-tile state: rows of this lane, its byte offset, the block range and the batch base
  struct Tile {
    int64_t m, obase;
    const TV *vbase;
    int voff, kb0, kb1;
    bool live;
  };
  auto decode = [&](int64_t id, Tile &t) {
    // (tile ids fit 31 bits — the launcher checks — so the three divisions are 32-bit ones: the 64-bit
    // forms are ~100 instructions each, four of them per tile beside a main loop of 13 column blocks)
    unsigned b = (unsigned)id;
    const int mtile = (int)(b % (unsigned)n_mtiles);
    b /= (unsigned)n_mtiles;
    const int split = (int)(b % (unsigned)nsplit);
    const int64_t batch = b / (unsigned)nsplit;
    const int64_t m0 = ((int64_t)mtile * 4 + wave) * (16 * VEC);
    t.live = m0 < M;  // wave-uniform
    t.m = m0 + (int64_t)VEC * j16;
    const int64_t m_ld = t.live ? min(t.m, M - VEC) : 0;  // clamped: lanes past the edge re-read
    t.voff = (int)(((int64_t)g * M + m_ld) * (int64_t)sizeof(TV));
    t.kb0 = split * kb_per_split;
    t.kb1 = min(nkb, t.kb0 + kb_per_split);
    t.vbase = V + batch * batch_stride;
    t.obase = split * out_split_stride + batch * out_batch_stride;
  };
#define PPALS_BUF_LOAD(vbase_, voff_, kb_, vv_, bb_)                                           \
  {                                                                                            \
    const int64_t boff_ = (int64_t)(kb_)*block_bytes;                                          \
    const int64_t rem_ = total_bytes - boff_;                                                  \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                      \
        (void *)((const char *)(vbase_) + boff_), 0, (int)min(rem_, block_bytes), 0x00020000); \
    _Pragma("unroll") for (int u = 0; u < VEC; u++) vv_[u] = __builtin_bit_cast(               \
        vec, __builtin_amdgcn_raw_buffer_load_b128(rs_, voff_, u * ustep, AUXV));              \
    _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bb_[nt] = __builtin_bit_cast(            \
        vec, __builtin_amdgcn_raw_buffer_load_b128(                                            \
                 rsrcP, voffP, (int)(((kb_)*NT + nt) * (4 * 16 * VEC) * (int)sizeof(TV)), 0)); \
  }

  Tile cur, nxt;
  int64_t id = blockIdx.x;
  for (; id < ntiles; id += gridDim.x) {  // first live tile of this wave
    decode(id, cur);
    if (cur.live && cur.kb0 < cur.kb1) break;
  }
  if (id >= ntiles) return;
  vec cv[VEC], cb[NT];
  PPALS_BUF_LOAD(cur.vbase, cur.voff, cur.kb0, cv, cb);

  for (;;) {
    bool has_next = false;
    int64_t nid = id + gridDim.x;
    for (; nid < ntiles; nid += gridDim.x) {
      decode(nid, nxt);
      if (nxt.live && nxt.kb0 < nxt.kb1) {
        has_next = true;
        break;
      }
    }
    acc_t acc[VEC][NT];
    double acc64[TR::NEEDS_FLUSH ? VEC : 1][TR::NEEDS_FLUSH ? NT : 1][4];
#pragma unroll
    for (int a = 0; a < VEC; a++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[a][nt][r] = 0;
        if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
          for (int r = 0; r < 4; r++) acc64[a][nt][r] = 0;
        }
      }
    for (int kc = cur.kb0; kc < cur.kb1; kc += FLUSH) {
      const int ke = min(cur.kb1, kc + FLUSH);
      for (int kb = kc; kb < ke; kb++) {
        vec nv[VEC], nb[NT];
        // what to request next: the following block of this tile, else the first block of the
        // next tile, else (very last block of the wave) this block again
        const bool same = kb + 1 < cur.kb1;
        const TV *pv = (same || !has_next) ? cur.vbase : nxt.vbase;
        const int po = (same || !has_next) ? cur.voff : nxt.voff;
        const int pk = same ? kb + 1 : (has_next ? nxt.kb0 : kb);
        PPALS_BUF_LOAD(pv, po, pk, nv, nb);
#pragma unroll
        for (int u = 0; u < VEC; u++)
#pragma unroll
          for (int jj = 0; jj < VEC; jj++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
              acc[jj][nt] = TR::mfma(cb[nt][u], cv[u][jj], acc[jj][nt]);
#pragma unroll
        for (int u = 0; u < VEC; u++) cv[u] = nv[u];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) cb[nt] = nb[nt];
      }
      if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
        for (int a = 0; a < VEC; a++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              acc64[a][nt][r] += (double)acc[a][nt][r];
              acc[a][nt][r] = 0;
            }
      }
    }
    // epilogue: a lane owns VEC consecutive rows of 4 output columns -> one vector store per
    // column (16 lanes x VEC rows contiguous); scalar stores only for unaligned strides
    const ScanRowMap rm = scan_row_map<VEC>(cur.m, M, row_ld, row_valid);
    const bool vec_ok =
        (((cur.obase | out_nstride | rm.mo) & (VEC - 1)) == 0) && rm.nvalid == VEC;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = 16 * nt + TR::row(lane, r);
        if (n < ncols && rm.nvalid > 0) {
          double val[VEC];
#pragma unroll
          for (int jj = 0; jj < VEC; jj++) {
            if constexpr (TR::NEEDS_FLUSH)
              val[jj] = acc64[jj][nt][r];
            else
              val[jj] = (double)acc[jj][nt][r];
          }
          int64_t idx = cur.obase + (int64_t)n * out_nstride + rm.mo;
          // OPT bit 1 (tools/place2_bench only): row-blocked result, 256 rows x ncols per block —
          // what one workgroup stores is then one contiguous piece instead of ncols pieces
          if constexpr (OPT & 2)
            idx = cur.obase + (cur.m >> 8) * ((int64_t)ncols << 8) + ((int64_t)n << 8) + (cur.m & 255);
          if (vec_ok && out32) {
            typedef float ovec_t __attribute__((ext_vector_type(VEC)));
            ovec_t ov;
#pragma unroll
            for (int jj = 0; jj < VEC; jj++) ov[jj] = (float)val[jj];
            // OPT bit 2: non-temporal result stores (a result of hundreds of MB: see the launcher)
            if constexpr (OPT & 4)
              __builtin_nontemporal_store(ov, reinterpret_cast<ovec_t *>(reinterpret_cast<float *>(out) + idx));
            else
              *reinterpret_cast<ovec_t *>(reinterpret_cast<float *>(out) + idx) = ov;
          } else if (vec_ok) {
#pragma unroll
            for (int jj = 0; jj < VEC; jj += 2) {
              f64x2 ov = {val[jj], val[jj + 1]};
              if constexpr (OPT & 4)
                __builtin_nontemporal_store(ov, reinterpret_cast<f64x2 *>(out + idx + jj));
              else
                *reinterpret_cast<f64x2 *>(out + idx + jj) = ov;
            }
          } else {
#pragma unroll
            for (int jj = 0; jj < VEC; jj++)
              if (jj < rm.nvalid) scan_store(out, idx + jj, val[jj], out32);
          }
        }
      }
    if (!has_next) break;
    cur = nxt;
    id = nid;
  }
#undef PPALS_BUF_LOAD
}

// ---------------------------------------------------------------------------------------------
// WIDE suffix scan: 65..128 result columns in ONE pass over the tensor (fp32 storage).
//
// Above 64 columns the scan is no longer HBM-bound: AI = 2R/4 >= 32 flop/B is past the fp32
// matrix-core ridge (157 TFLOP/s / 8 TB/s = 20), so the tensor must be read ONCE for all columns —
// the narrow kernels above would run one launch per 64 columns, each re-reading it — and what
// bounds the launch is the MFMA pipe. One lane cannot hold 8 n-tiles of accumulators for 4 rows
// (128 fp32 + 256 fp64 registers), so the rows a wave owns shrink and the tensor tile goes through
// LDS: a workgroup (8 waves, 2 x 4) owns 64 rows x all NT n-tiles; per 16-column k-block it copies
// the 64 x 16 tensor tile (16-byte loads, 256 B contiguous per column) and the NT packed Khatri-Rao
// tiles (the same packing as the narrow kernels: one ds_read_b128 per lane = the A operands of four
// MFMAs) into LDS, two k-blocks per barrier, double-buffered; wave (wm, wn) multiplies rows
// [32 wm, 32 wm + 32) by n-tiles 2 wn, 2 wn + 1: four accumulator tiles (16 fp32 + 32 fp64
// registers), so that FOUR waves share a SIMD: one wave's fp32 -> fp64 flush, barrier or LDS wait is
// covered by the others' matrix work (measured on the 4-wave form, profiles/r06f: the flush cost 10 %,
// barrier + staging 11 %, the loads 7 % of a launch whose bare LDS + MFMA loop runs at 0.90 of the
// fp32 matrix-core peak). Same numerics as the narrow kernels: exact fp32 products on
// v_mfma_f32_16x16x4_f32, chains of <= 64 terms, fp64 beyond.
// (At 64 columns and below the same structure LOSES to the register-streaming kernels above — cfg4's
// two-tile scan 22.4 against 17.6 ms, cfg2's 1.18 against 1.01 ms, R = 40 1.91 against 1.75, R = 64 2.24
// against 2.07: profiles/r06k_lds_staged_scan_below_64_columns.txt — so it is instantiated for 5..8
// n-tiles only; the template itself takes 1..8.)
// Preconditions (launcher): M % 4 == 0, M >= 4, V 16-byte aligned, 5 <= NT <= 8.
template <int NT, int OPT = 0>
__global__ __launch_bounds__(512) void k_scan_wide(
    const float *__restrict__ V, int64_t M, int64_t K, int64_t batch_stride,
    const float *__restrict__ P, int n_mtiles, int nsplit, int kb_per_split, int nkb,
    double *__restrict__ out, int64_t out_nstride, int64_t out_split_stride,
    int64_t out_batch_stride, int ncols, int out32, int64_t row_ld = 0, int64_t row_valid = 0) {
  constexpr int BM = 64, BK = 16;  // BK: one k-block = 16 columns = four k-quads
  constexpr int PITCH = BM + 16;   // floats; k rows 4q+g of one quad land in distinct bank groups
  constexpr int NTA = (NT + 3) / 4;  // n-tiles per wave (the four column groups of waves)
  // a STAGE = two k-blocks (32 MFMAs per wave between barriers); LDS: two stage buffers
  constexpr int VSZ = BK * PITCH, PSZ = 512 * 4;  // floats per k-block (512 items: both loader items
                                                  // of every thread land inside the block's image)
  __shared__ __attribute__((aligned(16))) float Vs[2][2 * VSZ];
  __shared__ __attribute__((aligned(16))) float Ps[2][2 * PSZ];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: descriptors and branches below)
  const int g = lane >> 4, j16 = lane & 15;
  const int wm = wave & 1, wn = wave >> 1;
  const int nt0 = NTA * wn, ntc = max(0, min(NTA, NT - nt0));  // (NT = 7: 2, 2, 2, 1 tiles)

  unsigned b = blockIdx.x;
  const int mtile = (int)(b % (unsigned)n_mtiles);
  b /= (unsigned)n_mtiles;
  const int split = (int)(b % (unsigned)nsplit);
  const int64_t batch = b / (unsigned)nsplit;
  const int64_t m0 = (int64_t)mtile * BM;
  const int kb0 = split * kb_per_split;
  const int kb1 = min(nkb, kb0 + kb_per_split);
  const int nb = kb1 - kb0;
  if (nb <= 0) return;  // (workgroup-uniform; the launcher makes no empty split)
  const int nstages = (nb + 1) >> 1;

  // loader roles, per stage: tensor tile — thread (mq = tid & 15, kk = (tid >> 4) & 15, hv = tid >> 8)
  // copies rows 4mq..4mq+3 of column kk of k-block hv; Khatri-Rao tiles — float4 item tid of the 512
  // item slots of each of the two k-blocks (NT * 64 of them are real).
  // BUFFER loads (wave-uniform descriptor + loop-invariant lane offset + scalar offset, as in
  // k_scan_suffix_buf): no address temporaries in VGPRs, so nothing aliases a register that a load
  // still in flight will write, and every load is unconditional, so the compiler's COUNTED vmcnt
  // waits keep the ring in flight. A block's descriptor ends at the last valid k column (the ragged
  // tail reads zeros in hardware) and has no records at all past the split's last block.
  const int mq = tid & 15, kk = (tid >> 4) & 15, hv = wave >> 2;
  const int64_t m_ld = min(m0 + 4 * mq, M - 4);  // clamped: rows past the edge re-read the last ones
  const char *__restrict__ vbase = (const char *)(V + batch * batch_stride);
  const int voffV = (int)(((int64_t)kk * M + m_ld) * 4);
  const int64_t block_bytes = (int64_t)BK * M * 4, total_bytes = K * M * 4;
  const int voffP = tid * 16;
  const int vdst = hv * VSZ + kk * PITCH + 4 * mq;
  // (a wave whose 64 item slots lie past the block's NT * 64 items gets a descriptor without records:
  // its loads return zeros without touching memory — no branch around a load, the waits stay counted)
  const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(
      (void *)P, 0, wave < NT ? (int)((int64_t)nkb * NT * 1024) : 0, 0x00020000);
  constexpr int AUXV = (OPT & 1) ? 2 : 0;  // nt: streamed once

  // register ring: two slot sets of one stage each. Stage j + 2 is requested at the top of stage j
  // (into the set stage j vacated when it went to LDS) and written to LDS at the end of stage j + 1:
  // two stages = ~1.7 us of matrix work between a request and its use (HBM latency under load ~2 us).
  f32x4 gv[2], gp[2][2];
#define PPALS_WIDE_LOAD(set_, st_)                                                                     \
  {                                                                                                    \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++) {                                                 \
      const int kb_ = kb0 + min(2 * (st_) + h_, nb - 1);                                               \
      gp[set_][h_] = __builtin_bit_cast(                                                               \
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcP, voffP, kb_ * (NT * 1024), 0));           \
    }                                                                                                  \
    const int i_ = 2 * (st_) + hv;             /* block of the split; past its end: no records */      \
    const int64_t boff_ = (int64_t)(kb0 + min(i_, nb - 1)) * block_bytes;                              \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                              \
        (void *)(vbase + boff_), 0, i_ < nb ? (int)min(total_bytes - boff_, block_bytes) : 0,          \
        0x00020000);                                                                                   \
    gv[set_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, voffV, 0, AUXV));  \
  }
#define PPALS_WIDE_STAGE(set_, buf_)                                                         \
  {                                                                                          \
    *reinterpret_cast<f32x4 *>(&Vs[buf_][vdst]) = gv[set_];                                  \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++)                                         \
        *reinterpret_cast<f32x4 *>(&Ps[buf_][h_ * PSZ + 4 * tid]) = gp[set_][h_];            \
  }

  f32x4 acc[2][NTA];
  double acc64[2][NTA][4];
#pragma unroll
  for (int ms = 0; ms < 2; ms++)
#pragma unroll
    for (int i = 0; i < NTA; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        acc[ms][i][r] = 0;
        acc64[ms][i][r] = 0;
      }

  PPALS_WIDE_LOAD(0, 0);
  PPALS_WIDE_LOAD(1, min(1, nstages - 1));
  PPALS_WIDE_STAGE(0, 0);
  __syncthreads();
  // two stages (= 4 k-blocks = chains of 64 terms) per trip: slot sets and LDS buffers are
  // compile-time constants; fp32 -> fp64 after each trip
  for (int j0 = 0; j0 < nstages; j0 += 2) {
#pragma unroll
    for (int sp = 0; sp < 2; sp++) {
      const int j = j0 + sp;
      if (j < nstages) {  // workgroup-uniform
        // the slot set of stage j went to LDS one stage ago: refill it with stage j + 2 (past the
        // split's last stage: that stage again, harmless)
        PPALS_WIDE_LOAD(sp, min(j + 2, nstages - 1));
#pragma unroll
        for (int h = 0; h < 2; h++) {
          // B operands: tensor values of this wave's two 16-row strips, four k-quads
          float vb[2][4];
          const float *vs = &Vs[sp][h * VSZ + g * PITCH + 32 * wm + j16];
#pragma unroll
          for (int q = 0; q < 4; q++)
#pragma unroll
            for (int ms = 0; ms < 2; ms++) vb[ms][q] = vs[4 * q * PITCH + 16 * ms];
          // A operands of all this wave's n-tiles first, then k-quad by k-quad over the 2 * ntc
          // accumulator tiles: consecutive MFMAs never touch the same accumulator
          f32x4 pa[NTA];
#pragma unroll
          for (int t = 0; t < NTA; t++)
            pa[t] = *reinterpret_cast<const f32x4 *>(
                &Ps[sp][h * PSZ + (min(nt0 + t, NT - 1) * 64 + lane) * 4]);
#pragma unroll
          for (int q = 0; q < 4; q++)
#pragma unroll
            for (int t = 0; t < NTA; t++) {
              if (t >= ntc) break;  // wave-uniform (the last column group of an odd NT)
#pragma unroll
              for (int ms = 0; ms < 2; ms++)
                acc[ms][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[t][q], vb[ms][q], acc[ms][t], 0, 0, 0);
            }
        }
        // stage j + 1 (requested two stages ago) -> the other LDS buffer
        PPALS_WIDE_STAGE(sp ^ 1, sp ^ 1);
        __syncthreads();
      }
    }
#pragma unroll
    for (int ms = 0; ms < 2; ms++)
#pragma unroll
      for (int t = 0; t < NTA; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          acc64[ms][t][r] += (double)acc[ms][t][r];
          acc[ms][t][r] = 0;
        }
  }
#undef PPALS_WIDE_LOAD
#undef PPALS_WIDE_STAGE

  // epilogue: lane (j16, g) holds row m of strip ms and columns 16 nt + 4 g + r
  const int64_t obase = split * out_split_stride + batch * out_batch_stride;
#pragma unroll
  for (int ms = 0; ms < 2; ms++) {
    const int64_t m = m0 + 32 * wm + 16 * ms + j16;
    const ScanRowMap rm = scan_row_map<1>(m, M, row_ld, row_valid);
    if (rm.nvalid == 0) continue;
#pragma unroll
    for (int i = 0; i < NTA; i++) {
      if (i >= ntc) break;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = 16 * (nt0 + i) + 4 * g + r;
        if (n < ncols)
          scan_store(out, obase + (int64_t)n * out_nstride + rm.mo,
                     acc64[ms][i][r] + (double)acc[ms][i][r], out32);
      }
    }
  }
}

template <typename TV, int NT, int OPT = 0>
__global__ __launch_bounds__(256) void k_scan_prefix_fast(
    const TV *__restrict__ V, int64_t M, int64_t K, const TV *__restrict__ P, int mb_per_split,
    int nmb, double *__restrict__ out, int64_t out_kstride, int64_t out_nstride,
    int64_t out_split_stride, int ncols, int out32) {
  typedef ScanTraits<TV> TR;
  typedef typename TR::vec vec;
  typedef typename TR::acc acc_t;
  constexpr int VEC = TR::VEC;
  constexpr int MB = 4 * VEC;
  constexpr int U = 4;      // reduction blocks per step (U x 16 B per lane in flight, x2 prefetch)
  constexpr int FLUSH = 2;  // steps between fp32 -> fp64 flushes (chains of <= 16*VEC*... terms)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  // OPT bit 3: the 4 waves of a workgroup share ONE group of 16 columns and interleave along the
  // reduction rows (wave w takes steps w, w+4, ...), so a workgroup step reads 4 x U x 64 B =
  // 1 KiB contiguous per column instead of 256 B from 64 different columns (DRAM page locality);
  // the four partial tiles are combined through LDS at the end. grid.x = ceil(K/16) then.
  constexpr bool INTERLEAVE = (OPT & 8) != 0;
  const int64_t k0 = INTERLEAVE ? (int64_t)blockIdx.x * 16 : ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (k0 >= K) return;  // wave-uniform (block-uniform when INTERLEAVE)
  const int64_t k = k0 + j16;
  const bool k_ok = k < K;
  const int split = blockIdx.y;
  const int mb0 = split * mb_per_split;
  const int mb1 = min(nmb, mb0 + mb_per_split);
  // OPT bit 2: the MFMA wants lane = column + 16*rowgroup, but a load coalesces best when the 4
  // lanes of a quad read 64 contiguous bytes of ONE column. So load as (column = lane>>2,
  // rowgroup = lane&3) and move every dword to its MFMA lane with ds_bpermute (LDS crossbar,
  // no LDS memory): lane l pulls from lane 4*(l&15) + (l>>4).
  constexpr bool PERMUTE = (OPT & 4) != 0;
  const int lcol = PERMUTE ? (lane >> 2) : j16;
  const int lg = PERMUTE ? (lane & 3) : g;
  const int pull = (4 * j16 + g) * 4;  // byte address of the source lane for ds_bpermute
  const TV *__restrict__ vc = V + min(k0 + lcol, K - 1) * M + (int64_t)VEC * lg;  // clamped column
  const TV *__restrict__ pp = P + ((int64_t)g * 16 + j16) * VEC;
  const int64_t m_last = M - VEC - (int64_t)VEC * lg;  // largest valid block offset for this lane

  acc_t acc[2][NT];
  double acc64[TR::NEEDS_FLUSH ? NT : 1][4];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      acc[0][nt][r] = 0;
      acc[1][nt][r] = 0;
    }
    if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
      for (int r = 0; r < 4; r++) acc64[nt][r] = 0.0;
    }
  }

  vec cv[U], cb[U][NT];
  // step at block mb_: blocks mb_..mb_+U-1. Steps whose blocks all lie below min(mb1, M/MB) are
  // addressed as lane-constant pointer + wave-uniform offset. Otherwise (last step): blocks past
  // mb1 are clamped to mb1-1 with a ZERO operand, rows past M are clamped (operand zero there).
  const int mfull = (int)min((int64_t)mb1, M / MB);
#define PPALS_LOAD_STEP(mb_, vv_, bb_)                                                       \
  {                                                                                          \
    if ((mb_) + U <= mfull) {                                                                \
      _Pragma("unroll") for (int u = 0; u < U; u++) {                                        \
        vv_[u] = scan_ld<vec, OPT>(reinterpret_cast<const vec *>(vc + (int64_t)((mb_) + u) * MB)); \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bb_[u][nt] =                       \
            *reinterpret_cast<const vec *>(pp + ((int64_t)((mb_) + u) * NT + nt) * (4 * 16 * VEC)); \
      }                                                                                      \
    } else {                                                                                 \
      _Pragma("unroll") for (int u = 0; u < U; u++) {                                        \
        const int mbu_ = min((mb_) + u, mb1 - 1);                                            \
        const bool dup_ = (mb_) + u > mb1 - 1;                                               \
        const int64_t off_ = min((int64_t)mbu_ * MB, m_last);                                \
        vv_[u] = scan_ld<vec, OPT>(reinterpret_cast<const vec *>(vc + off_));                \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {                                  \
          bb_[u][nt] = *reinterpret_cast<const vec *>(                                       \
              pp + ((int64_t)mbu_ * NT + nt) * (4 * 16 * VEC));                              \
          if (dup_) {                                                                        \
            _Pragma("unroll") for (int e = 0; e < VEC; e++) bb_[u][nt][e] = (TV)0;           \
          }                                                                                  \
        }                                                                                    \
      }                                                                                      \
    }                                                                                        \
  }
  constexpr int STRIDE = INTERLEAVE ? 4 * U : U;  // distance between this wave's steps
  const int mstart = INTERLEAVE ? mb0 + wave * U : mb0;
  if (mstart < mb1) PPALS_LOAD_STEP(mstart, cv, cb);
  for (int mc = mstart; mc < mb1; mc += STRIDE * FLUSH) {
    const int me = min(mb1, mc + STRIDE * FLUSH);
    for (int mb = mc; mb < me; mb += STRIDE) {
      vec nv[U], nb[U][NT];
      const int mn = min(mb + STRIDE, mb1 - 1);
      PPALS_LOAD_STEP(mn, nv, nb);
      if constexpr (PERMUTE) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          typedef int dwords_t __attribute__((ext_vector_type(4)));
          dwords_t d = __builtin_bit_cast(dwords_t, cv[u]);
#pragma unroll
          for (int e = 0; e < 4; e++) d[e] = __builtin_amdgcn_ds_bpermute(pull, d[e]);
          cv[u] = __builtin_bit_cast(vec, d);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int jj = 0; jj < VEC; jj++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
            acc[jj & 1][nt] = TR::mfma(cb[u][nt][jj], cv[u][jj], acc[jj & 1][nt]);
#pragma unroll
      for (int u = 0; u < U; u++) {
        cv[u] = nv[u];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) cb[u][nt] = nb[u][nt];
      }
    }
    if constexpr (TR::NEEDS_FLUSH) {
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          acc64[nt][r] += (double)acc[0][nt][r] + (double)acc[1][nt][r];
          acc[0][nt][r] = 0;
          acc[1][nt][r] = 0;
        }
    }
  }
#undef PPALS_LOAD_STEP

  const int64_t obase = split * out_split_stride;
  double val[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if constexpr (TR::NEEDS_FLUSH)
        val[nt][r] = acc64[nt][r];
      else
        val[nt][r] = (double)acc[0][nt][r] + (double)acc[1][nt][r];
    }
  if constexpr (INTERLEAVE) {
    __shared__ double red[3][NT][4][64];
    if (wave > 0) {
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave - 1][nt][r][lane] = val[nt][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        val[nt][r] += (red[0][nt][r][lane] + red[1][nt][r][lane]) + red[2][nt][r][lane];
  }
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int n = 16 * nt + TR::row(lane, r);
      if (n < ncols && k_ok)
        scan_store(out, obase + (int64_t)n * out_nstride + k * out_kstride, val[nt][r], out32);
    }
}

// ---------------------------------------------------------------------------------------------
// Rank-structured stream (K10, build_V / [diffV]): one pass over V[m,k] against the model
// vhat[m,k] = sum_r Q[m,r] P[k,r] (Q, P = Khatri-Rao products of the two halves of the factors,
// fp64), nothing materialised:
//   MODE 0  V = vhat                      (build_V, common.cxx:135-197)
//   MODE 1  partial[wg] = sum (V - vhat)^2   (als_CP.cxx:183-187)
//   MODE 2  partial[wg] = sum V^2
// The model tile comes off the matrix cores in fp64 (v_mfma_f64_16x16x4_f64, contraction over r):
// fp32 products would carry ~1e-7 of |vhat| and drown a converged residual (~3e-8 of |V| with fp32
// tensor storage). Tiling = the suffix scan's: a workgroup covers 256 consecutive rows (1 KiB of
// every column), a lane owns VEC consecutive rows and the columns k = 16*kb + 4*u + g of a block:
//   A[i = lane&15][kk = lane>>4] = P[16*kb + i, 4*rb + kk]   (packed: one 8-byte load per lane)
//   B[kk = lane>>4][j = lane&15] = Q[m0 + VEC*j + jj, 4*rb + kk]   (tile-invariant registers)
//   D[i = (lane>>4) + 4*reg][j]  -> the lane's element (row VEC*j + jj, column 16*kb + 4*reg + g)
// which is exactly where its 16-byte tensor loads put V. 2*ceil(R/4)*4 flops per element: at the
// 46 TFLOP/s this instruction sustains on an MI355X (tools/mfma64_rate.hip; not the data sheet's
// 78.6) the pipe is busy 0.83 ms per 6.4 GB at R = 10, beside 0.96 ms for the read alone — the
// kernel is bound by both (1.31 ms). Needs M % VEC == 0, R <= 32.
// REM (residual only): the last REM = R mod 4 (1 or 2) ranks are NOT padded to a fourth contraction step
// on the matrix cores — a step costs 4 instructions of ~110 cycles per 1024 elements — but added by REM
// multiply-adds per element on the vector pipe, their P columns waiting in LDS (16 * kb_per_chunk * REM
// doubles of dynamic LDS), their Q columns in registers: R = 10 is two steps + two ranks instead of three
// steps (the matrix pipe 0.57 ms instead of 0.86 ms per 6.4 GB). RB counts the matrix-core steps only.
template <typename TV, int MODE, int MAXRB, int REM = 0>
__global__ __launch_bounds__(256) void k_rank_mfma(TV *__restrict__ V, int64_t M, int64_t K,
                                                   const double *__restrict__ Q,
                                                   const double *__restrict__ Ppk, int R, int RB,
                                                   int kb_per_chunk, int nkb,
                                                   double *__restrict__ partial,
                                                   const double *__restrict__ Praw = nullptr) {
  typedef ScanTraits<TV> TR;
  typedef typename TR::vec vec;
  constexpr int VEC = TR::VEC;  // MAXRB: 4-wide contraction steps held in registers (R <= 4*MAXRB)
  static_assert(REM == 0 || MODE == 1, "the vector-pipe remainder belongs to the residual");
  __shared__ double red[4];
  extern __shared__ double rank_ps[];  // REM > 0: P[k, R - REM + t] at [(k - 16 kb0) * REM + t]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * VEC);
  const bool live = m0 < M;  // wave-uniform
  const int64_t m = m0 + (int64_t)VEC * j16;
  const bool row_ok = live && m < M;
  const int64_t m_ld = live ? min(m, M - VEC) : 0;
  const int kb0 = blockIdx.y * kb_per_chunk, kb1 = min(nkb, kb0 + kb_per_chunk);
  double bq[VEC][MAXRB];
  if constexpr (MODE != 2) {
#pragma unroll
    for (int jj = 0; jj < VEC; jj++)
#pragma unroll
      for (int rb = 0; rb < MAXRB; rb++) {
        const int r = 4 * rb + g;
        bq[jj][rb] = (rb < RB && r < R - REM && live) ? Q[m_ld + jj + M * (int64_t)r] : 0.0;
      }
  }
  double qrem[VEC][REM > 0 ? REM : 1];
  if constexpr (REM > 0) {
#pragma unroll
    for (int jj = 0; jj < VEC; jj++)
#pragma unroll
      for (int t = 0; t < REM; t++) qrem[jj][t] = live ? Q[m_ld + jj + M * (int64_t)(R - REM + t)] : 0.0;
    for (int e = threadIdx.x; e < (kb1 - kb0) * 16 * REM; e += blockDim.x) {
      const int64_t k = (int64_t)kb0 * 16 + e / REM;
      rank_ps[e] = k < K ? Praw[k + K * (int64_t)(R - REM + e % REM)] : 0.0;
    }
    __syncthreads();
  }
  double acc = 0.0;
  TV *__restrict__ vp = V + m_ld;
  // register double buffer: the tensor columns and the packed P values of block kb+1 are
  // requested before block kb is multiplied (the last block is simply requested twice)
  vec cv[4];
  double ca[MAXRB];
#define PPALS_RANK_LOAD(kb_, vv_, aa_)                                                        \
  {                                                                                           \
    if constexpr (MODE != 0) {                                                                \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                                         \
        const int64_t k_ = min((int64_t)(kb_)*16 + 4 * u + g, K - 1);                         \
        vv_[u] = __builtin_nontemporal_load(reinterpret_cast<const vec *>(vp + k_ * M));      \
      }                                                                                       \
    }                                                                                         \
    if constexpr (MODE != 2) {                                                                \
      _Pragma("unroll") for (int rb = 0; rb < MAXRB; rb++) if (rb < RB) aa_[rb] =             \
          Ppk[(((int64_t)(kb_)*RB + rb) * 4 + g) * 16 + j16];                                 \
    }                                                                                         \
  }
  if (live && kb0 < kb1) PPALS_RANK_LOAD(kb0, cv, ca);
  for (int kb = kb0; kb < kb1 && live; kb++) {
    vec nv[4];
    double na[MAXRB];
    const int kn = min(kb + 1, kb1 - 1);
    PPALS_RANK_LOAD(kn, nv, na);
    if constexpr (MODE == 0) {
      f64x4 d[VEC];
#pragma unroll
      for (int jj = 0; jj < VEC; jj++) d[jj] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int rb = 0; rb < MAXRB; rb++) {
        if (rb < RB) {
#pragma unroll
          for (int jj = 0; jj < VEC; jj++)
            d[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[rb], bq[jj][rb], d[jj], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t k = (int64_t)kb * 16 + 4 * u + g;
        if (k < K && row_ok) {
          vec o;
#pragma unroll
          for (int jj = 0; jj < VEC; jj++) o[jj] = (TV)d[jj][u];
          *reinterpret_cast<vec *>(V + m + k * M) = o;
        }
      }
    } else {
      // two rows of the lane at a time: two model accumulators live instead of VEC (16 registers
      // fewer: with the three-step instantiation the residual of R <= 12 runs 4 waves per SIMD)
#pragma unroll
      for (int jp = 0; jp < VEC; jp += 2) {
        f64x4 d0 = f64x4{0.0, 0.0, 0.0, 0.0}, d1 = d0;
        if constexpr (MODE == 1) {
#pragma unroll
          for (int rb = 0; rb < MAXRB; rb++) {
            if (rb < RB) {
              d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[rb], bq[jp][rb], d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[rb], bq[jp + 1][rb], d1, 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int64_t k = (int64_t)kb * 16 + 4 * u + g;
          const bool ok = k < K && row_ok;  // beyond K the loads were clamped: excluded here
          double e0 = (double)cv[u][jp] - d0[u], e1 = (double)cv[u][jp + 1] - d1[u];
          if constexpr (REM > 0) {
#pragma unroll
            for (int t = 0; t < REM; t++) {
              const double pk = rank_ps[((kb - kb0) * 16 + 4 * u + g) * REM + t];
              e0 -= qrem[jp][t] * pk;
              e1 -= qrem[jp + 1][t] * pk;
            }
          }
          const double e2 = e0 * e0 + e1 * e1;
          acc += ok ? e2 : 0.0;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) cv[u] = nv[u];
#pragma unroll
    for (int rb = 0; rb < MAXRB; rb++) ca[rb] = na[rb];
  }
#undef PPALS_RANK_LOAD
  if constexpr (MODE != 0) {
    acc = [&]() {
      double v = acc;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      return v;
    }();
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
      partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// The residual for ranks above 32 (the reference CLI's default rank is s/2): the model tile of a
// column block needs R/4 contraction steps, too many operand registers for one wave. The four
// waves of a workgroup take a QUARTER of the rank blocks each on the SAME 16*VEC rows (their share
// of Q stays in registers: 4 * MAXRBW doubles per row), leave their partial model tiles in LDS
// (double-buffered: one barrier per column block) and each finishes one of the four column quads
// of the block — so every tensor element is loaded once, by the wave that subtracts it. The
// partial tiles are added in wave order: deterministic. MFMA-bound (2 M K R fp64 flops).
// MODE 1: partial[blk] = the block's share of ||V - model||^2; MODE 0: V = model (tensor generation).
template <typename TV, int MODE, int MAXRBW>
__global__ __launch_bounds__(256) void k_rank_split(TV *__restrict__ V, int64_t M, int64_t K,
                                                    const double *__restrict__ Q,
                                                    const double *__restrict__ Ppk, int R, int RB,
                                                    int kb_per_chunk, int nkb,
                                                    double *__restrict__ partial) {
  typedef ScanTraits<TV> TR;
  typedef typename TR::vec vec;
  constexpr int VEC = TR::VEC;
  __shared__ double sd[2][4][VEC][4][64];  // [buffer][wave][jj][column quad][lane]
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j16 = lane & 15;
  const int64_t m0 = (int64_t)blockIdx.x * (16 * VEC);  // all four waves: the same rows
  const int64_t m = m0 + (int64_t)VEC * j16;
  const bool row_ok = m < M;
  const int64_t m_ld = min(m, M - VEC);
  const int kb0 = blockIdx.y * kb_per_chunk, kb1 = min(nkb, kb0 + kb_per_chunk);
  const int rbw = (RB + 3) / 4;  // rank blocks per wave
  const int rb0 = wave * rbw;
  double bq[VEC][MAXRBW];
#pragma unroll
  for (int jj = 0; jj < VEC; jj++)
#pragma unroll
    for (int i = 0; i < MAXRBW; i++) {
      const int rb = rb0 + i, r = 4 * rb + g;
      bq[jj][i] = (i < rbw && rb < RB && r < R) ? Q[m_ld + jj + M * (int64_t)r] : 0.0;
    }
  double acc = 0.0;
  const TV *__restrict__ vp = V + m_ld;
  vec cv = {};
  double ca[MAXRBW];
#define PPALS_SPLIT_LOAD(kb_, vv_, aa_)                                                   \
  {                                                                                       \
    if constexpr (MODE == 1) {                                                            \
      const int64_t k_ = min((int64_t)(kb_)*16 + 4 * wave + g, K - 1);                    \
      vv_ = __builtin_nontemporal_load(reinterpret_cast<const vec *>(vp + k_ * M));       \
    }                                                                                     \
    _Pragma("unroll") for (int i = 0; i < MAXRBW; i++) {                                  \
      const int rb = rb0 + i;                                                             \
      aa_[i] = (i < rbw && rb < RB) ? Ppk[(((int64_t)(kb_)*RB + rb) * 4 + g) * 16 + j16] : 0.0; \
    }                                                                                     \
  }
  if (kb0 < kb1) PPALS_SPLIT_LOAD(kb0, cv, ca);
  for (int kb = kb0; kb < kb1; kb++) {
    vec nv = {};  // (read only when MODE == 1)
    double na[MAXRBW];
    const int kn = min(kb + 1, kb1 - 1);
    PPALS_SPLIT_LOAD(kn, nv, na);
    f64x4 d[VEC];
#pragma unroll
    for (int jj = 0; jj < VEC; jj++) d[jj] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < MAXRBW; i++)
      if (i < rbw) {
#pragma unroll
        for (int jj = 0; jj < VEC; jj++)
          d[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[i], bq[jj][i], d[jj], 0, 0, 0);
      }
    const int buf = (kb - kb0) & 1;
#pragma unroll
    for (int jj = 0; jj < VEC; jj++)
#pragma unroll
      for (int u = 0; u < 4; u++) sd[buf][wave][jj][u][lane] = d[jj][u];
    __syncthreads();
    // this wave finishes column quad u = wave: columns 16 kb + 4 wave + g
    const int64_t k = (int64_t)kb * 16 + 4 * wave + g;
    double e2 = 0.0;
    vec o;
#pragma unroll
    for (int jj = 0; jj < VEC; jj++) {
      const double model = ((sd[buf][0][jj][wave][lane] + sd[buf][1][jj][wave][lane]) +
                            sd[buf][2][jj][wave][lane]) + sd[buf][3][jj][wave][lane];
      if constexpr (MODE == 1) {
        const double e = (double)cv[jj] - model;
        e2 += e * e;
      } else {
        o[jj] = (TV)model;
      }
    }
    if constexpr (MODE == 1) {
      acc += (k < K && row_ok) ? e2 : 0.0;
    } else {
      if (k < K && row_ok) *reinterpret_cast<vec *>(V + m + k * M) = o;
    }
    cv = nv;
#pragma unroll
    for (int i = 0; i < MAXRBW; i++) ca[i] = na[i];
  }
#undef PPALS_SPLIT_LOAD
  if constexpr (MODE == 1) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
      partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// Ppk[((kb*RB + rb)*4 + g)*16 + i] = P[16*kb + i, 4*rb + g]  (zero beyond K / R)
__global__ void k_rank_pack(const double *__restrict__ P, int64_t K, int R, int RB, int nkb,
                            double *__restrict__ Ppk) {
  const int64_t total = (int64_t)nkb * RB * 64;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e & 15), g = (int)((e >> 4) & 3);
    const int64_t t = e >> 6;
    const int rb = (int)(t % RB);
    const int64_t kb = t / RB;
    const int64_t k = kb * 16 + i;
    const int r = 4 * rb + g;
    Ppk[e] = (k < K && r < R) ? P[k + K * (int64_t)r] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// Khatri-Rao operand packing (layouts described at the top of this file).
struct KrpArgs {
  const double *ptr[MAX_ORDER];
  int64_t rows[MAX_ORDER];
  int64_t ld[MAX_ORDER];
  int nf;
};

template <typename TV>
__global__ void k_krp_pack(TV *__restrict__ P, int nblk, int NT, int prefix_layout, KrpArgs a,
                           int64_t J, int col0, int ncols) {
  constexpr int VEC = ScanTraits<TV>::VEC;
  const int64_t total = (int64_t)nblk * NT * 4 * 16 * VEC;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = e;
    const int u = (int)(t % VEC);
    t /= VEC;
    const int n = (int)(t % 16);
    t /= 16;
    const int g = (int)(t % 4);
    t /= 4;
    const int nt = (int)(t % NT);
    const int64_t blk = t / NT;
    const int64_t j = blk * 4 * VEC + (prefix_layout ? VEC * g + u : 4 * u + g);
    const int c = 16 * nt + n;
    double v = 0.0;
    if (j < J && c < ncols) {
      v = 1.0;
      int64_t rem = j;
      for (int f = 0; f < a.nf; f++) {
        const int64_t jf = rem % a.rows[f];
        rem /= a.rows[f];
        v *= a.ptr[f][jf + a.ld[f] * (col0 + c)];
      }
    }
    P[e] = (TV)v;
  }
}

// out[m'*out_mstride + out_rstride*n] = sum_s slab[s*split_stride + n*M + m],  n < ncols;
// m' = m, or the compact row of stored row m of a padded layout (scan_row_map)
__global__ void k_slab_reduce(const double *__restrict__ slab, int nsplit, int64_t split_stride,
                              int64_t M, int ncols, double *__restrict__ out, int64_t out_mstride,
                              int64_t out_rstride, int out32, int64_t row_ld, int64_t row_valid,
                              int64_t slab_batch_stride = 0, int64_t out_batch_stride = 0) {
  // blockIdx.y = batch (the K-split of a batched scan: slabs [split][batch][ncols][M])
  slab += (int64_t)blockIdx.y * slab_batch_stride;
  const int64_t obatch = (int64_t)blockIdx.y * out_batch_stride;
  const int64_t total = M * ncols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = e % M;
    const int64_t n = e / M;
    const ScanRowMap rm = scan_row_map<1>(m, M, row_ld, row_valid);
    if (rm.nvalid == 0) continue;
    double s = 0;
    for (int sp = 0; sp < nsplit; sp++) s += slab[sp * split_stride + e];
    scan_store(out, obatch + rm.mo * out_mstride + out_rstride * n, s, out32);
  }
}

}  // namespace ppals
