// engine.cpp — host control flow of the CP ALS sweep engine (dimension tree + pairwise
// perturbation) over abstract device ops. Mirrors the reference's orchestration:
//   Construct_Dimension_Tree  common.cxx:225-270     -> build_tree
//   mttkrp_map_DT             common.cxx:20-133      -> compute_node (fused per node, no s^(N-1)R
//                                                       intermediate, no copy of V)
//   alsCP_DT                  als_CP.cxx:127-320     -> run_dt / sweep_dt
//   Build_mttkrp_map          als_CP.cxx:352-409     -> pp_get
//   alsCP_DT_sub / PP_sub     als_CP.cxx:418-833     -> dt_sub / pp_sub
//   alsCP_PP                  als_CP.cxx:1082-1137   -> run_pp
//   alsCP_PP_partupdate       als_CP.cxx:852-1207    -> run_pp_partupdate
//   CPD<dtype,Optimizer>::als src/CP.cxx:100-186     -> run_class / update_modes
// Default sweep schedule: the multi-sweep dimension tree (MSDT) of the reference's class API
// (src/optimizer/cp_msdt_optimizer.cxx:172-207): ONE first-level contraction V x_r W_r is reused
// for the next N-1 mode updates, so an exact sweep costs N/(N-1) tensor scans instead of 2 — the
// very same ALS iterates (same update order, same normal equations), 1.5x fewer tensor bytes at
// N = 4. ppals_cp_set_schedule(0) selects the two-first-level-node tree of alsCP_DT instead.
// Multi-GPU: V is block-partitioned along mode 0; factors are replicated; each
// mode update combines the s x R partial MTTKRP rows over the communicator — one all-reduce plus
// the redundant fused update for small messages, reduce-scatter / row-block solve / all-gather
// otherwise (mode_update).
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

namespace ppals {

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ============================================================================ tensor helpers
int tensor_create(Ops &ops, Comm &comm, int order, const int64_t *glens, int dtype,
                  TensorDesc *out, std::string *err) {
  if (order < 2 || order > MAX_ORDER) {
    *err = "tensor order must be in [2, 8]";
    return -1;
  }
  TensorDesc t;
  t.order = order;
  t.dtype = dtype;
  int64_t rest = 1;
  for (int i = 0; i < order; i++) {
    if (glens[i] <= 0) {
      *err = "tensor extents must be positive";
      return -1;
    }
    t.glens[i] = glens[i];
    t.llens[i] = glens[i];
    if (i > 0) rest *= glens[i];
  }
  const int P = comm.size();
  const int64_t blk = block_rows(glens[0], P);
  t.row0 = blk * comm.rank();
  t.llens[0] = std::max<int64_t>(0, std::min(blk, glens[0] - t.row0));
  if (blk * (P - 1) >= glens[0]) {
    // decided from global quantities only, so EVERY rank returns the error (no rank is left
    // waiting in a collective): with row blocks of ceil(s0/P) the last rank would own no rows
    *err = "leading mode too short for this many ranks (the last rank would own no rows)";
    return -1;
  }
  t.nloc = t.llens[0] * rest;
  t.data = ops.alloc((size_t)t.nloc * dtype_size(dtype));
  *out = t;
  return 0;
}

static int64_t rest_of(const TensorDesc &V) {
  int64_t r = 1;
  for (int i = 1; i < V.order; i++) r *= V.glens[i];
  return r;
}

// device Khatri-Rao operands of the two halves of the mode set: Q over modes [0,h), P over [h,N)
static void split_krp(Ops &ops, const TensorDesc &V, int R, double *const *W, double *Q, double *P,
                      int64_t *M, int64_t *K) {
  const int N = V.order, h = N / 2;
  FactorRef fq[MAX_ORDER], fp[MAX_ORDER];
  int64_t m = 1, k = 1;
  for (int i = 0; i < h; i++) {
    fq[i].ptr = W[i] + (i == 0 ? V.row0 : 0);
    fq[i].rows = V.llens[i];
    fq[i].ld = V.glens[i];
    m *= V.llens[i];
  }
  for (int i = h; i < N; i++) {
    fp[i - h].ptr = W[i];
    fp[i - h].rows = V.glens[i];
    fp[i - h].ld = V.glens[i];
    k *= V.glens[i];
  }
  ops.krp(Q, fq, h, 0, R);
  ops.krp(P, fp, N - h, 0, R);
  *M = m;
  *K = k;
}
static void split_sizes(const TensorDesc &V, int64_t *M, int64_t *K) {
  const int N = V.order, h = N / 2;
  int64_t m = 1, k = 1;
  for (int i = 0; i < h; i++) m *= V.llens[i];
  for (int i = h; i < N; i++) k *= V.glens[i];
  *M = m;
  *K = k;
}

// `-tensor r` (test_ALS.cxx:275-286): V = [[W_true]], built on the device
void tensor_fill_cp(Ops &ops, const TensorDesc &V, int R, const double *Wtrue_flat) {
  const int N = V.order;
  std::vector<double *> W(N);
  const double *src = Wtrue_flat;
  for (int i = 0; i < N; i++) {
    size_t n = (size_t)V.glens[i] * R;
    W[i] = (double *)ops.alloc(n * sizeof(double));
    ops.h2d(W[i], src, n * sizeof(double));
    src += n;
  }
  int64_t M, K;
  split_sizes(V, &M, &K);
  double *Q = (double *)ops.alloc(sizeof(double) * M * R);
  double *Pm = (double *)ops.alloc(sizeof(double) * K * R);
  split_krp(ops, V, R, W.data(), Q, Pm, &M, &K);
  ops.fill_rank(V.data, V.dtype, M, K, Q, Pm, R);
  ops.sync();
  ops.free(Q);
  ops.free(Pm);
  for (auto p : W) ops.free(p);
}
void tensor_fill_uniform(Ops &ops, const TensorDesc &V, uint64_t seed, double lo, double hi) {
  ops.fill_uniform(V.data, V.dtype, V.llens[0], V.glens[0], V.row0, rest_of(V), seed, lo, hi);
  ops.sync();
}
void tensor_upload(Ops &ops, const TensorDesc &V, const double *host_full) {
  ops.upload_shard(V.data, V.dtype, host_full, V.llens[0], V.glens[0], V.row0, rest_of(V));
}
void tensor_download(Ops &ops, const TensorDesc &V, double *host_full) {
  ops.download_shard(V.data, V.dtype, host_full, V.llens[0], V.glens[0], V.row0, rest_of(V));
}
void tensor_fill_laplacian(Ops &ops, const TensorDesc &V, int ndigits, int s) {
  ops.fill_laplacian(V.data, V.dtype, V.llens[0], V.glens[0], V.row0, rest_of(V), ndigits, s);
  ops.sync();
}

// host-side counter RNG identical to the device / oracle generator
static inline uint64_t sm64_host(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline double u01_host(uint64_t seed, uint64_t idx) {
  uint64_t h = sm64_host(sm64_host(seed) ^ idx);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// Gen_collinearity (common.cxx:361-423): per mode j, the rank-i vector is re-drawn until its
// collinearity with every earlier vector of that mode lies in [col_min, col_max]; component i is
// weighted lambda_i = 0.2 + 0.6/R*(i+1) (folded into the mode-0 factor here).
void collinear_factors(const int64_t *lens, int N, int R, double col_min, double col_max,
                       uint64_t seed, double *Wflat) {
  double *W = Wflat;
  for (int j = 0; j < N; j++) {
    const int64_t s = lens[j];
    uint64_t draw = 0;
    for (int i = 0; i < R; i++) {
      double *vi = W + s * i;
      for (int attempt = 0; attempt < 100000; attempt++) {
        const uint64_t sd = seed + 7919ull * (uint64_t)j + 104729ull * (draw++);
        for (int64_t e = 0; e < s; e++) vi[e] = u01_host(sd, (uint64_t)e);
        bool ok = true;
        for (int k = 0; k < i && ok; k++) {
          const double *vk = W + s * k;
          double ip = 0, n1 = 0, n2 = 0;
          for (int64_t e = 0; e < s; e++) {
            ip += vi[e] * vk[e];
            n1 += vi[e] * vi[e];
            n2 += vk[e] * vk[e];
          }
          const double col = ip / (std::sqrt(n1) * std::sqrt(n2));
          if (col < col_min || col > col_max) ok = false;
        }
        if (ok) break;
      }
    }
    W += s * R;
  }
  for (int i = 0; i < R; i++) {
    const double lambda_ = 0.2 + 0.6 / R * (i + 1);
    for (int64_t e = 0; e < lens[0]; e++) Wflat[e + lens[0] * i] *= lambda_;
  }
}

void tensor_fill_collinear(Ops &ops, Comm &comm, const TensorDesc &V, int R, double col_min,
                           double col_max, double ratio_noise, uint64_t seed) {
  size_t tot = 0;
  for (int i = 0; i < V.order; i++) tot += (size_t)V.glens[i] * R;
  std::vector<double> W(tot);
  collinear_factors(V.glens, V.order, R, col_min, col_max, seed, W.data());
  tensor_fill_cp(ops, V, R, W.data());
  // V += ratio_noise * ||V|| / ||noise|| * noise, noise ~ U(-1,1)   (test_ALS.cxx:259-264)
  const double vnorm = tensor_norm(ops, comm, V);
  double *d = (double *)ops.alloc(sizeof(double));
  const uint64_t nseed = seed + 0x5eedull;
  ops.uniform_sumsq(V.llens[0], V.glens[0], V.row0, rest_of(V), nseed, -1.0, 1.0, d);
  if (comm.size() > 1) comm.allreduce_sum(d, 1);
  double nsq = 0;
  ops.d2h(&nsq, d, sizeof(double));
  ops.free(d);
  const double alpha = ratio_noise * vnorm / std::sqrt(nsq);
  ops.add_uniform_noise(V.data, V.dtype, V.llens[0], V.glens[0], V.row0, rest_of(V), nseed, -1.0,
                        1.0, alpha);
  ops.sync();
}

double tensor_norm(Ops &ops, Comm &comm, const TensorDesc &V) {
  int64_t M, K;
  split_sizes(V, &M, &K);
  double *d = (double *)ops.alloc(sizeof(double));
  ops.residual_sq(V.data, V.dtype, M, K, nullptr, nullptr, 0, d);
  if (comm.size() > 1) comm.allreduce_sum(d, 1);
  double h = 0;
  ops.d2h(&h, d, sizeof(double));
  ops.free(d);
  return std::sqrt(h);
}

// ============================================================================ CpEngine
CpEngine::CpEngine(Ops &ops, Comm &comm, const TensorDesc &V, int R)
    : ops_(ops), comm_(comm), V_(V), N_(V.order), R_(R), P_(comm.size()), rank_(comm.rank()) {
  dist_ = P_ > 1 || (force_comm_path() && !comm.is_self());
  if (R <= 0) throw std::runtime_error("ppals: rank must be positive");
  W_.resize(N_);
  gradW_.resize(N_);
  Wprev_.assign(N_, nullptr);
  Winit_.assign(N_, nullptr);
  dW_.assign(N_, nullptr);
  dM_.assign(N_, nullptr);
  Mm_.assign(N_, nullptr);
  for (int i = 0; i < N_; i++) {
    size_t n = (size_t)V_.glens[i] * R_ * sizeof(double);
    W_[i] = (double *)ops_.alloc(n);
    gradW_[i] = (double *)ops_.alloc(n);
    ops_.zero(W_[i], n);
    ops_.zero(gradW_[i], n);
    maxs_ = std::max(maxs_, V_.glens[i]);
    maxblk_ = std::max(maxblk_, block_rows(V_.glens[i], P_));
  }
  G_ = (double *)ops_.alloc(sizeof(double) * N_ * R_ * R_);
  S_ = (double *)ops_.alloc(sizeof(double) * R_ * R_);
  Sinv_ = (double *)ops_.alloc(sizeof(double) * R_ * R_);
  gradsq_ = (double *)ops_.alloc(sizeof(double) * MAX_ORDER);
  scal_ = (double *)ops_.alloc(sizeof(double) * 4 * MAX_ORDER);
  ops_.zero(gradsq_, sizeof(double) * MAX_ORDER);
  if (dist_) {
    size_t n = sizeof(double) * (size_t)maxblk_ * P_ * R_;
    sendbuf_ = (double *)ops_.alloc(n);
    recvbuf_ = (double *)ops_.alloc(sizeof(double) * (size_t)maxblk_ * R_);
    gatherbuf_ = (double *)ops_.alloc(n);
    ops_.zero(sendbuf_, n);
    ops_.zero(gatherbuf_, n);
  }
  if (const char *e = std::getenv("PPALS_COMM_SMALL_BYTES")) small_msg_bytes_ = std::atoll(e);
  if (const char *e = std::getenv("PPALS_TEST_BLOCKED_UPDATE")) test_blocks_ = std::atoi(e);
  if (N_ < 3) schedule_ = 0;
  if (const char *e = std::getenv("PPALS_PLACE_TUNE")) ms_tune_enabled_ = std::atoi(e) != 0;
  if (N_ >= 3) {  // multi-sweep structures exist for every session so the schedule can be switched
    ms_set_roots(ms_choose_roots());
    ms_scales_ = (double *)ops_.alloc(sizeof(double) * 32);
  }
  // The second resident layout is part of the session's set-up, like the tensor itself: built
  // here rather than at the first sweep, whose [dtime] would otherwise carry a one-off multi-GB
  // hipMalloc (≈ 1 s per 44 GB, several seconds on a fresh box) plus the transpose.
  if (V_.generation) tensor_gen_ = *V_.generation;
  {
    Layout nat;
    nat.ptr = V_.data;
    for (int m = 0; m < N_; m++) nat.order.push_back(m);
    lay_.push_back(nat);
  }
  // The block of the first-level intermediate is sized ONCE, for the largest root set plus the
  // slack the online placement choice moves inside (ms_start_step): growing it later would move
  // every root's result and void what the first visits measured (unequal mode extents,
  // [s/P, s, s, s] shards). A second candidate block is taken BEFORE the second resident layout, the
  // primary one after it: the layout's bytes lie between them. Nothing is launched or measured here.
  size_t xmax = 0;
  const bool place = schedule_ == 1 && N_ >= 3 && ms_tune_enabled_ && ms_X_slack() > 0;
  if (place) {
    for (int r = 0; r < N_; r++) {
      if (ms_set_excluded(r, ms_k_, ms_excl_)) continue;  // never a root set
      xmax = std::max(xmax, ms_X_bytes(r, ms_k_));
    }
    const size_t avail = ops_.mem_available();
    const double tensor_bytes = (double)V_.nloc * dtype_size(V_.dtype);
    if (avail == (size_t)-1 || (double)avail > 2.0 * tensor_bytes + 6.0 * (double)(xmax + ms_X_slack()) + 8e9)
      ms_X_alt_ = ops_.try_alloc(xmax + ms_X_slack());
  }
  ensure_transposed();
  if (place) {
    ms_X_base_ = big_alloc(xmax + ms_X_slack());
    ms_X_cap_ = xmax + ms_X_slack();
  }
  for (int i = 0; i < MAX_ORDER; i++) grad_replicated_[i] = !dist_;
  build_tree(0, N_ - 1, -1);
  leaf_.assign(N_, -1);
  for (size_t k = 0; k < nodes_.size(); k++)
    if (nodes_[k].lo == nodes_[k].hi) leaf_[nodes_[k].lo] = (int)k;
}

// 0 = the reference's two-first-level-node tree (alsCP_DT), 1 = multi-sweep tree (default)
void CpEngine::set_schedule(int schedule) {
  if (schedule != 0 && schedule != 1) throw std::runtime_error("ppals: unknown sweep schedule");
  schedule_ = (N_ < 3) ? 0 : schedule;
  for (auto &n : nodes_) n.valid = false;
  ms_invalidate();
}

CpEngine::~CpEngine() {
  try {
    ops_.sync();
  } catch (...) {
  }
  for (auto p : W_) ops_.free(p);
  for (auto p : gradW_) ops_.free(p);
  for (auto p : Wprev_) ops_.free(p);
  for (auto p : Winit_) ops_.free(p);
  for (auto p : dW_) ops_.free(p);
  for (auto p : dM_) ops_.free(p);
  for (auto p : Mm_) ops_.free(p);
  for (auto &n : nodes_) ops_.free(n.buf);
  lr_release();
  pp_clear();
  for (auto &kv : pp_pool_) ops_.free(kv.second.buf);
  ops_.free(pp_norms_);
  ops_.free(G_);
  ops_.free(S_);
  ops_.free(Sinv_);
  ops_.free(gradsq_);
  ops_.free(scal_);
  ops_.free(sendbuf_);
  ops_.free(recvbuf_);
  ops_.free(gatherbuf_);
  ops_.free(test_blkbuf_);
  ops_.free(Mbuf_);
  ops_.free(Qbuf_);
  ops_.free(Pbuf_);
  for (auto &l : lay_)
    if (l.owned) ops_.free(l.ptr);
  for (auto &ex : ms_place_)
    if (ex.timer >= 0) ops_.timer_read(ex.timer);
  ops_.free(ms_X_base_);
  ops_.free(ms_X_alt_);
  ops_.free(ms_scales_);
  for (auto &n : ms_nodes_) {
    ops_.free(n.t.buf);
    for (auto &t : n.tmp) ops_.free(t.buf);
  }
}

// Second resident layout of the tensor for the RIGHT first-level node: V viewed as the matrix
// [left modes (fastest) x right modes] is transposed once, so that the "cd" contraction streams
// rows of (c,d) contiguously — the same suffix-scan access pattern as the "ab" node (K1) instead
// of the column-strided prefix scan (K2). Costs one extra copy of V in HBM (288 GB are there for
// it; cfg2: +6.4 GB, cfg4: +102 GB) and one transpose per session; the bytes read per sweep do
// not change. Built once, when the session is created. PPALS_TRANSPOSED_COPY=0 or an allocation
// failure falls back to the prefix scan (and to single-mode root sets).
//
// Padded layouts (engine.h, Layout): the leading block of q modes is padded when that makes more
// root positions 128-B aligned than the plain order has (q < first aligned position) at a cost of
// at most PPALS_PAD_WASTE (default 3 %) extra bytes — s = 50: 2500 -> 2528 elements, +1.1 %.
// PPALS_PAD_LAYOUT=0 never pads, =1 pads whatever the tensor's size (tests); by default tensors
// below 100 MB are left alone. The padded copy in the tensor's own order is a THIRD copy of the
// tensor: built only if the allocation succeeds.
static int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

void CpEngine::fill_layout(const Layout &l) {
  // the layout is the tensor rotated by h modes: transposition of [first h modes | the rest]
  int h = l.order[0];
  const int64_t rows = h == 0 ? 1 : prod_ext(0, h - 1);
  const int64_t cols = h == 0 ? V_.nloc : prod_ext(h, N_ - 1);
  if (l.q == 0)
    ops_.transpose2d(V_.data, V_.dtype, rows, cols, l.ptr);
  else
    ops_.pad_layout(V_.data, V_.dtype, rows, cols, l.blk, l.ld, l.ptr);
}

void CpEngine::ensure_transposed() {
  if (vt_state_ != 0) return;
  vt_state_ = -1;
  const char *env = std::getenv("PPALS_TRANSPOSED_COPY");
  if (env && std::atoi(env) == 0) return;
  if (N_ < 3) return;
  const size_t esz = dtype_size(V_.dtype);
  const int64_t line = 128 / (int64_t)esz;  // elements per 128 B
  int pad_mode = -1;                         // auto
  if (const char *e = std::getenv("PPALS_PAD_LAYOUT")) pad_mode = std::atoi(e);
  const double max_waste = pad_mode == 1 ? 100.0 : 0.03;  // (forced: whatever the padding costs — tests)
  const bool may_pad = pad_mode == 1 || (pad_mode != 0 && (double)V_.nloc * esz >= 1e8);
  // leading block of a storage order worth padding: the smallest q whose padding is cheap, if the
  // plain order has no aligned position that early
  auto choose_pad = [&](Layout &l, int qmax) {
    l.q = 0;
    if (!may_pad) return;
    int q_nat = N_;
    int64_t b = 1;
    for (int p = 1; p < N_; p++) {
      b *= ext(l.order[p - 1]);
      if (b % line == 0) {
        q_nat = p;
        break;
      }
    }
    b = 1;
    for (int q = 1; q < q_nat && q <= qmax; q++) {
      b *= ext(l.order[q - 1]);
      const int64_t ld = round_up(b, line);
      if ((double)(ld - b) <= max_waste * (double)b) {
        l.q = q;
        l.blk = b;
        l.ld = ld;
        return;
      }
    }
  };
  auto bytes_of = [&](const Layout &l) {
    return l.q ? (size_t)l.ld * (size_t)(V_.nloc / l.blk) * esz : (size_t)V_.nloc * esz;
  };
  const int mid = (N_ - 1) / 2;
  Layout t;
  for (int m = mid + 1; m < N_; m++) t.order.push_back(m);
  for (int m = 0; m <= mid; m++) t.order.push_back(m);
  choose_pad(t, std::min(N_ - 2, N_ - 1 - mid));  // the block must lie inside the right half
  // An optional copy is taken only if what the session must still allocate fits behind it: the
  // first-level intermediate (+ placement slack), the level-1 scratch of the PP build of the same
  // size, tree nodes / pair operators (a few per cent of that) and a margin. (A Tucker session on
  // the same tensor, or a second CP session, may still find the device full: big_alloc then gives
  // the optional copies back.)
  int64_t min_ext = ext(0);
  for (int m = 1; m < N_; m++) min_ext = std::min(min_ext, ext(m));
  const size_t xbytes_max = (size_t)(V_.nloc / std::max<int64_t>(1, min_ext)) * R_ * esz;
  const size_t reserve = 3 * xbytes_max + ms_X_slack() + ((size_t)1 << 30);
  auto fits = [&](size_t bytes) {
    const size_t avail = ops_.mem_available();
    return avail == (size_t)-1 || avail >= bytes + reserve;
  };
  if (!fits(bytes_of(t))) return;
  t.ptr = ops_.try_alloc(bytes_of(t));
  if (!t.ptr) return;
  t.owned = true;
  t.bytes = bytes_of(t);
  fill_layout(t);
  lay_.push_back(t);
  vt_state_ = 1;
  Layout a;
  for (int m = 0; m < N_; m++) a.order.push_back(m);
  choose_pad(a, N_ - 2);
  if (a.q > 0 && fits(bytes_of(a))) {
    a.ptr = ops_.try_alloc(bytes_of(a));
    if (a.ptr) {
      a.owned = true;
      a.bytes = bytes_of(a);
      fill_layout(a);
      lay_.push_back(a);
    }
  }
}

// The scan that contracts modes first .. first+k-1 (cyclic) on the best resident layout: the set
// must be stored adjacently (no wrap-around) and, on a padded layout, behind the padded block.
// Among those, 128-B aligned columns of at least a tile of rows win, then the most rows in front
// of the set (long contiguous runs per reduction index; a leading set would be a column-strided
// prefix scan). false: no layout stores the set adjacently.
bool CpEngine::plan_scan(int first, int k, bool natural_only, ScanPlan &plan) {
  auto in_set = [&](int m) { return ((m - first + N_) % N_) < k; };
  const int64_t line = 128 / (int64_t)dtype_size(V_.dtype);
  const Layout *best = nullptr;
  int best_p0 = 0;
  int64_t best_L = 0;
  bool best_al = false;
  const size_t nlay = natural_only ? 1 : lay_.size();
  for (size_t li = 0; li < nlay; li++) {
    const Layout &l = lay_[li];
    int p0 = -1;
    for (int p = 0; p < N_; p++)
      if (in_set(l.order[p])) {
        p0 = p;
        break;
      }
    int len = 0;
    while (p0 + len < N_ && in_set(l.order[p0 + len])) len++;
    if (len != k || (l.q > 0 && p0 < l.q)) continue;
    int64_t L = 1;
    for (int p = 0; p < p0; p++) L *= ext(l.order[p]);
    const bool al = L >= 256 && (l.q > 0 || L % line == 0);
    if (!best || (al && !best_al) ||
        (al == best_al && (L > best_L || (L == best_L && l.q > 0 && best->q == 0)))) {
      best = &l;
      best_p0 = p0;
      best_L = L;
      best_al = al;
    }
  }
  if (!best) return false;
  plan = ScanPlan();
  plan.lay = best;
  for (int p = 0; p < N_; p++) {
    const int m = best->order[p];
    if (p < best_p0) {
      plan.Lc *= ext(m);
      plan.kept.push_back(m);
    } else if (p < best_p0 + k) {
      plan.J *= ext(m);
      plan.set.push_back(m);
    } else {
      plan.T *= ext(m);
      plan.kept.push_back(m);
    }
  }
  plan.L = plan.Lc;
  if (best->q > 0) {
    plan.L = plan.Lc / best->blk * best->ld;
    plan.pad.ld = best->ld;
    plan.pad.valid = best->blk;
  }
  return true;
}

// The tensor handle stays writable while sessions exist (ppals_tensor_fill_*/upload). Everything
// a session derived from the old contents — the second resident layout, tree nodes, the
// multi-sweep intermediate, PP operators — is rebuilt / dropped when the generation moved.
// Called wherever a session is about to read the tensor.
void CpEngine::check_tensor_generation() {
  if (!V_.generation || *V_.generation == tensor_gen_) return;
  tensor_gen_ = *V_.generation;
  for (const auto &l : lay_)
    if (l.owned) fill_layout(l);
  for (auto &n : nodes_) n.valid = false;
  ms_invalidate();
  pp_clear();
}

FactorRef CpEngine::fref(int m, double *const *W) const {
  FactorRef f;
  f.ptr = W[m] + (m == 0 ? V_.row0 : 0);
  f.rows = ext(m);
  f.ld = V_.glens[m];
  return f;
}
int64_t CpEngine::prod_ext(int lo, int hi) const {
  int64_t p = 1;
  for (int m = lo; m <= hi; m++) p *= ext(m);
  return p;
}

// Construct_Dimension_Tree (common.cxx:225-270): split [lo,hi] at (lo+hi)/2; every non-root range
// becomes a node that knows its parent and its sibling range.
void CpEngine::build_tree(int lo, int hi, int parent) {
  if (hi <= lo) return;
  const int mid = (lo + hi) / 2;
  const int ranges[2][2] = {{lo, mid}, {mid + 1, hi}};
  int idx[2];
  for (int c = 0; c < 2; c++) {
    Node n;
    n.lo = ranges[c][0];
    n.hi = ranges[c][1];
    n.parent = parent;
    n.slo = ranges[1 - c][0];
    n.shi = ranges[1 - c][1];
    n.elems = 0;
    nodes_.push_back(n);
    idx[c] = (int)nodes_.size() - 1;
  }
  build_tree(lo, mid, idx[0]);
  build_tree(mid + 1, hi, idx[1]);
}
int CpEngine::find_node(int lo, int hi) const {
  for (size_t k = 0; k < nodes_.size(); k++)
    if (nodes_[k].lo == lo && nodes_[k].hi == hi) return (int)k;
  return -1;
}

// mttkrp_map_DT (common.cxx:20-133). A node whose parent is the root scans V once with the
// Khatri-Rao product of ALL sibling modes (K1: sibling is a suffix, K2: a prefix); deeper nodes
// contract the cached parent tensor, which already carries the rank index.
void CpEngine::compute_node(int idx) {
  check_tensor_generation();
  Node &n = nodes_[idx];
  if (n.valid) return;
  n.elems = prod_ext(n.lo, n.hi);
  if (!n.buf) n.buf = (double *)ops_.alloc(sizeof(double) * (size_t)n.elems * R_);
  FactorRef f[MAX_ORDER];
  int nf = 0;
  for (int m = n.slo; m <= n.shi; m++) f[nf++] = fref(m, W_.data());
  const int64_t J = prod_ext(n.slo, n.shi);
  const bool sib_is_suffix = n.slo > n.hi;
  if (n.parent < 0) {
    // a half of the modes is contracted: a suffix scan of the tensor (right half) or of the second
    // resident layout (left half); without that layout the left half is a prefix scan
    ScanPlan pl;
    if (!plan_scan(n.slo, n.shi - n.slo + 1, false, pl))
      throw std::runtime_error("ppals: internal error (tree node not adjacent)");
    if (pl.Lc == 1 && !sib_is_suffix) {
      ops_.scan_contract(V_.data, V_.dtype, 1, J, n.elems, f, nf, R_, n.buf, F64, 1, n.elems);
    } else {
      if (pl.T != 1 || pl.Lc != n.elems || pl.J != J)
        throw std::runtime_error("ppals: internal error (tree node scan shape)");
      ops_.scan_contract(pl.lay->ptr, V_.dtype, pl.L, J, 1, f, nf, R_, n.buf, F64, n.elems, n.elems,
                         pl.pad);
    }
  } else {
    compute_node(n.parent);
    const Node &p = nodes_[n.parent];
    if (sib_is_suffix)
      ops_.mttv(p.buf, F64, n.elems, J, 1, f, nf, R_, n.buf, n.elems, 0, nullptr);
    else
      ops_.mttv(p.buf, F64, 1, J, n.elems, f, nf, R_, n.buf, n.elems, 0, nullptr);
  }
  n.valid = true;
}

void CpEngine::refresh_grams() {
  for (int i = 0; i < N_; i++)
    ops_.gram(W_[i], V_.glens[i], V_.glens[i], R_, G_ + (size_t)i * R_ * R_);
}

void CpEngine::set_factors(const double *Wflat, const double *gradWflat) {
  const double *w = Wflat, *g = gradWflat;
  double gs = 0, gsi[MAX_ORDER] = {0};
  for (int i = 0; i < N_; i++) {
    size_t n = (size_t)V_.glens[i] * R_;
    ops_.h2d(W_[i], w, n * sizeof(double));
    w += n;
    if (g) {
      ops_.h2d(gradW_[i], g, n * sizeof(double));
      for (size_t e = 0; e < n; e++) gsi[i] += g[e] * g[e];
      gs += gsi[i];
      g += n;
    }
  }
  // per-mode ||grad_W[i]||^2 of the caller's gradients: a partial sweep (class API steps) mixes
  // them with the modes updated so far, exactly like CPD::update_gradnorm (src/CP.cxx:89-96)
  ops_.h2d(gradsq_, gsi, sizeof(double) * MAX_ORDER);
  for (int i = 0; i < MAX_ORDER; i++) grad_replicated_[i] = true;
  // iter-0 [gradnorm] is the norm of the caller's initial grad_W (test_ALS.cxx:338,
  // als_CP.cxx:174-181)
  init_gradnorm_ = std::sqrt(gs);
  grad_from_sweep_ = false;
  refresh_grams();
  for (auto &n : nodes_) n.valid = false;
  ms_invalidate();
}

void CpEngine::get_factors(double *Wflat, double *gradWflat) {
  double *w = Wflat, *g = gradWflat;
  for (int i = 0; i < N_; i++) {
    size_t n = (size_t)V_.glens[i] * R_;
    if (w) {
      ops_.d2h(w, W_[i], n * sizeof(double));
      w += n;
    }
    if (g) {
      if (dist_ && grad_from_sweep_ && !grad_replicated_[i]) {
        // every rank holds only its own row block of grad_W: gather it
        const int64_t blk = block_rows(V_.glens[i], P_);
        ops_.pack_blocks(gradW_[i], V_.glens[i], V_.glens[i], R_, blk, P_, gatherbuf_);
        comm_.allgather(gatherbuf_ + (size_t)rank_ * blk * R_, gatherbuf_, blk * R_);
        ops_.unpack_blocks(gatherbuf_, V_.glens[i], V_.glens[i], R_, blk, P_, gradW_[i]);
      }
      ops_.d2h(g, gradW_[i], n * sizeof(double));
      g += n;
    }
  }
}

// one mode update: K4 (S), K5 (gradient with the pre-update W), K6 (solve), Gram refresh;
// with P > 1 ranks: C1 reduce-scatter of the partial rows, C2 all-gather of the updated rows.
void CpEngine::mode_update(int i, const double *M, int64_t ldm, double lambda, bool pp,
                           double ratio) {
  const int64_t s = V_.glens[i];
  double *Gi = G_ + (size_t)i * R_ * R_;
  if (!dist_) {
    if (test_blocks_ > 1 && s % test_blocks_ == 0 && !pp) {
      // test hook (PPALS_TEST_BLOCKED_UPDATE=P, one rank): the update reads its M in P row blocks, as the
      // sharded mode's update reads the all-gather's receive buffer on P ranks — the blocked
      // addressing of the fused launch with more than one block, on one GPU
      const int64_t blk = s / test_blocks_;
      const size_t bytes = sizeof(double) * (size_t)s * R_;
      if (!test_blkbuf_) {
        test_blkbuf_ = (double *)ops_.alloc(2 * sizeof(double) * (size_t)maxs_ * R_);
      }
      ops_.pack_blocks(M, s, ldm, R_, blk, test_blocks_, test_blkbuf_);
      ops_.cp_mode_update_blocked(G_, N_, i, R_, lambda, test_blkbuf_, blk, test_blocks_,
                                  test_blkbuf_ + bytes / sizeof(double), W_[i], s, gradW_[i], s, s,
                                  gradsq_ + i, nullptr, s, nullptr, s, ratio, S_, Sinv_);
      return;
    }
    ops_.cp_mode_update(G_, N_, i, R_, lambda, M, ldm, W_[i], s, gradW_[i], s, s, gradsq_ + i,
                        pp ? Winit_[i] : nullptr, s, pp ? dW_[i] : nullptr, s, ratio, S_, Sinv_,
                        (pp && pp_norms_) ? pp_norms_ + 2 * i : nullptr);
    return;
  } else if (i != 0 && (int64_t)sizeof(double) * s * R_ <= small_msg_bytes_) {
    // latency regime (s x R is a few KB): ONE all-reduce of the partial rows, then every rank runs
    // the whole (tiny) mode update redundantly — the reduce-scatter + all-gather pair below
    // collapsed into a single collective, and the same fused launch as on one GPU.
    if (ldm != s) throw std::runtime_error("ppals: partial MTTKRP must be contiguous");
    comm_.allreduce_sum(const_cast<double *>(M), s * R_);
    ops_.cp_mode_update(G_, N_, i, R_, lambda, M, ldm, W_[i], s, gradW_[i], s, s, gradsq_ + i,
                        pp ? Winit_[i] : nullptr, s, pp ? dW_[i] : nullptr, s, ratio, S_, Sinv_);
    grad_replicated_[i] = true;
    return;
  } else if (i == 0 && (int64_t)sizeof(double) * s * R_ <= small_msg_bytes_) {
    // same regime, sharded mode: the local rows are complete, so the owners' row blocks are
    // all-gathered (the cheapest collective: blk x R doubles per rank) and every rank runs the
    // fused update on the assembled s x R matrix.
    const int64_t blk = block_rows(s, P_);
    const int64_t nr = std::max<int64_t>(0, std::min(blk, s - blk * rank_));
    double *mine = gatherbuf_ + (size_t)rank_ * blk * R_;
    if (blk * P_ == s && ldm == blk && nr == blk) {
      // equal blocks: the local leaf IS this rank's block as the all-gather wants it (no pack launch),
      // and the fused update reads the gathered blocks as they lie (no unpack launch)
      comm_.allgather(M, gatherbuf_, blk * R_);
      ops_.cp_mode_update_blocked(G_, N_, i, R_, lambda, gatherbuf_, blk, P_, sendbuf_, W_[i], s, gradW_[i],
                                  s, s, gradsq_ + i, pp ? Winit_[i] : nullptr, s, pp ? dW_[i] : nullptr, s,
                                  ratio, S_, Sinv_);
      grad_replicated_[i] = true;
      return;
    }
    ops_.pack_blocks(M, nr, ldm, R_, blk, 1, mine);
    comm_.allgather(mine, gatherbuf_, blk * R_);
    ops_.unpack_blocks(gatherbuf_, s, s, R_, blk, P_, sendbuf_);
    ops_.cp_mode_update(G_, N_, i, R_, lambda, sendbuf_, s, W_[i], s, gradW_[i], s, s, gradsq_ + i,
                        pp ? Winit_[i] : nullptr, s, pp ? dW_[i] : nullptr, s, ratio, S_, Sinv_);
    grad_replicated_[i] = true;
    return;
  } else {
    grad_replicated_[i] = false;
    ops_.gram_system(G_, N_, i, R_, lambda, S_, Sinv_);
    const int64_t blk = block_rows(s, P_);
    const int64_t r0 = blk * rank_;
    const int64_t nr = std::max<int64_t>(0, std::min(blk, s - r0));
    const double *Mblk;
    int64_t ldb;
    if (i == 0) {  // rows of mode 0 are complete on their owner
      Mblk = M;
      ldb = ldm;
    } else {
      ops_.pack_blocks(M, s, ldm, R_, blk, P_, sendbuf_);
      comm_.reduce_scatter_sum(sendbuf_, recvbuf_, blk * R_);
      Mblk = recvbuf_;
      ldb = blk;
    }
    double *mine = gatherbuf_ + (size_t)rank_ * blk * R_;
    // SVD_solve_mod tail (common.cxx:753-756): the row block is written as W_init + ratio *
    // (M S^-1 - W_init) already, so after the gather dW = W - W_init holds for every row
    ops_.cp_update(Mblk, ldb, W_[i] + r0, s, mine, blk, gradW_[i] + r0, s, nr, R_, S_, Sinv_,
                   gradsq_ + i, pp ? Winit_[i] + r0 : nullptr, s, pp ? dW_[i] + r0 : nullptr, s,
                   ratio);
    comm_.allgather(mine, gatherbuf_, blk * R_);
    ops_.unpack_blocks(gatherbuf_, s, s, R_, blk, P_, W_[i]);
    if (pp) {
      double *A[1] = {W_[i]}, *B[1] = {Winit_[i]}, *D[1] = {dW_[i]};
      int64_t n[1] = {s * R_};
      ops_.diff_norms(A, B, n, 1, 1, D, 0, scal_ + 2 * MAX_ORDER);
    }
  }
  ops_.gram(W_[i], s, s, R_, Gi);
}

// cached MSDT tensors were built from the un-normalised factors of their contracted modes: a
// Normalize multiplies every live tensor's pending scalar by prod_{m contracted} f_m (same launch).
// Returns the set of live slots, marks them pending; skip_node: a leaf about to be consumed.
unsigned CpEngine::ms_collect_scales(unsigned *masks, unsigned *fresh, int skip_node) {
  unsigned active = 0;
  *fresh = 0;
  if (!(schedule_ == 1 && ms_root_ >= 0)) return 0;
  auto visit = [&](RTensor &t) {
    if (!t.valid) return;
    masks[t.slot] = t.contracted;
    active |= 1u << t.slot;
    if (!t.pending) *fresh |= 1u << t.slot;
    t.pending = true;
  };
  visit(ms_X_);
  for (size_t q = 0; q < ms_nodes_.size(); q++)
    if ((int)q != skip_node) visit(ms_nodes_[q].t);
  return active;
}

void CpEngine::normalize() {
  int64_t rows[MAX_ORDER];
  for (int i = 0; i < N_; i++) rows[i] = V_.glens[i];
  unsigned masks[32] = {0}, fresh = 0;
  const unsigned active = ms_collect_scales(masks, &fresh, -1);
  ops_.normalize_ms(W_.data(), rows, N_, R_, G_, ms_scales_, masks, active, fresh,
                    pp_norms_live_ ? pp_norms_ + 1 : nullptr);
}

// ---------------------------------------------------------------------------- multi-sweep tree
// Binary tree (same split rule as Construct_Dimension_Tree) over the POSITIONS 0..N-2 of the
// step's mode list; node [lo,hi] = X contracted with the listed modes outside [lo,hi].
void CpEngine::ms_build_tree(int lo, int hi, int parent) {
  if (hi < lo) return;
  if (hi == lo && parent < 0) {  // N == 2: single leaf directly under X
    MsNode n;
    n.lo = n.hi = lo;
    n.parent = -1;
    n.slo = 1;
    n.shi = 0;
    ms_nodes_.push_back(n);
    return;
  }
  if (hi <= lo) return;
  const int mid = (lo + hi) / 2;
  const int ranges[2][2] = {{lo, mid}, {mid + 1, hi}};
  int idx[2];
  for (int c = 0; c < 2; c++) {
    MsNode n;
    n.lo = ranges[c][0];
    n.hi = ranges[c][1];
    n.parent = parent;
    n.slo = ranges[1 - c][0];
    n.shi = ranges[1 - c][1];
    ms_nodes_.push_back(n);
    idx[c] = (int)ms_nodes_.size() - 1;
  }
  ms_build_tree(lo, mid, idx[0]);
  ms_build_tree(mid + 1, hi, idx[1]);
}

void CpEngine::ms_reserve(RTensor &t, size_t bytes) {
  if (t.cap < bytes) {
    ops_.free(t.buf);
    t.buf = ops_.alloc(bytes);
    t.cap = bytes;
  }
}

// ---- which root sets the multi-sweep schedule uses ----
bool CpEngine::ms_set_excluded(int first, int k, unsigned excl) const {
  for (int q = 0; q < k; q++)
    if ((excl >> ((first + q) % N_)) & 1u) return true;
  return false;
}

// The root set that serves the update of mode i when the running step cannot: the k modes updated
// last, slid back past excluded modes. A slid-back set is older but still current (none of its modes
// has been updated since); the step is then entered in the middle of its mode list and serves fewer
// updates (order 4, k = 1, mode 0 excluded: runs 3,3,2 -> 3 scans per 2 sweeps instead of 8/3).
int CpEngine::ms_next_root(int i, int k, unsigned excl) const {
  int first = (i - k + N_) % N_, guard = 0;
  while (ms_set_excluded(first, k, excl) && guard++ < N_) first = (first - 1 + N_) % N_;
  if (ms_set_excluded(first, k, excl) || ((i - first + N_) % N_) < k) return -1;
  return first;
}

// What a sweep costs under (k, excl), in reads of the tensor: the schedule is walked for a few sweeps
// (the transient of the root rotation dropped), every first-level scan costs 1 + 4.5 x (size of its
// X relative to the tensor): X is written once — a written byte costs ~2.5 read bytes on this part,
// profiles/README.md — and read twice by the step's tree.
double CpEngine::ms_schedule_cost(int k, unsigned excl) const {
  int root = -1;
  double cost = 0;
  const int sweeps = 2 * N_;
  for (int u = 0; u < (sweeps + 2) * N_; u++) {
    const int i = u % N_;
    if (root < 0 || ((i - root + N_) % N_) < k) {
      root = ms_next_root(i, k, excl);
      if (root < 0) return 1e300;
      if (u >= 2 * N_) {
        double J = 1;
        for (int q = 0; q < k; q++) J *= (double)ext((root + q) % N_);
        cost += 1.0 + 4.5 * (double)R_ / J;
      }
    }
  }
  return cost / sweeps;
}

// How many modes one first-level contraction removes (the "root set" of a step) and which modes are
// never roots. k = 1 with no exclusion wins at order 4 with equal extents (cfg2: 1.33 x 1.23 vs 2.0),
// k = 2 at order 6 with s = 50, R = 6 (1.2 x 1.54 vs 1.5 x 1.01); a mode of extent 3 at R = 10
// (coil-100: 3 x 128 x 128 x 7200) is excluded: 1.5 scans of the tensor per sweep instead of 1.33
// scans plus an X of 3.3 x the tensor written once and read twice every third step.
// PPALS_MSDT_ROOTS=k fixes k (the exclusion is still chosen by cost).
int CpEngine::ms_choose_roots() {
  const unsigned mandatory = (dist_ && N_ > 2) ? 1u : 0u;  // sharded: mode 0 is never in a root set
  // sharded: the root set is slid back past mode 0; it still has to end before the mode about to be
  // updated, which needs 2k < N
  const int kcap = dist_ ? std::max(1, (N_ - 1) / 2) : N_;
  int kfix = 0;
  if (const char *e = std::getenv("PPALS_MSDT_ROOTS")) {
    const int k = std::atoi(e);
    if (k >= 1 && k <= N_ - 2) kfix = std::min(k, kcap);
  }
  // candidates for exclusion: modes whose X would be at least 5 % of the tensor
  unsigned small = 0;
  for (int m = 0; m < N_; m++)
    if ((double)R_ >= 0.05 * (double)ext(m)) small |= 1u << m;
  int best = 1;
  unsigned best_excl = mandatory;
  double best_cost = 1e300;
  for (int k = 1; k <= std::max(1, N_ / 2) && k <= N_ - 2 && k <= kcap; k++) {
    if (kfix && k != kfix) continue;
    for (unsigned sub = small;; sub = (sub - 1) & small) {  // every subset of the short modes
      const unsigned excl = sub | mandatory;
      const double cost = ms_schedule_cost(k, excl);
      // (ties: fewer excluded modes, then smaller k — the subsets are walked from the fullest down, so a
      // tie has to be broken explicitly)
      const bool tie = cost <= best_cost * (1.0 + 1e-9) && cost >= best_cost * (1.0 - 1e-9);
      if (cost < best_cost * (1.0 - 1e-9) ||
          (tie && k == best && __builtin_popcount(excl) < __builtin_popcount(best_excl))) {
        best_cost = cost;
        best = k;
        best_excl = excl;
      }
      if (sub == 0) break;
    }
  }
  if (best_cost >= 1e299) {  // nothing schedulable with the requested k: single roots, mandatory exclusion only
    best = 1;
    best_excl = mandatory;
  }
  ms_excl_ = best_excl;
  return best;
}

// (re)build the step tree for k root modes: binary tree over the N - k positions of the step's
// mode list. Cached tensors of the old tree are released.
void CpEngine::ms_set_roots(int k) {
  for (auto &n : ms_nodes_) {
    ops_.free(n.t.buf);
    for (auto &t : n.tmp) ops_.free(t.buf);
  }
  ms_nodes_.clear();
  ms_k_ = k;
  ms_build_tree(0, N_ - k - 1, -1);
  ms_leaf_.assign(N_ - k, -1);
  for (size_t q = 0; q < ms_nodes_.size(); q++)
    if (ms_nodes_[q].lo == ms_nodes_[q].hi) ms_leaf_[ms_nodes_[q].lo] = (int)q;
  if (ms_nodes_.size() + 1 > 32) throw std::runtime_error("ppals: tensor order too large");
  ms_X_.slot = 0;
  for (size_t q = 0; q < ms_nodes_.size(); q++) ms_nodes_[q].t.slot = (int)q + 1;
  ms_invalidate();
}

// room for the placement candidates of X (ms_start_step): none for small tensors, where a scan is
// too short to time from the host and the effect does not matter
// tensors below this size are not worth a placement measurement (their scans are short against the
// host-side timing); PPALS_PLACE_MIN_MB lowers it so that the CPU tests exercise the machinery
double CpEngine::place_min_bytes() {
  const char *e = std::getenv("PPALS_PLACE_MIN_MB");
  return e ? std::atof(e) * 1048576.0 : 1.5e9;
}

size_t CpEngine::ms_X_slack() const {
  const double bytes = (double)V_.nloc * dtype_size(V_.dtype);
  return bytes >= place_min_bytes() ? ((size_t)64 << 20) + 4096 : 0;
}

// ---- online placement choice (engine.h, PlaceExplore) ----
// the sample of the visit in flight, if any, goes to its candidate
void CpEngine::ms_place_collect(PlaceExplore &ex) {
  if (ex.timer < 0) return;
  const double t = ops_.timer_read(ex.timer);
  ex.timer = -1;
  if (t <= 0 || ex.timer_cand < 0 || ex.timer_cand >= (int)ex.cands.size()) return;
  PlaceCand &c = ex.cands[ex.timer_cand];
  c.best = std::min(c.best, t);
  c.samples++;
  ex.worst = std::max(ex.worst, t);
}

// at the head of a visit: the candidate this visit runs (-1: none, the root has settled). Phase 0
// walks the offsets once (store kind by the back end's rule), phase 1 runs the three fastest
// offsets with both store kinds, then the root keeps the fastest sample of all; a later candidate
// must win by more than the timing noise. The sample of the previous visit is a whole cycle of the
// schedule old when it is read here: nothing waits.
int CpEngine::ms_place_pick(PlaceExplore &ex) {
  ms_place_collect(ex);
  // Evidence gate (round 6): on the driver's boxes five rounds of headlines showed the choice worth
  // nothing (tuned 537.6 against untuned 537.8 sweeps/s, fastest and slowest candidate 2.6 % apart),
  // on others 2 %. After the first 8 samples of a root, a spread below 4 % between its fastest and
  // slowest candidate ends the exploration: the root keeps the fastest sample IN THE FIRST BLOCK, so
  // that the second block goes back to the device at once.
  if (ex.phase == 0) {
    int n = 0, best0 = -1;
    double lo = 1e300, hi = 0;
    for (size_t q = 0; q < ex.cands.size(); q++)
      if (ex.cands[q].samples > 0) {
        n++;
        lo = std::min(lo, ex.cands[q].best);
        hi = std::max(hi, ex.cands[q].best);
        if (ex.cands[q].blk == 0 && (best0 < 0 || ex.cands[q].best < ex.cands[best0].best)) best0 = (int)q;
      }
    if (n >= kPlaceGateSamples && best0 >= 0 && hi < lo * (1.0 + kPlaceGateSpread)) {
      ex.chosen = best0;
      ex.gated = true;
      ex.phase = 2;
      ms_place_release_unchosen();
      return -1;
    }
  }
  if (ex.phase == 0 && ex.next >= ex.cands.size()) {
    std::vector<int> idx(ex.cands.size());
    for (size_t q = 0; q < idx.size(); q++) idx[q] = (int)q;
    std::stable_sort(idx.begin(), idx.end(),
                     [&](int a, int b) { return ex.cands[a].best < ex.cands[b].best; });
    const size_t nfin = std::min<size_t>(3, idx.size());
    for (size_t q = 0; q < nfin; q++)
      for (int kind = 0; kind < 2; kind++) {
        PlaceCand c;
        c.blk = ex.cands[idx[q]].blk;
        c.off = ex.cands[idx[q]].off;
        c.nt = kind;
        ex.cands.push_back(c);
      }
    ex.phase = 1;
  }
  if (ex.phase == 1 && ex.next >= ex.cands.size()) {
    int best = -1;
    for (size_t q = 0; q < ex.cands.size(); q++)
      if (ex.cands[q].samples > 0 && (best < 0 || ex.cands[q].best < ex.cands[best].best * 0.995))
        best = (int)q;
    ex.chosen = best;  // (-1: no stopwatch ever answered — offset 0, store kind by size)
    ex.phase = 2;
    ms_place_release_unchosen();
  }
  return ex.phase < 2 ? (int)ex.next++ : -1;
}

// once every exploring root has settled: a candidate block nobody chose goes back to the device
void CpEngine::ms_place_release_unchosen() {
  if (!ms_X_alt_) return;
  bool use[2] = {false, false};
  for (const auto &ex : ms_place_) {
    if (ex.phase == 0 || ex.phase == 1) return;  // somebody still explores
    if (ex.phase == 2) use[ex.chosen >= 0 ? ex.cands[ex.chosen].blk : 0] = true;
  }
  if (use[0] && use[1]) return;
  ops_.sync();
  if (!use[1]) {
    ops_.free(ms_X_alt_);
  } else {  // every root prefers the block in front of the layout: it becomes the only one
    ops_.free(ms_X_base_);
    ms_X_base_ = ms_X_alt_;
    for (auto &ex : ms_place_)
      for (auto &c : ex.cands) c.blk = 0;
  }
  ms_X_alt_ = nullptr;
  // (called at the head of a step, before its scan: no first-level intermediate is alive)
}

// bytes of the first-level intermediate of the root set first .. first+k-1 (layout-independent)
size_t CpEngine::ms_X_bytes(int first, int k) const {
  int64_t J = 1;
  for (int q = 0; q < k; q++) J *= ext((first + q) % N_);
  return (size_t)(V_.nloc / J) * R_ * dtype_size(V_.dtype);
}

// Allocation of one of the session's large buffers. The optional resident layouts (the padded
// third copy, then the second copy) were taken while the device still had room; when a MANDATORY
// buffer does not fit any more they are given back, one at a time, and the allocation is retried —
// the scans then read the layouts that are left (plan_scan), slower but correct. Callers must not
// hold a ScanPlan across this call.
void *CpEngine::big_alloc(size_t bytes) {
  for (;;) {
    void *p = ops_.try_alloc(bytes);
    if (p) return p;
    // first what is merely optional for SPEED OF CHOICE: the second candidate block of the placement
    // exploration (unless the intermediate alive right now lies in it) — a resident layout is worth
    // more to every later sweep than a few more candidates to ~24 visits
    if (ms_X_alt_ && !(ms_X_.valid && (char *)ms_X_.buf >= (char *)ms_X_alt_ &&
                       (char *)ms_X_.buf < (char *)ms_X_alt_ + ms_X_cap_)) {
      ops_.sync();
      for (auto &ex : ms_place_) {
        if (ex.timer >= 0) {  // a sample of a candidate in the block that goes away is dropped
          ops_.timer_read(ex.timer);
          ex.timer = -1;
        }
        for (auto &c : ex.cands) c.blk = 0;  // (its candidates now alias offsets of the first block)
      }
      ops_.free(ms_X_alt_);
      ms_X_alt_ = nullptr;
      continue;
    }
    if (lay_.size() > 1 && lay_.back().owned) {
      ops_.sync();
      ops_.free(lay_.back().ptr);
      lay_.pop_back();
      if (lay_.size() == 1) vt_state_ = -1;
      for (auto &n : nodes_) n.valid = false;
      continue;
    }
    return ops_.alloc(bytes);  // throws with the back end's message
  }
}

// new step: X = V contracted with the k root modes first, ..., first + k - 1 (cyclic) in ONE
// tensor scan on whichever resident layout stores them next to each other and behind at least one
// other mode; X is kept in the tensor's own precision
void CpEngine::ms_start_step(int first) {
  const int k = ms_k_;
  ms_root_ = first;
  ms_order_.clear();
  for (int q = k; q < N_; q++) ms_order_.push_back((first + q) % N_);
  for (auto &n : ms_nodes_) n.t.valid = false;
  // X at this root's offset inside the over-allocated block (see engine.h, PlaceExplore); sized
  // before the scan is planned (big_alloc may give a resident layout back)
  const size_t xbytes = ms_X_bytes(first, k);
  const size_t slack = ms_tune_enabled_ ? ms_X_slack() : 0;
  // (an override — the LR optimizers' per-root caches — receives the scan's result itself: no block)
  if (!ms_X_override_ && ms_X_cap_ < xbytes + slack) {
    // (the block moves: what the roots measured so far is void, they start over)
    for (auto &ex : ms_place_) {
      if (ex.timer >= 0) ops_.timer_read(ex.timer);
      ex = PlaceExplore();
    }
    ops_.free(ms_X_base_);
    ops_.free(ms_X_alt_);
    ms_X_base_ = ms_X_alt_ = nullptr;
    ms_X_cap_ = 0;
    ms_X_base_ = big_alloc(xbytes + slack);
    ms_X_cap_ = xbytes + slack;
  }
  ScanPlan pl;
  if (!plan_scan(first, k, false, pl)) {
    // a root set that wraps around the last mode is adjacent only in the second layout; without
    // it (allocation failed / PPALS_TRANSPOSED_COPY=0) fall back to single-mode roots for good
    if (k == 1) throw std::runtime_error("ppals: internal error (single root not adjacent)");
    ms_set_roots(1);
    // the mode just before the one about to be updated, slid back past modes that are never roots
    // (an older root is still current: none of the skipped modes' successors has been updated since)
    int r = (first + k - 1) % N_, guard = 0;
    while (((ms_excl_ >> r) & 1u) && guard++ < N_) r = (r - 1 + N_) % N_;
    ms_start_step(r);
    return;
  }
  const void *src = pl.lay->ptr;
  const int64_t L = pl.Lc, J = pl.J, T = pl.T;
  ms_X_.modes = pl.kept;
  std::vector<FactorRef> f;
  unsigned mask = 0;
  for (int m : pl.set) {
    f.push_back(fref(m, W_.data()));  // storage order: first listed = fastest
    mask |= 1u << m;
  }
  ms_X_.dt = V_.dtype;
  ms_X_.contracted = mask;
  if ((size_t)L * T * R_ * dtype_size(ms_X_.dt) != xbytes)
    throw std::runtime_error("ppals: internal error (first-level intermediate size)");
  auto launch_scan = [&](int blk, int64_t off, int nt) {
    ms_X_.buf = ms_X_override_ ? (char *)ms_X_override_
                               : (char *)(blk == 1 && ms_X_alt_ ? ms_X_alt_ : ms_X_base_) + off;
    ops_.scan_store_mode(nt);
    ops_.scan_contract(src, V_.dtype, pl.L, J, T, f.data(), (int)f.size(), R_, ms_X_.buf, ms_X_.dt,
                       L, L * T, pl.pad);
    ops_.scan_store_mode(-1);
  };
  PlaceExplore &ex = ms_place_[first];
  if (slack > 0 && !ms_X_override_ && ex.phase < 0) {
    // first visit of the root: is its placement worth choosing? Only where the result stream
    // matters — an HBM-bound scan (up to two n-tiles) of a tensor large enough for a launch to be
    // long against the stopwatch, that writes at least 1 % of what it reads.
    static const int64_t cand_small[] = {0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64};
    static const int64_t cand_large[] = {0, 3, 5, 12, 16, 24, 48, 64};
    const double bytes = (double)L * J * T * dtype_size(V_.dtype);
    ex.phase = 3;
    ex.layout = (int)(pl.lay - lay_.data());
    if (bytes >= place_min_bytes() && R_ <= 32 && (double)xbytes >= 0.01 * bytes) {
      const bool large = bytes >= 2e10;  // a scan takes >= 4 ms: fewer candidates
      const int64_t *mb = large ? cand_large : cand_small;
      for (int ci = 0; ci < (large ? 8 : 14); ci++) {
        if ((size_t)(mb[ci] << 20) > slack) break;
        PlaceCand c;
        c.off = mb[ci] << 20;
        ex.cands.push_back(c);
      }
      static const int64_t cand_alt[] = {0, 4, 16, 64};  // in the second block
      for (int ci = 0; ms_X_alt_ && ci < 4; ci++) {
        if ((size_t)(cand_alt[ci] << 20) > slack) break;
        PlaceCand c;
        c.blk = 1;
        c.off = cand_alt[ci] << 20;
        ex.cands.push_back(c);
      }
      ex.phase = 0;
    }
  }
  const int cand = (slack > 0 && !ms_X_override_ && ex.phase >= 0 && ex.phase < 2) ? ms_place_pick(ex) : -1;
  if (cand >= 0) {
    // an exploring visit: the sweep's own scan, at this candidate, under the stopwatch
    ex.visits++;
    ex.timer_cand = cand;
    ex.timer = ops_.timer_begin();
    launch_scan(ex.cands[cand].blk, ex.cands[cand].off, ex.cands[cand].nt);
    if (ex.timer >= 0) ops_.timer_end(ex.timer);
  } else if (ex.phase == 2 && ex.chosen >= 0 && slack > 0 && !ms_X_override_) {
    launch_scan(ex.cands[ex.chosen].blk, ex.cands[ex.chosen].off, ex.cands[ex.chosen].nt);
  } else {
    launch_scan(0, 0, -1);
  }
  ms_X_.pending = false;
  ms_X_.valid = true;
  if (const char *tr = std::getenv("PPALS_TRACE_STEPS")) {  // tests: which root sets were scanned
    if (FILE *f = std::fopen(tr, "a")) {
      const int li = (int)(pl.lay - lay_.data());
      std::fprintf(f, "rank=%d root=%d k=%d layout=%s%s L=%lld J=%lld T=%lld\n", rank_, first, k,
                   li == 1 ? "VT" : "V", pl.lay->q > 0 ? "pad" : "", (long long)L, (long long)J,
                   (long long)T);
      std::fclose(f);
    }
  }
}

// one mode update of the multi-sweep schedule: starts a new step when mode i belongs to the root
// set of the running one (its factor was frozen into X), i.e. after N - k updates
void CpEngine::ms_mode_update(int i, double lambda, bool last_of_sweep) {
  check_tensor_generation();
  if (ms_root_ < 0 || ((i - ms_root_ + N_) % N_) < ms_k_) {
    // the k modes updated last serve the next N - k updates — unless one of them is never a root
    // (ms_excl_: the partitioned mode of a sharded session, whose X would be a PARTIAL sum of full
    // global size — s^(N-1) R per rank whatever P is: 5.1 GB against a 12.8 GB shard at cfg4 / P = 8 —,
    // or a mode too short for X to be smaller than the tensor): then the set is slid back past it
    const int first = ms_next_root(i, ms_k_, ms_excl_);
    if (first < 0) throw std::runtime_error("ppals: no root set without the excluded modes");
    ms_start_step(first);
  }
  int pos = -1;
  for (size_t q = 0; q < ms_order_.size(); q++)
    if (ms_order_[q] == i) pos = (int)q;
  const int leaf = ms_leaf_[pos];
  // (S and S^-1 of this update depend on the other modes' Grams only: the contraction launched
  // next may prepare them on the side)
  ops_.arm_gram_system(G_, N_, i, R_, lambda, S_, Sinv_);
  ms_compute(leaf);
  ms_norm_fused_ = false;
  // the sweep's Normalize at the tail of its last update launch, with the pending scales of the cached
  // tensors that outlive the update (everything valid now but the leaf it consumes) — where the update
  // is the fused launch: one rank, or the one-all-reduce plan of a sharded session (never mode 0's)
  if (last_of_sweep && (!dist_ || (i != 0 && (int64_t)sizeof(double) * V_.glens[i] * R_ <= small_msg_bytes_))) {
    int64_t rows[MAX_ORDER];
    for (int q = 0; q < N_; q++) rows[q] = V_.glens[q];
    // (asked first with no cached tensors — a back end that cannot fold it must not see them marked)
    if (ops_.arm_normalize(W_.data(), rows, N_, R_, G_, i, nullptr)) {
      unsigned masks[32] = {0}, fresh = 0;
      const unsigned active = ms_collect_scales(masks, &fresh, leaf);
      ms_norm_fused_ = ops_.arm_normalize(W_.data(), rows, N_, R_, G_, i, nullptr, ms_scales_, masks, active, fresh);
    }
  }
  mode_update(i, (const double *)ms_nodes_[leaf].t.buf, ext(i), lambda, false, 1.0);
  ms_nodes_[leaf].t.valid = false;  // a leaf is consumed by its own update
}

// dst = src contracted with `mode` (rank index shared), fp64 result; in_scale = pending factor of
// src folded into dst, so dst starts with a clean scale of its own
void CpEngine::ms_contract(const RTensor &src, int mode, RTensor &dst, const double *in_scale) {
  int64_t L = 1, T = 1;
  bool before = true;
  dst.modes.clear();
  for (int m : src.modes) {
    if (m == mode) {
      before = false;
      continue;
    }
    (before ? L : T) *= ext(m);
    dst.modes.push_back(m);
  }
  dst.dt = F64;
  dst.contracted = src.contracted | (1u << mode);
  ms_reserve(dst, sizeof(double) * (size_t)L * T * R_);
  FactorRef f = fref(mode, W_.data());
  ops_.mttv(src.buf, src.dt, L, ext(mode), T, &f, 1, R_, (double *)dst.buf, L * T, 0, in_scale);
}

void CpEngine::ms_compute(int idx) {
  MsNode &n = ms_nodes_[idx];
  if (n.t.valid) return;
  const RTensor *src;
  if (n.parent < 0) {
    src = &ms_X_;
  } else {
    ms_compute(n.parent);
    src = &ms_nodes_[n.parent].t;
  }
  // contract the sibling's modes; the one stored slowest first (longest contiguous runs)
  std::vector<int> sib;
  for (int pos = n.slo; pos <= n.shi; pos++) sib.push_back(ms_order_[pos]);
  auto storage_pos = [&](int mode) {
    for (size_t k = 0; k < src->modes.size(); k++)
      if (src->modes[k] == mode) return (int)k;
    return -1;
  };
  std::sort(sib.begin(), sib.end(), [&](int a, int b) { return storage_pos(a) > storage_pos(b); });
  if (n.tmp.size() + 1 < sib.size()) n.tmp.resize(sib.size() - 1);
  const RTensor *cur = src;
  for (size_t k = 0; k < sib.size(); k++) {
    RTensor &dst = (k + 1 == sib.size()) ? n.t : n.tmp[k];
    ms_contract(*cur, sib[k], dst, k == 0 ? ms_scale_of(*src) : nullptr);
    cur = &dst;
  }
  if (sib.empty()) throw std::runtime_error("ppals: empty sibling set in the multi-sweep tree");
  n.t.pending = false;
  n.t.valid = true;
}

void CpEngine::sweep_msdt(double lambda) {
  for (int i = 0; i < N_; i++) ms_mode_update(i, lambda, i == N_ - 1);
  if (!ms_norm_fused_) normalize();
  ms_norm_fused_ = false;
  grad_from_sweep_ = true;
}

void CpEngine::sweep_dt(double lambda) {
  if (schedule_ == 1) {
    sweep_msdt(lambda);
    return;
  }
  for (auto &n : nodes_) n.valid = false;  // mttkrp_map.clear(), als_CP.cxx:215
  bool norm_fused = false;
  for (int i = 0; i < N_; i++) {
    compute_node(leaf_[i]);
    const Node &lf = nodes_[leaf_[i]];
    if (i == N_ - 1 && !dist_) {  // Normalize at the tail of the sweep's last update launch
      int64_t rows[MAX_ORDER];
      for (int q = 0; q < N_; q++) rows[q] = V_.glens[q];
      norm_fused = ops_.arm_normalize(W_.data(), rows, N_, R_, G_, i, nullptr);
    }
    mode_update(i, lf.buf, ext(i), lambda, false, 1.0);
    // every cached node that does NOT contain mode i stays valid; nodes containing mode i were
    // built from W_j, j != i only, so they stay valid too until the cache is cleared.
  }
  if (!norm_fused) normalize();
  grad_from_sweep_ = true;
}

// `count` consecutive mode updates starting at mode `first` (cyclic), NO Normalize — the step of
// the class-API optimizers (src/optimizer/*::step). The multi-sweep state carries over between
// calls, so a step that starts where the previous one ended reuses the cached contraction.
void CpEngine::update_modes(int first, int count, double lambda) {
  for (int k = 0; k < count; k++) {
    const int i = (first + k) % N_;
    if (schedule_ == 1) {
      ms_mode_update(i, lambda);
    } else {
      if (i == 0 || k == 0)
        for (auto &n : nodes_) n.valid = false;
      compute_node(leaf_[i]);
      mode_update(i, nodes_[leaf_[i]].buf, ext(i), lambda, false, 1.0);
    }
  }
  grad_from_sweep_ = true;
}

// CPD<dtype, Optimizer>::als (src/CP.cxx:100-186). The optimizer kind fixes the step granularity
// and the fractional sweep counter (Simple 1, DT 0.5 [modes 0..N-2 | mode N-1], MSDT (N-1)/N);
// the ALS iterates are the same cyclic mode updates for all three, computed with this engine's
// own contraction schedule. No Normalize (src/CP.cxx:171).
// ---------------------------------------------------------------------------- low-rank updates
void CpEngine::lr_release() {
  for (int m = 0; m < MAX_ORDER; m++) {
    ops_.free(lr_cache_[m]);
    lr_cache_[m] = nullptr;
    lr_have_[m] = false;
  }
  ops_.free(lr_X_);
  ops_.free(lr_Us_);
  ops_.free(lr_G2_);
  ops_.free(lr_small_);
  ops_.free(lr_T_);
  lr_X_ = lr_Us_ = lr_G2_ = lr_small_ = lr_T_ = nullptr;
  lr_T_cap_ = 0;
}

// mttkrp_map_init of the LR optimizers (cp_dt_lr_optimizer.cxx:36-94, cp_msdt_lr_optimizer.cxx:
// 31-75): the step's first contraction X = V x_left W_left in the root's own buffer — computed by
// one tensor scan, or (reuse) the kept one brought up to date with the rank-r change of W_left
// that lr_mode_update left in lr_Us_ (rows x r) and lr_small_ (VT, r x R).
void CpEngine::lr_step_begin(int left, bool reuse, int r) {
  check_tensor_generation();
  if (ms_k_ != 1) ms_set_roots(1);
  if (!lr_cache_[left]) lr_cache_[left] = big_alloc(ms_X_bytes(left, 1));
  if (reuse && lr_have_[left]) {
    ScanPlan pl;
    if (!plan_scan(left, 1, false, pl)) throw std::runtime_error("ppals: internal error (LR scan plan)");
    const int64_t n = pl.Lc * pl.T;
    if (lr_T_cap_ < sizeof(double) * (size_t)n * r) {
      ops_.free(lr_T_);
      lr_T_ = (double *)ops_.alloc(sizeof(double) * (size_t)n * r);
      lr_T_cap_ = sizeof(double) * (size_t)n * r;
    }
    FactorRef f;
    f.ptr = lr_Us_ + (left == 0 ? V_.row0 : 0);
    f.rows = ext(left);
    f.ld = V_.glens[left];
    ops_.scan_contract(pl.lay->ptr, V_.dtype, pl.L, ext(left), pl.T, &f, 1, r, lr_T_, F64, pl.Lc, n,
                       pl.pad);
    ops_.lowrank_accumulate(lr_cache_[left], V_.dtype, n, R_, lr_T_, r, lr_small_);
    ms_X_ = lr_desc_[left];
    ms_X_.buf = lr_cache_[left];
    ms_X_.valid = true;
    ms_X_.pending = false;
    ms_root_ = left;
    ms_order_.clear();
    for (int q = 1; q < N_; q++) ms_order_.push_back((left + q) % N_);
    for (auto &nd : ms_nodes_) nd.t.valid = false;
  } else {
    ms_X_override_ = lr_cache_[left];
    ms_start_step(left);
    ms_X_override_ = nullptr;
    lr_desc_[left] = ms_X_;
    lr_have_[left] = true;
  }
}

// One mode update of an LR step. r == 0: the exact update (cholesky_solve in the reference).
// Host helpers of the randomized variant (small dense algebra, R x r and r x r).
// Orthonormal basis of the columns of A (n x k, column-major, in place): modified Gram-Schmidt,
// twice (the second pass restores orthogonality to rounding when the columns are nearly dependent);
// a column that vanishes is left zero.
static void host_qr_mgs2(int n, int k, double *A) {
  for (int j = 0; j < k; j++) {
    double *aj = A + (size_t)n * j;
    for (int pass = 0; pass < 2; pass++)
      for (int p = 0; p < j; p++) {
        const double *ap = A + (size_t)n * p;
        double d = 0;
        for (int i = 0; i < n; i++) d += ap[i] * aj[i];
        for (int i = 0; i < n; i++) aj[i] -= d * ap[i];
      }
    double nn = 0;
    for (int i = 0; i < n; i++) nn += aj[i] * aj[i];
    nn = std::sqrt(nn);
    for (int i = 0; i < n; i++) aj[i] = nn > 0 ? aj[i] / nn : 0.0;
  }
}
// eigen-decomposition of a symmetric n x n matrix (column-major, destroyed) by cyclic Jacobi:
// ev descending, V columns = eigenvectors
static void host_jacobi_eig(int n, double *A, double *ev, double *V) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i + (size_t)n * j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) (i == j ? diag : off) += A[i + (size_t)n * j] * A[i + (size_t)n * j];
    if (off <= 1e-30 * (diag + 1e-300)) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = A[p + (size_t)n * q];
        if (apq == 0.0) continue;
        const double theta = (A[q + (size_t)n * q] - A[p + (size_t)n * p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; k++) {  // columns p, q
          const double akp = A[k + (size_t)n * p], akq = A[k + (size_t)n * q];
          A[k + (size_t)n * p] = c * akp - sn * akq;
          A[k + (size_t)n * q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {  // rows p, q
          const double apk = A[p + (size_t)n * k], aqk = A[q + (size_t)n * k];
          A[p + (size_t)n * k] = c * apk - sn * aqk;
          A[q + (size_t)n * k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          const double vkp = V[k + (size_t)n * p], vkq = V[k + (size_t)n * q];
          V[k + (size_t)n * p] = c * vkp - sn * vkq;
          V[k + (size_t)n * q] = sn * vkp + c * vkq;
        }
      }
  }
  std::vector<int> idx(n);
  for (int i = 0; i < n; i++) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return A[a + (size_t)n * a] > A[b + (size_t)n * b]; });
  std::vector<double> Vs((size_t)n * n);
  for (int k = 0; k < n; k++) {
    ev[k] = A[idx[k] + (size_t)n * idx[k]];
    for (int i = 0; i < n; i++) Vs[i + (size_t)n * k] = V[i + (size_t)n * idx[k]];
  }
  std::copy(Vs.begin(), Vs.end(), V);
}
static const uint64_t LR_RANDOM_SEED = 0x52414e44535644ull;  // "RANDSVD"

// r > 0: get_rankR_update_cholesky (common.cxx:768-786; random == true: lr_random_) relative to `base`
// (device, s x R; null: the current W_i): gamma = L L^T, X = (M - base gamma) L^-T, its leading r
// singular triplets, W_i = base + (U_r s_r)(VT_r L^-1). Leaves Us = U_r s_r in lr_Us_ (s x r) and
// VT in lr_small_ (r x R) for the next lr_step_begin. The R x R algebra runs on the host.
void CpEngine::lr_mode_update(int i, double lambda, int r, const double *base) {
  int pos = -1;
  for (size_t q = 0; q < ms_order_.size(); q++)
    if (ms_order_[q] == i) pos = (int)q;
  if (pos < 0) throw std::runtime_error("ppals: internal error (LR update of the root mode)");
  const int leaf = ms_leaf_[pos];
  ms_compute(leaf);
  const double *M = (const double *)ms_nodes_[leaf].t.buf;
  const int64_t s = V_.glens[i];
  if (r <= 0) {
    mode_update(i, M, ext(i), lambda, false, 1.0);
    ms_nodes_[leaf].t.valid = false;
    return;
  }
  const size_t nsr = (size_t)s * R_;
  if (!lr_X_) {
    lr_X_ = (double *)ops_.alloc(sizeof(double) * (size_t)maxs_ * R_);
    lr_Us_ = (double *)ops_.alloc(sizeof(double) * (size_t)maxs_ * R_);
    lr_G2_ = (double *)ops_.alloc(sizeof(double) * (size_t)maxs_ * R_);
    lr_small_ = (double *)ops_.alloc(sizeof(double) * 4 * (size_t)R_ * R_);
  }
  ops_.gram_system(G_, N_, i, R_, lambda, S_, Sinv_);
  // the gradient with the CURRENT factor (what [gradnorm] reports), W_new of the exact solve unused
  ops_.cp_update(M, s, W_[i], s, lr_X_, s, gradW_[i], s, s, R_, S_, Sinv_, gradsq_ + i, nullptr, 0,
                 nullptr, 0, 1.0);
  grad_replicated_[i] = true;
  const double *negrhs = gradW_[i];  // -(M - base gamma)
  if (base && base != W_[i]) {
    ops_.cp_update(M, s, base, s, lr_X_, s, lr_G2_, s, s, R_, S_, Sinv_, scal_ + 3 * MAX_ORDER,
                   nullptr, 0, nullptr, 0, 1.0);
    negrhs = lr_G2_;
  } else {
    base = W_[i];
  }
  // host: L = chol(gamma), T1 = -L^-T  (X = negrhs * T1)
  std::vector<double> Sg((size_t)R_ * R_), L((size_t)R_ * R_, 0.0), Li((size_t)R_ * R_, 0.0),
      T1((size_t)R_ * R_, 0.0);
  ops_.d2h(Sg.data(), S_, sizeof(double) * R_ * R_);
  for (int j = 0; j < R_; j++) {
    double d = Sg[j + (size_t)R_ * j];
    for (int k = 0; k < j; k++) d -= L[j + (size_t)R_ * k] * L[j + (size_t)R_ * k];
    if (!(d > 0)) throw std::runtime_error("ppals: low-rank update: S is not positive definite");
    d = std::sqrt(d);
    L[j + (size_t)R_ * j] = d;
    for (int q = j + 1; q < R_; q++) {
      double v = Sg[q + (size_t)R_ * j];
      for (int k = 0; k < j; k++) v -= L[q + (size_t)R_ * k] * L[j + (size_t)R_ * k];
      L[q + (size_t)R_ * j] = v / d;
    }
  }
  for (int c = 0; c < R_; c++)  // Li = L^-1 (lower), column by column
    for (int q = c; q < R_; q++) {
      double v = (q == c) ? 1.0 : 0.0;
      for (int k = c; k < q; k++) v -= L[q + (size_t)R_ * k] * Li[k + (size_t)R_ * c];
      Li[q + (size_t)R_ * c] = v / L[q + (size_t)R_ * q];
    }
  for (int a = 0; a < R_; a++)
    for (int b = 0; b < R_; b++) T1[a + (size_t)R_ * b] = -Li[b + (size_t)R_ * a];  // -(L^-1)^T
  double *dT1 = lr_small_ + (size_t)R_ * R_, *dGX = dT1 + (size_t)R_ * R_, *dV = dGX + (size_t)R_ * R_;
  ops_.h2d(dT1, T1.data(), sizeof(double) * R_ * R_);
  ops_.rows_times_small(negrhs, s, R_, dT1, R_, nullptr, lr_X_);    // X = rhs L^-T
  ops_.gram(lr_X_, s, s, R_, dGX);                                   // X^T X
  std::vector<double> Vr((size_t)R_ * r), VT((size_t)r * R_, 0.0);
  if (!lr_random_) {
    ops_.top_eigvecs(dGX, R_, r, dV);                                // leading right singular vectors
    ops_.d2h(Vr.data(), dV, sizeof(double) * R_ * r);
  } else {
    // randomized_svd(X, r, iter = 1) (common.cxx:691-709), on the R x R Gram G = X^T X alone:
    //   Q0 = qr(Omega), Omega ~ U(0,1)^{R x r};  Q = qr(G Q0);  B = X Q;  B = U s Vb^T
    //   => the rank-r factor is X (Q Vb) (Q Vb)^T: right vectors Q Vb, with Vb the eigenvectors of
    //   Q^T G Q (r x r). Everything but the two tall products stays on the host. Omega comes from
    //   the build's counter generator (CTF's stream is not reproducible): seed LR_RANDOM_SEED,
    //   one block of R*r draws per call.
    std::vector<double> G((size_t)R_ * R_), Q((size_t)R_ * r), Y((size_t)R_ * r), C((size_t)r * r),
        Vb((size_t)r * r), ev(r);
    ops_.d2h(G.data(), dGX, sizeof(double) * R_ * R_);
    const uint64_t base_idx = lr_random_calls_++ * (uint64_t)(R_ * r);
    for (int k = 0; k < r; k++)
      for (int a = 0; a < R_; a++) Q[a + (size_t)R_ * k] = u01_host(LR_RANDOM_SEED, base_idx + a + (uint64_t)R_ * k);
    host_qr_mgs2(R_, r, Q.data());
    for (int k = 0; k < r; k++)
      for (int a = 0; a < R_; a++) {
        double v = 0;
        for (int b = 0; b < R_; b++) v += G[a + (size_t)R_ * b] * Q[b + (size_t)R_ * k];
        Y[a + (size_t)R_ * k] = v;
      }
    host_qr_mgs2(R_, r, Y.data());  // Y = Q of the second factorisation
    for (int k = 0; k < r; k++)     // C = Y^T G Y
      for (int l = 0; l < r; l++) {
        double v = 0;
        for (int a = 0; a < R_; a++) {
          double ga = 0;
          for (int b = 0; b < R_; b++) ga += G[a + (size_t)R_ * b] * Y[b + (size_t)R_ * l];
          v += Y[a + (size_t)R_ * k] * ga;
        }
        C[k + (size_t)r * l] = v;
      }
    for (int k = 0; k < r; k++)  // exactly symmetric input for the Jacobi sweeps
      for (int l = k + 1; l < r; l++) C[k + (size_t)r * l] = C[l + (size_t)r * k] = 0.5 * (C[k + (size_t)r * l] + C[l + (size_t)r * k]);
    host_jacobi_eig(r, C.data(), ev.data(), Vb.data());
    for (int k = 0; k < r; k++)
      for (int a = 0; a < R_; a++) {
        double v = 0;
        for (int l = 0; l < r; l++) v += Y[a + (size_t)R_ * l] * Vb[l + (size_t)r * k];
        Vr[a + (size_t)R_ * k] = v;
      }
    ops_.h2d(dV, Vr.data(), sizeof(double) * R_ * r);
  }
  for (int k = 0; k < r; k++)  // VT[k, :] = v_k^T L^-1
    for (int c = 0; c < R_; c++) {
      double v = 0;
      for (int q = c; q < R_; q++) v += Vr[q + (size_t)R_ * k] * Li[q + (size_t)R_ * c];
      VT[k + (size_t)r * c] = v;
    }
  ops_.h2d(lr_small_, VT.data(), sizeof(double) * r * R_);
  ops_.rows_times_small(lr_X_, s, R_, dV, r, nullptr, lr_Us_);       // Us = X V_r = U_r s_r
  ops_.rows_times_small(lr_Us_, s, r, lr_small_, R_, base, W_[i]);   // W = base + Us VT
  ops_.gram(W_[i], s, s, R_, G_ + (size_t)i * R_ * R_);
  (void)nsr;
  ms_nodes_[leaf].t.valid = false;
}

int CpEngine::run_class(int kind, const CpOpts &o, double *sweeps_out, int *iters_out) {
  if (kind < 0 || kind > 4) throw std::runtime_error("ppals: unknown class-API optimizer");
  const bool lr = kind >= 3;
  if (lr) {
    if (dist_) throw std::runtime_error("ppals: the low-rank-update optimizers run on one GPU");
    if (N_ < 3) throw std::runtime_error("ppals: the low-rank-update optimizers need order >= 3");
    if (o.update_rank < 1 || o.update_rank > R_)
      throw std::runtime_error("ppals: update rank must be in [1, R]");
    for (int m = 0; m < MAX_ORDER; m++) lr_have_[m] = false;
    for (auto &nd : nodes_) nd.valid = false;
    ms_invalidate();
    lr_random_ = o.randomsvd != 0;
    lr_random_calls_ = 0;
  }
  // CPDTLROptimizer state (cp_dt_optimizer.cxx:24-37, cp_dt_lr_optimizer.cxx:9-33)
  bool lr_first = true, lr_low = false;
  int lr_left1 = N_ - 1, lr_left2 = N_ - 2, lr_special = 0, lr_count = 0;
  // CPMSDTLROptimizer state (cp_msdt_optimizer.cxx:28, cp_msdt_lr_optimizer.cxx:9-27)
  int lr_msleft = N_;
  bool lr_cached[MAX_ORDER] = {false};
  std::vector<double *> lr_oldW(N_, nullptr);
  const int ur = o.update_rank;
  auto lr_step = [&]() -> double {
    if (kind == 3) {
      const int left = lr_first ? lr_left1 : lr_left2;
      lr_step_begin(left, lr_low && lr_count > 1, ur);
      for (int p = 0; p < N_ - 1; p++) {
        if (lr_first && p < lr_special) continue;
        if (!lr_first && p > lr_special) break;
        const int mode = (left + 1 + p) % N_;
        if (((lr_first && p == N_ - 2) || (!lr_first && p == 0)) && lr_count >= 1) {
          lr_mode_update(mode, o.lambda, ur, nullptr);
          lr_low = true;
        } else {
          lr_mode_update(mode, o.lambda, 0, nullptr);
        }
      }
      if (!lr_first) lr_count++;
      if (lr_count == 5 && !lr_first) {
        lr_special = (lr_special + 1) % (N_ - 1);
        lr_count = 0;
        lr_low = false;
        if (lr_special != 0) {
          lr_left1 = (lr_left1 + N_ - 1) % N_;
          lr_left2 = (lr_left2 + N_ - 1) % N_;
        } else {
          lr_left1 = N_ - 1;
          lr_left2 = N_ - 2;
        }
        // the reference keeps two cached first contractions (cached_tensor1 / cached_tensor2,
        // cp_dt_lr_optimizer.cxx:21-35): the roots that just left the pair give theirs back
        // (s^(N-1) R each — at order 5-6 holding all N of them runs the device out of memory)
        for (int m = 0; m < N_; m++)
          if (m != lr_left1 && m != lr_left2 && lr_cache_[m]) {
            ops_.sync();
            ops_.free(lr_cache_[m]);
            lr_cache_[m] = nullptr;
            lr_have_[m] = false;
          }
      }
      lr_first = !lr_first;
      return 0.5;
    }
    lr_msleft = (lr_msleft + N_ - 1) % N_;
    const int left = lr_msleft;
    const bool reuse = lr_low && lr_cached[left];
    lr_step_begin(left, reuse, ur);
    if (!lr_oldW[left]) lr_oldW[left] = (double *)ops_.alloc(sizeof(double) * V_.glens[left] * R_);
    ops_.d2d(lr_oldW[left], W_[left], sizeof(double) * V_.glens[left] * R_);
    lr_cached[left] = true;
    for (int p = 0; p < N_ - 1; p++) {
      const int mode = (left + 1 + p) % N_;
      if (!lr_cached[mode] || p != N_ - 2) {
        lr_mode_update(mode, o.lambda, 0, nullptr);
      } else {
        lr_mode_update(mode, o.lambda, ur, lr_oldW[mode]);
        lr_low = true;
      }
    }
    return 1.0 * (N_ - 1) / N_;
  };
  struct FreeOld {
    Ops &ops;
    std::vector<double *> &v;
    ~FreeOld() {
      for (auto p : v) ops.free(p);
    }
  } free_old{ops_, lr_oldW};
  std::ofstream csv;
  std::ofstream *pcsv = nullptr;
  if (rank_ == 0 && !o.csv_path.empty()) {
    csv.open(o.csv_path, o.csv_append ? std::ios::app : std::ios::out);
    pcsv = &csv;
    if (!o.bench) csv << "[dim],[iter],[gradnorm],[tol],[pp_update],[diffV],[dtime]\n";
  }
  st_time_ = now();
  int iters = 0, next_mode = 0;
  bool first_subtree = true;
  double sweeps = 0, projnorm = 0, diffV = 1000.;
  const int maxsweep = o.maxiter;
  while ((int)sweeps <= maxsweep) {
    if (iters % o.resprint == 0 || sweeps >= maxsweep || sweeps == 0) {
      ops_.sync();
      const double st_time1 = now();
      projnorm = gradnorm();
      diffV = residual();
      st_time_ += now() - st_time1;
      const double dtime = now() - st_time_;
      if (rank_ == 0) {
        if (!o.bench) {
          if (o.verbose) {
            std::cout.precision(13);
            std::cout << "  [dim]=  " << V_.glens[0] << "  [sweeps]=  " << sweeps
                      << "  [gradnorm]  " << projnorm << "  [tol]  " << o.tol
                      << "  [pp_update]  " << 0 << "  [residual]  " << diffV << "  [dtime]  "
                      << dtime << "\n";
          }
          if (pcsv) {
            (*pcsv) << V_.glens[0] << "," << sweeps << "," << projnorm << "," << o.tol << "," << 0
                    << "," << diffV << "," << dtime << "\n";
            if (iters % 100 == 0 && iters != 0) (*pcsv) << std::endl;
          }
        } else if (iters != 0) {
          if (o.verbose) std::cout << "  [dimension tree step time]  " << dtime << "\n";
          if (pcsv) (*pcsv) << "[DTtime]" << "," << dtime << "\n";
        }
      }
      if (agree(projnorm < o.tol || now() - st_time_ > o.timelimit)) break;
    }
    int count;
    double frac;
    if (lr) {
      sweeps += lr_step();
      grad_from_sweep_ = true;
      iters += 1;
      if (iters % 10 == 0 && rank_ == 0 && o.verbose) printf(".");
      continue;
    }
    if (kind == 0) {
      count = N_;
      frac = 1.0;
    } else if (kind == 1) {
      count = first_subtree ? N_ - 1 : 1;
      first_subtree = !first_subtree;
      frac = 0.5;
    } else {
      count = N_ - 1;
      frac = 1.0 * (N_ - 1) / N_;
    }
    update_modes(next_mode, count, o.lambda);
    next_mode = (next_mode + count) % N_;
    sweeps += frac;
    iters += 1;
    if (iters % 10 == 0 && rank_ == 0 && o.verbose) printf(".");
  }
  ops_.sync();
  if (rank_ == 0 && o.verbose) {
    printf("\nIters = %d Final proj-grad norm %E \n", iters, projnorm);
    printf("tf took %lf seconds\n", now() - st_time_);
  }
  if (pcsv) csv.close();
  if (sweeps_out) *sweeps_out = sweeps;
  if (iters_out) *iters_out = iters;
  return sweeps == maxsweep + 1 ? 0 : 1;
}

double CpEngine::allreduce_scalar(double x) {
  if (!dist_) return x;
  ops_.h2d(scal_, &x, sizeof(double));
  comm_.allreduce_sum(scal_, 1);
  double y = 0;
  ops_.d2h(&y, scal_, sizeof(double));
  return y;
}

// A stop decision that involves this rank's own wall clock (the time limit) must be the same on
// every rank, or one rank leaves the loop while the others enter the next sweep's collective:
// any rank over the limit stops all of them (one scalar all-reduce, print blocks only).
bool CpEngine::agree(bool local) {
  if (!dist_) return local;
  return allreduce_scalar(local ? 1.0 : 0.0) > 0.0;
}

double CpEngine::gradnorm() {
  if (!grad_from_sweep_) return init_gradnorm_;
  double h[MAX_ORDER];
  ops_.d2h(h, gradsq_, sizeof(double) * N_);
  double part = 0, repl = 0;  // row-block partial sums vs. sums every rank already holds in full
  for (int i = 0; i < N_; i++) (grad_replicated_[i] ? repl : part) += h[i];
  return std::sqrt(allreduce_scalar(part) + repl);
}

double CpEngine::residual() {
  int64_t M, K;
  split_sizes(V_, &M, &K);
  if (!Qbuf_) Qbuf_ = (double *)ops_.alloc(sizeof(double) * M * R_);
  if (!Pbuf_) Pbuf_ = (double *)ops_.alloc(sizeof(double) * K * R_);
  split_krp(ops_, V_, R_, W_.data(), Qbuf_, Pbuf_, &M, &K);
  ops_.residual_sq(V_.data, V_.dtype, M, K, Qbuf_, Pbuf_, R_, scal_);
  if (dist_) comm_.allreduce_sum(scal_, 1);
  double h = 0;
  ops_.d2h(&h, scal_, sizeof(double));
  return std::sqrt(h);
}

// ---------------------------------------------------------------------------- kernel-level access
static bool parse_range(const std::string &key, int N, int *lo, int *hi) {
  if (key.empty()) return false;
  *lo = key[0] - 'a';
  *hi = key.back() - 'a';
  if (*lo < 0 || *hi >= N || *hi - *lo + 1 != (int)key.size()) return false;
  for (size_t k = 0; k < key.size(); k++)
    if (key[k] != 'a' + *lo + (int)k) return false;
  return true;
}
int64_t CpEngine::tree_node(const std::string &key, double *out_host) {
  int lo, hi;
  if (!parse_range(key, N_, &lo, &hi)) return -1;
  int idx = find_node(lo, hi);
  if (idx < 0) return -1;
  for (auto &n : nodes_) n.valid = false;
  compute_node(idx);
  const Node &n = nodes_[idx];
  if (out_host) ops_.d2h(out_host, n.buf, sizeof(double) * (size_t)n.elems * R_);
  return n.elems * R_;
}
void CpEngine::mttkrp(int mode, double *M_host) {
  for (auto &n : nodes_) n.valid = false;
  compute_node(leaf_[mode]);
  const Node &lf = nodes_[leaf_[mode]];
  const int64_t s = V_.glens[mode];
  if (!dist_) {
    ops_.d2h(M_host, lf.buf, sizeof(double) * s * R_);
    return;
  }
  const int64_t blk = block_rows(s, P_);
  if (mode == 0) {
    ops_.zero(sendbuf_, sizeof(double) * blk * P_ * R_);
    // place the owner's complete rows into its block, then sum (other blocks are zero)
    double *mine = sendbuf_ + (size_t)rank_ * blk * R_;
    for (int r = 0; r < R_; r++)
      ops_.d2d(mine + (size_t)r * blk, lf.buf + (size_t)r * ext(0), sizeof(double) * ext(0));
  } else {
    ops_.pack_blocks(lf.buf, s, s, R_, blk, P_, sendbuf_);
  }
  comm_.allreduce_sum(sendbuf_, blk * P_ * R_);
  double *nat = (double *)ops_.alloc(sizeof(double) * s * R_);
  ops_.unpack_blocks(sendbuf_, s, s, R_, blk, P_, nat);
  ops_.d2h(M_host, nat, sizeof(double) * s * R_);
  ops_.free(nat);
}
void CpEngine::gram_system(int mode, double lambda, double *S_host, double *Sinv_host) {
  ops_.gram_system(G_, N_, mode, R_, lambda, S_, Sinv_);
  ops_.d2h(S_host, S_, sizeof(double) * R_ * R_);
  ops_.d2h(Sinv_host, Sinv_, sizeof(double) * R_ * R_);
}

// ---------------------------------------------------------------------------- PP operators
// Build_mttkrp_map (als_CP.cxx:352-409): key = contracted modes in ascending order; built by
// dropping the last contracted mode. Level 1 scans V (K8), deeper levels contract the cache.
// Build_mttkrp_map (als_CP.cxx:385-390) derives the operator with the contracted modes `seq` from the
// one without seq's LAST mode, so every chain starts by contracting its lowest mode. Here the mode
// removed last is the SHORTEST of seq — the chain's level-1 tensor then contracts its longest mode and
// is the smallest possible (coil-100, 3 x 128 x 128 x 7200, R = 10: V x_a W_a is 3.3 x the tensor,
// 4.7 GB written once and read three times per build; V x_d W_d is 2 MB). Same operators, another
// association order; ties keep the reference's order, so equal extents change nothing.
int CpEngine::pp_last_mode(const std::string &seq) const {
  int best = seq.back() - 'a';
  for (int q = (int)seq.size() - 2; q >= 0; q--)
    if (ext(seq[q] - 'a') < ext(best)) best = seq[q] - 'a';
  return best;
}
std::string CpEngine::pp_parent(const std::string &seq) const {
  std::string p = seq;
  p.erase(p.find((char)('a' + pp_last_mode(seq))), 1);
  return p;
}

const CpEngine::PPOp &CpEngine::pp_get(const std::string &seq) {
  check_tensor_generation();
  auto it = pp_.find(seq);
  if (it != pp_.end()) return it->second;
  const int mode = pp_last_mode(seq);
  PPOp op;
  FactorRef f = fref(mode, W_.data());
  int64_t L = 1, T = 1;
  if (seq.size() == 1) {
    // level 1 = one tensor scan. Modes of the left half are contracted on the second resident
    // layout, where they sit behind the right half, so that EVERY level-1 scan is a
    // row-contiguous suffix-type scan; the s^(N-1) R result is kept in the tensor's precision.
    ScanPlan pl;
    if (!plan_scan(mode, 1, !(pp_fast_ && N_ >= 3), pl))
      throw std::runtime_error("ppals: internal error (single mode not adjacent)");
    L = pl.Lc;
    T = pl.T;
    op.modes = pl.kept;
    op.elems = L * T;
    // (at order 3 a level-1 result already IS a pair operator: those stay fp64 like all the others)
    op.dt = (pp_fast_ && N_ > 3) ? V_.dtype : F64;
    if (pp_fast_ && N_ > 3 && schedule_ == 1 && ms_k_ == 1 && ms_root_ == mode && ms_X_.valid &&
        ms_X_.dt == op.dt && !pp_no_borrow_) {
      // The exact sweep that ended just now left its first-level intermediate X = V x_mode W_mode
      // in the multi-sweep cache, and W_mode has not changed since (a step never updates its own
      // root; Normalize is the pending scalar): that IS this level-1 operator. One of the three
      // tensor scans of the build (als_CP.cxx:377-379) is already there.
      op.buf = ms_X_.buf;
      op.owned = false;
      op.scale = ms_scale_of(ms_X_);
      op.modes = ms_X_.modes;
    } else {
      op.buf = pp_buffer(seq, dtype_size(op.dt) * (size_t)(V_.nloc / ext(mode)) * R_);
      // (planned again: the allocation may have given a resident layout back)
      if (!plan_scan(mode, 1, !(pp_fast_ && N_ >= 3), pl))
        throw std::runtime_error("ppals: internal error (single mode not adjacent)");
      L = pl.Lc;
      T = pl.T;
      op.modes = pl.kept;
      ops_.scan_contract(pl.lay->ptr, V_.dtype, pl.L, ext(mode), T, &f, 1, R_, op.buf, op.dt, L,
                         op.elems, pl.pad);
    }
  } else {
    const PPOp &par = pp_get(pp_parent(seq));
    bool before = true;
    for (int m : par.modes) {
      if (m == mode) {
        before = false;
        continue;
      }
      (before ? L : T) *= ext(m);
      op.modes.push_back(m);
    }
    op.elems = L * T;
    op.buf = pp_buffer(seq, sizeof(double) * (size_t)op.elems * R_);
    ops_.mttv(par.buf, par.dt, L, ext(mode), T, &f, 1, R_, (double *)op.buf, op.elems, 0, par.scale);
  }
  pp_[seq] = op;
  return pp_[seq];
}
void CpEngine::pp_contract_pair(const PPOp &T, int cmode, const FactorRef &f, double *out,
                                int64_t out_rows) {
  if (T.modes.size() != 2 || T.dt != F64) throw std::runtime_error("ppals: not a pair operator");
  if (T.modes[0] == cmode)  // T[cmode, keep, r]
    ops_.mttv(T.buf, F64, 1, ext(cmode), ext(T.modes[1]), &f, 1, R_, out, out_rows, 1, nullptr);
  else  // T[keep, cmode, r]
    ops_.mttv(T.buf, F64, ext(T.modes[0]), ext(cmode), 1, &f, 1, R_, out, out_rows, 1, nullptr);
}
void CpEngine::pp_clear() { pp_.clear(); }  // the buffers stay in the pool
// Buffer of the operator `seq` from the grow-only pool. What the approximate sweeps read — the pair
// operators (N-2 modes contracted) and the full MTTKRPs — has a buffer of its own per key.
// Everything above them in the recursion of Build_mttkrp_map is scaffolding, read only while the
// operators below it are built; pp_build_all visits the keys in an order in which all users of a
// prefix are consecutive, so ONE buffer per level serves every scaffold of that level (order 4:
// one s^3 R level-1 tensor instead of three; order 6, s = 50, R = 6: 7.5 GB instead of 37.5 GB).
// A scaffold that takes the level's buffer over evicts the previous holder from the cache.
void *CpEngine::pp_buffer(const std::string &seq, size_t bytes) {
  const bool scaffold = (int)seq.size() < N_ - 2;
  const std::string key = scaffold ? "#" + std::to_string(seq.size()) : seq;
  if (scaffold) {
    for (auto it = pp_.begin(); it != pp_.end();) {
      if (it->first.size() == seq.size() && it->second.owned) {
        it = pp_.erase(it);
      } else {
        ++it;
      }
    }
  }
  PPBuf &b = pp_pool_[key];
  if (b.cap < bytes) {
    ops_.free(b.buf);
    b.buf = nullptr;
    b.cap = 0;
    b.buf = big_alloc(bytes);
    b.cap = bytes;
  }
  return b.buf;
}
static std::string all_but(int N, int i, int j = -1) {
  std::string s;
  for (int m = 0; m < N; m++)
    if (m != i && m != j) s.push_back((char)('a' + m));
  return s;
}
// als_CP.cxx:678-694: all pair operators, then all N full MTTKRPs
void CpEngine::pp_build_all() {
  struct Timer {  // (only while a caller asked for it: ppals_cp_pp_build_stats)
    CpEngine &e;
    double t0 = 0;
    explicit Timer(CpEngine &eng) : e(eng) {
      e.pp_builds_++;
      if (e.pp_build_timed_) {
        e.ops_.sync();
        t0 = now();
      }
    }
    ~Timer() {
      if (e.pp_build_timed_) {
        try {
          e.ops_.sync();
        } catch (...) {
        }
        e.pp_build_s_ += now() - t0;
      }
    }
  } timer(*this);
  pp_clear();
  // the pair operators in an order in which all users of a scaffold are consecutive (pp_buffer keeps
  // ONE buffer per scaffold level): sorted by their chain of ancestors, level 1 first
  std::vector<std::pair<std::string, std::string>> pairs;  // (chain, key)
  for (int ii = 0; ii < N_; ii++)
    for (int jj = ii + 1; jj < N_; jj++) {
      const std::string key = all_but(N_, ii, jj);
      std::string chain = key;
      for (std::string k = key; k.size() > 1;) {
        k = pp_parent(k);
        chain = k + "|" + chain;
      }
      pairs.emplace_back(chain, key);
    }
  std::stable_sort(pairs.begin(), pairs.end(),
                   [](const std::pair<std::string, std::string> &a, const std::pair<std::string, std::string> &b) {
                     return a.first < b.first;
                   });
  for (const auto &pk : pairs) pp_get(pk.second);
  for (int ii = 0; ii < N_; ii++) pp_get(all_but(N_, ii));
  // the approximate sweeps read the pair operators (N-2 modes contracted) and the full MTTKRPs
  // only; everything above them in the recursion was scaffolding — at order 6 that is 45 GB of
  // level-1 tensors
  for (auto it = pp_.begin(); it != pp_.end();) {
    if ((int)it->first.size() < N_ - 2) {
      it = pp_.erase(it);
    } else {
      ++it;
    }
  }
  // ... and its buffers go back to the device when they are large (above 2 GiB: below that, keeping
  // one per level saves the hipMalloc / hipFree of every phase)
  const size_t keep = (size_t)2 << 30;
  for (auto it = pp_pool_.begin(); it != pp_pool_.end();) {
    if (it->first[0] == '#' && it->second.cap > keep) {
      ops_.free(it->second.buf);
      it = pp_pool_.erase(it);
    } else {
      ++it;
    }
  }
}
std::string CpEngine::placement_report() const {
  char buf[400];
  std::string out = "{\"mode\": ";
  out += ms_tune_enabled_ && ms_X_slack() > 0 ? "\"online\"" : "\"off\"";
  out += ", \"setup_s\": 0.0, \"candidate_blocks_held\": ";
  out += ms_X_alt_ ? "2" : "1";
  out += ", \"roots\": [";
  bool firstrow = true;
  for (int r = 0; r < N_; r++) {
    const PlaceExplore &ex = ms_place_[r];
    if (ex.phase < 0 || ex.phase == 3 || ex.cands.empty()) continue;
    const PlaceCand *c = ex.chosen >= 0 ? &ex.cands[ex.chosen] : nullptr;
    double best = 1e300;
    for (const auto &q : ex.cands)
      if (q.samples > 0) best = std::min(best, q.best);
    snprintf(buf, sizeof buf,
             "%s{\"root\": %d, \"layout\": \"%s\", \"settled\": %s, \"visits\": %d, \"block\": %d, \"offset_mb\": %lld, "
             "\"store\": \"%s\", \"best_ms\": %.4f, \"worst_ms\": %.4f, \"gated\": %s}",
             firstrow ? "" : ", ", r, ex.layout == 1 ? "second (transposed) copy" : "tensor",
             ex.phase == 2 ? "true" : "false", ex.visits, c ? c->blk : 0, c ? (long long)(c->off >> 20) : 0LL,
             !c || c->nt < 0 ? "by size" : (c->nt == 1 ? "non-temporal" : "ordinary"),
             best < 1e299 ? best * 1e3 : 0.0, ex.worst * 1e3, ex.gated ? "true" : "false");
    out += buf;
    firstrow = false;
  }
  out += "]}";
  return out;
}

int64_t CpEngine::pp_operator(const std::string &contracted, double *out_host) {
  for (size_t k = 0; k < contracted.size(); k++) {
    int m = contracted[k] - 'a';
    if (m < 0 || m >= N_ || (k > 0 && contracted[k] <= contracted[k - 1])) return -1;
  }
  if (contracted.empty() || (int)contracted.size() >= N_) return -1;
  pp_clear();
  // (never the borrowed multi-sweep intermediate here: it carries a Normalize factor that only the
  // contractions below it apply — this entry point hands the operator itself to the caller)
  struct Restore {  // (pp_get may throw: the flags must not outlive the call)
    CpEngine &e;
    bool fast;
    ~Restore() {
      e.pp_fast_ = fast;
      e.pp_no_borrow_ = false;
    }
  } restore{*this, pp_fast_};
  pp_no_borrow_ = true;
  const PPOp *op = &pp_get(contracted);
  if (op->dt != F64 || !std::is_sorted(op->modes.begin(), op->modes.end())) {
    // the entry point returns fp64 with the remaining modes ascending: rebuild on the plain route
    pp_clear();
    pp_fast_ = false;
    op = &pp_get(contracted);
  }
  if (out_host) ops_.d2h(out_host, op->buf, sizeof(double) * (size_t)op->elems * R_);
  int64_t n = op->elems * R_;
  pp_clear();
  return n;
}

// one approximate sweep: als_CP.cxx:754-825. Per mode: ONE launch for M = M_i^0 + the N-1
// first-order corrections, ONE for the whole mode update (which also leaves ||dW_i||^2); the
// Normalize launch leaves ||W_i||^2: the restart test of the next iteration costs no launch.
void CpEngine::sweep_pp(double lambda, double ratio) {
  ms_invalidate();  // PP moves the factors without touching the multi-sweep cache
  if (!Mbuf_) Mbuf_ = (double *)ops_.alloc(sizeof(double) * (size_t)maxs_ * R_);
  if (!pp_norms_) {
    pp_norms_ = (double *)ops_.alloc(sizeof(double) * 2 * MAX_ORDER);
    ops_.zero(pp_norms_, sizeof(double) * 2 * MAX_ORDER);
  }
  bool norm_fused = false;
  for (int i = 0; i < N_; i++) {
    const int64_t si = ext(i);
    const PPOp &M0 = pp_get(all_but(N_, i));
    PPTerm terms[MAX_ORDER];
    int nt = 0;
    for (int ii = 0; ii < N_; ii++) {
      if (ii == i) continue;
      const PPOp &T = pp_get(all_but(N_, std::min(i, ii), std::max(i, ii)));
      if (T.modes.size() != 2 || T.dt != F64) throw std::runtime_error("ppals: not a pair operator");
      FactorRef f = fref(ii, dW_.data());  // als_CP.cxx:785 / :793
      terms[nt].T = (const double *)T.buf;
      terms[nt].ny = ext(ii);
      terms[nt].keep_first = T.modes[0] == i ? 1 : 0;
      terms[nt].dW = f.ptr;
      terms[nt].lddw = f.ld;
      nt++;
    }
    // (S and S^-1 of this mode depend on the other modes' Grams only: a back end may prepare
    // them on the side of the correction's launch)
    ops_.arm_gram_system(G_, N_, i, R_, lambda, S_, Sinv_);
    ops_.pp_correct((const double *)M0.buf, si, R_, terms, nt, Mbuf_);
    if (i == N_ - 1 && !dist_) {  // Normalize at the tail of the sweep's last update launch
      int64_t rows[MAX_ORDER];
      for (int q = 0; q < N_; q++) rows[q] = V_.glens[q];
      norm_fused = ops_.arm_normalize(W_.data(), rows, N_, R_, G_, i, pp_norms_ + 1);
    }
    mode_update(i, Mbuf_, si, lambda, true, ratio);
  }
  if (!norm_fused) {
    pp_norms_live_ = !dist_;
    normalize();
    pp_norms_live_ = false;
  }
  grad_from_sweep_ = true;
}

// ---------------------------------------------------------------------------- drivers
static void csv_row(std::ofstream *csv, int64_t dim, int iter, double gradnorm, double tol, int pp,
                    double diffV, double dtime) {
  if (!csv) return;
  (*csv) << dim << "," << iter << "," << gradnorm << "," << tol << "," << pp << "," << diffV << ","
         << dtime << "\n";
  if (iter % 100 == 0 && iter != 0) (*csv) << std::endl;  // als_CP.cxx:199-201
}

// print block: als_CP.cxx:166-213 / :457-498 / :697-752. Its own duration is subtracted from the
// elapsed time exactly as the reference does (st_time += ...).
bool CpEngine::print_block(const CpOpts &o, int iter, int pp_flag, double &projnorm, double &diffV,
                           std::ofstream *csv) {
  ops_.sync();  // the sweeps enqueued so far belong to [dtime]
  const double st_time1 = now();
  projnorm = gradnorm();
  diffV = residual();
  st_time_ += now() - st_time1;
  const double dtime = now() - st_time_;
  if (rank_ == 0) {
    if (o.verbose) {
      std::cout.precision(13);
      std::cout << "  [dim]=  " << V_.glens[0] << "  [iter]=  " << iter << "  [gradnorm]  "
                << projnorm << "  [tol]  " << o.tol << "  [pp_update]  " << pp_flag
                << "  [diffV]  " << diffV << "  [dtime]  " << dtime << "\n";
    }
    csv_row(csv, V_.glens[0], iter, projnorm, o.tol, pp_flag, diffV, dtime);
  }
  return agree((projnorm < o.tol) || (now() - st_time_ > o.timelimit));
}

int CpEngine::run_dt(const CpOpts &o, int *iters) {
  std::ofstream csv;
  std::ofstream *pcsv = nullptr;
  if (rank_ == 0 && !o.csv_path.empty()) {
    csv.open(o.csv_path, o.csv_append ? std::ios::app : std::ios::out);
    pcsv = &csv;
    if (!o.bench) csv << "[dim],[iter],[gradnorm],[tol],[pp_update],[diffV],[dtime]\n";
  }
  st_time_ = now();
  double projnorm = 0, diffV = 1000;
  int iter;
  for (iter = 0; iter <= o.maxiter; iter++) {
    if (iter % o.resprint == 0 || iter == o.maxiter) {
      if (!o.bench) {
        if (print_block(o, iter, 0, projnorm, diffV, pcsv)) break;
      } else {
        // pp_bench mode (als_CP.cxx:203-209): only the elapsed time of the sweep is reported
        ops_.sync();
        const double st_time1 = now();
        projnorm = gradnorm();
        diffV = residual();
        st_time_ += now() - st_time1;
        const double dtime = now() - st_time_;
        if (rank_ == 0 && iter != 0) {
          if (o.verbose) std::cout << "  [dimension tree step time]  " << dtime << "\n";
          if (pcsv) (*pcsv) << "[DTtime]" << "," << dtime << "\n";
        }
        if (agree(projnorm < o.tol || now() - st_time_ > o.timelimit)) break;
      }
    }
    sweep_dt(o.lambda);
    if (iter % 10 == 0 && rank_ == 0 && o.verbose) printf(".");
  }
  ops_.sync();
  if (rank_ == 0 && o.verbose) {
    printf("\nIter = %d Final proj-grad norm %E \n", iter, projnorm);
    printf("tf took %lf seconds\n", now() - st_time_);
  }
  if (pcsv) csv.close();
  if (iters) *iters = iter;
  return iter == o.maxiter + 1 ? 0 : 1;
}

// ||dW_i||^2 and ||W_i||^2 for all modes. dt_phase: dW = W - W_prev, W_prev = W (als_CP.cxx:594-
// 603); otherwise dW as left by the PP updates (als_CP.cxx:659-663).
void CpEngine::read_norms(bool dt_phase, std::vector<double> &nd, std::vector<double> &nw) {
  int64_t n[MAX_ORDER];
  for (int i = 0; i < N_; i++) n[i] = V_.glens[i] * R_;
  if (dt_phase)
    ops_.diff_norms(W_.data(), Wprev_.data(), n, N_, 1, dW_.data(), 1, scal_);
  else
    ops_.diff_norms(W_.data(), nullptr, n, N_, 0, dW_.data(), 0, scal_);
  double h[2 * MAX_ORDER];
  ops_.d2h(h, scal_, sizeof(double) * 2 * N_);
  nd.resize(N_);
  nw.resize(N_);
  for (int i = 0; i < N_; i++) {
    nd[i] = std::sqrt(h[2 * i]);
    nw[i] = std::sqrt(h[2 * i + 1]);
  }
}

double CpEngine::dt_sub(const CpOpts &o, double &projnorm, int &iter, std::ofstream *csv) {
  double diffV = 1000;
  for (int i = 0; i < N_; i++)  // W_prev starts at zero (als_CP.cxx:428-431)
    ops_.zero(Wprev_[i], sizeof(double) * V_.glens[i] * R_);
  std::vector<double> nd, nw;
  for (; iter <= o.maxiter; iter++) {
    if (iter % o.resprint == 0 || iter == o.maxiter) {
      if (print_block(o, iter, 0, projnorm, diffV, csv)) break;
    }
    sweep_dt(o.lambda);
    read_norms(true, nd, nw);
    int num_dw_break = 0;
    for (int i = 0; i < N_; i++)
      if (std::fabs(nd[i] / nw[i]) < o.tol_init) num_dw_break++;
    if (num_dw_break == N_) return diffV;  // iter is NOT incremented (als_CP.cxx:604-605)
    if (iter % 10 == 0 && rank_ == 0 && o.verbose) printf(".");
  }
  return diffV;
}

double CpEngine::pp_sub(const CpOpts &o, double &projnorm, int &iter, std::ofstream *csv) {
  const int init_iter = iter;
  double diffV = 1000;
  double dtime_first = 0;
  std::vector<double> nd, nw;
  for (; iter <= o.maxiter; iter++) {
    int num_dw_break = 0;
    if (!o.bench) {
      if (iter == init_iter || dist_ || !pp_norms_) {
        read_norms(false, nd, nw);  // dW as the exact phase left it
      } else {  // left behind by the launches of the previous approximate sweep
        double h[2 * MAX_ORDER];
        ops_.d2h(h, pp_norms_, sizeof(double) * 2 * N_);
        nd.resize(N_);
        nw.resize(N_);
        for (int i = 0; i < N_; i++) {
          nd[i] = std::sqrt(h[2 * i]);
          nw[i] = std::sqrt(h[2 * i + 1]);
        }
      }
      for (int i = 0; i < N_; i++)
        if (std::fabs(nd[i] / nw[i]) > o.tol_init) num_dw_break++;
    }
    if ((iter - init_iter) % 15 == 0 || num_dw_break > 0) {
      if (num_dw_break > 0 || iter != init_iter) return diffV;
      for (int j = 0; j < N_; j++) {  // W_init = W, dW = 0 (als_CP.cxx:672-675)
        size_t n = sizeof(double) * V_.glens[j] * R_;
        ops_.d2d(Winit_[j], W_[j], n);
        ops_.zero(dW_[j], n);
      }
      pp_build_all();
    }
    if (iter % o.resprint == 0 || iter == o.maxiter || iter == init_iter) {
      if (!o.bench) {
        if (print_block(o, iter, 1, projnorm, diffV, csv)) break;
      } else {
        // pp_bench mode (als_CP.cxx:735-748)
        ops_.sync();
        const double st_time1 = now();
        projnorm = gradnorm();
        diffV = residual();
        st_time_ += now() - st_time1;
        const double dtime = now() - st_time_;
        if (iter != o.maxiter) {
          dtime_first = dtime;
          st_time_ = now();
        } else {
          dtime_first = dtime_first + dtime;
          if (rank_ == 0) {
            if (o.verbose) {
              std::cout << "  [PP first time]  " << dtime_first << "\n";
              std::cout << "  [PP second time]  " << dtime << "\n";
            }
            if (csv) {
              (*csv) << "  [PPfirst]  " << "," << dtime_first << "\n";
              (*csv) << "  [PPsecond]  " << "," << dtime << "\n";
            }
          }
        }
        if (agree(projnorm < o.tol || now() - st_time_ > o.timelimit)) break;
      }
    }
    sweep_pp(o.lambda, o.ratio_step);
    if (iter % 10 == 0 && rank_ == 0 && o.verbose) printf(".");
  }
  if (o.bench) iter++;
  return diffV;
}

// alsCP_PP_partupdate_sub (als_CP.cxx:852-1073): PP phase that updates only the modes with the
// largest relative MTTKRP perturbation ||dM_i|| / ||M_i|| and propagates every update to the other
// modes' dM through the cached pair operators. Sharded: dM_i (i != 0) is a partial sum, completed
// on a copy for its norm; M_i is completed in place by its own mode update.
double CpEngine::pp_partupdate_sub(const CpOpts &o, double &projnorm, int &iter,
                                   std::ofstream *csv) {
  const int init_iter = iter;
  double diffV = 1000;
  std::vector<double> nd, nw, relpert(N_, 0.0);
  for (int i = 0; i < N_; i++) {
    size_t n = sizeof(double) * V_.glens[i] * R_;
    ops_.zero(dM_[i], n);
    ops_.zero(Mm_[i], n);
  }
  const int update_size = (int)(N_ * o.update_percentage);
  for (; iter <= o.maxiter; iter++) {
    int num_dw_break = 0;
    read_norms(false, nd, nw);
    for (int i = 0; i < N_; i++)
      if (std::fabs(nd[i] / nw[i]) > o.tol_init) num_dw_break++;
    if ((iter - init_iter) % 15 == 0 || num_dw_break > 0) {
      if (num_dw_break > 0 || iter != init_iter) return diffV;
      for (int j = 0; j < N_; j++) {
        size_t n = sizeof(double) * V_.glens[j] * R_;
        ops_.d2d(Winit_[j], W_[j], n);
        ops_.zero(dW_[j], n);
      }
      pp_build_all();
    }
    if (iter % o.resprint == 0 || iter == o.maxiter || iter == init_iter) {
      if (print_block(o, iter, 1, projnorm, diffV, csv)) break;
    }
    // sort_indexes (als_CP.cxx:835-843): descending, ties keep index order
    std::vector<int> idx(N_);
    for (int i = 0; i < N_; i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return relpert[a] > relpert[b]; });
    if (rank_ == 0 && o.verbose) std::cout << "new round" << std::endl;
    ms_invalidate();
    for (int t = 0; t < update_size; t++) {
      const int i = idx[t];
      if (rank_ == 0 && o.verbose) std::cout << i << std::endl;
      const int64_t si = ext(i);
      const PPOp &M0 = pp_get(all_but(N_, i));
      ops_.d2d(Mm_[i], M0.buf, sizeof(double) * si * R_);
      ops_.add_inplace(Mm_[i], dM_[i], si * R_);
      mode_update(i, Mm_[i], si, o.lambda, true, o.ratio_step);
      ops_.zero(dM_[i], sizeof(double) * si * R_);
      for (int ii = 0; ii < N_; ii++) {  // propagate (als_CP.cxx:1036-1053)
        if (ii == i) continue;
        const PPOp &T = pp_get(all_but(N_, std::min(i, ii), std::max(i, ii)));
        FactorRef f = fref(i, dW_.data());
        pp_contract_pair(T, i, f, dM_[ii], ext(ii));
      }
    }
    for (int i = 0; i < N_; i++) {
      if (dist_ && i != 0) {
        // dM_i is a partial sum over this rank's rows of the sharded mode: complete a copy of it.
        // (Mm_i was completed in place by the all-reduce inside its own mode update.)
        ops_.d2d(sendbuf_, dM_[i], sizeof(double) * ext(i) * R_);
        comm_.allreduce_sum(sendbuf_, ext(i) * R_);
        ops_.sumsq(sendbuf_, ext(i) * R_, scal_ + 2 * i);
      } else {
        ops_.sumsq(dM_[i], ext(i) * R_, scal_ + 2 * i);
      }
      ops_.sumsq(Mm_[i], ext(i) * R_, scal_ + 2 * i + 1);
    }
    // the sharded mode holds complete ROWS: its two sums of squares add up over the ranks
    if (dist_) comm_.allreduce_sum(scal_, 2);
    double h[2 * MAX_ORDER];
    ops_.d2h(h, scal_, sizeof(double) * 2 * N_);
    for (int i = 0; i < N_; i++) relpert[i] = std::sqrt(h[2 * i]) / std::sqrt(h[2 * i + 1]);
    normalize();
    grad_from_sweep_ = true;
    if (iter % 10 == 0 && rank_ == 0 && o.verbose) printf(".");
  }
  return diffV;
}

int CpEngine::run_pp(const CpOpts &o, int *iters) { return run_pp_common(o, iters, false); }
int CpEngine::run_pp_partupdate(const CpOpts &o, int *iters) {
  if (dist_) {
    // sharded: needs the plan in which every rank sees the complete s x R matrices
    for (int i = 0; i < N_; i++)
      if ((int64_t)sizeof(double) * V_.glens[i] * R_ > small_msg_bytes_)
        throw std::runtime_error(
            "ppals: -pp 2 on several GPUs needs s x R matrices below PPALS_COMM_SMALL_BYTES");
  }
  return run_pp_common(o, iters, true);
}

int CpEngine::run_pp_common(const CpOpts &o, int *iters, bool partupdate) {
  std::ofstream csv;
  std::ofstream *pcsv = nullptr;
  if (rank_ == 0 && !o.csv_path.empty()) {
    csv.open(o.csv_path, o.csv_append ? std::ios::app : std::ios::out);
    pcsv = &csv;
    if (!o.bench) csv << "[dim],[iter],[gradnorm],[tol],[pp_update],[diffV],[dtime]\n";
  }
  if (partupdate && rank_ == 0 && o.verbose) std::cout << "alsCP_PP_partupdate starts. " << std::endl;
  for (int i = 0; i < N_; i++) {
    size_t n = sizeof(double) * V_.glens[i] * R_;
    if (!Wprev_[i]) Wprev_[i] = (double *)ops_.alloc(n);
    if (!Winit_[i]) Winit_[i] = (double *)ops_.alloc(n);
    if (!dW_[i]) dW_[i] = (double *)ops_.alloc(n);
    if (partupdate && !dM_[i]) dM_[i] = (double *)ops_.alloc(n);
    if (partupdate && !Mm_[i]) Mm_[i] = (double *)ops_.alloc(n);
    ops_.zero(dW_[i], n);
  }
  st_time_ = now();
  int iter = 0;
  double gradnorm_v = 10.;
  while (gradnorm_v > o.tol && iter <= o.maxiter) {
    if (!o.bench) {
      if (rank_ == 0 && o.verbose) printf("DT starts from %d\n", iter);
      dt_sub(o, gradnorm_v, iter, pcsv);
    }
    if (rank_ == 0 && o.verbose) printf("pairwise perturbation starts from %d\n", iter);
    if (partupdate)
      pp_partupdate_sub(o, gradnorm_v, iter, pcsv);
    else
      pp_sub(o, gradnorm_v, iter, pcsv);
    // deviation from the reference: a timelimit hit terminates instead of looping forever
    // (als_CP.cxx:1105 with breaks at :496 and :750)
    if (agree(now() - st_time_ > o.timelimit)) break;
  }
  ops_.sync();
  pp_clear();
  if (rank_ == 0 && o.verbose) {
    printf("\nIter = %d Final grad norm %E \n", iter, gradnorm_v);
    printf("tf took %lf seconds\n", now() - st_time_);
  }
  if (pcsv) csv.close();
  if (iters) *iters = iter;
  return iter == o.maxiter + 1 ? 0 : 1;
}

}  // namespace ppals
