// tucker.cpp — host control flow of the Tucker HOOI engine. Mirrors:
//   ttmc_map_DT        als_Tucker.cxx:178-230  -> compute_node
//   alsTucker_DT       als_Tucker.cxx:240-424  -> sweep_dt / run_dt
//   hosvd              als_Tucker.cxx:12-70    -> hosvd
//   TTMc               als_Tucker.cxx:76-110   -> ttmc_chain
// Multi-GPU: V is block-partitioned along mode 0, factors are replicated. Every
// contraction over mode 0 uses the local rows of W_0 and yields a PARTIAL sum that stays partial
// down to the leaf tensor Y_i (TTMc is linear), where it is all-reduced (i != 0) or its rows are
// all-gathered (i = 0); Gram + eigenvectors are then computed redundantly on every rank.
// Order generalisation as for CP: a node is "first level" iff its parent is the root, which is
// the reference's length test for N = 4,6,7,8 and defines N = 3 (BASELINE config 5).
#include "tucker.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>

namespace ppals {

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

TuckerEngine::TuckerEngine(Ops &ops, Comm &comm, const TensorDesc &V, const int *ranks)
    : ops_(ops), comm_(comm), V_(V), N_(V.order) {
  P_ = comm.size();
  dist_ = P_ > 1 || (force_comm_path() && !comm.is_self());
  rank_ = comm.rank();
  eig_base_ = ops_.eig_session_new();  // this session's warm-start slots: base + {i, 8 + i, 16 + i}
  // (the back end sets up what lazy eigen-steps need now, not inside the first sweep)
  if (!dist_)
    for (int i = 0; i < N_ && i < MAX_ORDER; i++)
      if (V_.glens[i] > 64) ops_.eig_lazy(eig_base_ + i, true);
  int64_t maxs = 0;
  for (int i = 0; i < N_; i++) {
    if (ranks[i] <= 0 || ranks[i] > V_.glens[i])
      throw std::runtime_error("ppals: Tucker rank out of range");
    r_.push_back(ranks[i]);
    ncore_ *= ranks[i];
    maxs = std::max(maxs, V_.glens[i]);
    W_.push_back((double *)ops_.alloc(sizeof(double) * V_.glens[i] * ranks[i]));
    ops_.zero(W_.back(), sizeof(double) * V_.glens[i] * ranks[i]);
  }
  core_ = (double *)ops_.alloc(sizeof(double) * ncore_);
  core_prev_ = (double *)ops_.alloc(sizeof(double) * ncore_);
  ops_.zero(core_, sizeof(double) * ncore_);
  ops_.zero(core_prev_, sizeof(double) * ncore_);
  G_ = (double *)ops_.alloc(sizeof(double) * maxs * maxs);
  if (const char *e = std::getenv("PPALS_TUCKER_THIN")) thin_enabled_ = std::atoi(e) != 0;
  scal_ = (double *)ops_.alloc(sizeof(double) * 64);
  yend_elems_ = ncore_ / r_[N_ - 1] * V_.glens[N_ - 1];
  Yend_ = (double *)ops_.alloc(sizeof(double) * yend_elems_);
  build_tree(0, N_ - 1, -1);
  leaf_.assign(N_, -1);
  for (size_t k = 0; k < nodes_.size(); k++)
    if (nodes_[k].lo == nodes_[k].hi) leaf_[nodes_[k].lo] = (int)k;
  // second resident layout V^T[(right modes), (left modes)] (as in the CP engine): the first-level
  // node that contracts the LEFT half of the modes then runs as row-contiguous scans too, instead
  // of a column-strided scan of the leading mode followed by scans with a handful of rows
  const char *env = std::getenv("PPALS_TRANSPOSED_COPY");
  if (N_ >= 3 && !(env && std::atoi(env) == 0)) {
    const int mid = (N_ - 1) / 2;
    int64_t rows = 1, cols = 1;
    for (int m = 0; m <= mid; m++) rows *= ext(m);
    for (int m = mid + 1; m < N_; m++) cols *= ext(m);
    VT_ = ops_.try_alloc((size_t)rows * cols * dtype_size(V_.dtype));
    if (VT_) ops_.transpose2d(V_.data, V_.dtype, rows, cols, VT_);
    // the third rotation [1 2 | 0] for the order-3 multi-sweep schedule (tucker.h), room permitting
    const char *chain = std::getenv("PPALS_TUCKER_CHAIN");
    if (N_ == 3 && !dist_ && VT_ && !(chain && std::string(chain) == "tree")) {
      const size_t bytes = (size_t)V_.nloc * dtype_size(V_.dtype);
      const size_t avail = ops_.mem_available();
      if (avail == (size_t)-1 || (double)avail > 2.0 * (double)bytes + 2e9) VT2_ = ops_.try_alloc(bytes);
      if (VT2_) {
        ops_.transpose2d(V_.data, V_.dtype, ext(0), ext(1) * ext(2), VT2_);
        ms3_ = true;
      }
    }
  }
  if (V_.generation) tensor_gen_ = *V_.generation;
}

// The leaf of mode i, [s_i | the other two ranks], from the order-3 multi-sweep schedule; nullptr: the
// back end has no leading-mode product for this shape — the schedule is switched off for the session.
double *TuckerEngine::ms3_leaf(int i, int64_t *T) {
  check_tensor_generation();
  if (ms3_root_ < 0 || ms3_root_ == i || ms3_left_ <= 0) {
    // root = the mode updated last before i: serves i and the mode after it
    const int r = (i + 2) % 3;
    const void *lay = r == 2 ? V_.data : (r == 1 ? VT_ : VT2_);  // storage order (r+1, r+2, r)
    const int64_t d0 = ext((r + 1) % 3), d1 = ext((r + 2) % 3);
    const size_t need = dtype_size(V_.dtype) * (size_t)(d0 * d1 * r_[r]);
    if (ms3_cap_ < need) {
      ops_.free(ms3_X_);
      ms3_X_ = ops_.alloc(need);
      ms3_cap_ = need;
    }
    FactorRef f;
    f.ptr = wptr(r);
    f.rows = ext(r);
    f.ld = V_.glens[r];
    ops_.scan_contract(lay, V_.dtype, d0 * d1, ext(r), 1, &f, 1, r_[r], ms3_X_, V_.dtype, d0 * d1 * r_[r], d0 * d1);
    ms3_root_ = r;
    ms3_left_ = 2;
  }
  const int r = ms3_root_, m0 = (r + 1) % 3, m1 = (r + 2) % 3;  // X[s_m0, s_m1, rank_r]
  const int64_t d0 = ext(m0), d1 = ext(m1);
  const int64_t elems = (int64_t)V_.glens[i] * (i == m0 ? r_[m1] : r_[m0]) * r_[r];
  if (ms3_Ycap_[i] < elems) {
    ops_.free(ms3_Y_[i]);
    ms3_Y_[i] = (double *)ops_.alloc(sizeof(double) * elems);
    ms3_Ycap_[i] = elems;
  }
  if (i == m0) {  // keep the first mode in front, contract the second: [s_m0 | rank_m1, rank_r]
    ops_.ttm_keep(ms3_X_, V_.dtype, d0, d1, r_[r], wptr(m1), V_.glens[m1], r_[m1], ms3_Y_[i]);
    *T = (int64_t)r_[m1] * r_[r];
  } else {        // contract the leading mode, the second one comes to the front: [s_m1 | rank_m0, rank_r]
    if (!ops_.ttm_lead_front(ms3_X_, V_.dtype, d0, d1, r_[r], wptr(m0), V_.glens[m0], r_[m0], ms3_Y_[i])) {
      ms3_ = false;
      ms3_invalidate();
      return nullptr;
    }
    *T = (int64_t)r_[m0] * r_[r];
  }
  // rank indices behind the mode: (other, root). Ascending mode order — what ensure_core() reads — when
  // other < root
  ms3_perm_ = (i == m0 ? m1 : m0) > r;
  ms3_left_--;
  return ms3_Y_[i];
}

// see CpEngine::check_tensor_generation: the tensor handle stays writable while sessions exist
void TuckerEngine::check_tensor_generation() {
  if (!V_.generation || *V_.generation == tensor_gen_) return;
  tensor_gen_ = *V_.generation;
  if (VT_) {
    const int mid = (N_ - 1) / 2;
    int64_t rows = 1, cols = 1;
    for (int m = 0; m <= mid; m++) rows *= ext(m);
    for (int m = mid + 1; m < N_; m++) cols *= ext(m);
    ops_.transpose2d(V_.data, V_.dtype, rows, cols, VT_);
  }
  if (VT2_) ops_.transpose2d(V_.data, V_.dtype, ext(0), ext(1) * ext(2), VT2_);
  ms3_invalidate();
  for (auto &n : nodes_) n.valid = false;
  pp_clear();
}

TuckerEngine::~TuckerEngine() {
  try {
    ops_.sync();
  } catch (...) {
  }
  try {
    ops_.eig_session_free(eig_base_);
  } catch (...) {
  }
  for (auto p : W_) ops_.free(p);
  for (auto p : Wsave_) ops_.free(p);
  for (auto &n : nodes_) ops_.free(n.buf);
  ops_.free(core_);
  ops_.free(core_prev_);
  ops_.free(Yend_);
  ops_.free(G_);
  ops_.free(thin_);
  ops_.free(scal_);
  ops_.free(Yfull_);
  ops_.free(gather_);
  ops_.free(Ytmp_);
  ops_.free(Yacc_);
  ops_.free(chain_[0]);
  ops_.free(chain_[1]);
  ops_.free(VT_);
  ops_.free(VT2_);
  ops_.free(ms3_X_);
  for (auto p : ms3_Y_) ops_.free(p);
  for (auto p : Wprev_) ops_.free(p);
  for (auto p : Winit_) ops_.free(p);
  for (auto p : dW_) ops_.free(p);
  pp_clear();
}

void TuckerEngine::build_tree(int lo, int hi, int parent) {
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
    nodes_.push_back(n);
    idx[c] = (int)nodes_.size() - 1;
  }
  build_tree(lo, mid, idx[0]);
  build_tree(mid + 1, hi, idx[1]);
}

// a node [lo,hi] keeps modes lo..hi at full extent; every other mode is already contracted to rank
int64_t TuckerEngine::node_elems(const Node &n) const {
  int64_t e = 1;
  for (int m = 0; m < N_; m++) e *= (m >= n.lo && m <= n.hi) ? ext(m) : r_[m];
  return e;
}

// First-level node that contracts the left half [0, shi] of the modes, on the second layout
// [shi+1..N-1 | 0..shi]: the contracted modes are the trailing ones there, so they are removed
// last-first by row-contiguous scans; the result [right extents | left ranks] is transposed into
// the node's own order [left ranks | right extents] (a few hundred MB at most).
void TuckerEngine::compute_left_half_on_vt(Node &n) {
  const int nl = n.shi + 1, nr = N_ - nl;  // left (contracted) / right (kept) mode counts
  std::vector<int64_t> dims;              // extents in VT storage order
  for (int m = nl; m < N_; m++) dims.push_back(ext(m));
  for (int m = 0; m < nl; m++) dims.push_back(ext(m));
  const void *cur = VT_;
  int cur_dt = V_.dtype;
  int pp_slot = 0;
  // A LEAF (one mode on the right: order 3) keeps the chain's own order [s_mode | left ranks]: that
  // IS the unfolding with the mode in front which the Gram wants, so neither the transposition into
  // the tree's order nor the one back (2 x 7 us at cfg5) happens; the last scan writes the node.
  const bool front_leaf = (nr == 1) && !dist_;
  int64_t node_elems_total = 1;
  for (int m = 0; m < N_; m++) node_elems_total *= (m >= nl ? ext(m) : r_[m]);
  if (front_leaf && n.cap < node_elems_total) {
    ops_.free(n.buf);
    n.buf = (double *)ops_.alloc(sizeof(double) * node_elems_total);
    n.cap = node_elems_total;
  }
  for (int m = n.shi; m >= 0; m--) {
    const int p = nr + m;  // storage position of mode m
    int64_t L = 1, T = 1;
    for (int q = 0; q < p; q++) L *= dims[q];
    for (int q = p + 1; q < N_; q++) T *= dims[q];
    const bool last = (m == 0);
    const int dst_dt = last ? F64 : V_.dtype;
    const size_t need = dtype_size(dst_dt) * (size_t)(L * r_[m] * T);
    void *dst;
    if (last && front_leaf) {
      dst = n.buf;
    } else {
      if (chain_cap_[pp_slot] < need) {
        ops_.free(chain_[pp_slot]);
        chain_[pp_slot] = ops_.alloc(need);
        chain_cap_[pp_slot] = need;
      }
      dst = chain_[pp_slot];
      pp_slot ^= 1;
    }
    FactorRef f;
    f.ptr = wptr(m);
    f.rows = dims[p];
    f.ld = V_.glens[m];
    // (the step that reads the tensor is a scan; a later one reads a small intermediate and writes
    // fp64: the back end may run it as one batched GEMM, Ops::ttm_keep)
    if (cur != VT_ && dst_dt == F64)
      ops_.ttm_keep(cur, cur_dt, L, dims[p], T, f.ptr, f.ld, r_[m], (double *)dst);
    else
      ops_.scan_contract(cur, cur_dt, L, dims[p], T, &f, 1, r_[m], dst, dst_dt, L * r_[m], L);
    cur = dst;
    cur_dt = dst_dt;
    dims[p] = r_[m];
  }
  n.front = front_leaf;
  if (front_leaf) {
    n.valid = true;
    return;
  }
  int64_t rows = 1, cols = 1;
  for (int q = 0; q < nr; q++) rows *= dims[q];
  for (int q = nr; q < N_; q++) cols *= dims[q];
  if (n.cap < rows * cols) {
    ops_.free(n.buf);
    n.buf = (double *)ops_.alloc(sizeof(double) * rows * cols);
    n.cap = rows * cols;
  }
  ops_.transpose2d(cur, F64, rows, cols, n.buf);
  n.valid = true;
}

void TuckerEngine::compute_node(int idx) {
  check_tensor_generation();
  Node &n = nodes_[idx];
  if (n.valid) return;
  // (the sharded leaf of mode 0 keeps its own blocked layout: not this route)
  if (n.parent < 0 && n.slo == 0 && n.shi < N_ - 1 && VT_ && !(dist_ && n.lo == 0 && n.hi == 0)) {
    compute_left_half_on_vt(n);
    return;
  }
  n.front = false;
  // The leaf of mode 1 under a parent that keeps modes 0 and 1: one product with the LEADING mode,
  // written with mode 1 in front — the unfolding the Gram wants (no leading-mode scan + pack +
  // transposition: 25 -> 14 us at cfg5) — when the back end has the kernel.
  if (!dist_ && n.parent >= 0 && n.lo == 1 && n.hi == 1 && n.slo == 0 && n.shi == 0) {
    compute_node(n.parent);
    const Node &p = nodes_[n.parent];
    if (!p.front && p.lo == 0 && p.hi == 1) {
      int64_t T = 1;
      for (int m = 2; m < N_; m++) T *= r_[m];
      const int64_t elems = (int64_t)ext(1) * r_[0] * T;
      if (n.cap < elems) {
        ops_.free(n.buf);
        n.buf = (double *)ops_.alloc(sizeof(double) * elems);
        n.cap = elems;
      }
      if (ops_.ttm_lead_front(p.buf, F64, ext(0), ext(1), T, wptr(0), V_.glens[0], r_[0], n.buf)) {
        n.front = true;
        n.valid = true;
        return;
      }
    }
  }
  std::vector<int64_t> dims(N_);
  const void *src;
  int dt;
  if (n.parent < 0) {
    for (int m = 0; m < N_; m++) dims[m] = ext(m);
    src = V_.data;
    dt = V_.dtype;
  } else {
    compute_node(n.parent);
    const Node &p = nodes_[n.parent];
    for (int m = 0; m < N_; m++) dims[m] = (m >= p.lo && m <= p.hi) ? ext(m) : r_[m];
    src = p.buf;
    dt = F64;
  }
  // contract the sibling's modes one at a time (als_Tucker.cxx:216-227). The intermediates of the
  // chain live in two grow-only scratch buffers of the session (a per-call hipMalloc/hipFree of
  // these multi-GB tensors stalled whole seconds at order 6) and keep the tensor's own precision
  // (like the CP intermediate: fp32 tensor -> fp32 intermediates, fp64 accumulation inside the
  // scan, fp64 node result).
  const void *cur = src;
  int cur_dt = dt;
  int pp_slot = 0;
  // the sharded leaf of mode 0 is written with leading dimension blk (= rows per rank) so that it
  // is directly one block of the all-gather buffer (rows beyond the local extent stay zero)
  const bool leaf0_blocked = (dist_ && n.lo == 0 && n.hi == 0);
  const int64_t blk = block_rows(V_.glens[0], P_);
  // The mode products commute; the reference removes the sibling's modes in ascending order
  // (als_Tucker.cxx:216-227). The FIRST product is the one that reads the tensor: when its columns
  // are 128-B aligned only with the sibling's LAST mode in front (s = 50: 50^3 * 4 B is not a
  // multiple of 128, 50^5 * 4 B is), the chain runs in descending order — 0.80 instead of 0.66
  // of peak for that scan (profiles/r02s_stride_bench.txt). PPALS_TUCKER_CHAIN=asc|desc forces one.
  bool descending = false;
  if (n.parent < 0 && n.shi > n.slo && !leaf0_blocked) {
    const int64_t line = 128 / (int64_t)dtype_size(dt);
    int64_t la = 1, ld = 1;
    for (int q = 0; q < n.slo; q++) la *= dims[q];
    for (int q = 0; q < n.shi; q++) ld *= dims[q];
    descending = (ld % line == 0) && (la % line != 0);
    if (const char *e = std::getenv("PPALS_TUCKER_CHAIN")) descending = std::string(e) == "desc";
  }
  const int nsib = n.shi - n.slo + 1;
  for (int step = 0; step < nsib; step++) {
    const int m = descending ? n.shi - step : n.slo + step;
    int64_t L = 1, T = 1;
    for (int q = 0; q < m; q++) L *= dims[q];
    for (int q = m + 1; q < N_; q++) T *= dims[q];
    const bool last = (step == nsib - 1);
    const int64_t Lout = (last && leaf0_blocked) ? blk : L;
    const int64_t out_elems = Lout * r_[m] * T;
    void *dst;
    int dst_dt;
    if (last) {
      if (n.cap < out_elems) {
        ops_.free(n.buf);
        n.buf = (double *)ops_.alloc(sizeof(double) * out_elems);
        n.cap = out_elems;
      }
      dst = n.buf;
      dst_dt = F64;
      if (leaf0_blocked) ops_.zero(dst, sizeof(double) * out_elems);
    } else {
      dst_dt = V_.dtype;
      const size_t need = dtype_size(dst_dt) * (size_t)out_elems;
      if (chain_cap_[pp_slot] < need) {
        ops_.free(chain_[pp_slot]);
        chain_[pp_slot] = ops_.alloc(need);
        chain_cap_[pp_slot] = need;
      }
      dst = chain_[pp_slot];
      pp_slot ^= 1;
    }
    FactorRef f;
    f.ptr = wptr(m);
    f.rows = dims[m];
    f.ld = V_.glens[m];
    // out[l + Lout*(k + r*t)]: the mode product that keeps the mode in place (als_Tucker.cxx:224)
    if (cur != V_.data && dst_dt == F64 && Lout == L)
      ops_.ttm_keep(cur, cur_dt, L, dims[m], T, f.ptr, f.ld, r_[m], (double *)dst);
    else
      ops_.scan_contract(cur, cur_dt, L, dims[m], T, &f, 1, r_[m], dst, dst_dt, Lout * r_[m], Lout);
    cur = dst;
    cur_dt = dst_dt;
    dims[m] = r_[m];
  }
  n.valid = true;
}

// Make a leaf tensor complete on every rank. i == 0: the rows live on their owners -> all-gather
// of the [blk x rest] blocks, re-assembled to [s0 x rest]. i != 0: partial sums -> all-reduce.
double *TuckerEngine::complete_leaf(int i, double *Yloc, int64_t elems_local) {
  if (!dist_) return Yloc;
  if (i != 0) {
    comm_.allreduce_sum(Yloc, elems_local);
    return Yloc;
  }
  const int64_t s0 = V_.glens[0], blk = block_rows(s0, P_);
  int64_t rest = 1;
  for (int m = 1; m < N_; m++) rest *= r_[m];
  if (gather_cap_ < blk * rest * P_) {
    ops_.free(gather_);
    gather_ = (double *)ops_.alloc(sizeof(double) * blk * rest * P_);
    gather_cap_ = blk * rest * P_;
  }
  if (yfull_cap_ < s0 * rest) {
    ops_.free(Yfull_);
    Yfull_ = (double *)ops_.alloc(sizeof(double) * s0 * rest);
    yfull_cap_ = s0 * rest;
  }
  ops_.d2d(gather_ + (size_t)rank_ * blk * rest, Yloc, sizeof(double) * blk * rest);
  comm_.allgather(gather_ + (size_t)rank_ * blk * rest, gather_, blk * rest);
  ops_.unpack_blocks(gather_, s0, s0, (int)rest, blk, P_, Yfull_);
  return Yfull_;
}

// TTMc (als_Tucker.cxx:76-110): chain of mode products, skipping `skip`
double *TuckerEngine::ttmc_chain(int skip, int64_t *elems) {
  std::vector<int64_t> dims(N_);
  for (int m = 0; m < N_; m++) dims[m] = ext(m);
  const void *cur = V_.data;
  int cur_dt = V_.dtype;
  double *prev = nullptr;
  int pp_slot = 0;
  for (int m = 0; m < N_; m++) {
    if (m == skip) continue;
    int64_t L = 1, T = 1;
    for (int q = 0; q < m; q++) L *= dims[q];
    for (int q = m + 1; q < N_; q++) T *= dims[q];
    const size_t need = sizeof(double) * (size_t)(L * r_[m] * T);
    if (chain_cap_[pp_slot] < need) {
      ops_.free(chain_[pp_slot]);
      chain_[pp_slot] = ops_.alloc(need);
      chain_cap_[pp_slot] = need;
    }
    double *dst = (double *)chain_[pp_slot];
    pp_slot ^= 1;
    ops_.ttm_keep(cur, cur_dt, L, dims[m], T, wptr(m), V_.glens[m], r_[m], dst);
    prev = dst;
    cur = dst;
    cur_dt = F64;
    dims[m] = r_[m];
  }
  int64_t e = 1;
  for (int m = 0; m < N_; m++) e *= dims[m];
  *elems = e;
  return prev;  // lives in the session's chain scratch until the next chain runs
}

int64_t TuckerEngine::ttmc(int skip, double *Yhost) {
  settle_all();
  int64_t e;
  double *Y = ttmc_chain(skip, &e);
  // sharded: skip == 0 returns the local rows, every other result is summed over the ranks
  if (dist_ && skip != 0) comm_.allreduce_sum(Y, e);
  if (Yhost) ops_.d2h(Yhost, Y, sizeof(double) * e);
  return e;
}

void TuckerEngine::ensure_core() {
  if (!core_owed_) return;
  if (!yend_src_) {
    compute_core_full();
    return;
  }
  core_owed_ = false;
  const int64_t Lc = ncore_ / r_[N_ - 1], sN = V_.glens[N_ - 1];
  const int rN = r_[N_ - 1];
  if (yend_T_ == 1) {  // the tree's order [ranks before | s_{N-1}]
    ops_.ttm_keep(yend_src_, F64, Lc, sN, 1, W_[N_ - 1], sN, rN, core_);
  } else {
    // the last leaf kept the mode in front, [s_{N-1} | ranks]: core^T first, then its order
    double *tmp = (double *)ops_.alloc(sizeof(double) * ncore_);
    ops_.ttm_keep(yend_src_, F64, 1, sN, Lc, W_[N_ - 1], sN, rN, tmp);  // [r_{N-1}, ranks before]
    ops_.transpose2d(tmp, F64, rN, Lc, core_);
    ops_.sync();
    ops_.free(tmp);
  }
}

void TuckerEngine::compute_core_full() {
  core_owed_ = false;
  int64_t e;
  double *Y = ttmc_chain(-1, &e);
  if (dist_) comm_.allreduce_sum(Y, ncore_);
  ops_.d2d(core_, Y, sizeof(double) * ncore_);
}

// Eigenvectors on demand. A plain HOOI sweep lets the back end return ANY orthonormal basis of a
// mode's subspace (Ops::eig_lazy): the contractions, ||core|| and the convergence measure depend on
// W_i W_i^T only. Whoever needs the eigenvectors one by one, sorted — the caller reading the
// factors, the PP phases that difference them — asks here: W_i <- W_i Y_i, core <- core x_i Y_i.
void TuckerEngine::finalize_rotations() {
  settle_all();  // (no factor is read, rotated or published with an unchecked step behind it)
  ensure_core();
  for (int i = 0; i < N_; i++) {
    const double *Y = ops_.eig_pending_rotation(eig_base_ + i);
    if (!Y) continue;
    const int r = r_[i];
    ops_.rows_times_small(W_[i], V_.glens[i], r, Y, r, nullptr, W_[i]);
    int64_t L = 1, T = 1;
    for (int q = 0; q < i; q++) L *= r_[q];
    for (int q = i + 1; q < N_; q++) T *= r_[q];
    double *tmp = (double *)ops_.alloc(sizeof(double) * ncore_);
    ops_.ttm_keep(core_, F64, L, r, T, Y, r, r, tmp);
    ops_.d2d(core_, tmp, sizeof(double) * ncore_);
    ops_.sync();
    ops_.free(tmp);
    ops_.eig_rotation_done(eig_base_ + i);
    ms3_invalidate();  // (the multi-sweep intermediate may carry this mode's rank index in the old basis)
  }
}
void TuckerEngine::drop_rotations() {
  settle_all();
  for (int i = 0; i < N_; i++) ops_.eig_rotation_done(eig_base_ + i);
}

void TuckerEngine::set_factors(const double *Wflat) {
  settle_all();
  // (the core the last sweep owes belongs to the factors that are about to be replaced: the core object
  // travels on through a caller's repetitions, pp_bench.cxx:321-345)
  ensure_core();
  drop_rotations();  // (they belonged to the factors being replaced)
  ms3_invalidate();
  const double *w = Wflat;
  for (int i = 0; i < N_; i++) {
    size_t n = (size_t)V_.glens[i] * r_[i];
    ops_.h2d(W_[i], w, n * sizeof(double));
    w += n;
  }
}
// core == nullptr: core = V x_i W_i^T from the current factors (TTMc(core, V, W, -1)); the value
// becomes the `core` argument (and the initial core_prev) of the next alsTucker_DT / _PP call
void TuckerEngine::set_core(const double *core) {
  finalize_rotations();
  if (core) {
    core_owed_ = false;
    ops_.h2d(core_, core, sizeof(double) * ncore_);
  } else
    compute_core_full();
}
void TuckerEngine::get_factors(double *Wflat, double *core) {
  finalize_rotations();
  double *w = Wflat;
  for (int i = 0; i < N_; i++) {
    size_t n = (size_t)V_.glens[i] * r_[i];
    if (w) {
      ops_.d2h(w, W_[i], n * sizeof(double));
      w += n;
    }
  }
  if (core) ops_.d2h(core, core_, sizeof(double) * ncore_);
}

// hosvd (als_Tucker.cxx:12-70): W_i = leading eigenvectors of the Gram of the mode-i unfolding of
// V (K13), then core = V x_i W_i^T
void TuckerEngine::hosvd() {
  settle_all();
  ms3_invalidate();
  for (int i = 0; i < N_; i++) {
    int64_t L = 1, T = 1;
    for (int q = 0; q < i; q++) L *= ext(q);
    for (int q = i + 1; q < N_; q++) T *= ext(q);
    if (dist_ && i == 0) {
      // the Gram of the sharded mode needs every pair of rows: gather the shards once (needs room
      // for two extra copies of the tensor; HOSVD is a one-off initialisation)
      const int64_t s0 = V_.glens[0], blk = block_rows(s0, P_);
      const size_t esz = dtype_size(V_.dtype);
      const int64_t chunk_bytes = ((blk * T * (int64_t)esz + 7) / 8) * 8;
      char *stage = (char *)ops_.alloc((size_t)chunk_bytes * P_);
      void *full = ops_.alloc((size_t)s0 * T * esz);
      ops_.zero(stage, (size_t)chunk_bytes * P_);
      ops_.d2d(stage + (size_t)rank_ * chunk_bytes, V_.data, (size_t)V_.nloc * esz);
      comm_.allgather((const double *)(stage + (size_t)rank_ * chunk_bytes), (double *)stage,
                      chunk_bytes / 8);
      ops_.unpack_shards(stage, V_.dtype, s0, T, blk, P_, chunk_bytes, full);
      ops_.unfold_gram(full, V_.dtype, 1, s0, T, G_);
      ops_.sync();
      ops_.free(stage);
      ops_.free(full);
    } else {
      ops_.unfold_gram(V_.data, V_.dtype, L, V_.glens[i], T, G_);
      if (dist_) comm_.allreduce_sum(G_, V_.glens[i] * V_.glens[i]);
    }
    // (slots 8.. : a cold start of their own — the Gram of the full unfolding has little to do
    // with the one the first HOOI sweep will see in slot i)
    ops_.top_eigvecs_warm(G_, V_.glens[i], r_[i], W_[i], eig_base_ + MAX_ORDER + i);
  }
  compute_core_full();
  ops_.d2d(core_prev_, core_, sizeof(double) * ncore_);
  ops_.sync();
}

// K12 (als_Tucker.cxx:399-406: Gram of the unfolding by unroll_tensor_contraction, then
// `MTM.svd(U,S,VT,rank)`): the factor is the r_i leading left singular vectors of the s_i x (L*T)
// unfolding Y_(i). The reference always takes them from the s_i x s_i Gram Y_(i) Y_(i)^T. When the
// unfolding is TALL (s_i > L*T: the long mode of an image stack, 7200 x 300 for the coil-100
// shape) the same vectors come from the small side: eigenvectors v_k of the (L*T)^2 Gram
// Y_(i)^T Y_(i), then u_k = Y_(i) v_k / |Y_(i) v_k| — same squared conditioning, O(s c^2) instead
// of O(s^3) work. The columns are re-orthonormalised (they are orthogonal only to
// eps * lambda_1 / lambda_k); a numerically rank-deficient unfolding takes the s_i x s_i route.
void TuckerEngine::factor_update(int i, const double *Y, int64_t L, int64_t T) {
  const int64_t s = V_.glens[i], c = L * T;
  if (thin_enabled_ && c < s && r_[i] <= c) {
    const int64_t need = s * c + c * r_[i];
    if (thin_cap_ < need) {
      ops_.free(thin_);
      thin_ = (double *)ops_.alloc(sizeof(double) * need);
      thin_cap_ = need;
    }
    const double *Ym = Y;  // [s, c] with the mode fastest
    if (L > 1) {
      ops_.transpose_batched(Y, F64, L, s, T, thin_);
      Ym = thin_;
    }
    double *Vr = thin_ + s * c;
    ops_.unfold_gram(Ym, F64, s, c, 1, G_);  // c x c (c < s: fits G_)
    // (a slot of its own: the c x c problem and the s x s fallback below are different sequences)
    ops_.top_eigvecs_warm(G_, c, r_[i], Vr, eig_base_ + 2 * MAX_ORDER + i);
    ops_.ttm_keep(Ym, F64, s, c, 1, Vr, c, r_[i], W_[i]);
    if (ops_.orthonormalize(W_[i], s, r_[i])) return;
  }
  // (a slot that defers its checks keeps the Gram itself: they are finished on a second stream)
  double *G = ops_.eig_gram(eig_base_ + i, s);
  if (!G) G = G_;
  ops_.unfold_gram(Y, F64, L, s, T, G);
  ops_.top_eigvecs_warm(G, s, r_[i], W_[i], eig_base_ + i);
}

void TuckerEngine::sweep_dt() { sweep_body(nullptr); }

// one HOOI sweep; align_ref != nullptr: column signs aligned with that factor set after every
// eigen-step (alsTucker_DT_sub, als_Tucker.cxx:632-643)
void TuckerEngine::sweep_body(const std::vector<double *> *align_ref) {
  const bool may_defer = defer_enabled_ && align_ref == nullptr && !dist_;
  if (!may_defer) settle_all();
  for (auto &n : nodes_) n.valid = false;  // ttmc_map.clear(), als_Tucker.cxx:340
  for (int i = 0; i < N_; i++) {
    if (may_defer) settle_mode(i);
    mode_step(i, align_ref, may_defer);
  }
}

void TuckerEngine::mode_step(int i, const std::vector<double *> *align_ref, bool may_defer) {
  int64_t L = 1, T = 1;
  double *Y = ms3_ ? ms3_leaf(i, &T) : nullptr;
  const bool from_ms3 = Y != nullptr;
  if (!from_ms3) {
    compute_node(leaf_[i]);
    const Node &lf = nodes_[leaf_[i]];
    for (int q = 0; q < i; q++) L *= r_[q];
    T = 1;
    for (int q = i + 1; q < N_; q++) T *= r_[q];
    if (lf.front) {  // [s_i | ranks of the other modes] (compute_left_half_on_vt)
      T = L * T;
      L = 1;
    }
    Y = complete_leaf(i, lf.buf, L * V_.glens[i] * T);
  }
  if (i == N_ - 1) {  // als_Tucker.cxx:395
    yend_T_ = T;
    // (one GPU: the leaf's own buffer stays as it is until this mode is stepped again — no copy)
    if (dist_) {
      ops_.d2d(Yend_, Y, sizeof(double) * yend_elems_);
      yend_src_ = Yend_;
    } else {
      // (a multi-sweep leaf with its rank indices in descending mode order: the core is recomputed from
      // the tensor when somebody asks for it)
      yend_src_ = (from_ms3 && ms3_perm_) ? nullptr : Y;
    }
  }
  ops_.eig_lazy(eig_base_ + i, align_ref == nullptr && !dist_);
  ops_.eig_defer(eig_base_ + i, may_defer);
  if (may_defer) {
    // the step writes the new factor into the spare buffer; the one it replaces stays intact
    if (Wsave_.empty()) Wsave_.assign(N_, nullptr);
    if (!Wsave_[i]) Wsave_[i] = (double *)ops_.alloc(sizeof(double) * V_.glens[i] * r_[i]);
    std::swap(W_[i], Wsave_[i]);
  }
  factor_update(i, Y, L, T);  // K12
  ops_.eig_defer(eig_base_ + i, false);
  if (align_ref) ops_.sign_align(W_[i], (*align_ref)[i], V_.glens[i], r_[i]);
  // core = Y_end x_{N-1} W[N-1]  (als_Tucker.cxx:408): owed until somebody reads it (ensure_core) —
  // the print blocks recompute the core from the tensor anyway (als_Tucker.cxx:290), so inside a run
  // of sweeps this product (a launch-bound 30 us at cfg5) served nobody
  if (i == N_ - 1) core_owed_ = true;
  if (may_defer && (!defer_log_.empty() || ops_.eig_deferred(eig_base_ + i))) defer_log_.push_back(i);
}

void TuckerEngine::settle_mode(int i) {
  if (defer_log_.empty()) return;
  const int v = ops_.eig_verify(eig_base_ + i);
  if (v < 0) return;  // nothing of this mode is waiting
  if (defer_log_.front() != i) throw std::logic_error("ppals: deferred eigen-steps out of order");
  if (v == 1) {
    rollback_and_redo();
    return;
  }
  defer_log_.erase(defer_log_.begin());
  while (!defer_log_.empty() && !ops_.eig_deferred(eig_base_ + defer_log_.front()))
    defer_log_.erase(defer_log_.begin());  // (steps that were checked on the spot)
}

void TuckerEngine::settle_all() {
  while (!defer_log_.empty()) {
    const int i = defer_log_.front();
    if (ops_.eig_verify(eig_base_ + i) == 1) {
      rollback_and_redo();
      return;
    }
    defer_log_.erase(defer_log_.begin());
  }
}

// The oldest step of the log was not accepted: its factor and everything computed from it since
// is void. Every mode stepped since then gets back the factor that step replaced (a basis of the
// right subspace is all the sweep needs of it), unchecked steps among them are dropped, and the
// same steps are made again in the same order, each checked before the next one starts.
void TuckerEngine::rollback_and_redo() {
  const std::vector<int> steps = defer_log_;
  defer_log_.clear();
  defer_rollbacks_++;
  ms3_invalidate();
  ops_.sync();
  for (size_t k = 0; k < steps.size(); k++) {
    const int j = steps[k];
    if (k > 0) ops_.eig_verify(eig_base_ + j, true);
    std::swap(W_[j], Wsave_[j]);
    ops_.eig_rotation_done(eig_base_ + j);  // (the rotation owed belonged to the discarded factor)
  }
  for (int j : steps) {
    for (auto &n : nodes_) n.valid = false;
    ms3_invalidate();
    mode_step(j, nullptr, false);
  }
  ms3_invalidate();
  // (whatever the sweep in progress needs of the tree is rebuilt from the factors as they are now:
  // a leaf left valid here would be taken for this sweep's by the step that comes to it)
  for (auto &n : nodes_) n.valid = false;
}

// rank-agreed stop decision (see CpEngine::agree): the time limit reads a rank-local clock
bool TuckerEngine::agree(bool local) {
  if (!dist_) return local;
  double x = local ? 1.0 : 0.0, y = 0;
  ops_.h2d(scal_ + 3, &x, sizeof(double));
  comm_.allreduce_sum(scal_ + 3, 1);
  ops_.d2h(&y, scal_ + 3, sizeof(double));
  return y > 0.0;
}

double TuckerEngine::core_norm() {
  ensure_core();
  ops_.sumsq(core_, ncore_, scal_);
  ops_.sumsq(core_prev_, ncore_, scal_ + 1);
  double h[2];
  ops_.d2h(h, scal_, sizeof(double) * 2);
  return std::fabs(std::sqrt(h[0]) - std::sqrt(h[1]));
}

// ||core x_i W_i - V||_F (als_Tucker.cxx:296-310) without materialising the model tensor: expand
// every mode but the last into Q (prod s_0..s_{N-2} x r_{N-1}), then V^ = Q W_{N-1}^T is streamed.
double TuckerEngine::residual() {
  ensure_core();
  std::vector<int64_t> dims(N_);
  for (int m = 0; m < N_; m++) dims[m] = r_[m];
  const double *cur = core_;
  double *prev = nullptr;
  for (int m = 0; m < N_ - 1; m++) {
    int64_t L = 1, T = 1;
    for (int q = 0; q < m; q++) L *= dims[q];
    for (int q = m + 1; q < N_; q++) T *= dims[q];
    // expansion: out[l, a, t] = sum_k cur[l,k,t] * W_m[a,k]  ==  ttm_keep with W^T (J=r, K=s)
    // W^T as a column-major r x s matrix is W (s x r, ld = s) read with swapped strides; build it.
    const int64_t sg = V_.glens[m], s = ext(m), r0 = (m == 0 ? V_.row0 : 0);
    const int rk = r_[m];
    std::vector<double> Wh((size_t)sg * rk), WT((size_t)s * rk);
    ops_.d2h(Wh.data(), W_[m], sizeof(double) * sg * rk);
    for (int64_t a = 0; a < s; a++)  // local rows of the sharded mode only
      for (int k = 0; k < rk; k++) WT[k + (size_t)rk * a] = Wh[(a + r0) + sg * k];
    double *WTd = (double *)ops_.alloc(sizeof(double) * s * rk);
    ops_.h2d(WTd, WT.data(), sizeof(double) * s * rk);
    double *dst = (double *)ops_.alloc(sizeof(double) * L * s * T);
    ops_.ttm_keep(cur, F64, L, rk, T, WTd, rk, (int)s, dst);
    ops_.free(WTd);
    if (prev) ops_.free(prev);
    prev = dst;
    cur = dst;
    dims[m] = s;
  }
  int64_t M = 1;
  for (int m = 0; m < N_ - 1; m++) M *= ext(m);
  ops_.residual_sq(V_.data, V_.dtype, M, V_.glens[N_ - 1], cur, W_[N_ - 1], r_[N_ - 1], scal_ + 2);
  if (dist_) comm_.allreduce_sum(scal_ + 2, 1);
  double h = 0;
  ops_.d2h(&h, scal_ + 2, sizeof(double));
  if (prev) ops_.free(prev);
  return std::sqrt(h);
}

int TuckerEngine::run_dt(const CpOpts &o, int *iters) {
  std::ofstream csv;
  std::ofstream *pcsv = nullptr;
  if (rank_ == 0 && !o.csv_path.empty()) {
    csv.open(o.csv_path, o.csv_append ? std::ios::app : std::ios::out);
    pcsv = &csv;
    if (!o.bench) csv << "[dim],[iter],[diffnorm],[tol],[pp_update],[diffV],[dtime]\n";
  }
  const bool talk = o.verbose && rank_ == 0;
  double st_time = now();
  ensure_core();
  ops_.d2d(core_prev_, core_, sizeof(double) * ncore_);  // Tensor<> core_prev(core)
  double diffnorm = 1000, diffnorm_V = 1000;
  int iter;
  for (iter = 0; iter <= o.maxiter; iter++) {
    if ((iter % o.resprint == 0 && iter != 0) || iter == 1 || iter == o.maxiter) {
      settle_all();
      ops_.sync();
      const double st_time1 = now();
      compute_core_full();  // TTMc(core, V, W, -1)
      diffnorm = core_norm();
      diffnorm_V = residual();
      st_time += now() - st_time1;
      const double dtime = now() - st_time;
      if (!o.bench) {
        if (talk) {
          std::cout.precision(13);
          std::cout << "  [dim]=  " << V_.glens[0] << "  [iter]=  " << iter << "  [diffnorm]  "
                    << diffnorm << "  [tol]  " << o.tol << "  [pp_update]  " << 0 << "  [diffV]  "
                    << diffnorm_V << "  [dtime]  " << dtime << "\n";
        }
        if (pcsv) {
          (*pcsv) << V_.glens[0] << "," << iter << "," << diffnorm << "," << o.tol << "," << 0
                  << "," << diffnorm_V << "," << dtime << "\n";
          if (iter % 100 == 0 && iter != 0) (*pcsv) << std::endl;
        }
      } else {
        if (talk) std::cout << "  [dimension tree step time]  " << dtime << "\n";
        if (pcsv) (*pcsv) << "[DTtime]" << "," << dtime << "\n";
      }
      if (agree(diffnorm < o.tol || now() - st_time > o.timelimit)) break;
      ops_.d2d(core_prev_, core_, sizeof(double) * ncore_);
    }
    sweep_dt();
    if (iter % 10 == 0 && talk) printf(".");
  }
  settle_all();
  ops_.sync();
  if (talk) {
    printf("\nIter = %d Final Diff norm %E \n", iter, diffnorm);
    printf("tf took %lf seconds\n", now() - st_time);
  }
  if (pcsv) csv.close();
  if (iters) *iters = iter;
  return iter == o.maxiter + 1 ? 0 : 1;
}

// ============================================================================ Tucker PP
static std::string tk_all_but(int N, int i, int j = -1) {
  std::string s;
  for (int m = 0; m < N; m++)
    if (m != i && m != j) s.push_back((char)('a' + m));
  return s;
}

// Build_ttmc_map (als_Tucker.cxx:426-466): key = contracted modes (ascending), recursion drops the
// last one; every contraction keeps the tensor order. Level 1 scans V (K11), deeper levels
// contract the cached fp64 intermediate.
const TuckerEngine::PPOp &TuckerEngine::pp_get(const std::string &args) {
  check_tensor_generation();
  auto it = pp_.find(args);
  if (it != pp_.end()) return it->second;
  const int mode = args.back() - 'a';
  PPOp op;
  const void *src;
  int dt;
  if (args.size() == 1) {
    op.dims.resize(N_);
    for (int m = 0; m < N_; m++) op.dims[m] = ext(m);
    src = V_.data;
    dt = V_.dtype;
  } else {
    const PPOp &par = pp_get(args.substr(0, args.size() - 1));
    op.dims = par.dims;
    src = par.buf;
    dt = F64;
  }
  int64_t L = 1, T = 1;
  for (int q = 0; q < mode; q++) L *= op.dims[q];
  for (int q = mode + 1; q < N_; q++) T *= op.dims[q];
  const int64_t J = op.dims[mode];
  op.dims[mode] = r_[mode];
  op.elems = L * r_[mode] * T;
  op.buf = (double *)ops_.alloc(sizeof(double) * op.elems);
  ops_.ttm_keep(src, dt, L, J, T, wptr(mode), V_.glens[mode], r_[mode], op.buf);
  pp_[args] = op;
  return pp_[args];
}
void TuckerEngine::pp_clear() {
  for (auto &kv : pp_) ops_.free(kv.second.buf);
  pp_.clear();
}

// one approximate sweep (als_Tucker.cxx:824-891): Y_i = Y_i^0 + sum_{j != i} T_ij x_j dW_j
void TuckerEngine::sweep_pp() {
  ms3_invalidate();  // (PP moves the factors without touching the multi-sweep intermediate)
  for (int i = 0; i < N_; i++) {
    const PPOp &Y0 = pp_get(tk_all_but(N_, i));
    if (yacc_cap_ < Y0.elems) {
      ops_.free(Yacc_);
      Yacc_ = (double *)ops_.alloc(sizeof(double) * Y0.elems);
      yacc_cap_ = Y0.elems;
    }
    if (ytmp_cap_ < Y0.elems) {
      ops_.free(Ytmp_);
      Ytmp_ = (double *)ops_.alloc(sizeof(double) * Y0.elems);
      ytmp_cap_ = Y0.elems;
    }
    ops_.d2d(Yacc_, Y0.buf, sizeof(double) * Y0.elems);
    for (int ii = 0; ii < N_; ii++) {
      if (ii == i) continue;
      const PPOp &Tp = pp_get(tk_all_but(N_, std::min(i, ii), std::max(i, ii)));
      int64_t L = 1, T = 1;
      for (int q = 0; q < ii; q++) L *= Tp.dims[q];
      for (int q = ii + 1; q < N_; q++) T *= Tp.dims[q];
      // sharded mode: the local rows of dW_0 (the operator keeps the local extent of mode 0)
      const double *dwp = dW_[ii] + (ii == 0 ? V_.row0 : 0);
      ops_.ttm_keep(Tp.buf, F64, L, ext(ii), T, dwp, V_.glens[ii], r_[ii], Ytmp_);
      ops_.add_inplace(Yacc_, Ytmp_, Y0.elems);
    }
    int64_t L = 1, T = 1;
    for (int q = 0; q < i; q++) L *= r_[q];
    for (int q = i + 1; q < N_; q++) T *= r_[q];
    double *Y = Yacc_;
    if (dist_) {
      // same completion as the exact sweep: partial sums are all-reduced, the rows of the sharded
      // mode gathered (from a copy padded to the uniform block height)
      if (i == 0) {
        const int64_t blk = block_rows(V_.glens[0], P_);
        if (ytmp_cap_ < blk * T) {
          ops_.free(Ytmp_);
          Ytmp_ = (double *)ops_.alloc(sizeof(double) * blk * T);
          ytmp_cap_ = blk * T;
        }
        ops_.pack_blocks(Yacc_, ext(0), ext(0), (int)T, blk, 1, Ytmp_);
        Y = complete_leaf(0, Ytmp_, blk * T);
      } else {
        Y = complete_leaf(i, Yacc_, Y0.elems);
      }
    }
    if (i == N_ - 1) ops_.d2d(Yend_, Y, sizeof(double) * yend_elems_);
    // (the PP phase differences the eigenvectors one by one: never "any basis of the subspace",
    // whatever the slot's last exact sweep allowed — pp_bench enters here without a dt_sub)
    ops_.eig_lazy(eig_base_ + i, false);
    factor_update(i, Y, L, T);  // K12
    ops_.sign_align(W_[i], Winit_[i], V_.glens[i], r_[i]);  // als_Tucker.cxx:874-885
    double *A[1] = {W_[i]}, *B[1] = {Winit_[i]}, *D[1] = {dW_[i]};
    int64_t n[1] = {V_.glens[i] * r_[i]};
    ops_.diff_norms(A, B, n, 1, 1, D, 0, scal_ + 4);  // dW = W - W_init (als_Tucker.cxx:887)
  }
  int64_t L = ncore_ / r_[N_ - 1];
  core_owed_ = false;
  ops_.ttm_keep(Yend_, F64, L, V_.glens[N_ - 1], 1, W_[N_ - 1], V_.glens[N_ - 1], r_[N_ - 1],
                core_);
}

bool TuckerEngine::print_block(const CpOpts &o, int iter, int pp_flag, double &diffnorm,
                               double &diffV, std::ofstream *csv, double &st_time,
                               bool stop_at_maxiter) {
  settle_all();
  ops_.sync();
  const double st_time1 = now();
  compute_core_full();
  diffnorm = core_norm();
  diffV = residual();
  st_time += now() - st_time1;
  const double dtime = now() - st_time;
  if (rank_ == 0) {
    if (o.verbose) {
      std::cout.precision(13);
      std::cout << "  [dim]=  " << V_.glens[0] << "  [iter]=  " << iter << "  [diffnorm]  "
                << diffnorm << "  [tol]  " << o.tol << "  [pp_update]  " << pp_flag
                << "  [diffV]  " << diffV << "  [dtime]  " << dtime << "\n";
    }
    if (csv) {
      (*csv) << V_.glens[0] << "," << iter << "," << diffnorm << "," << o.tol << "," << pp_flag
             << "," << diffV << "," << dtime << "\n";
      if (iter % 100 == 0 && iter != 0) (*csv) << std::endl;
    }
  }
  if (agree(diffnorm < o.tol || now() - st_time > o.timelimit ||
            (stop_at_maxiter && iter == o.maxiter)))
    return true;
  ops_.d2d(core_prev_, core_, sizeof(double) * ncore_);
  return false;
}

void TuckerEngine::read_norms(bool dt_phase, std::vector<double> &nd, std::vector<double> &nw) {
  int64_t n[MAX_ORDER];
  for (int i = 0; i < N_; i++) n[i] = V_.glens[i] * r_[i];
  if (dt_phase)
    ops_.diff_norms(W_.data(), Wprev_.data(), n, N_, 1, dW_.data(), 1, scal_ + 8);
  else
    ops_.diff_norms(W_.data(), nullptr, n, N_, 0, dW_.data(), 0, scal_ + 8);
  double h[2 * MAX_ORDER];
  ops_.d2h(h, scal_ + 8, sizeof(double) * 2 * N_);
  nd.resize(N_);
  nw.resize(N_);
  for (int i = 0; i < N_; i++) {
    nd[i] = std::sqrt(h[2 * i]);
    nw[i] = std::sqrt(h[2 * i + 1]);
  }
}

// alsTucker_DT_sub (als_Tucker.cxx:476-669)
void TuckerEngine::dt_sub(const CpOpts &o, double tol_init, double &diffnorm, int &iter,
                          std::ofstream *csv, double &st_time) {
  double diffV = 1000;
  for (int i = 0; i < N_; i++) ops_.zero(Wprev_[i], sizeof(double) * V_.glens[i] * r_[i]);
  std::vector<double> nd, nw;
  for (; iter <= o.maxiter; iter++) {
    if ((iter % o.resprint == 0 && iter != 0) || iter == 1 || iter == o.maxiter) {
      if (print_block(o, iter, 0, diffnorm, diffV, csv, st_time, false)) break;
    }
    sweep_body(&Wprev_);
    read_norms(true, nd, nw);
    int num_dw_break = 0;
    for (int i = 0; i < N_; i++)
      if (std::fabs(nd[i] / nw[i]) < tol_init) num_dw_break++;
    if (num_dw_break == N_) return;
    if (iter % 10 == 0 && rank_ == 0 && o.verbose) printf(".");
  }
}

// alsTucker_PP_sub (als_Tucker.cxx:679-896); o.bench: pp_bench's form (:722-730 no restart test,
// :799-814 [PPfirst]/[PPsecond], :892-893 iter++ on exit)
void TuckerEngine::pp_sub(const CpOpts &o, double tol_init, double &diffnorm, int &iter,
                          std::ofstream *csv, double &st_time) {
  const int init_iter = iter;
  double diffV = 1000, dtime_first = 0;
  std::vector<double> nd, nw;
  for (; iter <= o.maxiter; iter++) {
    int num_dw_break = 0;
    if (!o.bench) {
      read_norms(false, nd, nw);
      for (int i = 0; i < N_; i++)
        if (std::fabs(nd[i] / nw[i]) > tol_init) num_dw_break++;
    }
    if (iter == init_iter || num_dw_break > 0) {
      if (num_dw_break > 0) return;
      for (int j = 0; j < N_; j++) {
        size_t n = sizeof(double) * V_.glens[j] * r_[j];
        ops_.d2d(Winit_[j], W_[j], n);
        ops_.zero(dW_[j], n);
      }
      pp_clear();
      for (int ii = 0; ii < N_; ii++)
        for (int jj = ii + 1; jj < N_; jj++) pp_get(tk_all_but(N_, ii, jj));
      for (int ii = 0; ii < N_; ii++) pp_get(tk_all_but(N_, ii));
    }
    if ((iter % o.resprint == 0 && iter != 0) || iter == 1 || iter == o.maxiter ||
        iter == init_iter) {
      if (!o.bench) {
        if (print_block(o, iter, 1, diffnorm, diffV, csv, st_time, true)) break;
      } else {
        ops_.sync();
        const double st_time1 = now();
        compute_core_full();
        diffnorm = core_norm();
        diffV = residual();
        st_time += now() - st_time1;
        const double dtime = now() - st_time;
        if (iter != o.maxiter) {
          dtime_first = dtime;
          st_time = now();
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
        if (agree(diffnorm < o.tol || now() - st_time > o.timelimit || iter == o.maxiter)) break;
        ops_.d2d(core_prev_, core_, sizeof(double) * ncore_);
      }
    }
    sweep_pp();
  }
  if (o.bench) iter++;
}

int TuckerEngine::run_pp(const CpOpts &o, int *iters) {
  finalize_rotations();  // the PP phases difference the eigenvectors themselves
  for (int i = 0; i < N_; i++) ops_.eig_lazy(eig_base_ + i, false);
  std::ofstream csv;
  std::ofstream *pcsv = nullptr;
  if (rank_ == 0 && !o.csv_path.empty()) {
    csv.open(o.csv_path, o.csv_append ? std::ios::app : std::ios::out);
    pcsv = &csv;
    if (!o.bench) csv << "[dim],[iter],[diffnorm],[tol],[pp_update],[diffV],[dtime]\n";
  }
  if (Wprev_.empty()) {
    for (int i = 0; i < N_; i++) {
      size_t n = sizeof(double) * V_.glens[i] * r_[i];
      Wprev_.push_back((double *)ops_.alloc(n));
      Winit_.push_back((double *)ops_.alloc(n));
      dW_.push_back((double *)ops_.alloc(n));
    }
  }
  for (int i = 0; i < N_; i++) ops_.zero(dW_[i], sizeof(double) * V_.glens[i] * r_[i]);
  double st_time = now();
  int iter = 0;
  ops_.d2d(core_prev_, core_, sizeof(double) * ncore_);  // Tensor<> core_prev(core)
  double diffnorm = 10.;
  double tol_init = o.tol_init;
  while (diffnorm > o.tol && iter <= o.maxiter) {
    if (!o.bench) {
      if (rank_ == 0 && o.verbose) printf("DT starts from %d\n", iter);
      dt_sub(o, tol_init, diffnorm, iter, pcsv, st_time);
    }
    if (rank_ == 0 && o.verbose) printf("pairwise perturbation starts from %d\n", iter);
    pp_sub(o, tol_init, diffnorm, iter, pcsv, st_time);
    if (tol_init > 5e-3) tol_init *= 0.9;  // als_Tucker.cxx:947-948
    if (agree(now() - st_time > o.timelimit)) break;
  }
  ops_.sync();
  pp_clear();
  if (rank_ == 0 && o.verbose) {
    printf("\nIter = %d Final Diff norm %E \n", iter, diffnorm);
    printf("tf took %lf seconds\n", now() - st_time);
  }
  if (pcsv) csv.close();
  if (iters) *iters = iter;
  return iter == o.maxiter + 1 ? 0 : 1;
}

}  // namespace ppals
