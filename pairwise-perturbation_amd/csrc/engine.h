// engine.h — ALS sweep engine for CP (dimension tree + pairwise perturbation) written against the
// abstract Ops/Comm interfaces (ops.h). Pure host control flow: which contraction, which cache,
// which collective, when to restart — the part the reference implements in als_CP.cxx.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "ops.h"

namespace ppals {

struct TensorDesc {
  int order = 0;
  int64_t glens[MAX_ORDER];  // global extents
  int64_t llens[MAX_ORDER];  // local extents (differs from glens only in mode 0)
  int64_t row0 = 0;          // first leading-mode row owned by this rank
  int dtype = F32;
  void *data = nullptr;
  int64_t nloc = 0;
  // bumped by every fill / upload of the tensor (owned by the ppals_tensor handle; nullptr: the
  // contents never change). Sessions compare it with the generation their derived data (second
  // resident layout, cached contractions) was built from and rebuild when it moved.
  const uint64_t *generation = nullptr;
};

// leading-mode block partition shared by the tensor shard and the factor-matrix row blocks
inline int64_t block_rows(int64_t s, int P) { return (s + P - 1) / P; }
// PPALS_FORCE_COMM=1: run the sharded code paths (pack, reduce-scatter, all-gather, all-reduce) even
// with one rank, so a single-GPU box exercises the RCCL plumbing end to end (tests only)
inline bool force_comm_path() {
  const char *v = getenv("PPALS_FORCE_COMM");
  return v && atoi(v) != 0;
}

int tensor_create(Ops &ops, Comm &comm, int order, const int64_t *glens, int dtype,
                  TensorDesc *out, std::string *err);
void tensor_fill_cp(Ops &ops, const TensorDesc &V, int R, const double *Wtrue_flat);
void tensor_fill_uniform(Ops &ops, const TensorDesc &V, uint64_t seed, double lo, double hi);
void tensor_upload(Ops &ops, const TensorDesc &V, const double *host_full);
void tensor_download(Ops &ops, const TensorDesc &V, double *host_full);
// `-tensor p/p2` (laplacian_tensor + fold_unfold, common.cxx:575-642,870-880): ndigits = -dim, s = -size
void tensor_fill_laplacian(Ops &ops, const TensorDesc &V, int ndigits, int s);
// `-tensor c` (Gen_collinearity + noise, common.cxx:361-423, test_ALS.cxx:246-264)
void collinear_factors(const int64_t *lens, int N, int R, double col_min, double col_max,
                       uint64_t seed, double *Wflat);
void tensor_fill_collinear(Ops &ops, Comm &comm, const TensorDesc &V, int R, double col_min,
                           double col_max, double ratio_noise, uint64_t seed);
double tensor_norm(Ops &ops, Comm &comm, const TensorDesc &V);

struct CpOpts {
  double tol = 0, timelimit = 5e3;
  int maxiter = 0;
  double lambda = 0;
  int resprint = 10;
  int bench = 0;
  double tol_init = 1e-2, ratio_step = 1.0;
  double update_percentage = 1.0;  // -pp 2: fraction of the modes updated per PP sweep
  int update_rank = 0;             // class API, low-rank-update optimizers (run.cxx -updaterank)
  int randomsvd = 0;               // ... and their randomized range finder (run.cxx -randomsvd)
  std::string csv_path;
  bool csv_append = false;
  bool verbose = false;
};

class CpEngine {
 public:
  CpEngine(Ops &ops, Comm &comm, const TensorDesc &V, int R);
  ~CpEngine();

  void set_factors(const double *Wflat, const double *gradWflat);
  void set_schedule(int schedule);
  int schedule() const { return schedule_; }
  // one JSON object: where the online placement choice put every root's first-level intermediate
  // (block, offset, store kind, fastest / slowest sample, settled or still exploring)
  std::string placement_report() const;
  // operator builds of the PP phases (Build_mttkrp_map, als_CP.cxx:678-694): how many, and — while
  // timing is on (a stream synchronisation on both sides of a build) — how long they took
  void pp_build_stats(int mode, int64_t *builds, double *seconds) {
    if (builds) *builds = pp_builds_;
    if (seconds) *seconds = pp_build_s_;
    if (mode != 0) {
      pp_builds_ = 0;
      pp_build_s_ = 0;
      pp_build_timed_ = mode > 0;
    }
  }
  void get_factors(double *Wflat, double *gradWflat);

  // body of the reference's sweep loop: als_CP.cxx:215-303 (clear cache, N mode updates,
  // Normalize). Asynchronous on the Ops stream.
  void sweep_dt(double lambda);
  double gradnorm();  // sqrt(sum_i ||grad_W[i]||^2), als_CP.cxx:174-181; synchronises
  double residual();  // ||V - [[W]]||_F, als_CP.cxx:183-187; synchronises

  // kernel-level access for parity tests
  int64_t tree_node(const std::string &key, double *out_host);
  void mttkrp(int mode, double *M_host);
  int64_t pp_operator(const std::string &contracted, double *out_host);
  void gram_system(int mode, double lambda, double *S_host, double *Sinv_host);

  // drivers
  int run_dt(const CpOpts &o, int *iters);  // alsCP_DT, als_CP.cxx:127-320
  int run_pp(const CpOpts &o, int *iters);  // alsCP_PP, als_CP.cxx:1082-1137
  int run_pp_partupdate(const CpOpts &o, int *iters);  // alsCP_PP_partupdate, als_CP.cxx:1146-1207
  // CPD<dtype,Optimizer>::als, src/CP.cxx:100-186; kind 0 Simple, 1 DT, 2 MSDT optimizer
  int run_class(int kind, const CpOpts &o, double *sweeps, int *iters);
  void update_modes(int first, int count, double lambda);


  int order() const { return N_; }
  int rank_r() const { return R_; }

 private:
  struct Node {
    int lo, hi;
    int parent = -1;   // index into nodes_, -1: parent is the root
    int slo, shi;      // sibling range
    double *buf = nullptr;
    int64_t elems = 0;  // without the rank index
    bool valid = false;
  };
  struct PPOp {
    void *buf = nullptr;
    int dt = F64;            // level-1 operators are kept in the tensor's own precision
    int64_t elems = 0;
    std::vector<int> modes;  // remaining modes in STORAGE order (first fastest)
    bool owned = true;       // false: borrowed from the multi-sweep cache (never freed here)
    const double *scale = nullptr;  // pending Normalize factor of a borrowed tensor (device scalar)
  };
  // grow-only pool of the PP operator buffers: a PP phase re-uses the buffers of the previous one
  // instead of allocating / freeing ~1 GB of operators around every phase. Keyed by operator for
  // what the approximate sweeps read, by LEVEL ("#1", "#2", ...) for the scaffolding above it
  struct PPBuf {
    void *buf = nullptr;
    size_t cap = 0;
  };
  std::map<std::string, PPBuf> pp_pool_;
  void *pp_buffer(const std::string &seq, size_t bytes);
  bool pp_norms_live_ = false;  // Normalize is asked for ||W_i||^2 (inside sweep_pp)
  double *pp_norms_ = nullptr;  // [2N]: ||dW_i||^2, ||W_i||^2 left by the fused PP mode updates

  // a cached intermediate of the multi-sweep schedule: modes in storage order (first fastest),
  // rank index last; `scale` (device scalar) is the Normalize factor still owed to its contents
  struct RTensor {
    void *buf = nullptr;
    size_t cap = 0;  // bytes
    int dt = F64;
    std::vector<int> modes;
    unsigned contracted = 0;
    int slot = -1;         // index into ms_scales_ (device scalars)
    bool pending = false;  // a Normalize happened after this tensor was built: scale is owed
    bool valid = false;
  };
  struct MsNode {
    int lo, hi, parent, slo, shi;  // ranges over positions of the step's cyclic mode list
    RTensor t;
    std::vector<RTensor> tmp;
  };
  void ms_build_tree(int lo, int hi, int parent);
  void ms_start_step(int root);
  void ms_compute(int idx);
  void ms_reserve(RTensor &t, size_t bytes);
  void ms_contract(const RTensor &src, int mode, RTensor &dst, const double *in_scale);
  void sweep_msdt(double lambda);
  void ms_invalidate() { ms_root_ = -1; }
  int schedule_ = 1;  // 1: multi-sweep dimension tree (default), 0: the reference's two-node tree
  int ms_root_ = -1;  // first mode of the running step's root set (-1: no step)
  int ms_k_ = 1;      // modes contracted by one first-level scan (ms_choose_roots)
  // modes that are never part of a root set: the partitioned mode 0 of a sharded session (its X would
  // be a partial sum of global size) and modes so short that X = V x_m W_m is no smaller than the tensor
  // (extent 3 at R = 10: X is 3.3 x the tensor — the reference's coil-100 / time-lapse shapes). The root
  // set is slid back past them (ms_next_root); the cost model prices both (ms_schedule_cost).
  int test_blocks_ = 0;            // PPALS_TEST_BLOCKED_UPDATE (test hook, see mode_update)
  double *test_blkbuf_ = nullptr;
  unsigned ms_excl_ = 0;
  bool ms_set_excluded(int first, int k, unsigned excl) const;
  int ms_next_root(int i, int k, unsigned excl) const;  // first mode of the root set that serves update i, -1: none
  double ms_schedule_cost(int k, unsigned excl) const;  // tensor-scan equivalents per sweep, 1e300: not schedulable
  int ms_choose_roots();
  void ms_set_roots(int k);
  void ms_mode_update(int i, double lambda, bool last_of_sweep = false);
  bool ms_norm_fused_ = false;  // the sweep's Normalize went into its last update launch (Ops::arm_normalize)
  unsigned ms_collect_scales(unsigned *masks, unsigned *fresh, int skip_node);
  RTensor ms_X_;
  // Placement of the first-level intermediate: the scan reads the tensor and writes X at the same
  // time, and how the two streams fall onto the HBM channels depends on where X lies relative to
  // the tensor buffer — 6-9 % of the launch between placements of one and the same kernel
  // (profiles/r02l_place_bench_*.txt) — and on the kind of store. X lives at a per-root offset
  // inside one of TWO blocks over-allocated by a slack of 64 MB — one taken before the session's
  // second resident layout, one after it, so that they lie several GB apart at no cost (which block
  // suits which source is a property of the pair, profiles/r03_place_regions.md). Block, offset and
  // store kind are found ONLINE: the first visits of a root run the sweep's REAL scan at one candidate each, timed
  // by a pair of events on the stream (Ops::timer_*), read when the root comes round again; after
  // ~24 visits the root keeps the fastest; a block no root chose is freed. No trial launches, no
  // set-up time, the results do not depend on where X lies. PPALS_PLACE_TUNE=0: one block, offset 0,
  // store kind by size.
  struct PlaceCand {
    int blk = 0;           // 0: the block allocated behind the second resident layout, 1: the one in front of it
    int64_t off = 0;
    int nt = -1;           // store kind: -1 the back end's rule, 0 ordinary, 1 non-temporal
    double best = 1e300;   // fastest sample, seconds
    int samples = 0;
  };
  struct PlaceExplore {
    int phase = -1;        // -1 not started, 0 offsets, 1 finalists x store kinds, 2 settled, 3 not worth it
    size_t next = 0;       // candidate of the next visit
    int timer = -1;        // stopwatch of the visit in flight (Ops::timer_begin)
    int timer_cand = -1;
    int chosen = -1;
    int layout = 0;        // which resident layout the root's scan reads
    double worst = 0;      // slowest sample seen, seconds
    int visits = 0;
    bool gated = false;    // settled early: the candidates' spread was inside the gate (ms_place_pick)
    std::vector<PlaceCand> cands;
  };
  static constexpr int kPlaceGateSamples = 8;       // samples of a root before the gate is consulted
  static constexpr double kPlaceGateSpread = 0.04;  // fastest / slowest candidate closer than this: stop
  PlaceExplore ms_place_[MAX_ORDER];
  void ms_place_collect(PlaceExplore &ex);
  int ms_place_pick(PlaceExplore &ex);
  void *ms_X_base_ = nullptr;
  void *ms_X_alt_ = nullptr;  // the second candidate block (nullptr: none / released)
  void ms_place_release_unchosen();
  size_t ms_X_cap_ = 0;
  int64_t pp_builds_ = 0;
  double pp_build_s_ = 0;
  bool pp_build_timed_ = false;
  bool ms_tune_enabled_ = true;
  size_t ms_X_slack() const;
  static double place_min_bytes();
  size_t ms_X_bytes(int first, int k) const;
  void *big_alloc(size_t bytes);  // gives optional resident layouts back when the device is full
  std::vector<MsNode> ms_nodes_;
  // ---- low-rank-update optimizers of the class API (CPDTLROptimizer / CPMSDTLROptimizer,
  // src/optimizer/cp_dt_lr_optimizer.cxx, cp_msdt_lr_optimizer.cxx; randomsvd = 0). The first
  // contraction of a step, V x_left W_left, is KEPT per root (lr_cache_) and, when W_left has
  // changed by a rank-r update Us * VT only, brought up to date with r tensor columns instead of
  // R: cached += (V x_left Us) * VT.
  void lr_step_begin(int left, bool reuse, int r);
  void lr_mode_update(int i, double lambda, int r, const double *base);
  void lr_release();
  void *lr_cache_[MAX_ORDER] = {nullptr};
  bool lr_have_[MAX_ORDER] = {false};
  bool lr_random_ = false;        // randomized_svd in the rank-r update (common.cxx:691-709)
  uint64_t lr_random_calls_ = 0;  // blocks of R*r draws consumed in this run
  RTensor lr_desc_[MAX_ORDER];
  void *ms_X_override_ = nullptr;  // ms_start_step writes the first-level intermediate here
  double *lr_X_ = nullptr, *lr_Us_ = nullptr, *lr_G2_ = nullptr, *lr_small_ = nullptr;
  double *lr_T_ = nullptr;
  size_t lr_T_cap_ = 0;


  std::vector<int> ms_order_;  // the N-k modes of the step in update order
  std::vector<int> ms_leaf_;   // node index of each list position
  double *ms_scales_ = nullptr;  // one pending-Normalize scalar per cached tensor (<= 32)
  const double *ms_scale_of(const RTensor &t) const { return t.pending ? ms_scales_ + t.slot : nullptr; }

  int64_t ext(int m) const { return m == 0 ? V_.llens[0] : V_.glens[m]; }
  FactorRef fref(int m, double *const *W) const;
  int64_t prod_ext(int lo, int hi) const;
  void build_tree(int lo, int hi, int parent);
  int find_node(int lo, int hi) const;
  void compute_node(int idx);
  void refresh_grams();
  void mode_update(int i, const double *M, int64_t ldm, double lambda, bool pp, double ratio);
  void normalize();
  const PPOp &pp_get(const std::string &seq);
  int pp_last_mode(const std::string &seq) const;      // the contracted mode of `seq` that is removed last
  std::string pp_parent(const std::string &seq) const;  // `seq` without it
  // out (+)= T contracted over `cmode` with f; T is a pair operator (two modes, any storage order)
  void pp_contract_pair(const PPOp &T, int cmode, const FactorRef &f, double *out, int64_t out_rows);
  bool pp_fast_ = true;  // both resident layouts + typed level-1 operators (pp_operator() switches it off for its fp64 hand-out)
  bool pp_no_borrow_ = false;  // pp_operator(): never hand out the (scaled) multi-sweep intermediate
  void pp_clear();
  void pp_build_all();
  void sweep_pp(double lambda, double ratio);
  double allreduce_scalar(double x);
  bool agree(bool local);
  void read_norms(bool dt_phase, std::vector<double> &nd, std::vector<double> &nw);
  bool print_block(const CpOpts &o, int iter, int pp_flag, double &projnorm, double &diffV,
                   std::ofstream *csv);
  double dt_sub(const CpOpts &o, double &projnorm, int &iter, std::ofstream *csv);
  double pp_sub(const CpOpts &o, double &projnorm, int &iter, std::ofstream *csv);
  double pp_partupdate_sub(const CpOpts &o, double &projnorm, int &iter, std::ofstream *csv);
  int run_pp_common(const CpOpts &o, int *iters, bool partupdate);

  Ops &ops_;
  Comm &comm_;
  TensorDesc V_;
  int N_, R_, P_, rank_;
  bool dist_ = false;  // take the collective code paths (P_ > 1, or PPALS_FORCE_COMM=1 for tests)
  std::vector<double *> W_, gradW_, Wprev_, Winit_, dW_, dM_, Mm_;
  double *G_ = nullptr;       // N Gram matrices, R*R each
  double *S_ = nullptr, *Sinv_ = nullptr;
  double *gradsq_ = nullptr;  // per-mode local sum of grad^2 (device)
  double *scal_ = nullptr;    // small device scalar scratch (>= 4*MAX_ORDER)
  double *sendbuf_ = nullptr, *recvbuf_ = nullptr, *gatherbuf_ = nullptr;
  double *Mbuf_ = nullptr;    // PP: M_i^0 + corrections
  double *Qbuf_ = nullptr, *Pbuf_ = nullptr;  // residual KRP operands
  // Resident storage orders of the local tensor that the scans may read. [0] is the tensor itself.
  // [1] (if built) lists the right-half modes first, so that contractions of left-half modes are
  // row-contiguous suffix scans too. When the tensor's column strides are not multiples of 128 B
  // (s = 50, 324: the reference scripts' shapes) — a scan then runs at 0.61 instead of 0.82 of
  // peak, profiles/r02s_stride_bench.txt — [1] is stored with its leading block of `q` modes
  // padded to 128 B, and [2] is such a padded copy in the tensor's own mode order; every root set
  // at a position >= q of a padded layout reads 128-B aligned columns, and the scan writes its
  // result compact (RowPad), so nothing else knows about the padding.
  struct Layout {
    void *ptr = nullptr;
    std::vector<int> order;   // modes, fastest first
    int q = 0;                // leading modes in the padded block (0: not padded)
    int64_t blk = 0, ld = 0;  // real / stored elements of that block
    bool owned = false;
    size_t bytes = 0;         // of an owned layout
  };
  std::vector<Layout> lay_;
  int vt_state_ = 0;          // 0 not tried, 1 second layout built, -1 unavailable (disabled / no memory)
  uint64_t tensor_gen_ = 0;   // generation of the tensor contents the layouts and caches were built from
  void ensure_transposed();   // builds lay_[1] (and [2]) once
  void fill_layout(const Layout &l);
  // one tensor scan that contracts the (cyclically) consecutive modes first .. first+k-1
  struct ScanPlan {
    const Layout *lay = nullptr;
    int64_t L = 1, Lc = 1, J = 1, T = 1;  // stored / compact rows in front, contracted, behind
    RowPad pad;
    std::vector<int> set, kept;  // contracted / kept modes in storage order
  };
  bool plan_scan(int first, int k, bool natural_only, ScanPlan &plan);
  void check_tensor_generation();
  // s x R partials up to this size use one all-reduce + redundant update instead of
  // reduce-scatter + row-block update + all-gather (PPALS_COMM_SMALL_BYTES overrides)
  int64_t small_msg_bytes_ = 1 << 20;
  bool grad_replicated_[MAX_ORDER];
  int64_t maxs_ = 0, maxblk_ = 0;
  std::vector<Node> nodes_;
  std::vector<int> leaf_;  // node index of each leaf
  std::map<std::string, PPOp> pp_;
  bool grad_from_sweep_ = false;
  double init_gradnorm_ = 0;
  double st_time_ = 0;
};

}  // namespace ppals
