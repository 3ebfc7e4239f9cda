// tucker.h — HOOI sweep engine for Tucker decomposition (als_Tucker.cxx) over abstract ops.
#pragma once
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "engine.h"

namespace ppals {

class TuckerEngine {
 public:
  TuckerEngine(Ops &ops, Comm &comm, const TensorDesc &V, const int *ranks);
  ~TuckerEngine();
  void set_factors(const double *Wflat);
  void get_factors(double *Wflat, double *core);
  void set_core(const double *core);  // nullptr: recompute from V and the factors
  void hosvd();                          // als_Tucker.cxx:12-70
  int64_t ttmc(int skip, double *Yhost);  // als_Tucker.cxx:76-110
  void sweep_dt();                       // als_Tucker.cxx:340-408
  void settle() { settle_all(); }        // every eigen-step behind the current factors is a checked one
  int rollbacks() const { return defer_rollbacks_; }
  int run_dt(const CpOpts &o, int *iters);  // alsTucker_DT, als_Tucker.cxx:240-424
  int run_pp(const CpOpts &o, int *iters);  // alsTucker_PP, als_Tucker.cxx:906-962

 private:
  struct Node {
    int lo, hi, parent, slo, shi;
    double *buf = nullptr;
    int64_t cap = 0;
    bool valid = false;
    bool front = false;  // a leaf stored [s_mode | ranks of the other modes] instead of the tree's order
  };
  void build_tree(int lo, int hi, int parent);
  // dims of a node's tensor: ranks on the contracted modes, full extent elsewhere
  void compute_node(int idx);
  void compute_left_half_on_vt(Node &n);
  double *ttmc_chain(int skip, int64_t *elems);  // result lives in the chain scratch
  int64_t ext(int m) const { return m == 0 ? V_.llens[0] : V_.glens[m]; }
  const double *wptr(int m) const { return W_[m] + (m == 0 ? V_.row0 : 0); }
  // leaf tensor Y_i complete on every rank (all-gather of the leading-mode rows for i = 0,
  // all-reduce of the partial sums otherwise); returns the buffer holding it
  double *complete_leaf(int i, double *Yloc, int64_t elems_local);
  // W_i = the r_i leading left singular vectors of the mode-i unfolding of Y = [L, s_i, T]
  void factor_update(int i, const double *Y, int64_t L, int64_t T);
  void compute_core_full();
  void ensure_core();  // the core the last exact sweep owes: Y_end x_{N-1} W_{N-1}
  bool core_owed_ = false;
  const double *yend_src_ = nullptr;
  int64_t yend_T_ = 1;  // layout of yend_src_: [ranks before, s_{N-1}] (1) or [s_{N-1} | ranks] (their product)
  double core_norm();
  bool agree(bool local);
  double residual();
  int64_t node_elems(const Node &n) const;

  Ops &ops_;
  Comm &comm_;
  TensorDesc V_;
  int N_;
  std::vector<int> r_;
  std::vector<double *> W_;
  double *core_ = nullptr, *core_prev_ = nullptr, *Yend_ = nullptr, *G_ = nullptr, *scal_ = nullptr;
  int64_t ncore_ = 1, yend_elems_ = 0;
  std::vector<Node> nodes_;
  std::vector<int> leaf_;
  std::vector<char> contracted_;  // scratch
  int P_ = 1, rank_ = 0;
  bool dist_ = false;
  // pairwise perturbation (als_Tucker.cxx:426-962)
  struct PPOp {
    double *buf = nullptr;
    int64_t elems = 0;
    std::vector<int64_t> dims;
  };
  std::map<std::string, PPOp> pp_;
  std::vector<double *> Wprev_, Winit_, dW_;
  double *Ytmp_ = nullptr, *Yacc_ = nullptr;
  double *thin_ = nullptr;  // [s_i x (L*T) unfolding | (L*T) x r_i right vectors] of the thin route
  int64_t thin_cap_ = 0;
  int eig_base_ = 0;  // block of warm-start slots drawn from the back end (Ops::eig_session_new)
  void finalize_rotations();
  void drop_rotations();
  bool thin_enabled_ = true;  // PPALS_TUCKER_THIN=0: always the s_i x s_i Gram (A/B, tests)
  void *VT_ = nullptr;  // second resident layout [(right modes), (left modes)], nullptr: not held
  // Order 3, one GPU: the multi-sweep dimension tree (as the CP engine's, cp_msdt_optimizer.cxx:172-207).
  // ONE first-level intermediate X_r = V x_r W_r serves the TWO mode updates that follow mode r's, then
  // the root moves on to the mode updated last: 3 tensor scans per 2 HOOI sweeps instead of 4, the same
  // factors in the same order (every product of a step sees exactly the factor versions alsTucker_DT's
  // tree gives it). Needs the tensor in all three rotations (V, VT_ = [2 | 0 1], VT2_ = [1 2 | 0]) so
  // that every root is contracted by a row-contiguous scan. PPALS_TUCKER_CHAIN=tree: the per-sweep tree.
  // Cost: 3 x the tensor's bytes resident for the session's life (the third rotation is taken only with
  // 2 x the tensor + 2 GB free at creation); X_r is kept in the TENSOR's precision, so with fp32 storage
  // the two schedules agree to ~1e-7 (X_r rounded like the tensor), with fp64 storage to rounding.
  bool ms3_ = false;
  void *VT2_ = nullptr;
  void *ms3_X_ = nullptr;
  size_t ms3_cap_ = 0;
  int ms3_root_ = -1, ms3_left_ = 0;
  bool ms3_perm_ = false;  // the leaf just served has its two rank indices in descending mode order
  double *ms3_Y_[3] = {nullptr, nullptr, nullptr};
  int64_t ms3_Ycap_[3] = {0, 0, 0};
  double *ms3_leaf(int i, int64_t *T);
  void ms3_invalidate() {
    ms3_root_ = -1;
    ms3_left_ = 0;
  }
  uint64_t tensor_gen_ = 0;  // generation of the tensor contents VT_ and the caches were built from
  void check_tensor_generation();
  void *chain_[2] = {nullptr, nullptr};  // ping-pong scratch of the mode-product chains
  size_t chain_cap_[2] = {0, 0};
  int64_t ytmp_cap_ = 0, yacc_cap_ = 0;
  const PPOp &pp_get(const std::string &args);
  void pp_clear();
  void sweep_body(const std::vector<double *> *align_ref);
  // one mode of a HOOI sweep: leaf tensor, eigen-step, (last mode) the core
  void mode_step(int i, const std::vector<double *> *align_ref, bool may_defer);
  // Deferred acceptance of eigen-steps (Ops::eig_defer / eig_verify): inside plain sweeps the host
  // does not wait for a step's checks — it enqueues ahead of the device — and asks for them when it
  // comes back to the mode (or before anything is published: print blocks, get_factors, the PP
  // phases). `defer_log_` = the modes stepped since the oldest unchecked step, in order; Wsave_[m] =
  // the factor the latest step of mode m replaced. A step that was NOT accepted: every factor
  // stepped since goes back to what it was and those steps are repeated, checked one by one.
  std::vector<int> defer_log_;
  std::vector<double *> Wsave_;
  bool defer_enabled_ = true;  // (the back end decides per slot: PPALS_EIG_DEFER=0 switches every deferral off)
  int defer_rollbacks_ = 0;
  void settle_mode(int i);  // before mode i is stepped again
  void settle_all();        // before anything derived from the factors is read
  void rollback_and_redo();
  void sweep_pp();
  bool print_block(const CpOpts &o, int iter, int pp_flag, double &diffnorm, double &diffV,
                   std::ofstream *csv, double &st_time, bool stop_at_maxiter);
  void read_norms(bool dt_phase, std::vector<double> &nd, std::vector<double> &nw);
  void dt_sub(const CpOpts &o, double tol_init, double &diffnorm, int &iter, std::ofstream *csv,
              double &st_time);
  void pp_sub(const CpOpts &o, double tol_init, double &diffnorm, int &iter, std::ofstream *csv,
              double &st_time);
  double *Yfull_ = nullptr, *gather_ = nullptr;
  int64_t yfull_cap_ = 0, gather_cap_ = 0;
};

}  // namespace ppals
