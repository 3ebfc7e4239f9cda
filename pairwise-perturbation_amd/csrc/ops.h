// ops.h — the device-operation and collective interfaces the ALS engine is written against.
//
// engine.cpp contains only control flow (which contraction, which cache, which collective, when to
// restart PP — the part of the reference that is "pure algorithm orchestration") and
// talks to the hardware through these two abstract classes. The product implements them with
// hand-written HIP kernels (hip_ops.hip) and RCCL (rccl_comm.cpp). tests/hostsim/ implements them
// with plain host loops + callbacks so the engine's host logic and the multi-rank shard plan can be
// exercised on a CPU-only box (test infrastructure, never shipped in libppals.so).
#pragma once
#include <cstddef>
#include <cstdint>

namespace ppals {

enum DType { F32 = 0, F64 = 1 };
inline size_t dtype_size(int dt) { return dt == F32 ? 4 : 8; }

// row blocks of a padded layout: `ld` stored rows per block, the first `valid` of them real
struct RowPad {
  int64_t ld = 0, valid = 0;  // ld == 0: not padded
};

constexpr int MAX_ORDER = 8;

// a factor-matrix operand of a Khatri-Rao product: `rows` rows starting at `ptr`, leading dim ld
struct FactorRef {
  const double *ptr;
  int64_t rows;
  int64_t ld;
};

// one first-order correction term of a PP mode update (als_CP.cxx:778-794): a pair operator
// T (two modes + the rank index, fp64) contracted over its mode of extent ny with dW (ny x R, lddw)
struct PPTerm {
  const double *T;
  int64_t ny;
  int keep_first;  // 1: T[x + rows*(y + ny*r)] (the kept mode is stored fastest), 0: T[y + ny*(x + rows*r)]
  const double *dW;
  int64_t lddw;
};

struct ProfileSlot {
  int64_t launches = 0;
  double ms = 0;
  double bytes = 0;
};

class Ops {
 public:
  virtual ~Ops() {}
  // ---- memory (all pointers returned by alloc are "device" pointers of this Ops) ----
  virtual void *alloc(size_t bytes) = 0;
  virtual void free(void *p) = 0;
  virtual void h2d(void *dst, const void *src, size_t bytes) = 0;  // returns when dst is usable
  virtual void d2h(void *dst, const void *src, size_t bytes) = 0;  // synchronises the stream
  virtual void d2d(void *dst, const void *src, size_t bytes) = 0;  // stream-ordered
  virtual void zero(void *p, size_t bytes) = 0;
  virtual void sync() = 0;
  virtual void *stream() { return nullptr; }
  virtual void bind() {}  // make this Ops' device current on the calling thread

  // ---- tensor generation / norms ----
  // V[e_local] = lo + (hi-lo)*u01(seed, global linear index); local shard = rows [row0,row0+l0)
  // of a leading mode of global extent g0; `rest` = product of the other extents
  virtual void fill_uniform(void *V, int dt, int64_t l0, int64_t g0, int64_t row0, int64_t rest,
                            uint64_t seed, double lo, double hi) = 0;
  // `-tensor p/p2` (laplacian_tensor, common.cxx:575-642): the element with global linear index e
  // has `ndigits` base-s digits (a_1,b_1,...,a_d,b_d); value = sum_k D[a_k,b_k] prod_{j!=k}
  // [a_j==b_j], D = tridiag(-1,2,-1)
  virtual void fill_laplacian(void *V, int dt, int64_t l0, int64_t g0, int64_t row0, int64_t rest,
                              int ndigits, int s) = 0;
  // V[e] += alpha * (lo + (hi-lo)*u01(seed, global index))   (`-tensor c` noise, test_ALS.cxx:259-264)
  virtual void add_uniform_noise(void *V, int dt, int64_t l0, int64_t g0, int64_t row0,
                                 int64_t rest, uint64_t seed, double lo, double hi,
                                 double alpha) = 0;
  // *out = sum over the local elements of (lo + (hi-lo)*u01(seed, global index))^2
  virtual void uniform_sumsq(int64_t l0, int64_t g0, int64_t row0, int64_t rest, uint64_t seed,
                             double lo, double hi, double *out) = 0;
  // V[m,k] = sum_r Q[m,r]*P[k,r]  (Q: M x R, P: K x R, column-major fp64)   (build_V)
  virtual void fill_rank(void *V, int dt, int64_t M, int64_t K, const double *Q, const double *P,
                         int R) = 0;
  // *out (device scalar) = sum_{m,k} (V[m,k] - sum_r Q[m,r]P[k,r])^2 ; Q==nullptr: sum V^2
  virtual void residual_sq(const void *V, int dt, int64_t M, int64_t K, const double *Q,
                           const double *P, int R, double *out) = 0;
  // upload fp64 host rows into the local shard (converting to dt): host is the FULL tensor
  virtual void upload_shard(void *V, int dt, const double *host_full, int64_t l0, int64_t g0,
                            int64_t row0, int64_t rest) = 0;

  // the reverse: the local shard's rows, widened to fp64, into their places of the FULL host tensor
  // (rows of other ranks are left untouched). Synchronises.
  virtual void download_shard(const void *V, int dt, double *host_full, int64_t l0, int64_t g0,
                              int64_t row0, int64_t rest) = 0;

  // dst[c + cols*r] = src[r + rows*c] (same element type dt): builds the second resident layout
  // of the tensor (right-half modes fastest) so that BOTH first-level tree nodes are suffix scans
  virtual void transpose2d(const void *src, int dt, int64_t rows, int64_t cols, void *dst) = 0;
  // `batch` independent transposes of consecutive rows x cols blocks:
  //   dst[c + cols*(r + rows*b)] = src[r + rows*(c + cols*b)]
  virtual void transpose_batched(const void *src, int dt, int64_t rows, int64_t cols, int64_t batch,
                                 void *dst) {
    const size_t step = (size_t)rows * cols * dtype_size(dt);
    for (int64_t b = 0; b < batch; b++)
      transpose2d((const char *)src + b * step, dt, rows, cols, (char *)dst + b * step);
  }
  // A resident layout whose leading block of `blk` elements is padded to `ld` (a multiple of
  // 128 B), optionally transposing on the way:
  //   dst[(c % blk) + ld*(c / blk + (cols / blk)*r)] = src[r + rows*c]    (cols % blk == 0)
  // rows == 1 is a plain pitched copy. Pad elements are set to zero. See CpEngine::build_layouts.
  virtual void pad_layout(const void *src, int dt, int64_t rows, int64_t cols, int64_t blk,
                          int64_t ld, void *dst) = 0;
  // full[a + s0*c] = stage_p[(a - p*blk) + l_p*c] for the rank p that owns row a (l_p rows each,
  // element type dt); stage_p starts at byte offset p*chunk_bytes. Re-assembles the leading-mode
  // shards of a tensor after an all-gather (Tucker HOSVD of the sharded mode).
  virtual void unpack_shards(const void *stage, int dt, int64_t s0, int64_t rest, int64_t blk,
                             int P, int64_t chunk_bytes, void *full) = 0;
  // alloc that returns nullptr instead of throwing when the device is out of memory
  virtual void *try_alloc(size_t bytes) { return alloc(bytes); }
  // how the next scans store a large result: -1 the back end's rule, 0 ordinary, 1 non-temporal
  virtual void scan_store_mode(int mode) { (void)mode; }
  // free device memory in bytes, (size_t)-1: unknown / unlimited
  virtual size_t mem_available() { return (size_t)-1; }

  // ---- Khatri-Rao product, plain: out[j + J*c] = prod_f W_f[j_f + ld_f*(col0+c)], fp64 ----
  virtual void krp(double *out, const FactorRef *f, int nf, int col0, int ncols) = 0;

  // ---- tensor scans (the only kernels that touch the s^N tensor) ----
  // V viewed as [L, J, T] (first fastest), B = KRP of `nf` factors with combined extent J:
  //   out[l + out_tstride*t + out_rstride*r] = sum_j V[l,j,t] * B[j,r]       r in [0,R)
  // This one primitive is K1 (T=1: contract a suffix), K2 (L=1: contract a prefix), the
  // single-mode TTM of the PP operator build (tstride=L, rstride=L*T: rank index last) and the
  // Tucker mode product that keeps the mode in place (tstride=L*R, rstride=L). V may also be a
  // cached fp64 intermediate (dt = F64).
  // The result is written as fp64 (out_dt = F64) or fp32 (out_dt = F32, used for the large
  // first-level intermediate of the multi-sweep schedule).
  // A PADDED resident layout (pad_layout below) stores its rows in blocks of `pad.ld` of which the
  // first `pad.valid` exist: L then counts the stored rows (a multiple of pad.ld), and the result
  // is written COMPACT — stored row l lands at (l / ld) * valid + l % ld, the pad rows nowhere —
  // so nothing downstream of a scan knows about the padding.
  virtual void scan_contract(const void *V, int dt, int64_t L, int64_t J, int64_t T,
                             const FactorRef *f, int nf, int R, void *out, int out_dt,
                             int64_t out_tstride, int64_t out_rstride, RowPad pad) = 0;
  void scan_contract(const void *V, int dt, int64_t L, int64_t J, int64_t T, const FactorRef *f,
                     int nf, int R, void *out, int out_dt, int64_t out_tstride,
                     int64_t out_rstride) {
    scan_contract(V, dt, L, J, T, f, nf, R, out, out_dt, out_tstride, out_rstride, RowPad());
  }
  // Tucker mode product keeping the mode in place (fp64 in/out or V-typed in):
  //   out[l + L*(k + Kc*t)] = sum_j X[l,j,t] * W[j + ldw*k]      (als_Tucker.cxx:102,224)
  virtual void ttm_keep(const void *X, int dt, int64_t L, int64_t J, int64_t T, const double *W,
                        int64_t ldw, int Kc, double *out) {
    FactorRef f;
    f.ptr = W;
    f.rows = J;
    f.ld = ldw;
    scan_contract(X, dt, L, J, T, &f, 1, Kc, out, F64, L * Kc, L);
  }

  // The product with the LEADING mode of a small tensor, the next mode moved to the front of the result:
  //   out[s + S*(k + Kc*t)] = sum_j X[j + J*(s + S*t)] * W[j + ldw*k]
  // (the leaf of mode 1 in a Tucker chain: what the Gram of the unfolding wants, with no transposition
  // behind a leading-mode scan). false: the back end has no such kernel — the caller takes the generic
  // route.
  virtual bool ttm_lead_front(const void * /*X*/, int /*dt*/, int64_t /*J*/, int64_t /*S*/, int64_t /*T*/,
                              const double * /*W*/, int64_t /*ldw*/, int /*Kc*/, double * /*out*/) {
    return false;
  }

  // ---- contraction of a cached intermediate (fp64) that already carries the rank index ----
  //   out[l + L*t + out_rstride*r] (+)= sum_j X[l + L*(j + J*(t + T*r))] * B[j,r]
  // B = KRP of `nf` factor refs (combined extent J). accumulate!=0: add into out.
  // X is fp64 or fp32 (xdt); out_scale (device scalar or nullptr) multiplies the contribution.
  virtual void mttv(const void *X, int xdt, int64_t L, int64_t J, int64_t T, const FactorRef *f,
                    int nf, int R, double *out, int64_t out_rstride, int accumulate,
                    const double *out_scale) = 0;
  // PP correction of one mode in one call (K9, als_CP.cxx:754-794):
  //   M[x + rows*r] = M0[x + rows*r] + sum_t sum_y T_t[x,y,r] * dW_t[y + lddw_t*r]
  // (fixed summation order: results are reproducible run to run)
  virtual void pp_correct(const double *M0, int64_t rows, int R, const PPTerm *terms, int nterms,
                          double *M) {
    d2d(M, M0, sizeof(double) * rows * R);
    for (int t = 0; t < nterms; t++) {
      FactorRef f;
      f.ptr = terms[t].dW;
      f.rows = terms[t].ny;
      f.ld = terms[t].lddw;
      if (terms[t].keep_first)
        mttv(terms[t].T, F64, rows, terms[t].ny, 1, &f, 1, R, M, rows, 1, nullptr);
      else
        mttv(terms[t].T, F64, 1, terms[t].ny, rows, &f, 1, R, M, rows, 1, nullptr);
    }
  }
  // pending-Normalize bookkeeping of cached tensors: *dst = (set_one ? 1 : *dst) * prod_{m in
  // mask} scales[m]; `scales` is what normalize() last applied (see normalize_scales()).
  virtual void scale_update(double *dst, const double *scales, unsigned mask, int set_one) = 0;
  // the same for up to 32 scalars in one launch: dst[k] for every k with (active >> k) & 1;
  // entries with (fresh >> k) & 1 start from 1.0 instead of their old value
  virtual void scale_update_many(double *dst, const double *scales, const unsigned *masks,
                                 unsigned active, unsigned fresh) {
    for (int k = 0; k < 32; k++)
      if ((active >> k) & 1u) scale_update(dst + k, scales, masks[k], (fresh >> k) & 1u);
  }
  // Normalize + the pending-scale update of the cached multi-sweep tensors (active == 0: none);
  // back ends may fold both into one launch
  // wsq (device array, stride 2, may be null): wsq[2i] = ||W_i||_F^2 after the rescaling
  virtual void normalize_ms(double *const *W, const int64_t *rows, int N, int R, double *Gall,
                            double *ms_dst, const unsigned *masks, unsigned active, unsigned fresh,
                            double *wsq = nullptr) {
    normalize(W, rows, N, R, Gall);
    if (active) scale_update_many(ms_dst, normalize_scales(), masks, active, fresh);
    if (wsq)
      for (int i = 0; i < N; i++) sumsq(W[i], rows[i] * R, wsq + 2 * i);
  }
  virtual const double *normalize_scales() = 0;  // device array [N] written by normalize()

  // ---- R x R normal-equation side ----
  // G = W^T W  (rows x R, ld)  ->  G[R*R]
  virtual void gram(const double *W, int64_t rows, int64_t ld, int R, double *G) = 0;
  // S = Hadamard_{j != mode} G_j (+ lambda I); Sinv = S^{-1} through a symmetric eigen-
  // decomposition without truncation (the reference's SVD_solve, common.cxx:717-722)
  virtual void gram_system(const double *Gall, int N, int mode, int R, double lambda, double *S,
                           double *Sinv) = 0;
  // for `rows` rows: grad = -M + Wold*S ; Wnew = M*Sinv ; *gradsq = sum grad^2 (overwritten; 0 when rows == 0)
  // if Winit != nullptr (SVD_solve_mod, common.cxx:739-758): dW = ratio*(Wnew-Winit) and, when
  // ratio != 1, Wnew = Winit + dW.
  virtual void cp_update(const double *M, int64_t ldm, const double *Wold, int64_t ldw,
                         double *Wnew, int64_t ldn, double *grad, int64_t ldg, int64_t rows,
                         int R, const double *S, const double *Sinv, double *gradsq,
                         const double *Winit, int64_t ldi, double *dW, int64_t ldd,
                         double ratio) = 0;
  // one whole single-rank mode update: gram_system + cp_update (W updated in place) + gram of the
  // new W into Gall[mode]. Backends may fuse it into one launch; S/Sinv may be nullptr.
  // dwsq (device scalar, may be null; needs Winit and ldd == rows): *dwsq = ||dW||_F^2 afterwards
  // (the restart test of the PP phase, als_CP.cxx:657-663, without a launch of its own).
  virtual void cp_mode_update(double *Gall, int N, int mode, int R, double lambda, const double *M,
                              int64_t ldm, double *W, int64_t ldw, double *grad, int64_t ldg,
                              int64_t rows, double *gradsq, const double *Winit, int64_t ldi,
                              double *dW, int64_t ldd, double ratio, double *S, double *Sinv,
                              double *dwsq = nullptr) {
    gram_system(Gall, N, mode, R, lambda, S, Sinv);
    cp_update(M, ldm, W, ldw, W, ldw, grad, ldg, rows, R, S, Sinv, gradsq, Winit, ldi, dW, ldd,
              ratio);
    gram(W, rows, ldw, R, Gall + (size_t)mode * R * R);
    if (dwsq && Winit) sumsq(dW, rows * R, dwsq);
  }
  // The same with M handed over in ROW BLOCKS (the receive buffer of an all-gather of the ranks' row
  // blocks: block p = rows [p*blk, (p+1)*blk) stored at Mblk + p*blk*R with leading dimension blk,
  // blk * P == rows). A back end may read the blocks as they are; the default re-assembles them in
  // `scratch` (rows x R) first.
  virtual void cp_mode_update_blocked(double *Gall, int N, int mode, int R, double lambda,
                                      const double *Mblk, int64_t blk, int P, double *scratch,
                                      double *W, int64_t ldw, double *grad, int64_t ldg, int64_t rows,
                                      double *gradsq, const double *Winit, int64_t ldi, double *dW,
                                      int64_t ldd, double ratio, double *S, double *Sinv) {
    unpack_blocks(Mblk, rows, rows, R, blk, P, scratch);
    cp_mode_update(Gall, N, mode, R, lambda, scratch, rows, W, ldw, grad, ldg, rows, gradsq, Winit, ldi,
                   dW, ldd, ratio, S, Sinv);
  }
  // Hint: the next cp_mode_update will be for `mode` with these arguments, and the Grams of the
  // other modes are final NOW — a back end may prepare S / S^-1 on the side of the contraction that
  // is launched next (mttv / pp_correct) instead of at the head of the update launch. Optional.
  virtual void arm_gram_system(const double * /*Gall*/, int /*N*/, int /*mode*/, int /*R*/,
                               double /*lambda*/, double * /*S*/, double * /*Sinv*/) {}
  // Hint: the next cp_mode_update is the last of a sweep and a Normalize of these N full factors
  // follows it immediately — a back end may fold it into that launch, together with the pending-scale
  // update of the cached multi-sweep tensors (ms_dst / masks / active / fresh as in normalize_ms;
  // active == 0: none alive). Returns true when it WILL (the caller then skips its normalize call).
  virtual bool arm_normalize(double *const * /*W*/, const int64_t * /*rows*/, int /*N*/, int /*R*/,
                             double * /*Gall*/, int /*mode*/, double * /*wsq*/,
                             double * /*ms_dst*/ = nullptr, const unsigned * /*masks*/ = nullptr,
                             unsigned /*active*/ = 0, unsigned /*fresh*/ = 0) {
    return false;
  }
  // Normalize (common.cxx:680-688) on N full factors using ||W_i||^2 = trace(G_i); rescales the
  // Grams consistently.
  virtual void normalize(double *const *W, const int64_t *rows, int N, int R, double *Gall) = 0;
  // out[2i] = ||A_i - B_i||^2 (B_i may be null -> ||A_i||^2 only in out[2i+1]), out[2i+1]=||A_i||^2
  // and if B_i != null and update_prev: B_i = A_i afterwards. (als_CP.cxx:594-603, :659-663)
  virtual void diff_norms(double *const *A, double *const *B, const int64_t *n, int N,
                          int store_diff, double *const *D, int update_prev, double *out) = 0;
  // row-block pack/unpack for reduce-scatter / all-gather of a column-major rows x R matrix:
  // blocked[p][x + blk*r] <-> nat[p*blk + x + ld*r]; rows beyond `rows` are zero in blocked
  virtual void pack_blocks(const double *nat, int64_t rows, int64_t ld, int R, int64_t blk, int P,
                           double *blocked) = 0;
  virtual void unpack_blocks(const double *blocked, int64_t rows, int64_t ld, int R, int64_t blk,
                             int P, double *nat) = 0;

  // ---- Tucker: Gram of the mode-`pos` unfolding and its leading eigenvectors ----
  // G[p + J*q] = sum_{l,t} X[l,p,t]*X[l,q,t]   (unroll_tensor_contraction, common.cxx:205-223)
  virtual void unfold_gram(const void *X, int dt, int64_t L, int64_t J, int64_t T, double *G) = 0;
  // U (J x rank, column-major) = leading eigenvectors of symmetric PSD G (J x J), descending
  virtual void top_eigvecs(double *G, int64_t J, int rank, double *U) = 0;
  // the same inside a HOOI iteration: `slot` names a sequence of calls on slowly changing matrices
  // (one per mode), so a back end may keep what it learnt from the previous call of the slot —
  // where the gap below the rank-th eigenvalue lies, the previous basis — and skip the full
  // eigen-decomposition. The result is the same invariant subspace, eigenvectors sorted descending.
  virtual void top_eigvecs_warm(double *G, int64_t J, int rank, double *U, int /*slot*/) {
    top_eigvecs(G, J, rank, U);
  }
  // Lazy eigenvectors. With eig_lazy(slot, true) a warm top_eigvecs_warm may return ANY orthonormal
  // basis of the invariant subspace (all a HOOI sweep needs) and finish the eigen-decomposition
  // inside the subspace off the critical path; eig_pending_rotation(slot) then returns the
  // rank x rank matrix Y (device, column-major, columns sorted by descending eigenvalue) with
  // U_eigenvectors = U_returned * Y — nullptr when U already holds the eigenvectors. The caller
  // applies Y (to the factor and to whatever it derived from it) when it needs eigenvectors and
  // says so with eig_rotation_done(slot).
  virtual void eig_lazy(int /*slot*/, bool /*on*/) {}
  // Deferred acceptance. With eig_defer(slot, true) a lazy warm step of a slot whose recent steps
  // were all accepted at the first attempt may RETURN BEFORE ITS CHECKS HAVE BEEN READ: nothing
  // inside a HOOI sweep then waits for the device, and the host enqueues ahead of it. The caller
  // asks with eig_verify(slot) before it uses the slot again and before it publishes anything that
  // was derived from the returned factor:
  //   -1  nothing pending,  0  the pending step has been accepted,
  //    1  it has NOT: the factor it returned is to be thrown away — the caller restores what it
  //       had before that call and repeats the work from there (the slot is back in the state it
  //       had before the step and takes the checked route next time).
  // discard = true: the pending step is dropped whatever its checks say (its input was wrong).
  virtual void eig_defer(int /*slot*/, bool /*on*/) {}
  // A deferring slot finishes its checks on a second stream, from the Gram the step was given — so
  // that Gram must outlive the call. eig_gram() hands out the slot's own J x J buffer to build the
  // Gram in (after waiting for whatever still reads it), or nullptr: use your own workspace.
  virtual double *eig_gram(int /*slot*/, int64_t /*J*/) { return nullptr; }
  virtual bool eig_deferred(int /*slot*/) { return false; }  // a step of the slot awaits eig_verify
  virtual int eig_verify(int /*slot*/, bool /*discard*/ = false) { return -1; }
  virtual const double *eig_pending_rotation(int /*slot*/) { return nullptr; }
  virtual void eig_rotation_done(int /*slot*/) {}
  // A session's block of 64 warm-start slots [base, base + 64) (what a back end remembers under a
  // slot dies with the session that drew the block)
  virtual int eig_session_new() { return 0; }
  virtual void eig_session_free(int /*base*/) {}
  // U (rows x r, column-major, ld = rows) -> orthonormal columns spanning the same nested
  // subspaces (column k stays in span(U[:, :k+1]) with a positive component on the old column k:
  // a QR factorisation's Q). Returns false, leaving U unspecified, when U is numerically rank
  // deficient.
  virtual bool orthonormalize(double *U, int64_t rows, int r) = 0;
  virtual void sumsq(const double *x, int64_t n, double *out) = 0;  // *out = sum x^2
  // ---- low-rank factor updates (class-API optimizers CPDTLR / CPMSDTLR, src/optimizer/) ----
  // out (rows x C) = A (rows x K) * B (K x C) [+ D (rows x C)]; all column-major fp64, ld = rows
  // for the tall ones, ld = K for B; D may be null or alias out
  virtual void rows_times_small(const double *A, int64_t rows, int K, const double *B, int C,
                                const double *D, double *out) = 0;
  // X[e + n*c] += sum_k T[e + n*k] * VT[k + r*c], c < R: the cached first contraction updated by a
  // rank-r change of the contracted factor (update_cached_tensor, cp_dt_lr_optimizer.cxx:142-168);
  // X is stored in xdt (the tensor's precision), T (n x r) and VT (r x R) are fp64
  virtual void lowrank_accumulate(void *X, int xdt, int64_t n, int R, const double *T, int r,
                                  const double *VT) = 0;
  virtual void add_inplace(double *dst, const double *src, int64_t n) = 0;  // dst += src
  // W[:,k] *= (<W[:,k], Wref[:,k]> > 0 ? +1 : -1)   (als_Tucker.cxx:632-643, :874-885)
  virtual void sign_align(double *W, const double *Wref, int64_t rows, int r) = 0;


  // A stopwatch on the launch stream (the online placement choice of the multi-sweep schedule):
  // timer_begin() marks the stream and returns a handle (-1: no stopwatch to be had), timer_end(h)
  // marks it again, timer_read(h) returns the seconds between the two marks once both have been
  // reached — waiting for them if need be — and gives the handle back. Nothing else synchronises.
  virtual int timer_begin() { return -1; }
  virtual void timer_end(int /*h*/) {}
  virtual double timer_read(int /*h*/) { return -1.0; }

  // profiling of the scan kernels (HIP events on the launch stream)
  virtual void profile_enable(int /*level*/) {}
  virtual void profile_collect() {}
  ProfileSlot prof[2];
};

class Comm {
 public:
  virtual ~Comm() {}
  virtual int rank() const = 0;
  virtual int size() const = 0;
  virtual bool is_self() const { return false; }  // the no-op single-process communicator
  // all on "device" pointers of the paired Ops, fp64, stream-ordered with the Ops' stream
  virtual void allreduce_sum(double *buf, int64_t n) = 0;
  virtual void reduce_scatter_sum(const double *send, double *recv, int64_t recvcount) = 0;
  virtual void allgather(const double *send, double *recv, int64_t sendcount) = 0;
};

class SelfComm : public Comm {
 public:
  int rank() const override { return 0; }
  int size() const override { return 1; }
  bool is_self() const override { return true; }
  void allreduce_sum(double *, int64_t) override {}
  void reduce_scatter_sum(const double *, double *, int64_t) override {}
  void allgather(const double *, double *, int64_t) override {}
};

}  // namespace ppals
