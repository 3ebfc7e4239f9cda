/* ppals.h — C ABI of the MI355X-native ALS sweep engine (libppals.so).
 *
 * Drop-in boundary for the reference's function API (LinjianMa/pairwise-perturbation): the
 * reference has no FFI layer; its boundary is the set of free C++ functions test_ALS.cxx:352-396
 * calls. Each entry point below names the reference interface it replaces. Plain pointers and
 * sizes only — no torch / CTF / C++ types cross this boundary.
 *
 * Conventions (identical to the reference's CTF objects):
 *   - tensors are dense, FIRST INDEX FASTEST: V[i0 + lens[0]*(i1 + lens[1]*(i2 + ...))]
 *   - factor matrix W_i is lens[i] x R, column-major; `Wflat` = W_0,...,W_{N-1} concatenated, fp64
 *   - the tensor lives in HBM as fp32 (PPALS_F32) or fp64 (PPALS_F64); all factor-matrix, Gram,
 *     solve and norm arithmetic is fp64 in both modes
 *   - multi-GPU: one process per GPU; the tensor is block-partitioned along its LEADING mode
 *     (rank p owns rows [p*ceil(s0/P), ...)), factor matrices are replicated
 *
 * Every function returns 0 on success and a negative code on failure (never aborts);
 * ppals_last_error() returns the message. There is NO CPU fallback: without a HIP device
 * ppals_ctx_create fails with PPALS_ERR_NO_DEVICE.
 */
#ifndef PPALS_H
#define PPALS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PPALS_F32 0
#define PPALS_F64 1

#define PPALS_OK 0
#define PPALS_ERR_NO_DEVICE (-1)
#define PPALS_ERR_HIP (-2)
#define PPALS_ERR_ARG (-3)
#define PPALS_ERR_COMM (-4)
#define PPALS_ERR_UNSUPPORTED (-5)
#define PPALS_MAX_ORDER 8
#define PPALS_UNIQUE_ID_BYTES 128

typedef struct ppals_ctx ppals_ctx;       /* device, stream, workspaces, communicator */
typedef struct ppals_tensor ppals_tensor; /* the (local shard of the) dense input tensor in HBM */
typedef struct ppals_cp ppals_cp;         /* a CP-ALS session: factors, Grams, tree caches in HBM */
typedef struct ppals_tucker ppals_tucker; /* a Tucker-HOOI session */

const char *ppals_last_error(void);
const char *ppals_version(void);

/* Tucker sessions with a mode extent above 64 use the vendor symmetric eigensolver (rocSOLVER
 * dsyevd). Its libraries register their code objects in milliseconds when they enter the process
 * BEFORE the HIP runtime is initialised, and in minutes afterwards (0.013 s vs 253 s measured).
 * Call this first thing — before ppals_ctx_create and before anything else touches the GPU (e.g.
 * torch.cuda) — in a process that will run such a session. Without it they are loaded on demand:
 * the library then prints one line on stderr (what is about to happen, how long it can take, how to
 * avoid it) before it stalls, and with PPALS_STRICT_PRELOAD=1 in the environment the call that needed
 * the solver fails with PPALS_ERR_UNSUPPORTED instead (csrc/preload_policy.h). */
int ppals_preload_eigensolver(void);

/* ---- context (replaces CTF::World dw, test_ALS.cxx:200) ---- */
int ppals_ctx_create(ppals_ctx **out, int device);
void ppals_ctx_destroy(ppals_ctx *ctx);
/* RCCL bootstrap: rank 0 calls ppals_get_unique_id and ships the 128 bytes to all ranks by any
 * means (torch.distributed, a file, a socket); then every rank calls ppals_ctx_init_comm. */
int ppals_get_unique_id(void *out128);
int ppals_ctx_init_comm(ppals_ctx *ctx, int rank, int nranks, const void *unique_id128);
int ppals_ctx_rank(const ppals_ctx *ctx);
int ppals_ctx_nranks(const ppals_ctx *ctx);
int ppals_ctx_sync(ppals_ctx *ctx);
/* HIP-event kernel timing on the engine's own stream (bench.py roofline leg).
 * which: 0 = tensor-scan kernels (K1/K2/K8), 1 = the other bracketed kernels.
 * level: 0 off, 1 = bracket the tensor scans only (an event pair costs ~10 us of stream time, so
 * the timed region of bench.py pays it on the dominant kernel alone), 2 = bracket both groups */
int ppals_profile_enable(ppals_ctx *ctx, int level);
int ppals_profile_read(ppals_ctx *ctx, int which, int64_t *launches, double *total_ms,
                       double *algo_bytes);
int ppals_profile_reset(ppals_ctx *ctx);

/* ---- tensor (replaces CTF::Tensor<> V and its initialisers, test_ALS.cxx:220-326) ---- */
int ppals_tensor_create(ppals_ctx *ctx, int order, const int64_t *global_lens, int dtype,
                        ppals_tensor **out);
void ppals_tensor_destroy(ppals_tensor *t);
int ppals_tensor_local_rows(const ppals_tensor *t, int64_t *lo, int64_t *n); /* leading-mode shard */
/* `-tensor r` (test_ALS.cxx:275-286): V = [[W_true]] built on the device (build_V, common.cxx:135) */
int ppals_tensor_fill_cp(ppals_tensor *t, int R, const double *Wtrue_flat);
/* `-tensor r2` (test_ALS.cxx:272): V[e] = lo + (hi-lo)*u01(seed, e), e = global linear index */
int ppals_tensor_fill_uniform(ppals_tensor *t, uint64_t seed, double lo, double hi);
/* `-tensor p` / `p2` (laplacian_tensor, common.cxx:575-642; `p` is the same data folded to order
 * dim/2 with extents size^2, fold_unfold common.cxx:870-880): the tensor must have been created
 * with those lens; ndigits = -dim, s = -size */
int ppals_tensor_fill_laplacian(ppals_tensor *t, int ndigits, int s);
/* `-tensor c` (Gen_collinearity + U(-1,1) noise of relative norm ratio_noise, common.cxx:361-423,
 * test_ALS.cxx:246-264) */
int ppals_tensor_fill_collinear(ppals_tensor *t, int R, double col_min, double col_max,
                                double ratio_noise, uint64_t seed);
/* the factor vectors Gen_collinearity draws (host only; lambda folded into mode 0) */
int ppals_collinear_factors(int order, const int64_t *lens, int R, double col_min, double col_max,
                            uint64_t seed, double *Wflat);
/* host data: the FULL tensor in fp64, first index fastest (the layout read_dense_from_file
 * implies, test_ALS.cxx:289-325); each rank keeps its own leading-mode rows */
int ppals_tensor_upload(ppals_tensor *t, const double *host_full);
/* A tensor may be re-filled / re-uploaded while CP or Tucker sessions created on it are alive:
 * every fill or upload bumps the tensor's generation, and a session rebuilds what it derived from
 * the old contents (its second resident layout; cached tree nodes and PP operators are dropped)
 * the next time it reads the tensor. Not while one of the session's calls is running. */
/* the reverse (the commented-out V.write_dense_to_file, test_ALS.cxx:347): this rank's leading-mode
 * rows, widened to fp64, into their places of the FULL host tensor; other ranks' rows untouched */
int ppals_tensor_download(ppals_tensor *t, double *host_full);
int ppals_tensor_norm(ppals_tensor *t, double *out); /* V.norm2(), test_ALS.cxx:328 */
/* same counter-based generator for host-side factor initialisation (W.fill_random(0,1)) */
void ppals_fill_uniform_host(double *out, int64_t n, uint64_t seed, uint64_t offset, double lo,
                             double hi);

/* ---- kernel-level entry points (per-kernel parity tests; single rank or sharded) ---- */
/* dimension-tree node, e.g. key "ab" -> T[a,b,r] (mttkrp_map_DT, common.cxx:20-133). Sharded:
 * nodes that keep mode 0 return the local rows, others the local PARTIAL sum. out may be NULL
 * to query the element count through *n. */
int ppals_tree_node(ppals_cp *s, const char *key, double *out, int64_t *n);
/* MTTKRP of one mode through the dimension tree (als_CP.cxx:239-284); full s_mode x R result
 * (summed over ranks) */
int ppals_mttkrp(ppals_cp *s, int mode, double *M);
/* PP operator: V contracted with the modes in `contracted` (Build_mttkrp_map, als_CP.cxx:352) */
int ppals_pp_operator(ppals_cp *s, const char *contracted, double *out, int64_t *n);
/* ||V - [[W]]||_F (als_CP.cxx:183-187), streaming, nothing materialised */
int ppals_cp_residual(ppals_cp *s, double *out);
/* S = Hadamard_{j!=mode} W_j^T W_j + lambda I and its inverse as the engine computes them */
int ppals_cp_gram_system(ppals_cp *s, int mode, double lambda, double *S, double *Sinv);

/* ---- CP sessions ---- */
int ppals_cp_create(ppals_ctx *ctx, ppals_tensor *V, int R, ppals_cp **out);
void ppals_cp_destroy(ppals_cp *s);
int ppals_cp_set_factors(ppals_cp *s, const double *Wflat, const double *gradWflat /*may be NULL*/);
int ppals_cp_get_factors(ppals_cp *s, double *Wflat, double *gradWflat /*may be NULL*/);
/* How an exact sweep walks the tensor — the ALS iterates are identical either way.
 * PPALS_SCHEDULE_DT: the two first-level nodes of alsCP_DT (mttkrp_map_DT, common.cxx:20-133), two
 * tensor scans per sweep. PPALS_SCHEDULE_MSDT (default): the multi-sweep tree of the class API
 * (cp_msdt_optimizer.cxx:172-207), N/(N-1) scans per sweep. */
#define PPALS_SCHEDULE_DT 0
#define PPALS_SCHEDULE_MSDT 1
int ppals_cp_set_schedule(ppals_cp *s, int schedule);
int ppals_cp_get_schedule(const ppals_cp *s);
/* Where the multi-sweep schedule's first-level intermediates lie (no counterpart in the reference:
 * CTF places its own buffers). The choice is made ONLINE: the first ~20 visits of a root run the
 * sweep's own scan at a different offset / store kind of the result, timed on the stream; then the
 * root keeps the fastest. No set-up time, the results do not depend on it. One JSON object
 * {"mode": "online"|"off", "setup_s": 0, "roots": [{"root", "layout", "settled", "visits",
 * "offset_mb", "store", "best_ms", "worst_ms"}]} written to buf (NUL-terminated).
 * PPALS_PLACE_TUNE=0 switches the choice off (offset 0, store kind by size). */
int ppals_cp_placement_report(const ppals_cp *s, char *buf, int cap);
/* The operator builds of the PP phases (Build_mttkrp_map + the N full MTTKRPs, als_CP.cxx:678-694)
 * since the last reset: their number and — while timing is on — their duration, the stream
 * synchronised on both sides of each. mode 0: read only; +1: read, then reset and turn timing on;
 * -1: read, then reset and turn it off. (The reference's [dtime] column counts a build inside the
 * interval that ends at the first PP row, als_CP.cxx:667-697.) */
int ppals_cp_pp_build_stats(ppals_cp *s, int mode, int64_t *builds, double *seconds);
/* n exact dimension-tree sweeps (body of alsCP_DT's loop incl. Normalize, als_CP.cxx:215-303),
 * enqueued asynchronously on the engine stream; no print block, no host sync */
int ppals_cp_sweeps_dt(ppals_cp *s, int n, double lambda);
/* sqrt(sum_i ||grad_W[i]||^2) of the last sweep (als_CP.cxx:174-181) */
int ppals_cp_gradnorm(ppals_cp *s, double *out);

typedef struct {
  double tol;        /* absolute: caller passes -tol * ||V|| (test_ALS.cxx:354) */
  double timelimit;  /* seconds */
  int maxiter;
  double lambda;     /* regularisation (-lambda) */
  int resprint;      /* print/CSV period (-resprint) */
  int bench;         /* pp_bench mode: emit [DTtime]/[PPfirst]/[PPsecond] instead of rows */
  double tol_init;   /* PP restart tolerance (-pp_res_tol) */
  double ratio_step; /* PP update magnitude (-magni) */
  const char *csv_path; /* NULL: no CSV; the file is opened, written and closed by the callee */
  int csv_append;    /* bench mode appends to an existing file */
  int verbose;       /* 1: console output identical to the reference's rank-0 cout/printf */
  double update_percentage; /* -pp 2 only (-update_percentage_pp): fraction of modes per sweep */
} ppals_cp_opts;

/* alsCP_DT (als_CP.h:30-32, als_CP.cxx:127-320). Returns 1 if it stopped before maxiter+1
 * (the reference's `true`), 0 if it ran out of iterations, <0 on error. */
int ppals_cp_dt(ppals_cp *s, const ppals_cp_opts *o, int *iters);
/* alsCP_PP (als_CP.h:105-108, als_CP.cxx:1082-1137) */
int ppals_cp_pp(ppals_cp *s, const ppals_cp_opts *o, int *iters);
/* alsCP_PP_partupdate (als_CP.h:117-122, als_CP.cxx:1146-1207): `-pp 2` */
int ppals_cp_pp_partupdate(ppals_cp *s, const ppals_cp_opts *o, int *iters);

/* ---- class API (src/CP.h, src/optimizer/) ----
 * CPD<dtype, Optimizer>::als(tol, timelimit, maxsweep, resprint, Plot_File, bench)
 * (src/CP.h:42-43, src/CP.cxx:100-186) after CPD::Init (= ppals_cp_set_factors; o->lambda is Init's
 * lambda). `optimizer` names the reference class whose step() cadence and fractional sweep counter
 * are reproduced: CPSimpleOptimizer (1 sweep/step, cp_simple_optimizer.cxx:21-56), CPDTOptimizer
 * (0.5, cp_dt_optimizer.cxx:195-237), CPMSDTOptimizer ((N-1)/N, cp_msdt_optimizer.cxx:172-207).
 * o->maxiter is maxsweep; no Normalize is applied (src/CP.cxx:171). *sweeps = final counter.
 * Returns 1 unless sweeps == maxsweep+1 (the reference's bool), <0 on error. */
#define PPALS_OPT_SIMPLE 0
#define PPALS_OPT_DT 1
#define PPALS_OPT_MSDT 2
int ppals_cpd_als(ppals_cp *s, int optimizer, const ppals_cp_opts *o, double *sweeps, int *iters);
/* The low-rank-update optimizers (run.cxx:401-407, `-pp 2` / `-pp 3`): CPDTLROptimizer
 * (cp_dt_lr_optimizer.cxx:170-236, 0.5 sweep/step) and CPMSDTLROptimizer
 * (cp_msdt_lr_optimizer.cxx:163-205, (N-1)/N sweep/step) with update_rank = run.cxx's -updaterank
 * and randomsvd = run.cxx's -randomsvd: 0 = get_rankR_update_cholesky with the full SVD
 * (common.cxx:768-786), 1 = with randomized_svd(X, r, 1) (common.cxx:691-709; its R x r start
 * matrix comes from this library's counter generator — CTF's stream is not reproducible — with a
 * fixed seed, a fresh block of draws per update, restarted by every call). The first contraction of
 * a step is kept per root and updated with update_rank tensor columns when the contracted factor
 * has changed by a low-rank update only (update_cached_tensor). Single GPU, order >= 3. */
#define PPALS_OPT_DT_LR 3
#define PPALS_OPT_MSDT_LR 4
int ppals_cpd_als_lr(ppals_cp *s, int optimizer, int update_rank, int randomsvd,
                     const ppals_cp_opts *o, double *sweeps, int *iters);

/* ---- Tucker sessions (als_Tucker.h) ---- */
int ppals_tucker_create(ppals_ctx *ctx, ppals_tensor *V, const int *ranks, ppals_tucker **out);
void ppals_tucker_destroy(ppals_tucker *s);
int ppals_tucker_set_factors(ppals_tucker *s, const double *Wflat);
/* the `core` argument of alsTucker_DT / alsTucker_PP (als_Tucker.h:46,89; it seeds core_prev):
 * prod(ranks) doubles, first index fastest; NULL: core = V x_i W_i^T of the current factors.
 * A new session's core is zero (pp_bench.cxx:327 hands over a fresh tensor). */
int ppals_tucker_set_core(ppals_tucker *s, const double *core);
int ppals_tucker_get_factors(ppals_tucker *s, double *Wflat, double *core);
/* hosvd (als_Tucker.h:14-15, als_Tucker.cxx:66): overwrites the factors and the core */
int ppals_tucker_hosvd(ppals_tucker *s);
/* TTMc skipping mode `skip` (-1: none) (als_Tucker.cxx:76-110) */
int ppals_tucker_ttmc(ppals_tucker *s, int skip, double *Y, int64_t *n);
int ppals_tucker_sweeps_dt(ppals_tucker *s, int n);
/* alsTucker_DT (als_Tucker.h:46-48, als_Tucker.cxx:240-424) */
int ppals_tucker_dt(ppals_tucker *s, const ppals_cp_opts *o, int *iters);
/* alsTucker_PP (als_Tucker.h:89-91, als_Tucker.cxx:906-962); o->tol_init = -pp_res_tol */
int ppals_tucker_pp(ppals_tucker *s, const ppals_cp_opts *o, int *iters);

#ifdef __cplusplus
}
#endif
#endif
