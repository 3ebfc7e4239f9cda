/* ppals_oracle.h — C ABI of the CPU checker (TEST INFRASTRUCTURE, see oracle/README.md).
 *
 * Parity status: UNPINNED for the floating-point path (the reference needs the external CTF
 * library and ships no golden vectors); the dimension tree and the `-pp 2` update order are pinned
 * against the reference's own STL-only functions (oracle/_ref/dimtree_ref, oracle/_ref/sortidx_ref ->
 * tests/golden/dimension_tree.json, tests/golden/sort_indexes.json).
 *
 * Conventions (identical to the reference's CTF objects):
 *   - dense tensors are fp64, FIRST INDEX FASTEST: V[i0 + lens[0]*(i1 + lens[1]*(i2 + ...))]
 *   - factor matrix W_i is lens[i] x R, column-major (row index fastest)
 *   - `Wflat` is W_0,W_1,...,W_{N-1} concatenated
 */
#ifndef PPALS_ORACLE_H
#define PPALS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* counter-based uniform generator shared bit-for-bit with the HIP engine (include/ppals.h):
 * out[i] = lo + (hi-lo) * u01(seed, offset+i) */
void ppo_fill_uniform(double *out, int64_t n, uint64_t seed, uint64_t offset, double lo, double hi);

/* Construct_Dimension_Tree (common.cxx:225-270). Writes "key:parent:sibling;" records into buf. */
int ppo_dimension_tree(int N, char *buf, int buflen);

/* build_V (common.cxx:135-197): V = [[W_0,...,W_{N-1}]] */
void ppo_build_V(int N, const int64_t *lens, int R, const double *Wflat, double *V);
/* ||V - [[W]]||_F (als_CP.cxx:183-187) without materialising the model tensor */
double ppo_residual(int N, const int64_t *lens, int R, const double *V, const double *Wflat);

/* MTTKRP for one mode. route 0: KhatriRao_contract (common.cxx:931-997);
 * route 1: dimension-tree route (common.cxx:20-133 + als_CP.cxx:239-284). M is lens[mode] x R. */
void ppo_mttkrp(int N, const int64_t *lens, int R, const double *V, const double *Wflat, int mode,
                int route, double *M);
/* dimension-tree node tensor, e.g. key "ab": T[a,b,r] (common.cxx:20-133). Returns #elements. */
int64_t ppo_tree_node(int N, const int64_t *lens, int R, const double *V, const double *Wflat,
                      const char *key, double *out);
/* PP operator: V contracted with the modes named in `key` (ascending, e.g. "bd" -> T[a,c,r]),
 * built by the recursion of Build_mttkrp_map (als_CP.cxx:352-409). Returns #elements. */
int64_t ppo_pp_operator(int N, const int64_t *lens, int R, const double *V, const double *Wflat,
                        const char *key, double *out);

/* S = Hadamard_{j != mode} (W_j^T W_j) + lambda*I, product order of als_CP.cxx:219-232,288-292 */
void ppo_gram_hadamard(int N, const int64_t *lens, int R, const double *Wflat, int mode,
                       double lambda, double *S);
/* SVD_solve (common.cxx:710-725): W = M * S^{-1} through a full SVD of S, no truncation */
void ppo_svd_solve(int rows, int R, const double *M, const double *S, double *W);
/* Normalize (common.cxx:680-688) in place */
void ppo_normalize(int N, const int64_t *lens, int R, double *Wflat);
/* thin SVD by one-sided Jacobi: A (m x n, col-major, m>=n) = U diag(s) V^T, s descending */
void ppo_svd(int m, int n, const double *A, double *U, double *s, double *Vm);

/* Full drivers. W/gradW are updated in place; CSV rows go to csv_path (NULL: none).
 * Return 1 if stopped before maxiter+1 (reference's `true`), 0 otherwise. *iters = final iter. */
int ppo_als_cp(int N, const int64_t *lens, int R, const double *V, double *Wflat, double *gradWflat,
               double tol, double timelimit, int maxiter, int verbose, int *iters); /* als_CP.cxx:20 */
int ppo_als_cp_dt(int N, const int64_t *lens, int R, const double *V, double *Wflat,
                  double *gradWflat, double tol, double timelimit, int maxiter, double lambda,
                  const char *csv_path, int resprint, int verbose, int *iters); /* als_CP.cxx:127 */
int ppo_als_cp_pp(int N, const int64_t *lens, int R, const double *V, double *Wflat,
                  double *gradWflat, double tol, double tol_init, double timelimit, int maxiter,
                  double lambda, double ratio_step, const char *csv_path, int resprint, int verbose,
                  int *iters); /* als_CP.cxx:1082 */

int ppo_als_cp_pp_partupdate(int N, const int64_t *lens, int R, const double *V, double *Wflat,
                             double *gradWflat, double tol, double tol_init, double timelimit,
                             int maxiter, double lambda, double ratio_step,
                             double update_percentage, const char *csv_path, int resprint,
                             int verbose, int *iters); /* als_CP.cxx:1146 */

/* Class API: CPD<dtype,Optimizer>::als (src/CP.cxx:100-186). kind 0 CPSimpleOptimizer, 1 CPDTOptimizer,
 * 2 CPMSDTOptimizer (src/optimizer/). No Normalize; *sweeps = the fractional sweep counter. */
int ppo_cpd_als(int N, const int64_t *lens, int R, const double *V, double *Wflat,
                double *gradWflat, int kind, double lambda, double tol, double timelimit,
                int maxsweep, int resprint, const char *csv_path, int verbose, double *sweeps_out,
                int *iters_out);

/* The low-rank-update optimizers of the class API, randomsvd = 0: kind 3 CPDTLROptimizer
 * (cp_dt_lr_optimizer.cxx:170-236), kind 4 CPMSDTLROptimizer (cp_msdt_lr_optimizer.cxx:163-205);
 * update_rank = run.cxx's -updaterank. Returns -1 on a bad argument. */
int ppo_cpd_als_lr(int N, const int64_t *lens, int R, const double *V, double *Wflat,
                   double *gradWflat, int kind, int update_rank, int randomsvd, double lambda,
                   double tol, double timelimit, int maxsweep, int resprint, const char *csv_path,
                   int verbose, double *sweeps_out, int *iters_out);

/* Tucker. ranks[N]; Wflat holds lens[i] x ranks[i] matrices; core is prod(ranks) */
void ppo_ttmc(int N, const int64_t *lens, const int *ranks, const double *V, const double *Wflat,
              int skip, double *Y);                                            /* als_Tucker.cxx:76 */
void ppo_hosvd(int N, const int64_t *lens, const int *ranks, const double *V, double *Wflat,
               double *core);                                                  /* als_Tucker.cxx:66 */
int ppo_als_tucker(int N, const int64_t *lens, const int *ranks, const double *V, double *Wflat,
                   double *core, double tol, double timelimit, int maxiter, int verbose,
                   int *iters);                                                /* als_Tucker.cxx:120 */
int ppo_als_tucker_dt(int N, const int64_t *lens, const int *ranks, const double *V, double *Wflat,
                      double *core, double tol, double timelimit, int maxiter, const char *csv_path,
                      int resprint, int verbose, int *iters);                  /* als_Tucker.cxx:240 */

int ppo_als_tucker_pp(int N, const int64_t *lens, const int *ranks, const double *V, double *Wflat,
                      double *core, double tol, double tol_init, double timelimit, int maxiter,
                      const char *csv_path, int resprint, int verbose, int *iters); /* als_Tucker.cxx:906 */

/* The same four drivers with the reference's `bool bench` argument (pp_bench.cxx:299-345 calls them
 * with maxiter = 1, bench = true): no CSV heading, [DTtime] / [PPfirst] / [PPsecond] lines instead
 * of rows (als_CP.cxx:203-209,735-748; als_Tucker.cxx:324-329,799-814) APPENDED to csv_path, the PP
 * phase without its restart test and with iter++ on exit (als_CP.cxx:656-664,829-830). */
int ppo_als_cp_dt_ex(int N, const int64_t *lens, int R, const double *V, double *Wflat,
                     double *gradWflat, double tol, double timelimit, int maxiter, double lambda,
                     const char *csv_path, int resprint, int verbose, int bench, int *iters);
int ppo_als_cp_pp_ex(int N, const int64_t *lens, int R, const double *V, double *Wflat,
                     double *gradWflat, double tol, double tol_init, double timelimit, int maxiter,
                     double lambda, double ratio_step, const char *csv_path, int resprint,
                     int verbose, int bench, int *iters);
int ppo_als_tucker_dt_ex(int N, const int64_t *lens, const int *ranks, const double *V,
                         double *Wflat, double *core, double tol, double timelimit, int maxiter,
                         const char *csv_path, int resprint, int verbose, int bench, int *iters);
int ppo_als_tucker_pp_ex(int N, const int64_t *lens, const int *ranks, const double *V,
                         double *Wflat, double *core, double tol, double tol_init, double timelimit,
                         int maxiter, const char *csv_path, int resprint, int verbose, int bench,
                         int *iters);

/* sort_indexes (als_CP.cxx:835-843): the update order of alsCP_PP_partupdate */
void ppo_sort_indexes(int n, const double *v, int *idx);

int ppo_num_threads(void);
void ppo_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
