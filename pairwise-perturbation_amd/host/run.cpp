// run.cpp — drop-in replacement of the reference's `run` driver (run.cxx:47-472): the class-API
// front end. Same flag parsing and defaults as test_ALS plus -updaterank / -randomsvd
// (run.cxx:129-156; -pp is only clamped below, run.cxx:79-85), the same echo block
// (run.cxx:222-240), and the dispatch of run.cxx:387-414 onto ppals_cpd_als:
//   -pp 0  CPD<double, CPDTOptimizer>     -pp 1  CPD<double, CPMSDTOptimizer>
//   -pp 2  CPD<double, CPDTLROptimizer>   -pp 3  CPD<double, CPMSDTLROptimizer>  (-updaterank;
//          -randomsvd 1 draws from CTF's generator in the reference and is refused here)
//   -pp 4  CPD<double, CPSimpleOptimizer>
// -model Tucker does nothing in the reference (commented out, run.cxx:421-455) and nothing here.
// Extra flags as in test_ALS: -prec 32|64, -seed N, -device N.
#include "driver_common.h"

int main(int argc, char **argv) {
  Args a = parse_args(argc, argv, 10);
  char *o;
  a.pp = (o = getCmdOption(argv, argv + argc, "-pp")) ? atoi(o) : 0;
  if (a.pp < 0) a.pp = 0;
  int update_rank = (o = getCmdOption(argv, argv + argc, "-updaterank")) ? atoi(o) : a.s / 2;
  if (update_rank < 0) update_rank = a.s / 2;
  int randomsvd = (o = getCmdOption(argv, argv + argc, "-randomsvd")) ? atoi(o) : 0;
  if (randomsvd < 0 || randomsvd > 1) randomsvd = 0;
  double start_time = wtime();
  if (a.rank == 0) {  // run.cxx:222-240
    cout << "  model=  " << a.model << "  tensor=  " << a.tensor << "  pp=  " << a.pp << endl;
    cout << "  dim=  " << a.dim << "  size=  " << a.s << "  rank=  " << a.R
         << "  updaterank=  " << update_rank << endl;
    cout << "  issparse=  " << a.issparse << "  tolerance=  " << a.tol << "  restarttol=  "
         << a.pp_res_tol << endl;
    cout << "  lambda=  " << a.lambda_ << "  magnitude=  " << a.magni << "  filename=  "
         << a.filename << endl;
    cout << "  col_min=  " << a.col_min << "  col_max=  " << a.col_max << "  rationoise  "
         << a.ratio_noise << endl;
    cout << "  timelimit=  " << a.timelimit << "  maxiter=  " << a.maxiter << "  resprint=  "
         << a.resprint << endl;
    cout << "  tensorfile=  " << a.tensorfile
         << "  update_percentage_pp=  " << a.update_percentage_pp << endl;
    cout << "  randomsvd=  " << randomsvd << endl;
  }
  if (a.resprint == 0) a.resprint = 10;
  if (a.model[0] == 'C' && (a.pp == 2 || a.pp == 3)) {
    if (update_rank < 1 || update_rank > a.R) {
      fprintf(stderr, "-updaterank must be in [1, rank] (got %d, rank %d)\n", update_rank, a.R);
      return 2;
    }
  }

  ppals_ctx *ctx = nullptr;
  ppals_tensor *V = nullptr;
  std::vector<int64_t> lens;
  if (int rc = make_ctx_and_tensor(a, 0.5, 1.0, &ctx, &V, lens)) return rc;

  double Vnorm = 0;
  CHECK(ppals_tensor_norm(V, &Vnorm));
  if (a.rank == 0) cout << "Vnorm= " << Vnorm << endl;

  if (a.model[0] == 'C' && a.pp >= 0 && a.pp <= 4) {
    ppals_cp_opts opt;
    memset(&opt, 0, sizeof(opt));
    opt.tol = a.tol * Vnorm;  // run.cxx:391
    opt.timelimit = a.timelimit;
    opt.maxiter = a.maxiter;  // maxsweep
    opt.lambda = 0.0;         // decom.Init(&V, W): lambda defaults to 0 (src/CP.h:30)
    opt.resprint = a.resprint;
    opt.csv_path = a.filename;
    opt.verbose = 1;
    // W[i] ~ U(0,1) (run.cxx:369-381); grad_W[i] ~ U(0,1) inside CPD::Init (src/CP.cxx:80-84)
    std::vector<double> W, G;
    init_factors_flat(lens, a.R, 2000 + 16 * a.seed, W);
    init_factors_flat(lens, a.R, 3000 + 16 * a.seed, G);
    ppals_cp *cp = nullptr;
    CHECK(ppals_cp_create(ctx, V, a.R, &cp));
    CHECK(ppals_cp_set_factors(cp, W.data(), G.data()));
    double sweeps = 0;
    int iters = 0;
    if (a.pp == 2 || a.pp == 3) {
      CHECK(ppals_cpd_als_lr(cp, a.pp == 2 ? PPALS_OPT_DT_LR : PPALS_OPT_MSDT_LR, update_rank,
                             randomsvd, &opt, &sweeps, &iters));
    } else {
      const int optimizer = a.pp == 0 ? PPALS_OPT_DT : (a.pp == 1 ? PPALS_OPT_MSDT : PPALS_OPT_SIMPLE);
      CHECK(ppals_cpd_als(cp, optimizer, &opt, &sweeps, &iters));
    }
    ppals_cp_destroy(cp);
  }

  if (a.rank == 0) printf("experiment took %lf seconds\n", wtime() - start_time);
  ppals_tensor_destroy(V);
  ppals_ctx_destroy(ctx);
  return 0;
}
