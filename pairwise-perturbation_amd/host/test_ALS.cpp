// test_ALS.cpp — drop-in replacement of the reference's test_ALS driver (test_ALS.cxx:22-415) on
// top of the C ABI (include/ppals.h): same -flag value parsing, defaults and silent re-defaulting
// (test_ALS.cxx:64-196), same echo block (:203-217), same console lines and CSV, with the tensor
// and every sweep on the GPU. No CTF, no MPI: one process per GPU; with WORLD_SIZE > 1 in the
// environment (torchrun-style RANK / WORLD_SIZE / LOCAL_RANK) the ranks bootstrap RCCL through a
// file under $PPALS_UID_DIR (default /tmp) and shard the tensor along its leading mode.
//
// Extra flags (unknown flags are ignored by the reference's parser, so command lines stay
// compatible): -prec 64|32 (tensor storage in HBM; default 64 = the reference's precision, 32 = the
// fast mode bench.py measures), -seed N (default 0),
// -device N (default LOCAL_RANK); file exchange with a reference run made elsewhere (raw
// little-endian fp64, first index fastest — what V.read_dense_from_file reads, test_ALS.cxx:302):
//   -dumpV f    the tensor (the reference's commented-out V.write_dense_to_file, test_ALS.cxx:347)
//   -dumpW0 f   the factors handed to the ALS routine: W_0..W_{N-1} (s_i x R, column-major), then
//               for CP grad_W_0..grad_W_{N-1}; Tucker: the HOSVD factors
//   -loadW0 f   read that file instead of drawing the initial factors (Tucker: instead of hosvd)
//   -dumpW f    the factors at return: CP W then grad_W, Tucker W then the core
//   -lens a,b,c,d   extents of the `-tensor o1|o2` file named by -tensorfile (default: the
//               datasets' hard-coded extents and file names, test_ALS.cxx:289-325)
//   -ranks r0,r1,..   Tucker core extents (default -rank for every mode; o1/o2: test_ALS.cxx:366-379)
// Not supported: -issparse 1 (the engine is dense).
#include "driver_common.h"

int main(int argc, char **argv) {
  Args a = parse_args(argc, argv, 10);
  double start_time = wtime();
  if (a.rank == 0) echo_args(a, true);
  if (a.resprint == 0) a.resprint = 10;
  // Tucker's eigen-step may need the vendor eigensolver: cheap to load only before the GPU is touched
  if (a.model[0] == 'T' || a.R > 64) CHECK(ppals_preload_eigensolver());

  ppals_ctx *ctx = nullptr;
  ppals_tensor *V = nullptr;
  std::vector<int64_t> lens;
  if (int rc = make_ctx_and_tensor(a, 0.5, 1.0, &ctx, &V, lens)) return rc;

  double Vnorm = 0;
  CHECK(ppals_tensor_norm(V, &Vnorm));
  if (a.rank == 0) cout << "Vnorm= " << Vnorm << endl;
  if (a.dumpV) {
    if (a.world > 1) {
      fprintf(stderr, "-dumpV needs a single-rank run (every rank holds only its own rows)\n");
      return 2;
    }
    size_t n = 1;
    for (auto l : lens) n *= (size_t)l;
    std::vector<double> host(n);
    CHECK(ppals_tensor_download(V, host.data()));
    if (!write_doubles(a.dumpV, host.data(), n)) {
      fprintf(stderr, "cannot write %s\n", a.dumpV);
      return 2;
    }
  }

  ppals_cp_opts opt;
  memset(&opt, 0, sizeof(opt));
  opt.tol = a.tol * Vnorm;  // test_ALS.cxx:354
  opt.timelimit = a.timelimit;
  opt.maxiter = a.maxiter;
  opt.lambda = a.lambda_;
  opt.resprint = a.resprint;
  opt.bench = 0;
  opt.tol_init = a.pp_res_tol;
  opt.ratio_step = a.magni;
  opt.csv_path = a.filename;
  opt.verbose = 1;
  opt.update_percentage = a.update_percentage_pp;
  int iters = 0;

  if (a.model[0] == 'C') {
    // W[i], grad_W[i] ~ U(0,1): the interleaved CTF draws of test_ALS.cxx:334-339 are replaced by
    // the counter-based generator (CTF's stream is not reproducible)
    std::vector<double> W, G;
    init_factors_flat(lens, a.R, 2000 + 16 * a.seed, W);
    init_factors_flat(lens, a.R, 3000 + 16 * a.seed, G);
    std::vector<double> WG(W);
    WG.insert(WG.end(), G.begin(), G.end());
    if (a.loadW0) {
      if (!read_doubles(a.loadW0, WG.data(), WG.size())) {
        fprintf(stderr, "%s does not hold exactly %zu doubles (W then grad_W)\n", a.loadW0, WG.size());
        return 2;
      }
      std::copy(WG.begin(), WG.begin() + W.size(), W.begin());
      std::copy(WG.begin() + W.size(), WG.end(), G.begin());
    }
    if (a.dumpW0 && a.rank == 0 && !write_doubles(a.dumpW0, WG.data(), WG.size())) {
      fprintf(stderr, "cannot write %s\n", a.dumpW0);
      return 2;
    }
    ppals_cp *cp = nullptr;
    CHECK(ppals_cp_create(ctx, V, a.R, &cp));
    CHECK(ppals_cp_set_factors(cp, W.data(), G.data()));
    if (a.pp == 0) {
      CHECK(ppals_cp_dt(cp, &opt, &iters));
    } else if (a.pp == 1) {
      CHECK(ppals_cp_pp(cp, &opt, &iters));
    } else {
      CHECK(ppals_cp_pp_partupdate(cp, &opt, &iters));  // test_ALS.cxx:359-362
    }
    if (a.dumpW) {
      CHECK(ppals_cp_get_factors(cp, WG.data(), WG.data() + W.size()));
      if (a.rank == 0 && !write_doubles(a.dumpW, WG.data(), WG.size())) {
        fprintf(stderr, "cannot write %s\n", a.dumpW);
        return 2;
      }
    }
    ppals_cp_destroy(cp);
  } else {
    std::vector<int> ranks(a.dim, a.R);
    if (a.tensor[0] == 'o' && strlen(a.tensor) > 1 && a.tensor[1] == '1') ranks = {3, 10, 10, 70};
    if (a.tensor[0] == 'o' && strlen(a.tensor) > 1 && a.tensor[1] == '2') ranks = {10, 100, 100, 5};
    if (!a.ranks_override.empty()) ranks = a.ranks_override;
    if ((int)ranks.size() != a.dim) {
      fprintf(stderr, "-ranks / the dataset's ranks need %d entries\n", a.dim);
      return 2;
    }
    size_t nW = 0, ncore = 1;
    for (int i = 0; i < a.dim; i++) {
      nW += (size_t)lens[i] * ranks[i];
      ncore *= (size_t)ranks[i];
    }
    std::vector<double> Wc(nW + ncore);
    ppals_tucker *tk = nullptr;
    CHECK(ppals_tucker_create(ctx, V, ranks.data(), &tk));
    if (a.loadW0) {
      if (!read_doubles(a.loadW0, Wc.data(), nW)) {
        fprintf(stderr, "%s does not hold exactly %zu doubles\n", a.loadW0, nW);
        return 2;
      }
      CHECK(ppals_tucker_set_factors(tk, Wc.data()));
      CHECK(ppals_tucker_set_core(tk, nullptr));  // what hosvd leaves: core = V x_i W_i^T
    } else {
      CHECK(ppals_tucker_hosvd(tk));  // test_ALS.cxx:388
    }
    if (a.dumpW0) {
      CHECK(ppals_tucker_get_factors(tk, Wc.data(), nullptr));
      if (a.rank == 0 && !write_doubles(a.dumpW0, Wc.data(), nW)) {
        fprintf(stderr, "cannot write %s\n", a.dumpW0);
        return 2;
      }
    }
    if (a.pp == 0) {
      CHECK(ppals_tucker_dt(tk, &opt, &iters));
    } else {
      CHECK(ppals_tucker_pp(tk, &opt, &iters));  // test_ALS.cxx:392-394
    }
    if (a.dumpW) {
      CHECK(ppals_tucker_get_factors(tk, Wc.data(), Wc.data() + nW));
      if (a.rank == 0 && !write_doubles(a.dumpW, Wc.data(), Wc.size())) {
        fprintf(stderr, "cannot write %s\n", a.dumpW);
        return 2;
      }
    }
    ppals_tucker_destroy(tk);
  }

  if (a.rank == 0) printf("experiment took %lf seconds\n", wtime() - start_time);
  ppals_tensor_destroy(V);
  ppals_ctx_destroy(ctx);
  return 0;
}
