// test_ALS.cpp — drop-in replacement of the reference's test_ALS driver (test_ALS.cxx:22-415) on
// top of the C ABI (include/ppals.h): same -flag value parsing, defaults and silent re-defaulting
// (test_ALS.cxx:64-196), same echo block (:203-217), same console lines and CSV, with the tensor
// and every sweep on the GPU. No CTF, no MPI: one process per GPU; with WORLD_SIZE > 1 in the
// environment (torchrun-style RANK / WORLD_SIZE / LOCAL_RANK) the ranks bootstrap RCCL through a
// file under $PPALS_UID_DIR (default /tmp) and shard the tensor along its leading mode.
//
// Extra flags (unknown flags are ignored by the reference's parser, so command lines stay
// compatible): -prec 32|64 (tensor storage in HBM, default 32), -seed N (default 0),
// -device N (default LOCAL_RANK).
// Not supported (SURVEY.md §8f "next"): -tensor p/p2/c, -pp 2, -issparse 1, Tucker -pp 1.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/ppals.h"

using namespace std;

static char *getCmdOption(char **begin, char **end, const std::string &option) {
  char **itr = std::find(begin, end, option);
  if (itr != end && ++itr != end) return *itr;
  return 0;
}
static double wtime() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CHECK(call)                                                          \
  do {                                                                       \
    int _rc = (call);                                                        \
    if (_rc < 0) {                                                           \
      fprintf(stderr, "test_ALS: %s failed (%d): %s\n", #call, _rc, ppals_last_error()); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}

// RCCL unique-id exchange through the file system (single node): rank 0 writes, the others poll
static int exchange_uid(int rank, unsigned char *uid) {
  std::string dir = getenv("PPALS_UID_DIR") ? getenv("PPALS_UID_DIR") : "/tmp";
  std::string tag = getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0";
  std::string path = dir + "/ppals_uid_" + tag + "_" + std::to_string((long)getppid());
  if (getenv("PPALS_UID_FILE")) path = getenv("PPALS_UID_FILE");
  if (rank == 0) {
    if (ppals_get_unique_id(uid) < 0) return -1;
    std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return -1;
    fwrite(uid, 1, PPALS_UNIQUE_ID_BYTES, f);
    fclose(f);
    rename(tmp.c_str(), path.c_str());
    return 0;
  }
  for (int tries = 0; tries < 6000; tries++) {
    FILE *f = fopen(path.c_str(), "rb");
    if (f) {
      size_t n = fread(uid, 1, PPALS_UNIQUE_ID_BYTES, f);
      fclose(f);
      if (n == PPALS_UNIQUE_ID_BYTES) return 0;
    }
    usleep(10000);
  }
  return -1;
}

int main(int argc, char **argv) {
  int const in_num = argc;
  char **input_str = argv;
  const char *model, *tensor, *filename, *tensorfile;
  int pp, dim, s, R, issparse, maxiter = 5e3, resprint = 1;
  double update_percentage_pp, tol, pp_res_tol, lambda_, magni, col_min, col_max, ratio_noise;
  double timelimit = 5e3;

  // ---- flag parsing: test_ALS.cxx:64-196 (same defaults, same silent resets) ----
  char *o;
  model = (o = getCmdOption(input_str, input_str + in_num, "-model")) ? o : "CP";
  if (model[0] != 'C' && model[0] != 'T') model = "CP";
  tensor = (o = getCmdOption(input_str, input_str + in_num, "-tensor")) ? o : "p";
  pp = (o = getCmdOption(input_str, input_str + in_num, "-pp")) ? atoi(o) : 0;
  if (pp < 0 || pp > 2) pp = 0;
  update_percentage_pp =
      (o = getCmdOption(input_str, input_str + in_num, "-update_percentage_pp")) ? atof(o) : 1.0;
  if (update_percentage_pp < 0 || update_percentage_pp > 1) update_percentage_pp = 1.0;
  dim = (o = getCmdOption(input_str, input_str + in_num, "-dim")) ? atoi(o) : 8;
  if (dim < 0) dim = 8;
  maxiter = (o = getCmdOption(input_str, input_str + in_num, "-maxiter")) ? atoi(o) : 5e3;
  if (maxiter < 0) maxiter = 5e3;
  timelimit = (o = getCmdOption(input_str, input_str + in_num, "-timelimit")) ? atof(o) : 5e3;
  if (timelimit < 0) timelimit = 5e3;
  s = (o = getCmdOption(input_str, input_str + in_num, "-size")) ? atoi(o) : 10;
  if (s < 0) s = 10;
  R = (o = getCmdOption(input_str, input_str + in_num, "-rank")) ? atoi(o) : s / 2;
  if (R < 0 || R > s) R = s / 2;
  issparse = (o = getCmdOption(input_str, input_str + in_num, "-issparse")) ? atoi(o) : 0;
  if (issparse < 0 || issparse > 1) issparse = 0;
  resprint = (o = getCmdOption(input_str, input_str + in_num, "-resprint")) ? atoi(o) : 10;
  if (resprint < 0) resprint = 10;
  tol = (o = getCmdOption(input_str, input_str + in_num, "-tol")) ? atof(o) : 1e-10;
  if (tol < 0 || tol > 1) tol = 1e-10;
  pp_res_tol = (o = getCmdOption(input_str, input_str + in_num, "-pp_res_tol")) ? atof(o) : 1e-2;
  if (pp_res_tol < 0 || pp_res_tol > 1) pp_res_tol = 1e-2;
  lambda_ = (o = getCmdOption(input_str, input_str + in_num, "-lambda")) ? atof(o) : 0.;
  if (lambda_ < 0) lambda_ = 0.;
  magni = (o = getCmdOption(input_str, input_str + in_num, "-magni")) ? atof(o) : 1.;
  if (magni < 0) magni = 1.;
  filename = (o = getCmdOption(input_str, input_str + in_num, "-filename")) ? o : "out.csv";
  tensorfile = (o = getCmdOption(input_str, input_str + in_num, "-tensorfile")) ? o : "test";
  col_min = (o = getCmdOption(input_str, input_str + in_num, "-colmin")) ? atof(o) : 0.5;
  col_max = (o = getCmdOption(input_str, input_str + in_num, "-colmax")) ? atof(o) : 0.9;
  ratio_noise = (o = getCmdOption(input_str, input_str + in_num, "-rationoise")) ? atof(o) : 0.01;
  if (ratio_noise < 0) ratio_noise = 0.01;
  // extra flags
  int prec = (o = getCmdOption(input_str, input_str + in_num, "-prec")) ? atoi(o) : 32;
  uint64_t seed = (o = getCmdOption(input_str, input_str + in_num, "-seed")) ? strtoull(o, 0, 10) : 0;
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
  int device = (o = getCmdOption(input_str, input_str + in_num, "-device"))
                   ? atoi(o)
                   : env_int("LOCAL_RANK", 0);

  double start_time = wtime();
  if (rank == 0) {  // echo block: test_ALS.cxx:203-217
    cout << "  model=  " << model << "  tensor=  " << tensor << "  pp=  " << pp << endl;
    cout << "  dim=  " << dim << "  size=  " << s << "  rank=  " << R << endl;
    cout << "  issparse=  " << issparse << "  tolerance=  " << tol << "  restarttol=  "
         << pp_res_tol << endl;
    cout << "  lambda=  " << lambda_ << "  magnitude=  " << magni << "  filename=  " << filename
         << endl;
    cout << "  col_min=  " << col_min << "  col_max=  " << col_max << "  rationoise  "
         << ratio_noise << endl;
    cout << "  timelimit=  " << timelimit << "  maxiter=  " << maxiter << "  resprint=  "
         << resprint << endl;
    cout << "  tensorfile=  " << tensorfile << "  update_percentage_pp=  " << update_percentage_pp
         << endl;
  }
  if (resprint == 0) resprint = 10;
  if (dim < 2 || dim > PPALS_MAX_ORDER) {
    fprintf(stderr, "test_ALS: -dim must be in [2,%d] for this engine\n", PPALS_MAX_ORDER);
    return 2;
  }
  if (issparse) {
    fprintf(stderr, "test_ALS: -issparse 1 is not supported (dense engine)\n");
    return 2;
  }

  ppals_ctx *ctx = nullptr;
  CHECK(ppals_ctx_create(&ctx, device));
  if (world > 1) {
    unsigned char uid[PPALS_UNIQUE_ID_BYTES];
    if (exchange_uid(rank, uid) != 0) {
      fprintf(stderr, "test_ALS: RCCL unique-id exchange failed\n");
      return 1;
    }
    CHECK(ppals_ctx_init_comm(ctx, rank, world, uid));
  }

  // ---- tensor: test_ALS.cxx:220-326 ----
  std::vector<int64_t> lens(dim, (int64_t)s);
  const int dtype = prec == 64 ? PPALS_F64 : PPALS_F32;
  ppals_tensor *V = nullptr;
  bool from_file = false;
  if (tensor[0] == 'o') {  // raw fp64, first index fastest (test_ALS.cxx:287-326)
    if (strlen(tensor) > 1 && tensor[1] == '1') {
      tensorfile = "coil-100.bin";
      lens = {3, 128, 128, 7200};
    } else if (strlen(tensor) > 1 && tensor[1] == '2') {
      tensorfile = "time-lapse.bin";
      lens = {33, 1344, 1024, 9};
    }
    dim = (int)lens.size();
    from_file = true;
  }
  CHECK(ppals_tensor_create(ctx, dim, lens.data(), dtype, &V));
  auto init_factors = [&](uint64_t sd, std::vector<double> &flat) {
    size_t tot = 0;
    for (int i = 0; i < dim; i++) tot += (size_t)lens[i] * R;
    flat.resize(tot);
    double *p = flat.data();
    for (int i = 0; i < dim; i++) {
      ppals_fill_uniform_host(p, lens[i] * R, sd + i, 0, 0.0, 1.0);
      p += lens[i] * R;
    }
  };
  if (from_file) {
    if (rank == 0) cout << "Read the tensor from file " << tensorfile << " ...... " << endl;
    size_t n = 1;
    for (auto l : lens) n *= (size_t)l;
    std::vector<double> host(n);
    FILE *f = fopen(tensorfile, "rb");
    if (!f || fread(host.data(), sizeof(double), n, f) != n) {
      fprintf(stderr, "test_ALS: cannot read %zu doubles from %s\n", n, tensorfile);
      return 2;
    }
    fclose(f);
    CHECK(ppals_tensor_upload(V, host.data()));
    if (rank == 0) cout << "Read dataset finished " << endl;
  } else if (tensor[0] == 'r' && strlen(tensor) > 1 && tensor[1] == '2') {
    CHECK(ppals_tensor_fill_uniform(V, 7000 + seed, 0.5, 1.0));  // test_ALS.cxx:272
  } else if (tensor[0] == 'r') {
    std::vector<double> Wtrue;  // test_ALS.cxx:279-284
    init_factors(1000 + 16 * seed, Wtrue);
    if (R > 64) {
      fprintf(stderr, "test_ALS: this build supports -rank <= 64\n");
      return 2;
    }
    CHECK(ppals_tensor_fill_cp(V, R, Wtrue.data()));
  } else {
    fprintf(stderr,
            "test_ALS: -tensor %s is not supported by this engine yet (supported: r, r2, o1, o2)\n",
            tensor);
    return 2;
  }

  double Vnorm = 0;
  CHECK(ppals_tensor_norm(V, &Vnorm));
  if (rank == 0) cout << "Vnorm= " << Vnorm << endl;

  ppals_cp_opts opt;
  memset(&opt, 0, sizeof(opt));
  opt.tol = tol * Vnorm;  // test_ALS.cxx:354
  opt.timelimit = timelimit;
  opt.maxiter = maxiter;
  opt.lambda = lambda_;
  opt.resprint = resprint;
  opt.bench = 0;
  opt.tol_init = pp_res_tol;
  opt.ratio_step = magni;
  opt.csv_path = filename;
  opt.verbose = 1;
  int iters = 0;

  if (model[0] == 'C') {
    // W[i], grad_W[i] ~ U(0,1), interleaved draw order of test_ALS.cxx:334-339 replaced by the
    // counter-based generator (CTF's stream is not reproducible)
    std::vector<double> W, G;
    init_factors(2000 + 16 * seed, W);
    init_factors(3000 + 16 * seed, G);
    ppals_cp *cp = nullptr;
    CHECK(ppals_cp_create(ctx, V, R, &cp));
    CHECK(ppals_cp_set_factors(cp, W.data(), G.data()));
    if (pp == 0) {
      CHECK(ppals_cp_dt(cp, &opt, &iters));
    } else if (pp == 1) {
      CHECK(ppals_cp_pp(cp, &opt, &iters));
    } else {
      fprintf(stderr, "test_ALS: -pp 2 (partial update) is not supported yet\n");
      return 2;
    }
    ppals_cp_destroy(cp);
  } else {
    std::vector<int> ranks(dim, R);
    if (tensor[0] == 'o' && strlen(tensor) > 1 && tensor[1] == '1') ranks = {3, 10, 10, 70};
    if (tensor[0] == 'o' && strlen(tensor) > 1 && tensor[1] == '2') ranks = {10, 100, 100, 5};
    ppals_tucker *tk = nullptr;
    CHECK(ppals_tucker_create(ctx, V, ranks.data(), &tk));
    CHECK(ppals_tucker_hosvd(tk));  // test_ALS.cxx:388
    if (pp == 0) {
      CHECK(ppals_tucker_dt(tk, &opt, &iters));
    } else {
      fprintf(stderr, "test_ALS: Tucker -pp 1 is not supported yet\n");
      return 2;
    }
    ppals_tucker_destroy(tk);
  }

  if (rank == 0) printf("experiment took %lf seconds\n", wtime() - start_time);
  ppals_tensor_destroy(V);
  ppals_ctx_destroy(ctx);
  return 0;
}
