// driver_common.h — flag parsing, echo block, RCCL bootstrap and tensor construction shared by the
// test_ALS and pp_bench drivers (reference: test_ALS.cxx:14-326, pp_bench.cxx:14-274).
#pragma once
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
      fprintf(stderr, "ppals driver: %s failed (%d): %s\n", #call, _rc, ppals_last_error()); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}

// RCCL unique-id exchange through the file system (single node), as a nonce handshake so that no
// rank can ever pick up the file of an earlier launch (same parent process, recycled PIDs, a
// crashed run's leftovers):
//   rank k > 0 publishes  <path>.r<k>  = its own 16-byte nonce (pid, start time in ns) and keeps it
//              in place (re-creating it if rank 0's start-up clean-up removed it);
//   rank 0     removes what earlier launches left under <path>*, waits for the world-1 hello files,
//              then writes <path> = uid (128 B) + every rank's nonce;
//   rank k     accepts <path> only when slot k carries ITS nonce, then removes its hello file;
//   rank 0     removes <path> once the communicator exists (uid_exchange_done: ncclCommInitRank
//              is collective, so every rank has read the file by then).
static std::string uid_path() {
  if (getenv("PPALS_UID_FILE")) return getenv("PPALS_UID_FILE");
  std::string dir = getenv("PPALS_UID_DIR") ? getenv("PPALS_UID_DIR") : "/tmp";
  std::string tag = getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0";
  std::string run = getenv("TORCHELASTIC_RUN_ID") ? getenv("TORCHELASTIC_RUN_ID") : "none";
  for (auto &c : run)
    if (c == '/' || c == ' ') c = '_';
  return dir + "/ppals_uid_" + tag + "_" + run + "_" + std::to_string((long)getppid());
}
static bool write_atomically(const std::string &path, const void *buf, size_t n) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(buf, 1, n, f) == n;
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
    unlink(tmp.c_str());
    return false;
  }
  return true;
}
static bool read_exactly(const std::string &path, void *buf, size_t n) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const size_t got = fread(buf, 1, n, f);
  const bool more = fgetc(f) != EOF;
  fclose(f);
  return got == n && !more;
}
static const int UID_NONCE_BYTES = 16;
static int exchange_uid(int rank, int world, unsigned char *uid) {
  const std::string path = uid_path();
  auto hello = [&](int k) { return path + ".r" + std::to_string(k); };
  const size_t total = PPALS_UNIQUE_ID_BYTES + (size_t)UID_NONCE_BYTES * world;
  std::vector<unsigned char> file(total, 0);
  const int max_tries = 100 * env_int("PPALS_UID_TIMEOUT_S", 120);
  if (rank == 0) {
    unlink(path.c_str());
    for (int k = 1; k < world; k++) unlink(hello(k).c_str());
    if (ppals_get_unique_id(uid) < 0) return -1;
    memcpy(file.data(), uid, PPALS_UNIQUE_ID_BYTES);
    for (int k = 1; k < world; k++) {
      unsigned char *slot = file.data() + PPALS_UNIQUE_ID_BYTES + (size_t)UID_NONCE_BYTES * k;
      int tries = 0;
      while (!read_exactly(hello(k), slot, UID_NONCE_BYTES)) {
        if (++tries > max_tries) return -1;
        usleep(10000);
      }
    }
    return write_atomically(path, file.data(), total) ? 0 : -1;
  }
  unsigned char nonce[UID_NONCE_BYTES];
  const uint64_t pid = (uint64_t)getpid();
  const uint64_t t_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                            std::chrono::system_clock::now().time_since_epoch()).count();
  memcpy(nonce, &pid, 8);
  memcpy(nonce + 8, &t_ns, 8);
  unlink(hello(rank).c_str());
  for (int tries = 0; tries < max_tries; tries++) {
    unsigned char mine[UID_NONCE_BYTES];
    if (!read_exactly(hello(rank), mine, UID_NONCE_BYTES) || memcmp(mine, nonce, UID_NONCE_BYTES))
      write_atomically(hello(rank), nonce, UID_NONCE_BYTES);
    if (read_exactly(path, file.data(), total) &&
        !memcmp(file.data() + PPALS_UNIQUE_ID_BYTES + (size_t)UID_NONCE_BYTES * rank, nonce,
                UID_NONCE_BYTES)) {
      memcpy(uid, file.data(), PPALS_UNIQUE_ID_BYTES);
      unlink(hello(rank).c_str());
      return 0;
    }
    usleep(10000);
  }
  unlink(hello(rank).c_str());
  return -1;
}
static void uid_exchange_done(int rank) {
  if (rank == 0) unlink(uid_path().c_str());
}


struct Args {
  const char *model, *tensor, *filename, *tensorfile;
  int pp, dim, s, R, issparse, maxiter, resprint, prec, device, rank, world;
  double update_percentage_pp, tol, pp_res_tol, lambda_, magni, col_min, col_max, ratio_noise,
      timelimit;
  uint64_t seed;
  // file exchange with a run of the reference made elsewhere (raw little-endian fp64, first index
  // fastest = the layout V.read_dense_from_file / write_dense_to_file use, test_ALS.cxx:302,347)
  const char *dumpV, *dumpW0, *loadW0, *dumpW;
  std::vector<int64_t> lens_override;  // -lens a,b,c,..: extents of the `-tensor o*` file
  std::vector<int> ranks_override;     // -ranks r0,r1,..: Tucker core extents
};

static std::vector<int64_t> parse_int_list(const char *str) {
  std::vector<int64_t> v;
  if (!str) return v;
  const char *p = str;
  while (*p) {
    char *end;
    const long long x = strtoll(p, &end, 10);
    if (end == p) break;
    v.push_back(x);
    p = (*end == ',') ? end + 1 : end;
    if (*end != ',') break;
  }
  return v;
}
// raw fp64 files of the exchange flags
static bool write_doubles(const char *path, const double *p, size_t n) {
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  const bool ok = fwrite(p, sizeof(double), n, f) == n;
  return fclose(f) == 0 && ok;
}
static bool read_doubles(const char *path, double *p, size_t n) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  const bool ok = fread(p, sizeof(double), n, f) == n && fgetc(f) == EOF;
  fclose(f);
  return ok;
}

// test_ALS.cxx:64-196 — same defaults, same silent resets. resprint_default: 10 in test_ALS
// (test_ALS.cxx:133-139), 1 in pp_bench (pp_bench.cxx).
static Args parse_args(int argc, char **argv, int resprint_default) {
  Args a;
  char **b = argv, **e = argv + argc;
  char *o;
  a.model = (o = getCmdOption(b, e, "-model")) ? o : "CP";
  if (a.model[0] != 'C' && a.model[0] != 'T') a.model = "CP";
  a.tensor = (o = getCmdOption(b, e, "-tensor")) ? o : "p";
  a.pp = (o = getCmdOption(b, e, "-pp")) ? atoi(o) : 0;
  if (a.pp < 0 || a.pp > 2) a.pp = 0;
  a.update_percentage_pp = (o = getCmdOption(b, e, "-update_percentage_pp")) ? atof(o) : 1.0;
  if (a.update_percentage_pp < 0 || a.update_percentage_pp > 1) a.update_percentage_pp = 1.0;
  a.dim = (o = getCmdOption(b, e, "-dim")) ? atoi(o) : 8;
  if (a.dim < 0) a.dim = 8;
  a.maxiter = (o = getCmdOption(b, e, "-maxiter")) ? atoi(o) : 5e3;
  if (a.maxiter < 0) a.maxiter = 5e3;
  a.timelimit = (o = getCmdOption(b, e, "-timelimit")) ? atof(o) : 5e3;
  if (a.timelimit < 0) a.timelimit = 5e3;
  a.s = (o = getCmdOption(b, e, "-size")) ? atoi(o) : 10;
  if (a.s < 0) a.s = 10;
  a.R = (o = getCmdOption(b, e, "-rank")) ? atoi(o) : a.s / 2;
  if (a.R < 0 || a.R > a.s) a.R = a.s / 2;
  a.issparse = (o = getCmdOption(b, e, "-issparse")) ? atoi(o) : 0;
  if (a.issparse < 0 || a.issparse > 1) a.issparse = 0;
  a.resprint = (o = getCmdOption(b, e, "-resprint")) ? atoi(o) : resprint_default;
  if (a.resprint < 0) a.resprint = 10;
  a.tol = (o = getCmdOption(b, e, "-tol")) ? atof(o) : 1e-10;
  if (a.tol < 0 || a.tol > 1) a.tol = 1e-10;
  a.pp_res_tol = (o = getCmdOption(b, e, "-pp_res_tol")) ? atof(o) : 1e-2;
  if (a.pp_res_tol < 0 || a.pp_res_tol > 1) a.pp_res_tol = 1e-2;
  a.lambda_ = (o = getCmdOption(b, e, "-lambda")) ? atof(o) : 0.;
  if (a.lambda_ < 0) a.lambda_ = 0.;
  a.magni = (o = getCmdOption(b, e, "-magni")) ? atof(o) : 1.;
  if (a.magni < 0) a.magni = 1.;
  a.filename = (o = getCmdOption(b, e, "-filename")) ? o : "out.csv";
  a.tensorfile = (o = getCmdOption(b, e, "-tensorfile")) ? o : "test";
  a.col_min = (o = getCmdOption(b, e, "-colmin")) ? atof(o) : 0.5;
  a.col_max = (o = getCmdOption(b, e, "-colmax")) ? atof(o) : 0.9;
  a.ratio_noise = (o = getCmdOption(b, e, "-rationoise")) ? atof(o) : 0.01;
  if (a.ratio_noise < 0) a.ratio_noise = 0.01;
  // extra flags (the reference's parser ignores unknown flags)
  // -prec: storage type of the tensor in HBM. Default 64 = the reference's precision, so that a
  // default command line stops where the fp64 reference stops (-tol 1e-10 * ||V|| is far below
  // what fp32 storage can resolve: its gradient norm floors near 1e-5 * ||V||). -prec 32 halves
  // the bytes every sweep streams (the configuration bench.py measures).
  a.prec = (o = getCmdOption(b, e, "-prec")) ? atoi(o) : 64;
  if (a.prec != 32 && a.prec != 64) a.prec = 64;
  a.seed = (o = getCmdOption(b, e, "-seed")) ? strtoull(o, 0, 10) : 0;
  a.dumpV = getCmdOption(b, e, "-dumpV");
  a.dumpW0 = getCmdOption(b, e, "-dumpW0");
  a.loadW0 = getCmdOption(b, e, "-loadW0");
  a.dumpW = getCmdOption(b, e, "-dumpW");
  a.lens_override = parse_int_list(getCmdOption(b, e, "-lens"));
  for (auto r : parse_int_list(getCmdOption(b, e, "-ranks"))) a.ranks_override.push_back((int)r);
  a.rank = env_int("RANK", 0);
  a.world = env_int("WORLD_SIZE", 1);
  a.device = (o = getCmdOption(b, e, "-device")) ? atoi(o) : env_int("LOCAL_RANK", 0);
  return a;
}

// test_ALS.cxx:203-217
static void echo_args(const Args &a, bool with_files) {
  cout << "  model=  " << a.model << "  tensor=  " << a.tensor << "  pp=  " << a.pp << endl;
  cout << "  dim=  " << a.dim << "  size=  " << a.s << "  rank=  " << a.R << endl;
  cout << "  issparse=  " << a.issparse << "  tolerance=  " << a.tol << "  restarttol=  "
       << a.pp_res_tol << endl;
  cout << "  lambda=  " << a.lambda_ << "  magnitude=  " << a.magni << "  filename=  "
       << a.filename << endl;
  cout << "  col_min=  " << a.col_min << "  col_max=  " << a.col_max << "  rationoise  "
       << a.ratio_noise << endl;
  cout << "  timelimit=  " << a.timelimit << "  maxiter=  " << a.maxiter << "  resprint=  "
       << a.resprint << endl;
  if (with_files)
    cout << "  tensorfile=  " << a.tensorfile
         << "  update_percentage_pp=  " << a.update_percentage_pp << endl;
}

static void init_factors_flat(const std::vector<int64_t> &lens, int R, uint64_t sd,
                              std::vector<double> &flat) {
  size_t tot = 0;
  for (auto l : lens) tot += (size_t)l * R;
  flat.resize(tot);
  double *p = flat.data();
  for (size_t i = 0; i < lens.size(); i++) {
    ppals_fill_uniform_host(p, lens[i] * R, sd + i, 0, 0.0, 1.0);
    p += lens[i] * R;
  }
}

// context (+ RCCL bootstrap) and tensor: test_ALS.cxx:220-326. r2_lo/r2_hi: (0.5,1) in test_ALS
// (test_ALS.cxx:272), (-1,1) in pp_bench. Returns 0 on success.
static int make_ctx_and_tensor(Args &a, double r2_lo, double r2_hi, ppals_ctx **ctx_out,
                               ppals_tensor **V_out, std::vector<int64_t> &lens) {
  const bool folded_p = a.tensor[0] == 'p' && !(strlen(a.tensor) > 1 && a.tensor[1] == '2');
  if (a.dim < 2 || (!folded_p && a.dim > PPALS_MAX_ORDER) || (folded_p && a.dim > 2 * PPALS_MAX_ORDER)) {
    fprintf(stderr, "-dim must be in [2,%d] for this engine\n", PPALS_MAX_ORDER);
    return 2;
  }
  if (a.issparse) {
    fprintf(stderr, "-issparse 1 is not supported (dense engine)\n");
    return 2;
  }
  ppals_ctx *ctx = nullptr;
  CHECK(ppals_ctx_create(&ctx, a.device));
  if (a.world > 1) {
    unsigned char uid[PPALS_UNIQUE_ID_BYTES];
    if (exchange_uid(a.rank, a.world, uid) != 0) {
      fprintf(stderr, "RCCL unique-id exchange failed\n");
      return 1;
    }
    CHECK(ppals_ctx_init_comm(ctx, a.rank, a.world, uid));
    uid_exchange_done(a.rank);
  }
  lens.assign(a.dim, (int64_t)a.s);
  const int ndigits = a.dim;  // -tensor p / p2: number of base-`size` digits of an element index
  if (a.tensor[0] == 'p' && !(strlen(a.tensor) > 1 && a.tensor[1] == '2')) {
    // p: Poisson operator of order `dim`, folded to order dim/2 with extents size^2
    // (test_ALS.cxx:231-245, fold_unfold common.cxx:870-880)
    if (a.dim % 2) {
      fprintf(stderr, "-tensor p needs an even -dim\n");
      return 2;
    }
    a.dim = a.dim / 2;
    lens.assign(a.dim, (int64_t)a.s * a.s);
  }
  const int dtype = a.prec == 64 ? PPALS_F64 : PPALS_F32;
  if (dtype == PPALS_F32 && a.tol < 1e-6 && a.rank == 0)
    fprintf(stderr,
            "ppals: -prec 32 stores the tensor in fp32: [gradnorm] floors near 1e-5*||V||, so "
            "-tol %g cannot be met; the run ends at -maxiter / -timelimit (use -prec 64 for the "
            "reference's stop criterion)\n", a.tol);
  const char *tensor = a.tensor;
  bool from_file = false;
  if (tensor[0] == 'o') {  // raw fp64, first index fastest (test_ALS.cxx:287-326)
    // -lens a,b,c,d (extra flag): the same code path on a file of other extents, named by
    // -tensorfile — a dump of this driver (-dumpV) or of a CTF run, a down-scaled dataset
    const bool custom = !a.lens_override.empty();
    if (strlen(tensor) > 1 && tensor[1] == '1') {
      if (!custom) a.tensorfile = "coil-100.bin";
      lens = {3, 128, 128, 7200};
    } else if (strlen(tensor) > 1 && tensor[1] == '2') {
      if (!custom) a.tensorfile = "time-lapse.bin";
      lens = {33, 1344, 1024, 9};
    }
    if (custom) lens = a.lens_override;
    for (auto l : lens)
      if (l <= 0) {
        fprintf(stderr, "-lens: extents must be positive\n");
        return 2;
      }
    a.dim = (int)lens.size();
    from_file = true;
  }
  ppals_tensor *V = nullptr;
  CHECK(ppals_tensor_create(ctx, a.dim, lens.data(), dtype, &V));
  if (from_file) {
    if (a.rank == 0) cout << "Read the tensor from file " << a.tensorfile << " ...... " << endl;
    size_t n = 1;
    for (auto l : lens) n *= (size_t)l;
    std::vector<double> host(n);
    if (!read_doubles(a.tensorfile, host.data(), n)) {
      fprintf(stderr, "%s does not hold exactly %zu doubles\n", a.tensorfile, n);
      return 2;
    }
    CHECK(ppals_tensor_upload(V, host.data()));
    if (a.rank == 0) cout << "Read dataset finished " << endl;
  } else if (tensor[0] == 'r' && strlen(tensor) > 1 && tensor[1] == '2') {
    CHECK(ppals_tensor_fill_uniform(V, 7000 + a.seed, r2_lo, r2_hi));
  } else if (tensor[0] == 'r') {
    std::vector<double> Wtrue;  // test_ALS.cxx:279-284
    init_factors_flat(lens, a.R, 1000 + 16 * a.seed, Wtrue);
    CHECK(ppals_tensor_fill_cp(V, a.R, Wtrue.data()));
  } else if (tensor[0] == 'p') {
    if (ndigits % 2) {
      fprintf(stderr, "-tensor p2 needs an even -dim\n");
      return 2;
    }
    CHECK(ppals_tensor_fill_laplacian(V, ndigits, a.s));  // test_ALS.cxx:222-245
  } else if (tensor[0] == 'c') {
    CHECK(ppals_tensor_fill_collinear(V, a.R, a.col_min, a.col_max, a.ratio_noise,
                                      5000 + a.seed));  // test_ALS.cxx:246-264
  } else {
    fprintf(stderr, "-tensor %s is not a tensor source of the reference (p, p2, c, r, r2, o1, o2)\n",
            tensor);
    return 2;
  }
  *ctx_out = ctx;
  *V_out = V;
  return 0;
}
