// ppals_api.cpp — the C ABI declared in include/ppals.h: thin, exception-free glue between plain
// pointers/sizes and the engine. The backend (device ops + communicator) comes from backend.h:
// libppals.so links the HIP/RCCL backend; there is no other backend in the product.
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <set>
#include <string>

#include "../../include/ppals.h"
#include "backend.h"
#include "preload_policy.h"
#include "engine.h"
#include "tucker.h"

using namespace ppals;

static thread_local std::string g_err;

struct ppals_ctx {
  Ops *ops = nullptr;
  Comm *comm = nullptr;
  SelfComm self;
  Comm &c() { return comm ? *comm : self; }
  // Live children. A caller should destroy sessions, then tensors, then the context; a garbage-
  // collected binding (or an exception on the way out) may not. ppals_ctx_destroy therefore tears
  // down whatever is still alive itself and leaves the orphaned handles DEAD (eng / data null):
  // destroying them later is a no-op, using them an error — never a use-after-free of the device.
  std::set<ppals_cp *> cps;
  std::set<ppals_tucker *> tks;
  std::set<ppals_tensor *> tensors;
};
struct ppals_tensor {
  ppals_ctx *ctx;
  TensorDesc d;
  uint64_t generation = 1;  // bumped by every fill / upload; sessions rebuild what they derived
};
struct ppals_cp {
  ppals_ctx *ctx;
  CpEngine *eng;
};
struct ppals_tucker {
  ppals_ctx *ctx;
  TuckerEngine *eng;
};

#define API_BEGIN try {
#define API_END(code)                     \
  }                                       \
  catch (const ppals::Unsupported &e) {   \
    g_err = e.what();                     \
    return PPALS_ERR_UNSUPPORTED;         \
  }                                       \
  catch (const std::exception &e) {       \
    g_err = e.what();                     \
    return code;                          \
  }                                       \
  catch (...) {                           \
    g_err = "ppals: unknown exception";   \
    return code;                          \
  }

static int fail(int code, const char *msg) {
  g_err = msg;
  return code;
}

extern "C" {

const char *ppals_last_error(void) { return g_err.c_str(); }
const char *ppals_version(void) { return backend_name(); }

int ppals_preload_eigensolver(void) {
  API_BEGIN
  backend_preload_eigensolver();
  return PPALS_OK;
  API_END(PPALS_ERR_UNSUPPORTED)
}
int ppals_ctx_create(ppals_ctx **out, int device) {
  if (!out) return fail(PPALS_ERR_ARG, "ppals_ctx_create: out is NULL");
  *out = nullptr;
  API_BEGIN
  std::unique_ptr<ppals_ctx> c(new ppals_ctx);
  c->ops = backend_make_ops(device);  // throws when no HIP device: there is no CPU fallback
  *out = c.release();
  return PPALS_OK;
  API_END(PPALS_ERR_NO_DEVICE)
}
void ppals_ctx_destroy(ppals_ctx *ctx) {
  if (!ctx) return;
  for (ppals_cp *s : ctx->cps) {
    delete s->eng;
    s->eng = nullptr;
    s->ctx = nullptr;
  }
  for (ppals_tucker *s : ctx->tks) {
    delete s->eng;
    s->eng = nullptr;
    s->ctx = nullptr;
  }
  for (ppals_tensor *t : ctx->tensors) {
    try {
      ctx->ops->free(t->d.data);
    } catch (...) {
    }
    t->d.data = nullptr;
    t->ctx = nullptr;
  }
  delete ctx->comm;
  delete ctx->ops;
  delete ctx;
}
int ppals_get_unique_id(void *out128) {
  API_BEGIN
  backend_unique_id(out128);
  return PPALS_OK;
  API_END(PPALS_ERR_COMM)
}
int ppals_ctx_init_comm(ppals_ctx *ctx, int rank, int nranks, const void *uid) {
  if (!ctx) return fail(PPALS_ERR_ARG, "ctx is NULL");
  API_BEGIN
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(PPALS_ERR_ARG, "bad rank / nranks");
  // one rank needs no communicator; PPALS_FORCE_COMM=1 (tests) still builds a real one
  if (nranks == 1 && !force_comm_path()) return PPALS_OK;
  ctx->comm = backend_make_comm(ctx->ops, rank, nranks, uid);
  return PPALS_OK;
  API_END(PPALS_ERR_COMM)
}
int ppals_ctx_rank(const ppals_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->rank() : 0; }
int ppals_ctx_nranks(const ppals_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->size() : 1; }
int ppals_ctx_sync(ppals_ctx *ctx) {
  API_BEGIN
  ctx->ops->sync();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_profile_enable(ppals_ctx *ctx, int on) {
  API_BEGIN
  ctx->ops->profile_enable(on < 0 ? 0 : on);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_profile_read(ppals_ctx *ctx, int which, int64_t *launches, double *total_ms,
                       double *algo_bytes) {
  if (which < 0 || which > 1) return fail(PPALS_ERR_ARG, "which must be 0 or 1");
  API_BEGIN
  ctx->ops->profile_collect();
  if (launches) *launches = ctx->ops->prof[which].launches;
  if (total_ms) *total_ms = ctx->ops->prof[which].ms;
  if (algo_bytes) *algo_bytes = ctx->ops->prof[which].bytes;
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_profile_reset(ppals_ctx *ctx) {
  API_BEGIN
  ctx->ops->profile_collect();
  ctx->ops->prof[0] = ProfileSlot();
  ctx->ops->prof[1] = ProfileSlot();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}

// ------------------------------------------------------------------ tensor
int ppals_tensor_create(ppals_ctx *ctx, int order, const int64_t *global_lens, int dtype,
                        ppals_tensor **out) {
  if (!ctx || !global_lens || !out) return fail(PPALS_ERR_ARG, "NULL argument");
  if (dtype != PPALS_F32 && dtype != PPALS_F64) return fail(PPALS_ERR_ARG, "bad dtype");
  API_BEGIN
  std::unique_ptr<ppals_tensor> t(new ppals_tensor);
  t->ctx = ctx;
  std::string err;
  if (tensor_create(*ctx->ops, ctx->c(), order, global_lens, dtype, &t->d, &err) != 0) {
    g_err = "ppals_tensor_create: " + err;
    return PPALS_ERR_ARG;
  }
  t->d.generation = &t->generation;
  ctx->tensors.insert(t.get());
  *out = t.release();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
void ppals_tensor_destroy(ppals_tensor *t) {
  if (!t) return;
  if (t->ctx) {
    try {
      t->ctx->ops->free(t->d.data);
    } catch (...) {
    }
    t->ctx->tensors.erase(t);
  }
  delete t;
}
int ppals_tensor_local_rows(const ppals_tensor *t, int64_t *lo, int64_t *n) {
  if (!t || !t->ctx) return fail(PPALS_ERR_ARG, "NULL tensor");
  if (lo) *lo = t->d.row0;
  if (n) *n = t->d.llens[0];
  return PPALS_OK;
}
int ppals_tensor_fill_cp(ppals_tensor *t, int R, const double *Wtrue_flat) {
  if (!t || !t->ctx || !Wtrue_flat || R <= 0) return fail(PPALS_ERR_ARG, "bad argument");
  API_BEGIN
  t->generation++;
  tensor_fill_cp(*t->ctx->ops, t->d, R, Wtrue_flat);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tensor_fill_uniform(ppals_tensor *t, uint64_t seed, double lo, double hi) {
  if (!t || !t->ctx) return fail(PPALS_ERR_ARG, "NULL tensor");
  API_BEGIN
  t->generation++;
  tensor_fill_uniform(*t->ctx->ops, t->d, seed, lo, hi);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tensor_fill_laplacian(ppals_tensor *t, int ndigits, int s) {
  if (!t || !t->ctx || ndigits < 2 || ndigits % 2 || s < 1) return fail(PPALS_ERR_ARG, "bad argument");
  API_BEGIN
  double total = 1, want = 1;
  for (int i = 0; i < t->d.order; i++) total *= (double)t->d.glens[i];
  for (int i = 0; i < ndigits; i++) want *= s;
  if (total != want) return fail(PPALS_ERR_ARG, "tensor extents do not hold size^dim elements");
  t->generation++;
  tensor_fill_laplacian(*t->ctx->ops, t->d, ndigits, s);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tensor_fill_collinear(ppals_tensor *t, int R, double col_min, double col_max,
                                double ratio_noise, uint64_t seed) {
  if (!t || !t->ctx || R <= 0) return fail(PPALS_ERR_ARG, "bad argument (rank must be positive)");
  API_BEGIN
  t->generation++;
  tensor_fill_collinear(*t->ctx->ops, t->ctx->c(), t->d, R, col_min, col_max, ratio_noise, seed);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_collinear_factors(int order, const int64_t *lens, int R, double col_min, double col_max,
                            uint64_t seed, double *Wflat) {
  if (!lens || !Wflat || order < 1 || R <= 0) return fail(PPALS_ERR_ARG, "bad argument");
  API_BEGIN
  collinear_factors(lens, order, R, col_min, col_max, seed, Wflat);
  return PPALS_OK;
  API_END(PPALS_ERR_ARG)
}
int ppals_tensor_upload(ppals_tensor *t, const double *host_full) {
  if (!t || !t->ctx || !host_full) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  t->generation++;
  tensor_upload(*t->ctx->ops, t->d, host_full);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tensor_download(ppals_tensor *t, double *host_full) {
  if (!t || !t->ctx || !host_full) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  tensor_download(*t->ctx->ops, t->d, host_full);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tensor_norm(ppals_tensor *t, double *out) {
  if (!t || !t->ctx || !out) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  *out = tensor_norm(*t->ctx->ops, t->ctx->c(), t->d);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}

static inline uint64_t sm64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
void ppals_fill_uniform_host(double *out, int64_t n, uint64_t seed, uint64_t offset, double lo,
                             double hi) {
  const uint64_t s = sm64(seed);
  for (int64_t i = 0; i < n; i++) {
    uint64_t h = sm64(s ^ (offset + (uint64_t)i));
    out[i] = lo + (hi - lo) * ((double)(h >> 11) * (1.0 / 9007199254740992.0));
  }
}

// ------------------------------------------------------------------ CP
int ppals_cp_create(ppals_ctx *ctx, ppals_tensor *V, int R, ppals_cp **out) {
  if (!ctx || !V || !out) return fail(PPALS_ERR_ARG, "NULL argument");
  if (R <= 0) return fail(PPALS_ERR_ARG, "rank must be positive");
  API_BEGIN
  std::unique_ptr<ppals_cp> s(new ppals_cp);
  s->ctx = ctx;
  s->eng = new CpEngine(*ctx->ops, ctx->c(), V->d, R);
  ctx->cps.insert(s.get());
  *out = s.release();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
void ppals_cp_destroy(ppals_cp *s) {
  if (!s) return;
  delete s->eng;
  if (s->ctx) s->ctx->cps.erase(s);
  delete s;
}
int ppals_cp_set_factors(ppals_cp *s, const double *Wflat, const double *gradWflat) {
  if (!s || !s->eng || !Wflat) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  s->eng->set_factors(Wflat, gradWflat);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_get_factors(ppals_cp *s, double *Wflat, double *gradWflat) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  s->eng->get_factors(Wflat, gradWflat);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_set_schedule(ppals_cp *s, int schedule) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL session");
  if (schedule != PPALS_SCHEDULE_DT && schedule != PPALS_SCHEDULE_MSDT)
    return fail(PPALS_ERR_ARG, "schedule must be PPALS_SCHEDULE_DT or PPALS_SCHEDULE_MSDT");
  API_BEGIN
  s->eng->set_schedule(schedule);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_get_schedule(const ppals_cp *s) { return s && s->eng ? s->eng->schedule() : PPALS_ERR_ARG; }
int ppals_cp_placement_report(const ppals_cp *s, char *buf, int cap) {
  if (!s || !s->eng || !buf || cap <= 0) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  const std::string r = s->eng->placement_report();
  if ((int)r.size() + 1 > cap) return fail(PPALS_ERR_ARG, "buffer too small");
  std::memcpy(buf, r.c_str(), r.size() + 1);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_pp_build_stats(ppals_cp *s, int mode, int64_t *builds, double *seconds) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  s->eng->pp_build_stats(mode, builds, seconds);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_sweeps_dt(ppals_cp *s, int n, double lambda) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  for (int i = 0; i < n; i++) s->eng->sweep_dt(lambda);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_gradnorm(ppals_cp *s, double *out) {
  if (!s || !s->eng || !out) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  *out = s->eng->gradnorm();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_residual(ppals_cp *s, double *out) {
  if (!s || !s->eng || !out) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  *out = s->eng->residual();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tree_node(ppals_cp *s, const char *key, double *out, int64_t *n) {
  if (!s || !s->eng || !key) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  int64_t c = s->eng->tree_node(key, out);
  if (c < 0) return fail(PPALS_ERR_ARG, "ppals_tree_node: not a node of the dimension tree");
  if (n) *n = c;
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_mttkrp(ppals_cp *s, int mode, double *M) {
  if (!s || !s->eng || !M) return fail(PPALS_ERR_ARG, "NULL argument");
  if (mode < 0 || mode >= s->eng->order()) return fail(PPALS_ERR_ARG, "mode out of range");
  API_BEGIN
  s->eng->mttkrp(mode, M);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_pp_operator(ppals_cp *s, const char *contracted, double *out, int64_t *n) {
  if (!s || !s->eng || !contracted) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  int64_t c = s->eng->pp_operator(contracted, out);
  if (c < 0) return fail(PPALS_ERR_ARG, "ppals_pp_operator: bad mode string");
  if (n) *n = c;
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_gram_system(ppals_cp *s, int mode, double lambda, double *S, double *Sinv) {
  if (!s || !s->eng || !S || !Sinv) return fail(PPALS_ERR_ARG, "NULL argument");
  if (mode < 0 || mode >= s->eng->order()) return fail(PPALS_ERR_ARG, "mode out of range");
  API_BEGIN
  s->eng->gram_system(mode, lambda, S, Sinv);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}

static CpOpts to_opts(const ppals_cp_opts *o) {
  CpOpts c;
  c.tol = o->tol;
  c.timelimit = o->timelimit;
  c.maxiter = o->maxiter;
  c.lambda = o->lambda;
  c.resprint = o->resprint > 0 ? o->resprint : 10;
  c.bench = o->bench;
  c.tol_init = o->tol_init;
  c.ratio_step = o->ratio_step;
  if (o->csv_path) c.csv_path = o->csv_path;
  c.csv_append = o->csv_append != 0;
  c.verbose = o->verbose != 0;
  c.update_percentage = o->update_percentage;
  return c;
}
int ppals_cp_dt(ppals_cp *s, const ppals_cp_opts *o, int *iters) {
  if (!s || !s->eng || !o) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  return s->eng->run_dt(to_opts(o), iters);
  API_END(PPALS_ERR_HIP)
}
int ppals_cp_pp(ppals_cp *s, const ppals_cp_opts *o, int *iters) {
  if (!s || !s->eng || !o) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  return s->eng->run_pp(to_opts(o), iters);
  API_END(PPALS_ERR_HIP)
}

int ppals_cp_pp_partupdate(ppals_cp *s, const ppals_cp_opts *o, int *iters) {
  if (!s || !s->eng || !o) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  return s->eng->run_pp_partupdate(to_opts(o), iters);
  API_END(PPALS_ERR_HIP)
}
int ppals_cpd_als(ppals_cp *s, int optimizer, const ppals_cp_opts *o, double *sweeps, int *iters) {
  if (!s || !s->eng || !o) return fail(PPALS_ERR_ARG, "NULL argument");
  if (optimizer < PPALS_OPT_SIMPLE || optimizer > PPALS_OPT_MSDT)
    return fail(PPALS_ERR_ARG, "optimizer must be PPALS_OPT_SIMPLE, PPALS_OPT_DT or PPALS_OPT_MSDT");
  API_BEGIN
  return s->eng->run_class(optimizer, to_opts(o), sweeps, iters);
  API_END(PPALS_ERR_HIP)
}

int ppals_cpd_als_lr(ppals_cp *s, int optimizer, int update_rank, int randomsvd,
                     const ppals_cp_opts *o, double *sweeps, int *iters) {
  if (!s || !s->eng || !o) return fail(PPALS_ERR_ARG, "NULL argument");
  if (optimizer != PPALS_OPT_DT_LR && optimizer != PPALS_OPT_MSDT_LR)
    return fail(PPALS_ERR_ARG, "optimizer must be PPALS_OPT_DT_LR or PPALS_OPT_MSDT_LR");
  if (update_rank < 1 || update_rank > s->eng->rank_r())
    return fail(PPALS_ERR_ARG, "update_rank must be in [1, R]");
  if (randomsvd < 0 || randomsvd > 1) return fail(PPALS_ERR_ARG, "randomsvd must be 0 or 1");
  API_BEGIN
  CpOpts c = to_opts(o);
  c.update_rank = update_rank;
  c.randomsvd = randomsvd;
  return s->eng->run_class(optimizer, c, sweeps, iters);
  API_END(PPALS_ERR_HIP)
}

// ------------------------------------------------------------------ Tucker
int ppals_tucker_create(ppals_ctx *ctx, ppals_tensor *V, const int *ranks, ppals_tucker **out) {
  if (!ctx || !V || !ranks || !out) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  std::unique_ptr<ppals_tucker> s(new ppals_tucker);
  s->ctx = ctx;
  s->eng = new TuckerEngine(*ctx->ops, ctx->c(), V->d, ranks);
  ctx->tks.insert(s.get());
  *out = s.release();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
void ppals_tucker_destroy(ppals_tucker *s) {
  if (!s) return;
  delete s->eng;
  if (s->ctx) s->ctx->tks.erase(s);
  delete s;
}
int ppals_tucker_set_factors(ppals_tucker *s, const double *Wflat) {
  if (!s || !s->eng || !Wflat) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  s->eng->set_factors(Wflat);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tucker_set_core(ppals_tucker *s, const double *core) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  s->eng->set_core(core);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tucker_get_factors(ppals_tucker *s, double *Wflat, double *core) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  s->eng->get_factors(Wflat, core);
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tucker_hosvd(ppals_tucker *s) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  s->eng->hosvd();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tucker_ttmc(ppals_tucker *s, int skip, double *Y, int64_t *n) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  int64_t c = s->eng->ttmc(skip, Y);
  if (n) *n = c;
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tucker_sweeps_dt(ppals_tucker *s, int n) {
  if (!s || !s->eng) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  for (int i = 0; i < n; i++) s->eng->sweep_dt();
  s->eng->settle();
  return PPALS_OK;
  API_END(PPALS_ERR_HIP)
}
int ppals_tucker_dt(ppals_tucker *s, const ppals_cp_opts *o, int *iters) {
  if (!s || !s->eng || !o) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  return s->eng->run_dt(to_opts(o), iters);
  API_END(PPALS_ERR_HIP)
}

int ppals_tucker_pp(ppals_tucker *s, const ppals_cp_opts *o, int *iters) {
  if (!s || !s->eng || !o) return fail(PPALS_ERR_ARG, "NULL argument");
  API_BEGIN
  return s->eng->run_pp(to_opts(o), iters);
  API_END(PPALS_ERR_HIP)
}

}  // extern "C"
