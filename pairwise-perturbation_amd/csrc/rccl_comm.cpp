// rccl_comm.cpp — ppals::Comm on RCCL (one process per GPU, collectives enqueued on the engine's
// HIP stream so they order with the kernels and need no host synchronisation).
//
// The reference issues no collectives itself (every one is implicit inside a CTF contraction,
// SURVEY.md §2.1 C1-C3); these three are the explicit replacements:
//   C1 reduce_scatter_sum : partial s x R MTTKRP rows  -> row block of the owning rank
//   C2 allgather          : updated factor-matrix row blocks -> every rank
//   C3 allreduce_sum      : handfuls of fp64 scalars (norms) and R x R partials
#include <dlfcn.h>

#include <cstring>
#include <stdexcept>
#include <string>

#include "hip_ops.h"

#include "roctx_ranges.h"

namespace ppals {
namespace {

typedef struct {
  char internal[128];
} ncclUniqueId_t;
typedef void *ncclComm_h;
enum { kNcclSum = 0, kNcclFloat64 = 8 };

struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId_t *) = nullptr;
  int (*CommInitRank)(ncclComm_h *, int, ncclUniqueId_t, int) = nullptr;
  int (*CommDestroy)(ncclComm_h) = nullptr;
  int (*CommCount)(const ncclComm_h, int *) = nullptr;
  int (*CommUserRank)(const ncclComm_h, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_h, void *) = nullptr;
  int (*ReduceScatter)(const void *, void *, size_t, int, int, ncclComm_h, void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_h, void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};

RcclApi &api() {
  static RcclApi a;
  if (a.lib) return a;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (a.lib) break;
  }
  if (!a.lib) throw std::runtime_error(std::string("ppals: cannot load RCCL: ") + dlerror());
#define SYM(field, name)                                                   \
  *(void **)(&a.field) = dlsym(a.lib, name);                               \
  if (!a.field) throw std::runtime_error("ppals: RCCL symbol missing: " name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(CommUserRank, "ncclCommUserRank");
  SYM(AllReduce, "ncclAllReduce");
  SYM(ReduceScatter, "ncclReduceScatter");
  SYM(AllGather, "ncclAllGather");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return a;
}

void check(int rc, const char *what) {
  if (rc != 0)
    throw std::runtime_error(std::string("ppals: RCCL ") + what + " failed: " +
                             api().GetErrorString(rc));
}

class RcclComm : public Comm {
 public:
  RcclComm(int rank, int nranks, const void *uid, void *stream)
      : rank_(rank), size_(nranks), stream_(stream) {
    ncclUniqueId_t id;
    std::memcpy(id.internal, uid, 128);
    check(api().CommInitRank(&comm_, nranks, id, rank), "ncclCommInitRank");
    // what RCCL itself says about the communicator it made: size() and rank() answer with ITS
    // numbers from here on, and a disagreement with what the caller asked for ends the set-up
    int cnt = -1, ur = -1;
    check(api().CommCount(comm_, &cnt), "ncclCommCount");
    check(api().CommUserRank(comm_, &ur), "ncclCommUserRank");
    if (cnt != nranks || ur != rank) {
      api().CommDestroy(comm_);
      comm_ = nullptr;
      throw std::runtime_error("ppals: RCCL communicator reports " + std::to_string(cnt) + " ranks / rank " +
                               std::to_string(ur) + ", asked for " + std::to_string(nranks) + " / " +
                               std::to_string(rank));
    }
    size_ = cnt;
    rank_ = ur;
  }
  ~RcclComm() override {
    if (comm_) api().CommDestroy(comm_);
  }
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  void allreduce_sum(double *buf, int64_t n) override {
    RoctxRange roctx_("C1/C3 all-reduce");
    check(api().AllReduce(buf, buf, (size_t)n, kNcclFloat64, kNcclSum, comm_, stream_),
          "ncclAllReduce");
  }
  void reduce_scatter_sum(const double *send, double *recv, int64_t recvcount) override {
    RoctxRange roctx_("C1 reduce-scatter");
    check(api().ReduceScatter(send, recv, (size_t)recvcount, kNcclFloat64, kNcclSum, comm_,
                              stream_),
          "ncclReduceScatter");
  }
  void allgather(const double *send, double *recv, int64_t sendcount) override {
    RoctxRange roctx_("C2 all-gather");
    check(api().AllGather(send, recv, (size_t)sendcount, kNcclFloat64, comm_, stream_),
          "ncclAllGather");
  }

 private:
  int rank_, size_;
  void *stream_;
  ncclComm_h comm_ = nullptr;
};

}  // namespace

void rccl_get_unique_id(void *out128) {
  ncclUniqueId_t id;
  check(api().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, id.internal, 128);
}

Comm *make_rccl_comm(int rank, int nranks, const void *uid, void *stream) {
  return new RcclComm(rank, nranks, uid, stream);
}

}  // namespace ppals
