// hipsim_backend.cpp — TEST INFRASTRUCTURE: the product's HIP device ops (hip_ops.hip, the very
// kernels libppals.so ships) paired with a STAGED CALLBACK communicator, so that the sharded code
// paths of engine.cpp / tucker.cpp meet the real kernels at P > 1 on a box with ONE GPU.
//
// P ranks (processes, or threads of one process) share the one device, each with its own
// ppals_ctx, stream and leading-mode shard. A collective is: D2H of the send buffer into pinned
// memory on the engine's stream -> stream synchronise -> the test's callback on host buffers
// (torch.distributed/gloo between processes, a barrier + fixed-order sum between threads:
// tests/hipsim_util.py) -> H2D of the result on the engine's stream. Everything the engine enqueues
// after the collective is ordered behind the H2D exactly as it is ordered behind the RCCL kernel in
// the product (rccl_comm.cpp). What this does NOT exercise is RCCL itself (tests/test_gpu_rccl.py
// runs it on a one-rank communicator) and xGMI.
//
// Never linked into libppals.so; the version string says so and ppals/__init__.py refuses to load
// a library with that string as the product.
#include <hip/hip_runtime.h>

#include <cstring>
#include <stdexcept>
#include <string>

#include "backend.h"
#include "hip_ops.h"

namespace ppals {
namespace {

#define HS_CHECK(x)                                                                              \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess)                                                                        \
      throw std::runtime_error(std::string("ppals hipsim: ") + #x + ": " + hipGetErrorString(e_)); \
  } while (0)

// same three callbacks as tests/hostsim/host_ops.cpp: fp64 host buffers
struct CommCallbacks {
  void (*allreduce)(double *buf, int64_t n);
  void (*reduce_scatter)(const double *send, double *recv, int64_t recvcount);
  void (*allgather)(const double *send, double *recv, int64_t sendcount);
};

class StagedCallbackComm : public Comm {
 public:
  StagedCallbackComm(Ops *ops, int rank, int size, const CommCallbacks &cb)
      : ops_(ops), rank_(rank), size_(size), cb_(cb), st_((hipStream_t)ops->stream()) {}
  ~StagedCallbackComm() override {
    if (send_) hipHostFree(send_);
    if (recv_) hipHostFree(recv_);
  }
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  void allreduce_sum(double *buf, int64_t n) override {
    if (n <= 0) return;
    stage(n, 0);
    down(send_, buf, n);
    cb_.allreduce(send_, n);
    up(buf, send_, n);
  }
  void reduce_scatter_sum(const double *s, double *r, int64_t n) override {
    if (n <= 0) return;
    stage(n * size_, n);
    down(send_, s, n * size_);
    cb_.reduce_scatter(send_, recv_, n);
    up(r, recv_, n);
  }
  void allgather(const double *s, double *r, int64_t n) override {
    if (n <= 0) return;
    stage(n, n * size_);
    down(send_, s, n);
    cb_.allgather(send_, recv_, n);
    up(r, recv_, n * size_);
  }

 private:
  void grow(double *&p, int64_t &cap, int64_t n) {
    if (n <= cap) return;
    // a staging block that an H2D may still read is only replaced behind a stream synchronise
    HS_CHECK(hipStreamSynchronize(st_));
    if (p) HS_CHECK(hipHostFree(p));
    p = nullptr;
    cap = n + n / 2 + 512;
    HS_CHECK(hipHostMalloc((void **)&p, sizeof(double) * cap, hipHostMallocDefault));
  }
  void stage(int64_t nsend, int64_t nrecv) {
    ops_->bind();
    grow(send_, cap_send_, nsend);
    if (nrecv) grow(recv_, cap_recv_, nrecv);
  }
  // device -> pinned, complete when this returns (the previous collective's H2D — which read the
  // same staging blocks — is on the same stream and hence complete as well)
  void down(double *h, const double *d, int64_t n) {
    HS_CHECK(hipMemcpyAsync(h, d, sizeof(double) * n, hipMemcpyDeviceToHost, st_));
    HS_CHECK(hipStreamSynchronize(st_));
  }
  void up(double *d, const double *h, int64_t n) {
    HS_CHECK(hipMemcpyAsync(d, h, sizeof(double) * n, hipMemcpyHostToDevice, st_));
  }
  Ops *ops_;
  int rank_, size_;
  CommCallbacks cb_;
  hipStream_t st_;
  double *send_ = nullptr, *recv_ = nullptr;
  int64_t cap_send_ = 0, cap_recv_ = 0;
};

}  // namespace

const char *backend_name() {
  return "ppals hipsim (TEST INFRASTRUCTURE: HIP gfx950 ops + staged callback communicator)";
}
Ops *backend_make_ops(int device) { return make_hip_ops(device); }
void backend_unique_id(void *out128) { std::memset(out128, 0, 128); }
void backend_preload_eigensolver() { hip_preload_eigensolver(); }
// as in the hostsim library, the "unique id" argument carries the three callback pointers
Comm *backend_make_comm(Ops *ops, int rank, int nranks, const void *uid128) {
  CommCallbacks cb;
  std::memcpy(&cb, uid128, sizeof(cb));
  if (!cb.allreduce || !cb.reduce_scatter || !cb.allgather)
    throw std::runtime_error("ppals hipsim: the unique id must carry three callback pointers");
  return new StagedCallbackComm(ops, rank, nranks, cb);
}

}  // namespace ppals
