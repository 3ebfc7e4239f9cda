// backend_hip.cpp — product backend: HIP kernels (hip_ops.hip) + RCCL (rccl_comm.cpp).
#include "backend.h"
#include "hip_ops.h"

namespace ppals {
const char *backend_name() { return "ppals 0.1 (HIP gfx950 + RCCL)"; }
Ops *backend_make_ops(int device) { return make_hip_ops(device); }
void backend_unique_id(void *out128) { rccl_get_unique_id(out128); }
void backend_preload_eigensolver() { hip_preload_eigensolver(); }
Comm *backend_make_comm(Ops *ops, int rank, int nranks, const void *uid128) {
  ops->bind();  // ncclCommInitRank binds the communicator to the CURRENT device
  return make_rccl_comm(rank, nranks, uid128, ops->stream());
}
}  // namespace ppals
