// hip_ops.h — factory of the product Ops (HIP kernels, gfx950) and of the RCCL communicator.
#pragma once
#include "ops.h"

namespace ppals {
// throws std::runtime_error when no HIP device is present (there is no CPU fallback)
Ops *make_hip_ops(int device);
// RCCL over xGMI; the library is resolved with dlopen at first use so that a host process that
// already carries librccl.so.1 (PyTorch-ROCm) shares it
void rccl_get_unique_id(void *out128);
Comm *make_rccl_comm(int rank, int nranks, const void *unique_id128, void *hip_stream);
}  // namespace ppals
