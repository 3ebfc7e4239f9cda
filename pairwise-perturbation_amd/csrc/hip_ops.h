// hip_ops.h — factory of the product Ops (HIP kernels, gfx950) and of the RCCL communicator.
#pragma once
#include "ops.h"

namespace ppals {
// throws std::runtime_error when no HIP device is present (there is no CPU fallback)
Ops *make_hip_ops(int device);
// dlopen librocblas / librocsolver (Tucker modes > 64). Registering their code objects costs
// milliseconds BEFORE the HIP runtime is initialised in the process and minutes after (measured:
// 0.013 s vs 253 s, tools/eig_dlopen_probe.cpp), hence this explicit early entry point.
void hip_preload_eigensolver();
// RCCL over xGMI; the library is resolved with dlopen at first use so that a host process that
// already carries librccl.so.1 (PyTorch-ROCm) shares it
void rccl_get_unique_id(void *out128);
Comm *make_rccl_comm(int rank, int nranks, const void *unique_id128, void *hip_stream);
}  // namespace ppals
