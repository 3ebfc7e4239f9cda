// backend.h — the one place where the C ABI glue obtains its device ops and communicator.
// libppals.so links backend_hip.cpp (HIP kernels + RCCL). tests/hostsim links a host stand-in so
// the engine's control flow can be exercised on a CPU-only box; that library is test
// infrastructure and is never loaded by the product.
#pragma once
#include "ops.h"

namespace ppals {
const char *backend_name();
Ops *backend_make_ops(int device);
void backend_unique_id(void *out128);
Comm *backend_make_comm(Ops *ops, int rank, int nranks, const void *uid128);
// load the vendor eigensolver libraries now (see ppals_preload_eigensolver); throws on failure
void backend_preload_eigensolver();
}  // namespace ppals
