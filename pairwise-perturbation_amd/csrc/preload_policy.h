// preload_policy.h — what the library does when the vendor eigensolver (rocBLAS + rocSOLVER) has to
// be loaded AFTER the HIP runtime is up in the process (include/ppals.h, ppals_preload_eigensolver):
// registering their code objects then takes minutes instead of milliseconds (253 s against 0.013 s,
// tools/eig_dlopen_probe.cpp). The library says so once on stderr before it stalls, and refuses
// (PPALS_ERR_UNSUPPORTED) under PPALS_STRICT_PRELOAD=1. Pure host logic, in a header of its own so that
// a CPU-only test can exercise it (tests/test_tools_cpu.py).
#pragma once
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace ppals {

// thrown for a request the library understands but will not carry out: the C ABI maps it to
// PPALS_ERR_UNSUPPORTED
struct Unsupported : std::runtime_error {
  explicit Unsupported(const std::string &m) : std::runtime_error(m) {}
};

enum LatePreload { kPreloadSilent = 0, kPreloadWarn = 1, kPreloadRefuse = 2 };

inline LatePreload late_preload_policy(bool libs_loaded, bool hip_runtime_up, const char *strict_env) {
  if (libs_loaded || !hip_runtime_up) return kPreloadSilent;
  return (strict_env && std::atoi(strict_env) != 0) ? kPreloadRefuse : kPreloadWarn;
}

inline const char *late_preload_message() {
  return "ppals: loading rocBLAS / rocSOLVER AFTER the HIP runtime is up: registering their code objects "
         "now can stall this process for minutes (253 s measured, against 0.013 s before the runtime "
         "starts). Call ppals_preload_eigensolver() before ppals_ctx_create and before anything else "
         "touches the GPU; PPALS_STRICT_PRELOAD=1 turns this stall into an error (PPALS_ERR_UNSUPPORTED).";
}

}  // namespace ppals
