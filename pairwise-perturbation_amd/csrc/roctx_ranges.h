// roctx_ranges.h — named ranges around the hot-path kernels (K1..K14) and collectives (C1..C3) of
// SURVEY.md §2.1 for `rocprofv3 --marker-trace` (the reference's own instrumentation is CTF's
// Timer scopes, common.cxx:136,712,728,741,1010). Off unless PPALS_ROCTX=1: the library is
// resolved with dlopen on first use, a disabled range costs one branch.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace ppals {

struct RoctxApi {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  bool on = false;
  RoctxApi() {
    const char *e = std::getenv("PPALS_ROCTX");
    if (!e || std::atoi(e) == 0) return;
    void *h = nullptr;
    for (const char *name : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1",
                             "librocprofiler-sdk-roctx.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return;
    *(void **)(&push) = dlsym(h, "roctxRangePushA");
    *(void **)(&pop) = dlsym(h, "roctxRangePop");
    on = push && pop;
  }
};
inline RoctxApi &roctx_api() {
  static RoctxApi api;
  return api;
}
// RAII range: `RoctxRange r("K1 scan");`
struct RoctxRange {
  bool live;
  explicit RoctxRange(const char *name) : live(roctx_api().on) {
    if (live) roctx_api().push(name);
  }
  ~RoctxRange() {
    if (live) roctx_api().pop();
  }
};

}  // namespace ppals
