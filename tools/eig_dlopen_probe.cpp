// eig_dlopen_probe — does it matter WHEN librocsolver/librocblas enter the process? Loads them
// with dlopen either before or after the HIP runtime is initialised and times the first dsyevd.
// usage: eig_dlopen_probe pre|post [n]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
  const bool pre = argc > 1 && !strcmp(argv[1], "pre");
  const int n = argc > 2 ? atoi(argv[2]) : 400;
  void *lb = nullptr, *ls = nullptr;
  double t0 = now();
  auto load = [&]() {
    lb = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
    if (!lb) lb = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    ls = dlopen("librocsolver.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!ls) ls = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lb || !ls) {
      fprintf(stderr, "dlopen failed: %s\n", dlerror());
      exit(1);
    }
  };
  if (pre) load();
  printf("[%s] dlopen before HIP init: %.3f s\n", argv[1], now() - t0);
  fflush(stdout);
  double *dA, *dW, *dE;
  int *dinfo;
  t0 = now();
  hipMalloc(&dA, sizeof(double) * n * n);
  hipMalloc(&dW, sizeof(double) * n);
  hipMalloc(&dE, sizeof(double) * n);
  hipMalloc(&dinfo, sizeof(int));
  std::vector<double> G((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) G[i + (size_t)n * i] = 1.0 + i;
  hipMemcpy(dA, G.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  printf("HIP init + alloc: %.3f s\n", now() - t0);
  fflush(stdout);
  if (!pre) {
    t0 = now();
    load();
    printf("dlopen after HIP init: %.3f s\n", now() - t0);
    fflush(stdout);
  }
  auto create = (int (*)(void **))dlsym(lb, "rocblas_create_handle");
  auto dsyevd = (int (*)(void *, int, int, int, double *, int, double *, double *, int *))dlsym(
      ls, "rocsolver_dsyevd");
  void *h = nullptr;
  t0 = now();
  create(&h);
  printf("rocblas_create_handle: %.3f s\n", now() - t0);
  fflush(stdout);
  for (int rep = 0; rep < 3; rep++) {
    t0 = now();
    int rc = dsyevd(h, 211, 121, n, dA, n, dW, dE, dinfo);
    hipDeviceSynchronize();
    printf("dsyevd rep %d: %.3f s rc=%d\n", rep, now() - t0, rc);
    fflush(stdout);
  }
  return 0;
}
