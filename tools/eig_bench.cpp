// eig_bench — times the vendor symmetric eigensolvers on one n x n fp64 Gram (Tucker K13):
// rocsolver_dsyevd (divide & conquer) vs rocsolver_dsyevj (Jacobi). usage: eig_bench [n]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 400;
  std::vector<double> Y((size_t)n * n), G((size_t)n * n, 0.0);
  srand(1);
  for (auto &y : Y) y = 0.5 + 0.5 * rand() / (double)RAND_MAX;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += Y[i + (size_t)n * k] * Y[j + (size_t)n * k];
      G[i + (size_t)n * j] = s;
    }
  double *dG, *dA, *dW, *dE, *dres;
  int *dinfo, *dsw;
  hipMalloc(&dG, sizeof(double) * n * n);
  hipMalloc(&dA, sizeof(double) * n * n);
  hipMalloc(&dW, sizeof(double) * n);
  hipMalloc(&dE, sizeof(double) * n);
  hipMalloc(&dres, sizeof(double));
  hipMalloc(&dinfo, sizeof(int));
  hipMalloc(&dsw, sizeof(int));
  hipMemcpy(dG, G.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  rocblas_handle h;
  double t0 = now();
  rocblas_create_handle(&h);
  printf("rocblas_create_handle: %.3f s\n", now() - t0);
  fflush(stdout);
  for (int alg = 0; alg < 2; alg++) {
    for (int rep = 0; rep < 6; rep++) {
      hipMemcpy(dA, dG, sizeof(double) * n * n, hipMemcpyDeviceToDevice);
      hipDeviceSynchronize();
      t0 = now();
      int rc;
      if (alg == 0)
        rc = rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dW, dE, dinfo);
      else
        rc = rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper,
                              n, dA, n, 1e-14, dres, 30, dsw, dW, dinfo);
      hipDeviceSynchronize();
      double dt = now() - t0;
      int info = -1, sw = -1;
      hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost);
      if (alg == 1) hipMemcpy(&sw, dsw, sizeof(int), hipMemcpyDeviceToHost);
      double w[2];
      hipMemcpy(w, dW + n - 2, 2 * sizeof(double), hipMemcpyDeviceToHost);
      printf("%s n=%d rep %d: %.3f ms rc=%d info=%d sweeps=%d top eigenvalues %.6e %.6e\n",
             alg == 0 ? "dsyevd" : "dsyevj", n, rep, dt * 1e3, rc, info, sw, w[1], w[0]);
      fflush(stdout);
    }
  }
  return 0;
}
