// Kernel-level A/B harness for the library's own kernels (includes ks_kernels.hpp directly).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../arnoldimethod.jl_amd/csrc/ks_kernels.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace ksd;
template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char** argv) {
  const long n = 216L * 216 * 216, ld = n;
  const int NC = 42;
  double *V, *coef, *partial, *partial2;
  CK(hipMalloc(&V, sizeof(double) * ld * NC)); CK(hipMalloc(&coef, 8 * 256)); CK(hipMalloc(&partial, 8 * 64 * 4096)); CK(hipMalloc(&partial2, 8 * 4096));
  CK(hipMemset(V, 0, sizeof(double) * ld * NC)); CK(hipMemset(coef, 0, 8 * 256));
  std::vector<double> h(1 << 20);
  for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
  for (long off = 0; off + (long)h.size() <= ld * NC; off += h.size() * 29) CK(hipMemcpy(V + off, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  const int cu = 256;
  double* w = V + ld * 40;
  const int j = argc > 1 ? atoi(argv[1]) : 40;
  const double GB = (double)ld * 8 / 1e6;  // per column, in GB*1e3/ms units
  for (int bpc : {4, 6}) {
    float ms;
    ms = timeit([&] { k_dots<double, 10><<<cu * 3, 256>>>(V, ld, j, w, partial, 4096, j, 1, nullptr); }, 5);
    printf("bpc=%d  k_dots<10> (3/CU)       %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 1) / ms);
    ms = timeit([&] { k_axpy<double><<<cu * bpc, 256>>>(V, ld, j, w, coef, partial2, 1, nullptr); }, 5);
    printf("bpc=%d  k_axpy                  %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 1><<<cu * std::min(bpc, 5), 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr); }, 5);
    printf("bpc=%d  k_axpy_dots_cs<10,1>    %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 2><<<cu * 3, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr); }, 5);
    printf("bpc=%d  k_axpy_dots_cs<10,2>    %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots<10, 1><<<cu * 2, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr); }, 5);
    printf("bpc=%d  k_axpy_dots<10,1>       %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
#ifdef EXTRA
    EXTRA
#endif
  }
  return 0;
}
