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
  if (argc > 2) {  // placement experiment: where does the updated vector live relative to V?
    double* wsep; CK(hipMalloc(&wsep, sizeof(double) * (ld + (1 << 22))));
    CK(hipMemset(wsep, 0, sizeof(double) * (ld + (1 << 22))));
    for (int rep = 0; rep < 2; ++rep)
      for (long off : {0L, 512L, 4096L, 65536L, 262144L, 1048576L, 2097152L}) {
        float ms = timeit([&] { k_axpy<double><<<cu * 6, 256>>>(V, ld, j, wsep + off, coef, partial2, 1, nullptr); }, 5);
        printf("k_axpy w separate +%8ld doubles: %.3f ms  %.0f GB/s\n", off, ms, GB * (j + 2) / ms);
      }
    for (int col : {40, 41}) {
      float ms = timeit([&] { k_axpy<double><<<cu * 6, 256>>>(V, ld, j, V + ld * col, coef, partial2, 1, nullptr); }, 5);
      printf("k_axpy w = column %d of V: %.3f ms  %.0f GB/s\n", col, ms, GB * (j + 2) / ms);
    }
    // does the leading dimension matter?  (columns 0..j-1 at stride ld2 < ld inside the same allocation)
    for (long pad : {0L, 64L, 512L, 4096L, 32768L}) {
      const long ld2 = ld - 65536 + pad;
      float ms = timeit([&] { k_axpy<double><<<cu * 6, 256>>>(V, ld2, j, V + ld2 * 40, coef, partial2, 1, nullptr); }, 5);
      float md = timeit([&] { k_dots<double, 10><<<cu * 3, 256>>>(V, ld2, j, V + ld2 * 40, partial, 4096, j, 1, nullptr); }, 5);
      printf("ld = %ld (pad %ld): k_axpy %.3f ms %.0f GB/s | k_dots %.3f ms %.0f GB/s\n", ld2, pad, ms, GB * (j + 2) / ms * ld2 / ld, md, GB * (j + 1) / md * ld2 / ld);
    }
    return 0;
  }
  for (int bpc : {4, 6}) {
    float ms;
    ms = timeit([&] { k_dots<double, 10><<<cu * 3, 256>>>(V, ld, j, w, partial, 4096, j, 1, nullptr); }, 5);
    printf("bpc=%d  k_dots<10> (3/CU)       %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 1) / ms);
    ms = timeit([&] { k_axpy<double><<<cu * bpc, 256>>>(V, ld, j, w, coef, partial2, 1, nullptr); }, 5);
    printf("bpc=%d  k_axpy                  %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 1><<<cu * std::min(bpc, 5), 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0); }, 5);
    printf("bpc=%d  k_axpy_dots_cs<10,1>    %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 2><<<cu * 3, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0); }, 5);
    printf("bpc=%d  k_axpy_dots_cs<10,2>    %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 2, false><<<cu * 3, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0); }, 5);
    printf("bpc=%d  cs<10,2> plain store    %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 2, true, 4><<<cu * 4, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0); }, 5);
    printf("bpc=%d  cs<10,2> minw4 (4/CU)   %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 1, true><<<cu * 4, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0); }, 5);
    printf("bpc=%d  cs<10,1> 4/CU           %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 1, true><<<cu * 5, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0); }, 5);
    printf("bpc=%d  cs<10,1> 5/CU           %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<10, 4, true><<<cu * 2, 256>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0); }, 5);
    printf("bpc=%d  cs<10,4> 2/CU           %.3f ms  %.0f GB/s\n", bpc, ms, GB * (j + 2) / ms);
#ifdef EXTRA
    EXTRA
#endif
  }
  return 0;
}
