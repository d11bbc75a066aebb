// Does the speed of the streaming update kernels depend on WHERE the basis lives?  One allocation of
// V + slack; the second-pass update kernel (k_axpy<double,8>, j = 40) and the fused kernel run on V placed
// at base + delta for a sweep of deltas.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I arnoldimethod.jl_amd/csrc
//                                                  tools/placement_probe.hip -o tools/_build/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ks_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)
using namespace ksd;

int main(int argc, char** argv) {
  const int64_t n = 10077696, ld = n;  // multiple of 64
  const int j = 40, ncol = 41;
  const size_t vbytes = (size_t)ld * ncol * 8;
  const size_t slack = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
  const size_t step = (size_t)(argc > 2 ? atof(argv[2]) : 64.0) * (1ull << 20);
  const size_t pre = (size_t)(argc > 3 ? atof(argv[3]) : 0.0) * (1ull << 30);
  void* junk = nullptr;
  if (pre) CK(hipMalloc(&junk, pre));
  char* buf = nullptr;
  CK(hipMalloc(&buf, vbytes + slack));
  CK(hipMemset(buf, 0, vbytes + slack));
  double *coef, *partial2, *partial;
  CK(hipMalloc(&coef, 1024)); CK(hipMemset(coef, 0, 1024));
  CK(hipMalloc(&partial2, 8 * 4096)); CK(hipMalloc(&partial, 8 * 4096 * 48));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int nb_axpy = 256 * 3, nb_fused = 256 * 2;
  printf("buf=%p vbytes=%.3f GiB\n", (void*)buf, vbytes / 1073741824.0);
  for (size_t d = 0; d <= slack; d += step) {
    double* V = reinterpret_cast<double*>(buf + d);
    double* w = V + (size_t)j * ld;
    float ms1 = 0, ms2 = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a, s));
      for (int r = 0; r < 4; ++r) k_axpy<double, 8><<<nb_axpy, kBlock, 0, s>>>(V, ld, j, w, coef, partial2, 1, nullptr);
      CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms1, a, b));
      CK(hipEventRecord(a, s));
      for (int r = 0; r < 4; ++r) k_axpy_dots_cs<10, 4, true, 1, 8><<<nb_fused, kBlock, 0, s>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0);
      CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms2, a, b));
    }
    printf("delta %7.1f MiB  V=%p  (mod 1GiB = %4zu MiB)  axpy %.1f us  fused %.1f us\n", d / 1048576.0, (void*)V,
           ((size_t)V % (1ull << 30)) >> 20, ms1 * 250.0, ms2 * 250.0);
    fflush(stdout);
  }
  return 0;
}
