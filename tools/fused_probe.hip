// What limits k_axpy_dots_cs?  The library's own kernels (ks_kernels.hpp) on a 216^3-sized basis with the library's column
// stride rule: k_dots vs the projection kernel with / without its write stream, at several grid sizes and tile shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fused_probe.hip -o /tmp/fused_probe && /tmp/fused_probe [j]
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
  const long n = 216L * 216 * 216;
  long ld = (n + 63) / 64 * 64;
  const long colb = ld * 8, window = 0x20000, target = 0xF800;
  if (!(argc > 2 && atoi(argv[2]) == 0)) ld += (((target - colb % window) % window + window) % window) / 8;
  const int NC = 43;
  double *V, *coef, *partial, *partial2;
  CK(hipMalloc(&V, sizeof(double) * ld * NC)); CK(hipMalloc(&coef, 8 * 256)); CK(hipMalloc(&partial, 8 * 64 * 8192)); CK(hipMalloc(&partial2, 8 * 8192));
  CK(hipMemset(V, 0, sizeof(double) * ld * NC)); CK(hipMemset(coef, 0, 8 * 256));
  std::vector<double> h(1 << 20);
  for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
  for (long off = 0; off + (long)h.size() <= ld * NC; off += h.size() * 29) CK(hipMemcpy(V + off, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  const int cu = 256;
  const int j = argc > 1 ? atoi(argv[1]) : 40;
  double* y = V + ld * 41;   // scratch (the operator's product)
  double* w = V + ld * j;    // column j (output)
  const double GB = (double)n * 8 / 1e6;
  printf("ld = %ld (stride mod 128K = 0x%lx), j = %d\n", ld, (ld * 8) % window, j);
  for (int rep = 0; rep < 2; ++rep) {
    float ms = 0;
    for (int g : {1, 2, 3, 4, 6}) {
      ms = timeit([&] { k_dots<double, 10><<<cu * g, 256>>>(V, ld, j, y, partial, 8192, j, 1, nullptr); }, 5);
      printf("k_dots<10> %d/CU                         %.3f ms  %.0f GB/s\n", g, ms, GB * (j + 1) / ms);
    }
    for (int g : {2}) {
      for (int ps : {0, 1, 2, 3}) {
        ms = timeit([&] { k_axpy_dots_cs<double, 10, 4, 8><<<cu * g, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, ps); }, 5);
        printf("cs<10,U=4,WB=8> %d/CU store=%s   %.3f ms  %.0f GB/s\n", g, ps == 0 ? "nt   " : ps == 1 ? "plain" : ps == 2 ? "none " : "64KiB", ms, GB * (j + 2) / ms);
      }
    }
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 2, 8><<<cu * 3, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 0); }, 5);
    printf("cs<10,U=2,WB=8> 3/CU store=nt      %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 2, 16><<<cu * 4, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 0); }, 5);
    printf("cs<10,U=2,WB=16> 4/CU store=nt     %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 4, 4><<<cu * 2, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 0); }, 5);
    printf("cs<10,U=4,WB=4> 2/CU store=nt      %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 4, 16><<<cu * 1, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 0); }, 5);
    printf("cs<10,U=4,WB=16> 1/CU store=nt     %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 4, 16><<<cu * 2, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 0); }, 5);
    printf("cs<10,U=4,WB=16> 2/CU store=nt     %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 4, 24><<<cu * 1, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 0); }, 5);
    printf("cs<10,U=4,WB=24> 1/CU store=nt     %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 4, 30><<<cu * 1, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 0); }, 5);
    printf("cs<10,U=4,WB=30> 1/CU store=nt     %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
    ms = timeit([&] { k_axpy_dots_cs<double, 10, 4, 24><<<cu * 1, 256>>>(V, ld, j, y, coef, partial, 8192, partial2, nullptr, 0, w, 2); }, 5);
    printf("cs<10,U=4,WB=24> 1/CU store=none   %.3f ms  %.0f GB/s\n", ms, GB * (j + 2) / ms);
  }
  // the restart rotation (41 columns in, r out, in place): tile-bound or byte-bound?
  {
    double* Qd; CK(hipMalloc(&Qd, 8 * 44 * 48)); CK(hipMemset(Qd, 0, 8 * 44 * 48));
    const int c = 41;
    for (int r : {8, 16, 17, 21, 32}) {
      const int ntile = (r + 15) / 16;
      const size_t smem = (size_t)ntile * 16 * (4 * 11 + 1) * 8;
      float ms = timeit([&] { k_rotate_mfma<11, 2><<<cu * 4, 256, smem>>>(V, ld, c, r, Qd, c, 0, -1); }, 5);
      printf("k_rotate_mfma<11,2> 41 -> %2d columns  %.3f ms  %.0f GB/s (on %d columns)\n", r, ms, GB * (c + r) / ms, c + r);
    }
    {
      const int r = 21, ntile = 2;
      const size_t smem = (size_t)ntile * 16 * (4 * 11 + 1) * 8;
      for (int g : {2, 3, 4}) {
        float ms = timeit([&] { k_rotate_mfma<11, 4><<<cu * g, 256, smem>>>(V, ld, c, r, Qd, c, 0, -1); }, 5);
        printf("k_rotate_mfma<11,RT=4> %d/CU 41 -> 21 columns  %.3f ms  %.0f GB/s\n", g, ms, GB * (c + r) / ms);
        ms = timeit([&] { k_rotate_mfma<11, 1><<<cu * g * 2, 256, smem>>>(V, ld, c, r, Qd, c, 0, -1); }, 5);
        printf("k_rotate_mfma<11,RT=1> %d/CU 41 -> 21 columns  %.3f ms  %.0f GB/s\n", g * 2, ms, GB * (c + r) / ms);
      }
    }
  }
  return 0;
}
