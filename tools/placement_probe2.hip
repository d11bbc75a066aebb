// K simultaneous allocations of a V-sized buffer in ONE process: is the speed of the streaming kernels a
// property of the allocation (physical placement)?  Build like placement_probe.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ks_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)
using namespace ksd;
int main(int argc, char** argv) {
  const int64_t n = 10077696;
  const int64_t pad = argc > 2 ? atoll(argv[2]) : 0;
  const int64_t ld = n + pad;
  const int j = 40, ncol = 41;
  const size_t vbytes = (size_t)ld * ncol * 8;
  const int K = argc > 1 ? atoi(argv[1]) : 8;
  std::vector<char*> bufs(K);
  const int contig = argc > 3 ? atoi(argv[3]) : 0;  // 1: hipDeviceMallocContiguous for the odd-numbered buffers
  for (int k = 0; k < K; ++k) {
    if (contig && (k & 1)) { hipError_t e = hipExtMallocWithFlags((void**)&bufs[k], vbytes, hipDeviceMallocContiguous); if (e != hipSuccess) { printf("contiguous alloc failed: %s\n", hipGetErrorString(e)); CK(hipMalloc(&bufs[k], vbytes)); } }
    else CK(hipMalloc(&bufs[k], vbytes));
    CK(hipMemset(bufs[k], 0, vbytes));
  }
  double *coef, *partial2, *partial;
  CK(hipMalloc(&coef, 1024)); CK(hipMemset(coef, 0, 1024));
  CK(hipMalloc(&partial2, 8 * 4096)); CK(hipMalloc(&partial, 8 * 4096 * 48));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int round = 0; round < 1; ++round)
    for (int k = 0; k < K; ++k) {
      double* V = reinterpret_cast<double*>(bufs[k]);
      double* w = V + (size_t)j * ld;
      float ms1 = 0, ms2 = 0, ms3 = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, s));
        for (int r = 0; r < 4; ++r) k_axpy<double, 8><<<768, kBlock, 0, s>>>(V, ld, j, w, coef, partial2, 1, nullptr);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms1, a, b));
        CK(hipEventRecord(a, s));
        for (int r = 0; r < 4; ++r) k_axpy_dots_cs<10, 4, true, 1, 8><<<512, kBlock, 0, s>>>(V, ld, j, w, coef, partial, 4096, partial2, nullptr, 0);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms2, a, b));
        CK(hipEventRecord(a, s));
        for (int r = 0; r < 4; ++r) k_dots<double, 10, 1><<<512, kBlock, 0, s>>>(V, ld, j, w, partial, 4096, 40, 1, nullptr);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms3, a, b));
      }
      printf("pad %lld round %d buf %d @%p  axpy %.1f  fused %.1f  dots %.1f us\n", (long long)pad, round, k, (void*)V, ms1 * 250.0, ms2 * 250.0, ms3 * 250.0);
      fflush(stdout);
    }
  return 0;
}
