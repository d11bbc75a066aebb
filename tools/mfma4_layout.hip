// Lane layout of v_mfma_f64_4x4x4_4b_f64 (gfx950), found by experiment: for every pair of lanes (la, lb) the kernel runs the
// instruction with A = one-hot at lane la, B = one-hot at lane lb and records which result lanes become non-zero.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma4_layout.hip -o tools/_build/mfma4_layout && tools/_build/mfma4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_probe(double* out) {
  const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[(size_t)blockIdx.x * 64 + lane] = d;
}

int main() {
  double* out;
  CK(hipMalloc(&out, sizeof(double) * 4096 * 64));
  hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  std::vector<double> h(4096 * 64);
  CK(hipMemcpy(h.data(), out, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  // for every A lane: the B lanes it meets and the result lane of each meeting
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d meets:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int ld = 0; ld < 64; ++ld)
        if (h[((size_t)la * 64 + lb) * 64 + ld] != 0.0) printf("  B%02d->D%02d", lb, ld);
    printf("\n");
  }
  // hypothesis check:  A lane = 16 b + 4 k + i ... printed as fitted indices
  // D(b,i,j) = sum_k A(b,i,k) B(b,k,j).  From the meetings: la and lb meet iff same block and same k; ld determines (i, j).
  return 0;
}
