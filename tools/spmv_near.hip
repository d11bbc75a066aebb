// k_spmv_stencil2 against an experimental k_spmv_stencil2n (near taps from the neighbouring lanes: five loads per lane instead of
// eight) on the 216^3 Laplacian: time (caches flushed between launches, plain and shifted forms) and bit-identity of the results.
// Outcome (profiles/r05_spmv_near.txt): bit-identical, and SLOWER -- 51-52 us against 43-45 us, 53 against 44 us in a chain of 20:
// the two one-lane loads and the lane exchanges cost more than the three gathers they replace.  Withdrawn; the kernel lives here.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/spmv_near.hip -o tools/_build/spmv_near && tools/_build/spmv_near [m]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../arnoldimethod.jl_amd/csrc/ks_kernels.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace ksd;
namespace ksd {
__device__ __forceinline__ double shfl_up_(double v, int) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double shfl_dn_(double v, int) { return __shfl_down(v, 1, 64); }
__device__ __forceinline__ cd shfl_up_(cd v, int) { return cd{__shfl_up(v.x, 1, 64), __shfl_up(v.y, 1, 64)}; }
__device__ __forceinline__ cd shfl_dn_(cd v, int) { return cd{__shfl_down(v.x, 1, 64), __shfl_down(v.y, 1, 64)}; }

// The same product with the NEAR taps (column - row = -1, 0, +1) taken from the lane's own pair and its neighbours' instead of
// three more 16-byte gathers: lane l holds x[r], x[r+1] (one load); x[r-1] is the previous lane's second element, x[r+2] the
// next lane's first (lane 0 / 63 of a wave fetch theirs with a one-lane load).  Why: the kernel is bound by latency x
// occupancy (profiles/r04_spmv_counters.txt), and what the wave slots hold in flight is mostly REDUNDANT -- seven gathers per
// pair of rows for one pair of new values; with five loads per lane the registers allow more waves and a larger share of the
// bytes in flight is new.  Same products, same order of additions as k_spmv_stencil2: bit-identical y.
// im1 / i0 / ip1: dictionary slots of the deltas -1 / 0 / +1 (the host launches this form only when all three exist).
template <class T, class MT2>
__global__ void __launch_bounds__(kBlock)
    k_spmv_stencil2n(const MT2* __restrict__ mask2, const StencilDict<T> d, int nslots, const T* __restrict__ x,
                     T* __restrict__ y, int64_t n, int ntiles, const DevState* __restrict__ st, int shifted, T theta, double sigma,
                     int im1, int i0, int ip1) {
  if (st && st->breakdown >= 0) return;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int64_t r = 2 * ((int64_t)tile * kBlock + threadIdx.x);  // rows r, r + 1
  if (r >= n) return;
  const bool two = r + 1 < n;
  const int lane = threadIdx.x & 63;
  constexpr int MB = (int)sizeof(MT2) * 4;  // mask bits per row
  const MT2 mm = mask2[r >> 1];
  const uint32_t m0 = (uint32_t)(mm & (MT2)((((uint64_t)1) << MB) - 1)), m1 = (uint32_t)((uint64_t)mm >> MB);
  const int64_t cmax = n - 2;               // last admissible pair start
  // own pair (clamped like every other pair: n >= 2 is the host's condition for this form)
  T o0, o1;
  {
    const int64_t lo = r > cmax ? cmax : r;
    T a, b;
    ld_pair_u(x + lo, a, b);
    o0 = r == lo ? a : b;   // (r == n - 1: the pair starts one row early)
    o1 = b;
  }
  // x[r - 1], x[r + 2]: the neighbours' elements; the wave's first / last lane load theirs (rows outside [0, n) belong to
  // absent slots: never added)
  T up = shfl_up_(o1, lane), dn = shfl_dn_(o0, lane);
  if (lane == 0) up = r > 0 ? x[r - 1] : zero_of(T{});
  if (lane == 63) dn = r + 2 < n ? x[r + 2] : zero_of(T{});
  T s0 = zero_of(T{}), s1 = zero_of(T{});
  constexpr int UN = 8;
#pragma unroll
  for (int k0 = 0; k0 < kStencilSlots; k0 += UN) {
    if (k0 < nslots) {  // uniform
      T xa[UN], xb[UN];
      int sh[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int k = k0 + u;
        sh[u] = 0;
        xa[u] = zero_of(T{});
        xb[u] = zero_of(T{});
        if (k != im1 && k != i0 && k != ip1 && k < nslots) {  // uniform
          const int64_t c = r + d.delta[k];
          const int64_t lo = c < 0 ? 0 : (c > cmax ? cmax : c);
          sh[u] = (int)(c - lo);
          ld_pair_u(x + lo, xa[u], xb[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int k = k0 + u;
        T v0, v1;
        if (k == im1) { v0 = up; v1 = o0; }
        else if (k == i0) { v0 = o0; v1 = o1; }
        else if (k == ip1) { v0 = o1; v1 = dn; }
        else { v0 = sh[u] == 1 ? xb[u] : xa[u]; v1 = sh[u] == -1 ? xa[u] : xb[u]; }
        const T p0 = mul_nc(d.val[k], v0), p1 = mul_nc(d.val[k], v1);
        s0 = ((m0 >> k) & 1u) ? add_(s0, p0) : s0;
        s1 = ((m1 >> k) & 1u) ? add_(s1, p1) : s1;
      }
    }
  }
  if (shifted) {
    s0 = scl(sub_s(s0, mul_(theta, o0)), sigma);
    s1 = scl(sub_s(s1, mul_(theta, two ? o1 : o0)), sigma);
    if (shifted & 2) {
      if (two) { if constexpr (sizeof(T) == 8) st_pack(y + r, make_double2(s0, s1)); else { y[r] = s0; y[r + 1] = s1; } }
      else y[r] = s0;
      return;
    }
  }
  if (two) {
    if constexpr (sizeof(T) == 8) st_pack_nt(y + r, make_double2(s0, s1));
    else { st_elem_nt(y + r, s0); st_elem_nt(y + r + 1, s1); }
  } else {
    st_elem_nt(y + r, s0);
  }
}

}  // namespace ksd
__global__ void k_flush(const double* __restrict__ a, double* __restrict__ out, long n) {
  double s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(a + i);
  if (s == 1.2345) out[0] = s;
}
int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 216;
  const int my = argc > 2 ? atoi(argv[2]) : m, mz = argc > 3 ? atoi(argv[3]) : m;
  const long n = (long)m * my * mz;
  StencilDict<double> d{};
  const long del[7] = {-(long)m * my, -m, -1, 0, 1, m, (long)m * my};
  for (int k = 0; k < 7; ++k) { d.delta[k] = (int)del[k]; d.val[k] = k == 3 ? 6.0 : -1.0; }
  std::vector<uint16_t> mask((n + 1) / 2 + 1, 0);
  for (long r = 0; r < n; ++r) {
    const long xx = r % m, yy = (r / m) % my, zz = r / ((long)m * my);
    unsigned b = 8;
    if (zz > 0) b |= 1; if (yy > 0) b |= 2; if (xx > 0) b |= 4; if (xx < m - 1) b |= 16; if (yy < my - 1) b |= 32; if (zz < mz - 1) b |= 64;
    if (r & 1) mask[r >> 1] |= (uint16_t)(b << 8); else mask[r >> 1] = (uint16_t)b;
  }
  uint16_t* dm; double *xs, *y1, *y2, *big, *out;
  CK(hipMalloc(&dm, mask.size() * 2)); CK(hipMemcpy(dm, mask.data(), mask.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&xs, n * 8)); CK(hipMalloc(&y1, n * 8)); CK(hipMalloc(&y2, n * 8)); CK(hipMalloc(&big, (1L << 28) * 8)); CK(hipMalloc(&out, 64));
  std::vector<double> h(n), r1(n), r2(n);
  for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
  CK(hipMemcpy(xs, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(big, 0, (1L << 28) * 8));
  const int nt = (int)((n + 511) / 512);
  const double MB = (1.0 * n + 16.0 * n) / 1e6;
  for (int shifted : {0, 1, 3}) {
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 8; ++rep)
      for (int which = 0; which < 2; ++which) {
        k_flush<<<2048, 256>>>(big, out, 1L << 28);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        if (which == 0) k_spmv_stencil2<double, uint16_t><<<nt, 256>>>(dm, d, 7, xs, y1, n, nt, nullptr, shifted, 0.37, 0.125);
        else k_spmv_stencil2n<double, uint16_t><<<nt, 256>>>(dm, d, 7, xs, y2, n, nt, nullptr, shifted, 0.37, 0.125, 2, 3, 4);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep) best[which] = ms < best[which] ? ms : best[which];
      }
    CK(hipMemcpy(r1.data(), y1, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), y2, n * 8, hipMemcpyDeviceToHost));
    const bool same = memcmp(r1.data(), r2.data(), n * 8) == 0;
    printf("%d x %d x %d shifted=%d: k_spmv_stencil2 %.1f us (%.0f GB/s)   k_spmv_stencil2n %.1f us (%.0f GB/s)   results %s\n", m, my, mz, shifted,
           best[0] * 1e3, MB / best[0], best[1] * 1e3, MB / best[1], same ? "bit-identical" : "DIFFER");
  }
  // back to back (a chain: the product of one launch is the input of the next, cacheable stores), 20 launches
  for (int which = 0; which < 2; ++which) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_flush<<<2048, 256>>>(big, out, 1L << 28);
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) {
      const double* src = i == 0 ? xs : (i & 1 ? y1 : y2);
      double* dst = (i & 1) ? y2 : y1;
      if (which == 0) k_spmv_stencil2<double, uint16_t><<<nt, 256>>>(dm, d, 7, src, dst, n, nt, nullptr, 3, 0.37, 0.125);
      else k_spmv_stencil2n<double, uint16_t><<<nt, 256>>>(dm, d, 7, src, dst, n, nt, nullptr, 3, 0.37, 0.125, 2, 3, 4);
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("chain of 20 (%s): %.1f us per product\n", which ? "k_spmv_stencil2n" : "k_spmv_stencil2", ms * 1e3 / 20);
  }
  return 0;
}
