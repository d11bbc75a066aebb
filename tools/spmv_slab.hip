// k_spmv_stencil_slab (csrc/ks_spmv_slab.hpp) against the library's k_spmv_stencil2 on an m x my x mz 7-point Laplacian:
// time with the caches flushed between launches (plain / shifted / shifted with cacheable stores), time of a chain of 20
// products, and bit-identity of the results, for several (waves per workgroup, ring slots, workgroup order, segments) shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/spmv_slab.hip -o tools/_build/spmv_slab
//   tools/_build/spmv_slab [m [my [mz]]]                all variants
//   tools/_build/spmv_slab m my mz only V reps           only variant V (0 = k_spmv_stencil2), `reps` launches (for rocprofv3 --pmc)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "spmv_slab_kernel.hpp"
#include "../arnoldimethod.jl_amd/csrc/ks_spmv_march.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace ksd;

__global__ void k_flush(const double* __restrict__ a, double* __restrict__ out, long n) {
  double s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(a + i);
  if (s == 1.2345) out[0] = s;
}

// reference: what the memory system gives a plain copy in the same chain (16-byte loads, cacheable or streaming 16-byte stores)
__global__ void __launch_bounds__(256) k_copy_ref(const double* __restrict__ x, double* __restrict__ y, long n, int nt_store) {
  typedef double v2 __attribute__((ext_vector_type(2)));
  const long np = n / 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < np; i += (long)gridDim.x * 256) {
    const v2 v = *reinterpret_cast<const v2*>(x + 2 * i);
    if (nt_store) __builtin_nontemporal_store(v, reinterpret_cast<v2*>(y + 2 * i));
    else *reinterpret_cast<v2*>(y + 2 * i) = v;
  }
}
struct Problem {
  long n, P; int nz; uint16_t* dm; long nmask; StencilDict<double> d; double* xs;
};
static int g_allodd = 0, g_dbg = 0;
struct Variant { const char* name; int nw, ns, order, wg_per_cu; };

template <int NW, int NS>
void launch_slab(const Problem& pr, const double* x, double* y, int order, int wgs, int shifted, int nseg_override) {
  using G = SlabGeom<NW, NS>;
  static int attr = 0;
  const unsigned odd = slab_odd_mask(pr.d.delta, 7, pr.P);
  auto kern = odd == 0x14u ? &k_spmv_stencil_slab<NW, NS, 0x14u> : (odd == 0x36u ? &k_spmv_stencil_slab<NW, NS, 0x36u> : &k_spmv_stencil_slab<NW, NS, 0xffu>);
  if (g_allodd) kern = &k_spmv_stencil_slab<NW, NS, 0xffu>;
  if (attr < 8) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); ++attr; }
  const int nslab = (pr.nz + NW - 1) / NW;
  int nseg = nseg_override > 0 ? nseg_override : (wgs / nslab > 0 ? wgs / nslab : 1);
  long sl = (pr.P + nseg - 1) / nseg;
  sl = (sl + 3) & ~3L;
  nseg = (int)((pr.P + sl - 1) / sl);
  kern<<<nslab * nseg, NW * 64, G::lds_bytes>>>(pr.dm, pr.nmask, pr.d, 7, x, pr.n, y, pr.n, pr.P, pr.nz, (int)sl, nseg, nslab, order | (g_dbg << 1), nullptr, shifted, 0.37, 0.125);
}

static int g_nseg = 0, g_G = 768, g_nzr = 16, g_nb = 3;
static uint16_t* g_m1 = nullptr;   // all-ones masks (slot ladder)
static StencilDict<double> g_d1{}, g_d3{}, g_d5{}, g_d3f{};
void launch(int var, const Problem& pr, const double* x, double* y, int shifted) {
  const int nt = (int)((pr.n + 511) / 512);
  switch (var) {
    case 0: k_spmv_stencil2<double, uint16_t><<<nt, 256>>>(pr.dm, pr.d, 7, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 1: launch_slab<8, 7>(pr, x, y, 0, 256, shifted, g_nseg); break;
    case 2: launch_slab<8, 7>(pr, x, y, 1, 256, shifted, g_nseg); break;
    case 3: launch_slab<8, 6>(pr, x, y, 0, 256, shifted, g_nseg); break;
    case 4: launch_slab<8, 5>(pr, x, y, 0, 256, shifted, g_nseg); break;
    case 5: launch_slab<8, 4>(pr, x, y, 0, 256, shifted, g_nseg); break;
    case 6: launch_slab<12, 5>(pr, x, y, 0, 256, shifted, g_nseg); break;
    case 7: launch_slab<16, 4>(pr, x, y, 0, 256, shifted, g_nseg); break;
    case 8: launch_slab<4, 6>(pr, x, y, 0, 512, shifted, g_nseg); break;    // 78 KB: two workgroups per CU
    case 9: launch_slab<6, 7>(pr, x, y, 0, 256, shifted, g_nseg); break;
    case 10: launch_slab<4, 4>(pr, x, y, 0, 768, shifted, g_nseg); break;   // 52 KB: three workgroups per CU
    case 11: launch_slab<8, 7>(pr, x, y, 0, 512, shifted, g_nseg); break;   // twice the workgroups (two rounds)
    case 12: k_spmv_stencil_march<7, 3><<<256 * 4, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 13: k_spmv_stencil_march<7, 3><<<256 * 6, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 14: k_spmv_stencil_march<7, 3><<<256 * 8, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 15: k_spmv_stencil_march<7, 3><<<256 * 3, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 16: k_spmv_stencil_march<7, 3><<<256 * 5, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 25: {
      const int ntp = (int)((pr.P + 511) / 512), cmax = (ntp + 7) / 8;
      if (g_nb == 5) {   // 512-thread workgroups: tiles of 1024 in-plane offsets
        const int ntp2 = (int)((pr.P + 1023) / 1024), cmax2 = (ntp2 + 7) / 8;
        k_spmv_stencil_marchz<7, 0x3eu, 0x14u, 3, 0, 6, 3, 512><<<8 * cmax2 * g_nzr, 512>>>(pr.dm, pr.d, x, y, pr.n, g_nzr, nullptr, shifted, 0.37, 0.125);
      } else if (g_nb == 4) k_spmv_stencil_marchz<7, 0x3eu, 0x14u, 3, 0, 6, 4><<<8 * cmax * g_nzr, 256>>>(pr.dm, pr.d, x, y, pr.n, g_nzr, nullptr, shifted, 0.37, 0.125);
      else k_spmv_stencil_marchz<7, 0x3eu, 0x14u, 3, 0, 6><<<8 * cmax * g_nzr, 256>>>(pr.dm, pr.d, x, y, pr.n, g_nzr, nullptr, shifted, 0.37, 0.125);
      break;
    }
    case 24: k_spmv_stencil_marchw<7, 0x3eu, 0x14u, 3><<<g_G, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 23: k_spmv_stencil_march<7, 3, 2, 4><<<g_G, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 19: k_spmv_stencil_march<1, 0><<<g_G, 256>>>(g_m1, g_d1, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 20: k_spmv_stencil_march<3, 1><<<g_G, 256>>>(g_m1, g_d3, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 21: k_spmv_stencil_march<5, 2><<<g_G, 256>>>(g_m1, g_d5, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 22: k_spmv_stencil_march<3, 1><<<g_G, 256>>>(g_m1, g_d3f, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    case 18: k_copy_ref<<<g_G, 256>>>(x, y, pr.n, (shifted & 2) ? 0 : 1); break;
    case 17: k_spmv_stencil_march<7, 3><<<g_G, 256>>>(pr.dm, pr.d, x, y, pr.n, nt, nullptr, shifted, 0.37, 0.125); break;
    default: printf("no variant %d\n", var); exit(1);
  }
}
const char* vname(int var) {
  static const char* names[] = {"k_spmv_stencil2", "slab<8,7> z-order", "slab<8,7> seg-order", "slab<8,6>", "slab<8,5>", "slab<8,4>", "slab<12,5>",
                                "slab<16,4>", "slab<4,6> 2/CU", "slab<6,7>", "slab<4,4> 3/CU", "slab<8,7> 512 wgs", "march 4 wg/CU", "march 6 wg/CU", "march 8 wg/CU", "march 3 wg/CU", "march 5 wg/CU"};
  return names[var];
}
constexpr int kNVar = 17;

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 216;
  const int my = argc > 2 ? atoi(argv[2]) : m, mz = argc > 3 ? atoi(argv[3]) : m;
  const long n = (long)m * my * mz;
  Problem pr{};
  pr.n = n; pr.P = (long)m * my; pr.nz = mz;
  const long del[7] = {-(long)m * my, -m, -1, 0, 1, m, (long)m * my};
  for (int k = 0; k < 7; ++k) { pr.d.delta[k] = (int)del[k]; pr.d.val[k] = k == 3 ? 6.0 : -1.0; }
  if (slab_far_stride(pr.d.delta, 7, n) != pr.P) { printf("the slab form does not take this grid (P = %ld)\n", pr.P); return 1; }
  std::vector<uint16_t> mask((n + 1) / 2 + 1, 0);
  for (long r = 0; r < n; ++r) {
    const long xx = r % m, yy = (r / m) % my, zz = r / ((long)m * my);
    unsigned b = 8;
    if (zz > 0) b |= 1; if (yy > 0) b |= 2; if (xx > 0) b |= 4; if (xx < m - 1) b |= 16; if (yy < my - 1) b |= 32; if (zz < mz - 1) b |= 64;
    if (r & 1) mask[r >> 1] |= (uint16_t)(b << 8); else mask[r >> 1] = (uint16_t)b;
  }
  pr.nmask = (long)mask.size();
  double *y1, *y2, *big, *out;
  CK(hipMalloc(&pr.dm, mask.size() * 2)); CK(hipMemcpy(pr.dm, mask.data(), mask.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&pr.xs, n * 8)); CK(hipMalloc(&y1, n * 8)); CK(hipMalloc(&y2, n * 8)); CK(hipMalloc(&big, (1L << 28) * 8)); CK(hipMalloc(&out, 64));
  std::vector<double> h(n), r1(n), r2(n);
  srand(1);
  for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
  CK(hipMemcpy(pr.xs, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(big, 0, (1L << 28) * 8));
  const double MB = (1.0 * n + 16.0 * n) / 1e6;
  if (argc > 6 && !strcmp(argv[4], "only")) {
    const int var = atoi(argv[5]), reps = atoi(argv[6]);
    if (argc > 7) g_nseg = atoi(argv[7]);
    if (argc > 8) g_dbg = atoi(argv[8]);
    for (int i = 0; i < reps; ++i) { k_flush<<<2048, 256>>>(big, out, 1L << 28); launch(var, pr, pr.xs, y1, 1); }
    CK(hipDeviceSynchronize());
    printf("%s: %d launches\n", vname(var), reps);
    return 0;
  }
  if (argc > 4 && !strcmp(argv[4], "columns")) {
    // the chain the solver runs: product i reads column i and writes column i + 1 of a 21-column basis (1.7 GB at 216^3: nothing
    // stays in the 256-MB Infinity Cache but the column just written), cacheable stores, three passes; per-launch HIP events.
    //   ... columns V0 [S1 S2 ...]: variant V0 (0 = k_spmv_stencil2) and the marching kernel at S1, S2, ... slots per XCD
    // slot ladder (variants 19-22; S + 20000 / 30000 / 40000 / 50000): the marching kernel with 1 slot (delta 0: a scaled copy), 3 slots
    // (-1, 0, +1), 5 slots (+- nx added), 3 slots (-P, 0, +P), all masks set -- where the time of the 7-slot product goes
    { std::vector<uint16_t> ones(mask.size(), 0xffff); CK(hipMalloc(&g_m1, ones.size() * 2)); CK(hipMemcpy(g_m1, ones.data(), ones.size() * 2, hipMemcpyHostToDevice));
      g_d1.delta[0] = 0; g_d1.val[0] = 6.0;
      const long d3[3] = {-1, 0, 1}, d5[5] = {-m, -1, 0, 1, m}, d3f[3] = {-pr.P, 0, pr.P};
      for (int k = 0; k < 3; ++k) { g_d3.delta[k] = (int)d3[k]; g_d3.val[k] = k == 1 ? 6.0 : -1.0; g_d3f.delta[k] = (int)d3f[k]; g_d3f.val[k] = k == 1 ? 6.0 : -1.0; }
      for (int k = 0; k < 5; ++k) { g_d5.delta[k] = (int)d5[k]; g_d5.val[k] = k == 2 ? 6.0 : -1.0; } }
    const int ncol = 21;
    const long ld = (n + 511) / 512 * 512 + 512;
    double* V; CK(hipMalloc(&V, (size_t)ncol * ld * 8));
    CK(hipMemset(V, 0, (size_t)ncol * ld * 8));
    CK(hipMemcpy(V, h.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<hipEvent_t> ev(21);
    for (auto& evt : ev) CK(hipEventCreate(&evt));
    for (int ai = 5; ai < argc; ++ai) {
      int sarg = atoi(argv[ai]);
      int shf = 3;
      if (sarg < 0) { sarg = -sarg; shf = 1; }            // negative: streaming (nt) stores instead of cacheable ones
      int var = sarg == 0 ? 0 : 17;
      if (sarg >= 100000) { g_nzr = sarg - 100000; g_nb = 5; sarg = 1; var = 25; }
      else if (sarg >= 90000) { g_nzr = sarg - 90000; g_nb = 4; sarg = 1; var = 25; }
      else if (sarg >= 80000) { g_nzr = sarg - 80000; g_nb = 3; sarg = 1; var = 25; }
      else if (sarg >= 70000) { sarg -= 70000; var = 24; }
      else if (sarg >= 60000) { sarg -= 60000; var = 23; }
      else if (sarg >= 50000) { sarg -= 50000; var = 22; }
      else if (sarg >= 40000) { sarg -= 40000; var = 21; }
      else if (sarg >= 30000) { sarg -= 30000; var = 20; }
      else if (sarg >= 20000) { sarg -= 20000; var = 19; }
      else if (sarg >= 10000) { sarg -= 10000; shf = 1; }
      if (sarg >= 5000) { sarg -= 5000; var = 18; }       // 5000 + S: the plain copy with 8 S workgroups
      if (sarg && var != 25) g_G = sarg * 8;
      double tot = 0; float mn = 1e9f, mx = 0; std::vector<float> per(20, 0.f);
      for (int pass = 0; pass < 4; ++pass) {
        for (int i = 0; i < 20; ++i) { CK(hipEventRecord(ev[i])); launch(var, pr, V + (size_t)i * ld, V + (size_t)(i + 1) * ld, shf); }
        CK(hipEventRecord(ev[20])); CK(hipEventSynchronize(ev[20]));
        if (!pass) continue;
        for (int i = 0; i < 20; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); tot += ms; mn = ms < mn ? ms : mn; mx = ms > mx ? ms : mx; per[i] += ms / 3; }
      }
      printf("%d x %d x %d  columns chain  %s stores  %-18s S=%-4d  mean %.1f us  min %.1f  max %.1f   per product:", m, my, mz, shf == 3 ? "cacheable" : "streaming", var == 18 ? "plain copy" : var == 19 ? "march 1 slot" : var == 20 ? "march -1 0 +1" : var == 21 ? "march 5 near" : var == 22 ? "march -P 0 +P" : var == 23 ? "march near-by-dpp" : var == 24 ? "march LDS window" : var == 25 ? "z-march (S = z-ranges)" : var ? "march" : "k_spmv_stencil2", var == 25 ? g_nzr : var ? g_G / 8 : 0, tot / 60 * 1e3, mn * 1e3, mx * 1e3);
      for (int i = 0; i < 20; ++i) printf(" %.0f", per[i] * 1e3);
      printf("\n");
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 4 && !strcmp(argv[4], "sweep")) {
    // the marching kernel over the number of workgroups G (S = G / 8 tiles per XCD and round): cold time and chain time
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(0, pr, pr.xs, y1, 3);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r1.data(), y1, n * 8, hipMemcpyDeviceToHost));
    for (int ai = 5; ai < argc; ++ai) {
      int sv = atoi(argv[ai]);
      int mvar = sv < 0 ? 23 : 17;      // negative: the form with the +-1 taps from the neighbouring lanes
      if (sv >= 100000) { g_nzr = sv - 100000; g_nb = 5; sv = 1; mvar = 25; }   // 100000 + z-ranges: 512-thread workgroups
      else if (sv >= 90000) { g_nzr = sv - 90000; g_nb = 4; sv = 1; mvar = 25; }   // 90000 + z-ranges: ... with four window buffers
      else if (sv >= 80000) { g_nzr = sv - 80000; g_nb = 3; sv = 1; mvar = 25; }   // 80000 + z-ranges: the z-marching form
      else if (sv >= 70000) { sv -= 70000; mvar = 24; }   // 70000 + S: the window form
      g_G = (sv < 0 ? -sv : sv) * 8;
      float best = 1e9f;
      CK(hipMemset(y2, 0xff, n * 8));
      for (int rep = 0; rep < 6; ++rep) {
        k_flush<<<2048, 256>>>(big, out, 1L << 28);
        CK(hipEventRecord(a));
        launch(mvar, pr, pr.xs, y2, 3);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) best = ms < best ? ms : best;
      }
      CK(hipMemcpy(r2.data(), y2, n * 8, hipMemcpyDeviceToHost));
      const bool same = !memcmp(r1.data(), r2.data(), n * 8);
      k_flush<<<2048, 256>>>(big, out, 1L << 28);
      CK(hipEventRecord(a));
      for (int i = 0; i < 20; ++i) launch(mvar, pr, i == 0 ? pr.xs : (i & 1 ? y1 : y2), (i & 1) ? y2 : y1, 3);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%d x %d x %d  march%s S=%d (G=%d, %.2f wg/CU, plane = %.1f tiles)  cold %.1f us  chain %.1f us  %s\n", m, my, mz, mvar == 23 ? " (near taps by dpp)" : mvar == 24 ? " (LDS window)" : mvar == 25 ? " (z-march; S = z-ranges)" : "", mvar == 25 ? g_nzr : g_G / 8, g_G, g_G / 256.0, pr.P / 512.0,
             best * 1e3, ms * 1e3 / 20, same ? "bit-identical" : "DIFFER");
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 4) g_nseg = atoi(argv[4]);
  if (argc > 5) g_allodd = atoi(argv[5]);
  if (argc > 6) g_dbg = atoi(argv[6]);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int shifted : {0, 1, 3}) {
    launch(0, pr, pr.xs, y1, shifted);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r1.data(), y1, n * 8, hipMemcpyDeviceToHost));
    for (int var = 0; var < kNVar; ++var) {
      float best = 1e9f, sum = 0;
      CK(hipMemset(y2, 0xff, n * 8));
      for (int rep = 0; rep < 7; ++rep) {
        k_flush<<<2048, 256>>>(big, out, 1L << 28);
        CK(hipEventRecord(a));
        launch(var, pr, pr.xs, y2, shifted);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { printf("%s: %s\n", vname(var), hipGetErrorString(e)); break; }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) { best = ms < best ? ms : best; sum += ms; }
      }
      CK(hipMemcpy(r2.data(), y2, n * 8, hipMemcpyDeviceToHost));
      long bad = 0, first = -1;
      for (long i = 0; i < n; ++i) if (memcmp(&r1[i], &r2[i], 8)) { if (first < 0) first = i; ++bad; }
      printf("%d x %d x %d shifted=%d  %-22s best %.1f us (%.0f GB/s)  mean %.1f us   %s", m, my, mz, shifted, vname(var), best * 1e3, MB / best, sum / 6 * 1e3,
             bad ? "DIFFER" : "bit-identical");
      if (bad) printf(" (%ld rows, first %ld = plane %ld offset %ld)", bad, first, first / pr.P, first % pr.P);
      printf("\n");
      fflush(stdout);
    }
  }
  // back to back (a chain: the product of one launch is the input of the next, cacheable stores), 20 launches
  for (int var = 0; var < kNVar; ++var) {
    k_flush<<<2048, 256>>>(big, out, 1L << 28);
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) {
      const double* src = i == 0 ? pr.xs : (i & 1 ? y1 : y2);
      double* dst = (i & 1) ? y2 : y1;
      launch(var, pr, src, dst, 3);
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("chain of 20 (%s): %.1f us per product\n", vname(var), ms * 1e3 / 20);
  }
  return 0;
}
