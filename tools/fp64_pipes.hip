// Issue rates of the FP64 pipes of gfx950 (MI355X), register-only, plus the LDS read rates the block kernels depend on.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fp64_pipes.hip -o tools/_build/fp64_pipes && tools/_build/fp64_pipes
// Every test launches 256 x OCC workgroups of 256 threads (one wave per SIMD and workgroup; OCC workgroups per CU ->
// OCC waves per SIMD) and reports chip-wide TFLOP/s and the cycles one SIMD spends per instruction at the clock the
// run achieved (measured with s_memrealtime against wall_clock: see `clk`).
//   vfma     v_fma_f64, 16 independent chains per lane                (128 flop per wave instruction)
//   mfma16   v_mfma_f64_16x16x4_f64, 4 independent accumulators       (2048 flop)
//   mfma4    v_mfma_f64_4x4x4_4b_f64, 8 independent accumulators      (512 flop)
//   mix      waves 0..OCC/2-1 of a SIMD run mfma, the others vfma: do the two pipes add up?
//   inter    ONE wave interleaves mfma and independent v_fma_f64
//   lds      ds_read_b128 / ds_read_b64, per-lane addresses vs one address for all lanes (broadcast)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int UNROLL> __device__ __forceinline__ void vfma_body(double (&a)[16], double x, double y, int iters) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], x, y);
  }
}

__global__ void __launch_bounds__(256) k_vfma(double* out, int iters, double x, double y) {
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  vfma_body<4>(a, x, y, iters);
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456) out[0] = s;
}

__global__ void __launch_bounds__(256) k_mfma16(double* out, int iters, double x, double y) {
  d4 c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = d4{0, 0, 0, 0};
  double a = x + threadIdx.x, b = y;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
  if (s == 123.456) out[0] = s;
}

__global__ void __launch_bounds__(256) k_mfma4(double* out, int iters, double x, double y) {
  double c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = 0;
  double a = x + threadIdx.x, b = y;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i];
  if (s == 123.456) out[0] = s;
}

// workgroups with an even index run the matrix pipe, odd ones the vector pipe (the dispatcher places consecutive
// workgroups of 256 threads on the same CU until it is full: OCC even -> half of each SIMD's waves per kind)
template <int KIND> __global__ void __launch_bounds__(256) k_mix(double* out, int iters_m, int iters_v, double x, double y, unsigned long long* tm) {
  const bool matrix = (blockIdx.x & 1) == 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  double s = 0;
  if (matrix) {
    if constexpr (KIND == 16) {
      d4 c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = d4{0, 0, 0, 0};
      double a = x + threadIdx.x, b = y;
      for (int it = 0; it < iters_m; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
    } else {
      double c[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = 0;
      double a = x + threadIdx.x, b = y;
      for (int it = 0; it < iters_m; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += c[i];
    }
  } else {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
    vfma_body<4>(a, x, y, iters_v);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) tm[blockIdx.x] = t1 - t0;
  if (s == 123.456) out[0] = s;
}

// one wave: per loop trip 4 mfma16 (or 8 mfma4) and NV independent v_fma_f64
template <int KIND, int NV> __global__ void __launch_bounds__(256) k_inter(double* out, int iters, double x, double y) {
  d4 c[4];
  double c4[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = d4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) c4[i] = 0;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  double aa = x + threadIdx.x, bb = y;
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb, c[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) a[(i * (NV / 4) + j) & 15] = __builtin_fma(a[(i * (NV / 4) + j) & 15], x, y);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(aa, bb, c4[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / 8; ++j) a[(i * (NV / 8) + j) & 15] = __builtin_fma(a[(i * (NV / 8) + j) & 15], x, y);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c4[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456) out[0] = s;
}

// LDS reads.  MODE 0: ds_read_b128 at lane * 16; 1: ds_read_b128, one address for the wave; 2: ds_read_b64 at lane * 8;
// 3: ds_read_b64 one address; 4: ds_read_b64 at (lane & 15) * STRIDE + (lane >> 4) * 8 (the operand pattern of a 16x16x4 MFMA
// from a column-major tile, column stride STRIDE bytes)
template <int MODE> __global__ void __launch_bounds__(256) k_lds(double* out, int iters, int stride) {
  __shared__ __attribute__((aligned(16))) double buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t addr = (uint32_t)(uintptr_t)buf;
  if constexpr (MODE == 0) addr += lane * 16;
  if constexpr (MODE == 2) addr += lane * 8;
  if constexpr (MODE == 4) addr += (lane & 15) * stride + (lane >> 4) * 8;
  double s = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE <= 1) {
      d4 r[8];  // (only the low half is loaded)
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(addr), "n"(1024 * 0) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) s += q[i].x;
      (void)r;
    } else {
      double q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(q[i]) : "v"(addr) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) s += q[i];
    }
  }
  if (s == 123.456) out[0] = s;
}

static double time_ms(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch(ctx);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) launch(ctx);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

struct Ctx { int grid, iters, iters2, stride; double* out; unsigned long long* tm; };

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("# %s: %d CUs, clockRate %.3f GHz (cycles below are at this clock)\n", p.name, ncu, ghz);
  double* out;
  unsigned long long* tm;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&tm, sizeof(unsigned long long) * ncu * 16));
  const int iters = 4000;
  auto report = [&](const char* name, int occ, double ms, double instr_per_wave, double flop_per_instr) {
    const double waves_per_simd = occ;  // one wave per SIMD and workgroup
    const double instr_per_simd = instr_per_wave * waves_per_simd;
    const double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
    const double tf = instr_per_simd * 4 * ncu * flop_per_instr / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d  %8.3f ms  %7.2f cycles/instr/SIMD  %7.2f TFLOP/s\n", name, occ, ms, cyc, tf);
  };
  for (int occ : {1, 2, 4}) {
    Ctx c{ncu * occ, iters, 0, 0, out, tm};
    double ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_vfma, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 1.0000001, 1e-9); }, &c, 5);
    report("vfma  v_fma_f64", occ, ms, iters * 64.0, 128);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_mfma16, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 1.0000001, 1e-9); }, &c, 5);
    report("mfma16 v_mfma_f64_16x16x4", occ, ms, iters * 16.0, 2048);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_mfma4, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 1.0000001, 1e-9); }, &c, 5);
    report("mfma4  v_mfma_f64_4x4x4_4b", occ, ms, iters * 16.0, 512);
  }
  // mixed occupancy: OCC workgroups per CU, even ones matrix, odd ones vector.  Tuned so that both kinds take about the
  // same time alone: 16 mfma16 per trip vs 64 vfma per trip.
  for (int kind : {16, 4}) {
    for (int occ : {2, 4}) {
      Ctx c{ncu * occ, iters, iters, 0, out, tm};
      double ms;
      if (kind == 16) ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_mix<16>, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, c->iters2, 1.0000001, 1e-9, c->tm); }, &c, 5);
      else ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_mix<4>, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, c->iters2, 1.0000001, 1e-9, c->tm); }, &c, 5);
      std::vector<unsigned long long> h(c.grid);
      CK(hipMemcpy(h.data(), tm, sizeof(unsigned long long) * c.grid, hipMemcpyDeviceToHost));
      double tmx = 0, tvx = 0;
      for (int b = 0; b < c.grid; ++b) { if (b & 1) tvx += h[b]; else tmx += h[b]; }
      tmx /= c.grid / 2;
      tvx /= c.grid / 2;
      const double fm = (kind == 16 ? 2048.0 : 512.0) * 16 * iters, fv = 128.0 * 64 * iters;
      const double tf = (fm + fv) * (occ / 2) * 4 * ncu / (ms * 1e-3) / 1e12;
      printf("mix mfma%-2d + vfma  waves/SIMD %d (half each)  %8.3f ms  both pipes together %7.2f TFLOP/s  (wave life, shader-clock ticks: matrix %.0f, vector %.0f)\n",
             kind, occ, ms, tf, tmx, tvx);
    }
  }
  for (int occ : {1, 2}) {
    Ctx c{ncu * occ, iters, 0, 0, out, tm};
    double ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL((k_inter<16, 16>), dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 1.0000001, 1e-9); }, &c, 5);
    printf("inter 4 mfma16 + 16 vfma per trip  waves/SIMD %d  %8.3f ms  %7.1f cycles per trip and wave slot  (%.2f TFLOP/s)\n", occ, ms,
           ms * 1e-3 * ghz * 1e9 / (iters * occ), (4 * 2048.0 + 16 * 128.0) * iters * occ * 4 * ncu / (ms * 1e-3) / 1e12);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL((k_inter<16, 32>), dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 1.0000001, 1e-9); }, &c, 5);
    printf("inter 4 mfma16 + 32 vfma per trip  waves/SIMD %d  %8.3f ms  %7.1f cycles per trip and wave slot  (%.2f TFLOP/s)\n", occ, ms,
           ms * 1e-3 * ghz * 1e9 / (iters * occ), (4 * 2048.0 + 32 * 128.0) * iters * occ * 4 * ncu / (ms * 1e-3) / 1e12);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL((k_inter<4, 16>), dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 1.0000001, 1e-9); }, &c, 5);
    printf("inter 8 mfma4  + 16 vfma per trip  waves/SIMD %d  %8.3f ms  %7.1f cycles per trip and wave slot  (%.2f TFLOP/s)\n", occ, ms,
           ms * 1e-3 * ghz * 1e9 / (iters * occ), (8 * 512.0 + 16 * 128.0) * iters * occ * 4 * ncu / (ms * 1e-3) / 1e12);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL((k_inter<4, 32>), dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 1.0000001, 1e-9); }, &c, 5);
    printf("inter 8 mfma4  + 32 vfma per trip  waves/SIMD %d  %8.3f ms  %7.1f cycles per trip and wave slot  (%.2f TFLOP/s)\n", occ, ms,
           ms * 1e-3 * ghz * 1e9 / (iters * occ), (8 * 512.0 + 32 * 128.0) * iters * occ * 4 * ncu / (ms * 1e-3) / 1e12);
  }
  // LDS: bytes per cycle and CU
  for (int occ : {1, 2}) {
    Ctx c{ncu * occ, iters, 0, 0, out, tm};
    auto lds_report = [&](const char* name, double ms, double bytes_per_instr) {
      const double instr = iters * 8.0 * 4 * occ;  // per CU
      const double cyc = ms * 1e-3 * ghz * 1e9;
      printf("lds %-44s waves/SIMD %d  %8.3f ms  %6.2f cycles/instr/CU  %7.1f B/cycle/CU\n", name, occ, ms, cyc / instr, instr * bytes_per_instr / cyc);
    };
    double ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_lds<0>, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 0); }, &c, 5);
    lds_report("ds_read_b128 lane*16", ms, 1024);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_lds<1>, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 0); }, &c, 5);
    lds_report("ds_read_b128 broadcast (one address)", ms, 16);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_lds<2>, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 0); }, &c, 5);
    lds_report("ds_read_b64 lane*8", ms, 512);
    ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_lds<3>, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, 0); }, &c, 5);
    lds_report("ds_read_b64 broadcast (one address)", ms, 8);
    for (int stride : {1024, 1040, 1032, 1056, 2064}) {
      c.stride = stride;
      ms = time_ms([](void* v) { Ctx* c = (Ctx*)v; hipLaunchKernelGGL(k_lds<4>, dim3(c->grid), dim3(256), 0, 0, c->out, c->iters, c->stride); }, &c, 5);
      char nm[64];
      snprintf(nm, sizeof nm, "ds_read_b64 mfma pattern, column stride %d", stride);
      lds_report(nm, ms, 512);
    }
  }
  return 0;
}
