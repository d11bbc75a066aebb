// How fast does gfx950 move R column streams in and W column streams out when the work is laid out like the matrix-instruction
// block kernels (ks_block_mfma.hpp): 256 workgroups of 8 waves, a wave copies 128-byte pieces (8 lanes x 16 bytes) of every
// column?  No arithmetic: what is measured is the memory system's answer to the ADDRESS PATTERN of the writes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rw_streams.hip -o tools/_build/rw_streams && tools/_build/rw_streams [m=216]
// Parameters varied:
//   chunk   slabs (16 rows) a wave reads before it writes: its writes are chunk x 128 contiguous bytes per column
//   wavecon 0: wave w of a workgroup owns slab w of every 8-slab tile (pieces of one wave are 1 KiB apart: chunk only delays)
//           1: a wave owns `chunk` CONSECUTIVE slabs of a super-tile of 8 x chunk slabs (its burst is contiguous)
//   align   the workgroup ranges start at multiples of `align` packs (1: as block_range, 64: whole tiles)
//   nt      non-temporal stores
//   oop     write into columns behind the ones read (0: in place over the last W columns read)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));

template <int R, int W, int CHUNK, bool WAVECON, bool NT>
__global__ void __launch_bounds__(512, 2) k_rw(const double* __restrict__ V, long ld, double* __restrict__ Wb, long npacks, long per, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long pb = blockIdx.x * per, pe = pb + per;
  if (pe > npacks) pe = npacks;
  if (pb > npacks) pb = npacks;
  const long nslab = (pe - pb + 7) / 8;                       // slabs of 8 packs in the range
  const long nsuper = (nslab + 8 * CHUNK - 1) / (8 * CHUNK);  // super-tiles of 8 x CHUNK slabs
  constexpr int NJ = (R + 7) / 8;
  d2v acc = {0.0, 0.0};
  for (long st = 0; st < nsuper; ++st) {
    d2v keep[CHUNK];
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
      const long slab = WAVECON ? st * 8 * CHUNK + wave * CHUNK + c : st * 8 * CHUNK + c * 8 + wave;
      const long p = pb + slab * 8 + (lane & 7);
      d2v s = {0.0, 0.0};
      if (p < pe) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = 8 * j + (lane >> 3);
          if (col < R) {
            const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(V + (long)col * ld + p * 2));
            s += v;
          }
        }
      }
      keep[c] = s;
      acc += s;
    }
    // writes: W columns x CHUNK slabs x 8 packs
    if (WAVECON) {
      const long first = pb + (st * 8 * CHUNK + wave * CHUNK) * 8;   // first pack of the wave's contiguous run
      constexpr int PER = CHUNK * 8;                                  // packs per column
#pragma unroll
      for (int j = 0; j < (W * PER + 63) / 64; ++j) {
        const int idx = j * 64 + lane, col = idx / PER, w = idx % PER;
        const long p = first + w;
        if (col < W && p < pe) {
          d2v* dst = reinterpret_cast<d2v*>(Wb + (long)col * ld + p * 2);
          const d2v v = keep[(w / 8) % CHUNK] + (double)col;
          if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < CHUNK; ++c) {
        const long p = pb + (st * 8 * CHUNK + c * 8 + wave) * 8 + (lane & 7);
#pragma unroll
        for (int j = 0; j < (W + 7) / 8; ++j) {
          const int col = 8 * j + (lane >> 3);
          if (col < W && p < pe) {
            d2v* dst = reinterpret_cast<d2v*>(Wb + (long)col * ld + p * 2);
            const d2v v = keep[c] + (double)col;
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
          }
        }
      }
    }
  }
  if (acc.x + acc.y == 123.456) out[0] = acc.x;
}

struct Args { const double* V; long ld; double* Wb; long npacks, per; double* out; };


// the copy structure of ks_block_mfma.hpp: per-wave ring of RING slabs filled by asynchronous global -> LDS copies
// (global_load_lds_dwordx4, counted by hand), stores of 16 bytes per lane from registers (LDS read back), out of place
__device__ __forceinline__ void glds16_nt(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void waitvm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory"); }
template <int R, int W, int RING, bool ASMST, bool NTL = true>
__global__ void __launch_bounds__(512, 2) k_rw_dma(const double* __restrict__ V, long ld, double* __restrict__ Wb, long npacks, long per, double* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long pb = blockIdx.x * per, pe = pb + per;
  if (pe > npacks) pe = npacks;
  if (pb > npacks) pb = npacks;
  const int niter = (int)((pe - pb + 63) / 64);
  constexpr int NJ = (R + 7) / 8, NST = (W + 7) / 8, SLAB = NJ * 1024;
  unsigned char* myring = lds + (size_t)wave * RING * SLAB;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)myring;
  auto issue = [&](int it, int sl) {
    const long p = pb + (long)it * 64 + wave * 8 + (lane & 7);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = 8 * j + (lane >> 3);
      const double* src = (p < pe && col < R) ? V + (long)col * ld + p * 2 : V;
      if (NTL) glds16_nt(src, ring_lds + (uint32_t)(sl * SLAB + j * 1024)); else glds16(src, ring_lds + (uint32_t)(sl * SLAB + j * 1024));
    }
  };
  for (int it = 0; it < RING - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = RING - 1;
  for (int it = 0; it < niter; ++it) {
    if (it == 0) waitvm<(RING - 2) * NJ>();
    else if (it == 1 && RING > 2) waitvm<(RING - 2) * NJ + NST>();
    else waitvm<(RING - 2) * NJ + (RING - 1) * NST>();
    issue(it + RING - 1, sl_new);
    const unsigned char* slab = myring + (size_t)sl_cur * SLAB;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == RING ? 0 : sl_cur + 1;
    const long p = pb + (long)it * 64 + wave * 8 + (lane & 7);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int col = 8 * j + (lane >> 3);
      const d2v v = *reinterpret_cast<const d2v*>(slab + (NJ - NST + j) * 1024 + lane * 16);
      double* dst = Wb + (long)col * ld + p * 2;
      if (ASMST) {
        // always issued (the count must hold): lanes out of range write to their own slot of a dump area behind the columns
        double* d2 = (col < W && p < pe) ? dst : Wb + (long)W * ld + lane * 2;
        asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(d2), "v"(v) : "memory");
      } else {
        if (col < W && p < pe) __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(dst));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 77 && out) out[1] = 1.0;
}
template <int R, int W, int RING, bool ASMST, bool NTL = true> double run_dma(const Args& a, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  constexpr int SLAB = ((R + 7) / 8) * 1024;
  const size_t smem = (size_t)8 * RING * SLAB;
  auto kern = k_rw_dma<R, W, RING, ASMST, NTL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int i = 0; i < 2; ++i) kern<<<256, 512, smem>>>(a.V, a.ld, a.Wb, a.npacks, a.per, a.out);
  CK(hipGetLastError());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<256, 512, smem>>>(a.V, a.ld, a.Wb, a.npacks, a.per, a.out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}


// plain loads into registers one slab ahead, ds_write into the wave's LDS slab, stores from LDS (what the block kernels would
// do instead of the asynchronous copies)
template <int R, int W, bool NTL, bool STFIRST = false>
__global__ void __launch_bounds__(512, 2) k_rw_reg(const double* __restrict__ V, long ld, double* __restrict__ Wb, long npacks, long per, double* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long pb = blockIdx.x * per, pe = pb + per;
  if (pe > npacks) pe = npacks;
  if (pb > npacks) pb = npacks;
  const int niter = (int)((pe - pb + 63) / 64);
  constexpr int NJ = (R + 7) / 8, NST = (W + 7) / 8, SLAB = NJ * 1024;
  unsigned char* slab = lds + (size_t)wave * SLAB;
  d2v nx[NJ];
  auto load = [&](int it) {
    const long p = pb + (long)it * 64 + wave * 8 + (lane & 7);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = 8 * j + (lane >> 3);
      const d2v* src = reinterpret_cast<const d2v*>((p < pe && col < R) ? V + (long)col * ld + p * 2 : V);
      nx[j] = NTL ? __builtin_nontemporal_load(src) : *src;
    }
  };
  load(0);
  for (int it = 0; it < niter; ++it) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<d2v*>(slab + j * 1024 + lane * 16) = nx[j];
    if (!STFIRST) load(it + 1);
    const long p = pb + (long)it * 64 + wave * 8 + (lane & 7);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int col = 8 * j + (lane >> 3);
      const d2v v = *reinterpret_cast<const d2v*>(slab + (NJ - NST + j) * 1024 + ((lane * 16 + 64) & 1023));
      if (col < W && p < pe) __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(Wb + (long)col * ld + p * 2));
    }
    if (STFIRST) { asm volatile("" ::: "memory"); load(it + 1); }
  }
  if (lds[threadIdx.x] == 77 && out) out[1] = nx[0].x;
}
template <int R, int W, bool NTL, bool STFIRST = false> double run_reg(const Args& a, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  constexpr int SLAB = ((R + 7) / 8) * 1024;
  const size_t smem = (size_t)8 * SLAB;
  auto kern = k_rw_reg<R, W, NTL, STFIRST>;
  for (int i = 0; i < 2; ++i) kern<<<256, 512, smem>>>(a.V, a.ld, a.Wb, a.npacks, a.per, a.out);
  CK(hipGetLastError());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<256, 512, smem>>>(a.V, a.ld, a.Wb, a.npacks, a.per, a.out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int R, int W, int CHUNK, bool WAVECON, bool NT> double run(const Args& a, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_rw<R, W, CHUNK, WAVECON, NT>), dim3(256), dim3(512), 0, 0, a.V, a.ld, a.Wb, a.npacks, a.per, a.out);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_rw<R, W, CHUNK, WAVECON, NT>), dim3(256), dim3(512), 0, 0, a.V, a.ld, a.Wb, a.npacks, a.per, a.out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const long m = argc > 1 ? atol(argv[1]) : 216;
  const long n = m * m * m;
  long ld = (n + 63) / 64 * 64;
  {  // the library's column-stride rule (ks_hip.hip: 0xF800 modulo 128 KiB)
    const long colb = ld * 8, target = 0xF800, window = 0x20000;
    ld += (((target - colb % window) % window + window) % window) / 8;
  }
  constexpr int R = 41, W = 20;
  const long ncols = R + W + 4;
  double* V;
  double* out;
  CK(hipMalloc(&V, sizeof(double) * ld * ncols));
  CK(hipMemset(V, 0, sizeof(double) * ld * ncols));
  CK(hipMalloc(&out, 64));
  const long npacks = ld / 2;
  const double gb_r = 8.0 * n * R / 1e9, gb_w = 8.0 * n * W / 1e9;
  printf("# n = %ld (ld %ld), %d columns read (%.2f GB), %d written (%.2f GB); 256 workgroups x 8 waves\n", n, ld, R, gb_r, W, gb_w);
  auto per_for = [&](long align) { long per = (npacks + 255) / 256; return (per + align - 1) / align * align; };
  auto line = [&](const char* name, double ms, double gb) { printf("%-72s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, gb / ms * 1e-0 * 1e3 / 1e3 * 1.0); };
  (void)line;
#define RUN(CH, WC, NTS, OOP, ALIGN)                                                                                          \
  {                                                                                                                            \
    Args a{V, ld, V + (OOP ? (long)R : (long)(R - W)) * ld, npacks, per_for(ALIGN), out};                                     \
    const double ms = run<R, W, CH, WC, NTS>(a, 5);                                                                            \
    printf("chunk %2d wavecon %d nt %d oop %d align %3d   %8.1f us  %6.0f GB/s\n", CH, (int)WC, (int)NTS, OOP, ALIGN, ms * 1e3, (gb_r + gb_w) / ms);  \
  }
  {  // reads only
    Args a{V, ld, V + (long)R * ld, npacks, per_for(1), out};
    const double ms = run<R, 0, 1, false, true>(a, 5);
    printf("reads only (align 1)                             %8.1f us  %6.0f GB/s\n", ms * 1e3, gb_r / ms);
    a.per = per_for(64);
    const double ms2 = run<R, 0, 1, false, true>(a, 5);
    printf("reads only (align 64)                            %8.1f us  %6.0f GB/s\n", ms2 * 1e3, gb_r / ms2);
  }
  RUN(1, false, true, 1, 1)
  RUN(1, false, true, 0, 1)
  RUN(1, false, false, 1, 1)
  RUN(1, false, true, 1, 8)
  RUN(1, false, true, 1, 64)
  RUN(2, false, true, 1, 1)
  RUN(4, false, true, 1, 1)
  RUN(8, false, true, 1, 1)
  RUN(2, true, true, 1, 1)
  RUN(4, true, true, 1, 1)
  RUN(8, true, true, 1, 1)
  RUN(4, true, true, 1, 8)
  RUN(4, true, true, 1, 64)
  RUN(8, true, true, 1, 64)
  RUN(4, true, false, 1, 1)
  RUN(4, true, true, 0, 1)
  RUN(8, true, true, 0, 1)
  {
    Args a{V, ld, V + (long)R * ld, npacks, per_for(1), out};
    printf("dma ring 3, compiler stores (vmcnt drained by the compiler), oop   %8.1f us\n", run_dma<R, W, 3, false>(a, 5) * 1e3);
    printf("dma ring 3, counted asm stores, oop                                %8.1f us\n", run_dma<R, W, 3, true>(a, 5) * 1e3);
    printf("dma ring 2, counted asm stores, oop                                %8.1f us\n", run_dma<R, W, 2, true>(a, 5) * 1e3);
    printf("dma ring 3, no stores                                              %8.1f us\n", run_dma<R, 0, 3, true>(a, 5) * 1e3);
    printf("dma ring 3, plain (cacheable) copies, counted asm stores, oop      %8.1f us\n", run_dma<R, W, 3, true, false>(a, 5) * 1e3);
    printf("reg prefetch, nt loads, oop                                        %8.1f us\n", run_reg<R, W, true>(a, 5) * 1e3);
    printf("reg prefetch, nt loads, stores BEFORE the next loads, oop           %8.1f us\n", run_reg<R, W, true, true>(a, 5) * 1e3);
    printf("reg prefetch, plain loads, oop                                     %8.1f us\n", run_reg<R, W, false>(a, 5) * 1e3);
    printf("reg prefetch, nt loads, no stores                                  %8.1f us\n", run_reg<R, 0, true>(a, 5) * 1e3);
    a.Wb = V + (long)(R - W) * ld;
    printf("reg prefetch, nt loads, in place                                   %8.1f us\n", run_reg<R, W, true>(a, 5) * 1e3);
    printf("dma ring 3, counted asm stores, in place                           %8.1f us\n", run_dma<R, W, 3, true>(a, 5) * 1e3);
  }
  return 0;
}
