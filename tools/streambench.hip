// Micro-benchmark: how fast can gfx950 stream S interleaved column streams of doubles?
// Informs the layout of the tall-skinny kernels (k_dots / k_axpy*).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double2 ldnt(const double* p);
// S columns, U consecutive 16-byte packs per lane per column per iteration, contiguous block ranges
template <int S, int U, bool NT>
__global__ void __launch_bounds__(256) k_read(const double* __restrict__ V, long ld, double* out) {
  const long npk = ld / 2;
  const long per = (npk + gridDim.x - 1) / gridDim.x;
  long pb = (long)blockIdx.x * per, pe = pb + per;
  if (pe > npk) pe = npk;
  double acc = 0.0;
  for (long p = pb + threadIdx.x; p < pe; p += 256 * U) {
    double2 v[S][U];
#pragma unroll
    for (int c = 0; c < S; ++c)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long q = p + u * 256; if (q >= pe) q = p;
        const double2* a = reinterpret_cast<const double2*>(V + (long)c * ld + 2 * q);
        if (NT) { v[c][u].x = __builtin_nontemporal_load(&a->x); v[c][u].y = __builtin_nontemporal_load(&a->y); }
        else v[c][u] = *a;
      }
#pragma unroll
    for (int c = 0; c < S; ++c)
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[c][u].x + v[c][u].y;
  }
  if (acc == 123.456) out[0] = acc;
}

// streaming update: w -= sum_c V_c * g ; S read streams + 1 read/write stream
template <int S>
__global__ void __launch_bounds__(256) k_update(const double* __restrict__ V, long ld, double* __restrict__ w) {
  const long npk = ld / 2;
  const long per = (npk + gridDim.x - 1) / gridDim.x;
  long pb = (long)blockIdx.x * per, pe = pb + per;
  if (pe > npk) pe = npk;
  for (long p = pb + threadIdx.x; p < pe; p += 256) {
    double2 s = *reinterpret_cast<double2*>(w + 2 * p);
#pragma unroll
    for (int c = 0; c < S; ++c) {
      const double2 v = *reinterpret_cast<const double2*>(V + (long)c * ld + 2 * p);
      s.x -= 1e-3 * v.x; s.y -= 1e-3 * v.y;
    }
    *reinterpret_cast<double2*>(w + 2 * p) = s;
  }
}

// U packs per lane per iteration, stores delayed to the end of the iteration; optional separate output
template <int S, int U, bool NT, bool NTS>
__global__ void __launch_bounds__(256) k_update4(const double* __restrict__ V, long ld, double* __restrict__ w) {
  const long npk = ld / 2;
  const long per = (npk + gridDim.x - 1) / gridDim.x;
  long pb = (long)blockIdx.x * per, pe = pb + per;
  if (pe > npk) pe = npk;
  for (long p = pb + threadIdx.x; p < pe; p += 256 * U) {
    double2 s[U];
    long q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { q[u] = p + u * 256; if (q[u] >= pe) q[u] = p; s[u] = *reinterpret_cast<double2*>(w + 2 * q[u]); }
#pragma unroll
    for (int c = 0; c < S; ++c) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double2 v = NT ? ldnt(V + (long)c * ld + 2 * q[u]) : *reinterpret_cast<const double2*>(V + (long)c * ld + 2 * q[u]);
        s[u].x -= 1e-3 * v.x; s[u].y -= 1e-3 * v.y;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double* o = w + 2 * q[u];
      if (NTS) { __builtin_nontemporal_store(s[u].x, o); __builtin_nontemporal_store(s[u].y, o + 1); }
      else *reinterpret_cast<double2*>(o) = s[u];
    }
  }
}

template <int S, int U, bool NT, bool SEP>
__global__ void __launch_bounds__(256) k_update2(const double* __restrict__ V, long ld, double* __restrict__ w, double* __restrict__ wout) {
  const long npk = ld / 2;
  const long per = (npk + gridDim.x - 1) / gridDim.x;
  long pb = (long)blockIdx.x * per, pe = pb + per;
  if (pe > npk) pe = npk;
  for (long p = pb + threadIdx.x; p < pe; p += 256 * U) {
    double2 s[U];
    long q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { q[u] = p + u * 256; if (q[u] >= pe) q[u] = p; s[u] = *reinterpret_cast<double2*>(w + 2 * q[u]); }
#pragma unroll
    for (int c = 0; c < S; ++c) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double2* a = reinterpret_cast<const double2*>(V + (long)c * ld + 2 * q[u]);
        double2 v;
        if (NT) { v.x = __builtin_nontemporal_load(&a->x); v.y = __builtin_nontemporal_load(&a->y); } else v = *a;
        s[u].x -= 1e-3 * v.x; s[u].y -= 1e-3 * v.y;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) *reinterpret_cast<double2*>((SEP ? wout : w) + 2 * q[u]) = s[u];
  }
}

__device__ __forceinline__ double2 ldnt(const double* p) {
  const double2* a = reinterpret_cast<const double2*>(p);
  double2 v; v.x = __builtin_nontemporal_load(&a->x); v.y = __builtin_nontemporal_load(&a->y); return v;
}
// software-pipelined update: the first PF column loads (and w) of iteration i+1 are issued BEFORE the store of
// iteration i, so the in-order vmcnt never makes a load wait behind a store acknowledgement.
template <int S, int PF, bool NT>
__global__ void __launch_bounds__(256) k_update3(const double* __restrict__ V, long ld, double* __restrict__ w) {
  const long npk = ld / 2;
  const long per = (npk + gridDim.x - 1) / gridDim.x;
  long pb = (long)blockIdx.x * per, pe = pb + per;
  if (pe > npk) pe = npk;
  long p = pb + threadIdx.x;
  if (p >= pe) return;
  double2 wn = *reinterpret_cast<double2*>(w + 2 * p);
  double2 nx[PF];
#pragma unroll
  for (int c = 0; c < PF; ++c) nx[c] = NT ? ldnt(V + (long)c * ld + 2 * p) : *reinterpret_cast<const double2*>(V + (long)c * ld + 2 * p);
  while (true) {
    double2 s = wn;
    double2 v[S - PF];
#pragma unroll
    for (int c = PF; c < S; ++c) v[c - PF] = NT ? ldnt(V + (long)c * ld + 2 * p) : *reinterpret_cast<const double2*>(V + (long)c * ld + 2 * p);
#pragma unroll
    for (int c = 0; c < PF; ++c) { s.x -= 1e-3 * nx[c].x; s.y -= 1e-3 * nx[c].y; }
    const long pn = p + 256;
    const bool more = pn < pe;
    const long pq = more ? pn : p;
    wn = *reinterpret_cast<double2*>(w + 2 * pq);
#pragma unroll
    for (int c = 0; c < PF; ++c) nx[c] = NT ? ldnt(V + (long)c * ld + 2 * pq) : *reinterpret_cast<const double2*>(V + (long)c * ld + 2 * pq);
#pragma unroll
    for (int c = PF; c < S; ++c) { s.x -= 1e-3 * v[c - PF].x; s.y -= 1e-3 * v[c - PF].y; }
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<double2*>(w + 2 * p) = s;
    if (!more) break;
    p = pn;
  }
}

// read-only twin of k_update (same arithmetic, result reduced instead of stored): isolates the store cost
template <int S>
__global__ void __launch_bounds__(256) k_update_nostore(const double* __restrict__ V, long ld, const double* __restrict__ w, double* out) {
  const long npk = ld / 2;
  const long per = (npk + gridDim.x - 1) / gridDim.x;
  long pb = (long)blockIdx.x * per, pe = pb + per;
  if (pe > npk) pe = npk;
  double acc = 0;
  for (long p = pb + threadIdx.x; p < pe; p += 256) {
    double2 s = *reinterpret_cast<const double2*>(w + 2 * p);
#pragma unroll
    for (int c = 0; c < S; ++c) {
      const double2 v = *reinterpret_cast<const double2*>(V + (long)c * ld + 2 * p);
      s.x -= 1e-3 * v.x; s.y -= 1e-3 * v.y;
    }
    acc += s.x + s.y;
  }
  if (acc == 123.456) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_copy(const double2* __restrict__ a, double2* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
  const long n = 216L * 216 * 216;  // 10,077,696
  const long ld = n;
  const int NC = 41;
  double *V, *out, *w;
  CK(hipMalloc(&V, sizeof(double) * ld * NC)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&w, sizeof(double) * ld));
  CK(hipMemset(V, 0, sizeof(double) * ld * NC)); CK(hipMemset(w, 0, sizeof(double) * ld));
  std::vector<double> h(1 << 20);
  for (auto& x : h) x = rand() / (double)RAND_MAX;
  for (long off = 0; off + (long)h.size() <= ld * NC; off += h.size() * 37) CK(hipMemcpy(V + off, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cu = prop.multiProcessorCount;
  printf("CUs %d\n", cu);
  {
    float ms = timeit([&] { k_copy<<<cu * 8, 256>>>((const double2*)V, (double2*)(V + ld * 20), ld * 20 / 2); }, 5);
    printf("copy 20 cols -> 20 cols: %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * ld * 20 * 8 / ms / 1e6);
  }
#define RUN(S, U, NT, BPC) { float ms = timeit([&] { k_read<S, U, NT><<<cu * BPC, 256>>>(V, ld, out); }, 5); \
    printf("read S=%2d U=%d nt=%d bpc=%d: %.3f ms  %.0f GB/s\n", S, U, (int)NT, BPC, ms, (double)ld * S * 8 / ms / 1e6); }
  RUN(1, 1, false, 8) RUN(1, 4, false, 8) RUN(1, 8, false, 4) RUN(1, 8, false, 8)
  RUN(8, 1, false, 4) RUN(8, 1, false, 8) RUN(8, 2, false, 4) RUN(8, 4, false, 4)
  RUN(40, 1, false, 2) RUN(40, 1, false, 3) RUN(40, 1, false, 4) RUN(40, 1, false, 8)
  RUN(40, 1, true, 4) RUN(40, 1, true, 8)
  RUN(20, 2, false, 4) RUN(20, 1, false, 8) RUN(10, 4, false, 4) RUN(10, 2, false, 8) RUN(10, 1, false, 8)
#define RUNU(S, BPC) { float ms = timeit([&] { k_update<S><<<cu * BPC, 256>>>(V, ld, w); }, 5); \
    printf("update S=%2d bpc=%d: %.3f ms  %.0f GB/s\n", S, BPC, ms, (double)ld * (S + 2) * 8 / ms / 1e6); }
  RUNU(40, 4) RUNU(40, 8) RUNU(20, 8) RUNU(8, 8)
  { float ms = timeit([&] { k_update_nostore<40><<<cu * 4, 256>>>(V, ld, w, out); }, 5);
    printf("update-nostore S=40 bpc=4: %.3f ms  %.0f GB/s\n", ms, (double)ld * 41 * 8 / ms / 1e6); }
#define RUNU2(S, U, NT, SEP, BPC) { float ms = timeit([&] { k_update2<S, U, NT, SEP><<<cu * BPC, 256>>>(V, ld, w, V + ld * 40); }, 5); \
    printf("update2 S=%2d U=%d nt=%d sep=%d bpc=%d: %.3f ms  %.0f GB/s\n", S, U, (int)NT, (int)SEP, BPC, ms, (double)ld * (S + 2) * 8 / ms / 1e6); }
#define RUNU3(S, PF, NT, BPC) { float ms = timeit([&] { k_update3<S, PF, NT><<<cu * BPC, 256>>>(V, ld, w); }, 5); \
    printf("update3 S=%2d PF=%d nt=%d bpc=%d: %.3f ms  %.0f GB/s\n", S, PF, (int)NT, BPC, ms, (double)ld * (S + 2) * 8 / ms / 1e6); }
#define RUNU4(S, U, NT, NTS, BPC) { float ms = timeit([&] { k_update4<S, U, NT, NTS><<<cu * BPC, 256>>>(V, ld, w); }, 5); \
    printf("update4 S=%2d U=%d nt=%d nts=%d bpc=%d: %.3f ms  %.0f GB/s\n", S, U, (int)NT, (int)NTS, BPC, ms, (double)ld * (S + 2) * 8 / ms / 1e6); }
  RUNU4(40, 1, true, true, 4) RUNU4(40, 2, true, true, 4) RUNU4(40, 4, true, true, 4) RUNU4(40, 8, true, true, 4) RUNU4(40, 4, false, true, 4) RUNU4(40, 4, true, false, 4)
  RUNU4(40, 4, true, true, 3) RUNU4(40, 4, true, true, 6) RUNU4(40, 8, true, true, 2) RUNU4(20, 4, true, true, 4) RUNU4(30, 4, true, true, 4)
  RUNU3(40, 4, false, 4) RUNU3(40, 8, false, 4) RUNU3(40, 16, false, 4) RUNU3(40, 8, true, 4) RUNU3(40, 16, true, 4) RUNU3(40, 8, true, 8) RUNU3(40, 8, true, 3)
  RUNU2(40, 1, false, false, 4) RUNU2(40, 2, false, false, 4) RUNU2(40, 4, false, false, 4) RUNU2(40, 4, false, false, 2)
  RUNU2(40, 1, true, false, 4) RUNU2(40, 2, true, false, 4) RUNU2(40, 4, true, false, 4)
  RUNU2(40, 1, false, true, 4) RUNU2(40, 2, false, true, 4) RUNU2(40, 2, true, true, 4) RUNU2(40, 4, true, true, 8)
  return 0;
}
