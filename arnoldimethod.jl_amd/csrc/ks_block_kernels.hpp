// gfx950 kernels of the S-STEP (block) Arnoldi expansion: s steps of iterate_arnoldi! (src/expansion.jl:116-133) with TWO
// passes over the basis per BLOCK instead of two per step.  The algebra is pinned on the CPU by tests/sstep_model.py against
// the oracle (tests/test_sstep_model.py); DESIGN.md section 3 "s-step expansion".
//
//   z_0 = S[:, k-1]  (last stored column),   z_i = sigma (A z_{i-1} - theta_i z_{i-1}),  i = 1..s      k_shift_scale / fused SpMV
//   pass 1   P_raw = S_k^H Z,  G_Z = Z^H Z                                                             k_bdots
//            P = T^H P_raw;  G_1 = G_Z - P^H P = R_1^H R_1;  coef' = (T P) R_1^-1                       k_fin_blk (stage 1)
//   pass 2   Qt = Z R_1^-1 - S_k coef'   (written over Z);  C_raw = S_k^H Qt,  G_t = Qt^H Qt           k_bupdate
//            C = T^H C_raw;  G_2 = G_t - C^H C = R_2^H R_2 (~ I);  T <- [[T, -T C R_2^-1], [0, R_2^-1]]
//            H columns k-1 .. k+s-2 from the basis recurrence                                           k_fin_blk (stage 2)
//
// Both streaming kernels use the column-split form of k_axpy_dots_cs: the k basis columns are dealt round-robin to the 4
// waves of a workgroup, all waves walk the same rows; a lane keeps NCW column slices (16-byte packs) and NCW x S
// accumulators in registers.  Per-lane accumulators are summed over the wave by HALVING exchanges (fold_wave: each stage
// trades half of the values with the partner lane, N values cost ~N shuffles instead of 6 N), every accumulator belongs to
// exactly one wave, so there is no cross-wave reduction and no barrier at the end.  Deterministic: fixed lane <-> row map,
// fixed exchange pattern.
#pragma once

#include "ks_kernels.hpp"

namespace ksd {

constexpr int kBlkSMax = 20;                                   // largest block (steps per block): Float64 20, ComplexF64 10 (blk_smax)
template <class T> constexpr int blk_smax() { return sizeof(T) == 8 ? kBlkSMax : 10; }
constexpr int kBlkGram = kBlkSMax * (kBlkSMax + 1) / 2;        // upper triangle of an s x s Gram matrix
constexpr int kBlkHLds = 2048;                                 // ... and of H[0:k, 0:k-1)
constexpr int kBlkTLds = 2048;                                 // elements of T the block algebra stages in LDS (16 / 32 KiB)
constexpr int kBlkKMax = kTMax;                                // columns the factored path covers (maxdim <= 64 -> 65 columns)

// ---- per-lane accumulators -> per-wave totals by halving exchanges ---------------------------------------------------------
// a[0..N) on every lane; afterwards (P = N, a power of two):
//   P >= 64: lane L holds the wave totals of the P/64 original indices  j + (P/64) L,  j < P/64, in a[j]
//   P <  64: lane L holds the wave total of original index  L / (64/P)  in a[0]
template <int N, int OFF> __device__ __forceinline__ void fold_stage(double* a, int lane) {
  if constexpr (OFF >= 1) {
    if constexpr (N >= 2) {
      const bool hi = (lane & OFF) != 0;
#pragma unroll
      for (int i = 0; i < N / 2; ++i) {
        const double lo_v = a[i], hi_v = a[i + N / 2];
        const double send = hi ? lo_v : hi_v, keep = hi ? hi_v : lo_v;
        a[i] = keep + __shfl_xor(send, OFF, 64);
      }
      fold_stage<N / 2, OFF / 2>(a, lane);
    } else {
      a[0] += __shfl_xor(a[0], OFF, 64);
      fold_stage<1, OFF / 2>(a, lane);
    }
  }
}
constexpr int next_pow2(int n) { int p = 1; while (p < n) p *= 2; return p; }

// doubles per element
template <class T> struct Dpe { static constexpr int value = (int)(sizeof(T) / 8); };
__device__ __forceinline__ void put_acc(double* f, int idx, double v) { f[idx] = v; }
__device__ __forceinline__ void put_acc(double* f, int idx, cd v) { f[2 * idx] = v.x; f[2 * idx + 1] = v.y; }

// packs per lane and iteration, and the occupancy the kernels are compiled for: a lane keeps NCW x SB accumulators, NCW x U
// basis packs and 2 S U block packs; while that fits 256 registers TWO waves share a SIMD (measured on the 216^3 basis:
// pass 1 6.2-6.5 TB/s with two resident workgroups, 3.0-5.0 with one; pass 2 5.0 against 3.8) -- what counts is the number
// of bytes in flight per CU.
//
// WIDE form (NW = 8 waves, WB = 2): the waves form a WA x WB grid (WA = NW / WB); wave (wa, wb) keeps the basis columns
// c = wa (mod WA) and accumulates against the block columns i = wb (mod WB) -- NCW x ceil(S / WB) accumulators, half of
// what the four-wave form needs at the same k.  That is what lets blocks of 10 run with two waves per SIMD (the four-wave
// instantiations need > 256 registers there: one wave per SIMD, 2.6-3.5 TB/s).  Each basis column is loaded by WB waves
// (the second hit is served by the CU's L1).
template <class T, int NCW, int S, int NW = 4> constexpr int blk_u() { return S <= 5 ? 2 : (NW == 8 && NCW <= 6 && S <= 10 ? 2 : 1); }
template <class T, int NCW, int S, int U> constexpr int blk_regs() {
  constexpr int D = (int)(sizeof(T) / 8);
  return 2 * (D * NCW * S + 2 * NCW * U + 2 * S * U + D * ((S * (S + 1) / 2 + 3) / 4)) + 44;
}
// workgroups per CU the kernels are compiled for (four-wave form: 2 while the registers allow; wide form: always 1 = two
// waves per SIMD)
template <class T, int NCW, int S, int U, int NW = 4> constexpr int blk_wpe() { return NW == 8 ? 1 : (blk_regs<T, NCW, S, U>() <= 256 ? 2 : 1); }
// second argument of __launch_bounds__: waves per SIMD
template <class T, int NCW, int S, int U, int NW> constexpr int blk_wps() { return blk_wpe<T, NCW, S, U, NW>() * NW / 4; }

// upper-triangle index of Gram entry (i, i2), i <= i2
__host__ __device__ __forceinline__ constexpr int gram_idx(int i, int i2) { return i2 * (i2 + 1) / 2 + i; }

template <int V> using blk_ic = std::integral_constant<int, V>;
// f(blk_ic<wb>) for the wave's (uniform, run-time) wb with a compile-time argument: register arrays stay statically indexed
template <int WB, class F> __device__ __forceinline__ void blk_by_wb(int wb, F&& f) {
  if constexpr (WB == 1) f(blk_ic<0>{});
  else if constexpr (WB == 2) { if (wb == 0) f(blk_ic<0>{}); else f(blk_ic<1>{}); }
  else { if (wb == 0) f(blk_ic<0>{}); else if (wb == 1) f(blk_ic<1>{}); else if (wb == 2) f(blk_ic<2>{}); else f(blk_ic<3>{}); }
}

// f(blk_ic<w>) for a wave-uniform run-time w < N: ONE jump instead of a compare-and-branch per dealt item (the per-entry
// branches of the Gram triangle cost more than its arithmetic: 55 of them at s = 10 for 7 products per wave)
template <int N, class F> __device__ __forceinline__ void blk_by_idx(int w, F&& f) {
  static_assert(N == 1 || N == 2 || N == 4 || N == 8, "dealt over 1, 2, 4 or 8");
  if constexpr (N == 1) f(blk_ic<0>{});
  else if constexpr (N == 2) { if (w == 0) f(blk_ic<0>{}); else f(blk_ic<1>{}); }
  else if constexpr (N == 4) {
    switch (w) { case 0: f(blk_ic<0>{}); break; case 1: f(blk_ic<1>{}); break; case 2: f(blk_ic<2>{}); break; default: f(blk_ic<3>{}); break; }
  } else {
    switch (w) {
      case 0: f(blk_ic<0>{}); break; case 1: f(blk_ic<1>{}); break; case 2: f(blk_ic<2>{}); break; case 3: f(blk_ic<3>{}); break;
      case 4: f(blk_ic<4>{}); break; case 5: f(blk_ic<5>{}); break; case 6: f(blk_ic<6>{}); break; default: f(blk_ic<7>{}); break;
    }
  }
}

// Write the folded accumulators of one wave.  Flattened accumulator index (in elements of T):
//   e < NCW*SB       : column c = wa + WA (e / SB), right-hand side i = wb + WB (e % SB)  -> partial entry  i*k + c
//   e = NCW*SB + gi  : Gram entry g = NW gi + wave (if < ng)                              -> partial entry  k*S + g
template <class T, int NCW, int S, int NGW, int NW, int WB>
__device__ __forceinline__ void store_folded(const double* f, int lane, int wave, int k, T* __restrict__ partial, int pnb) {
  constexpr int D = Dpe<T>::value;
  constexpr int WA = NW / WB, SB = (S + WB - 1) / WB;
  constexpr int NE = NCW * SB + NGW;           // elements
  constexpr int P = next_pow2(NE * D);         // doubles, padded
  constexpr int NG = S * (S + 1) / 2;
  const int wa = wave % WA, wb = wave / WA;
  auto put = [&](int didx, double v) {
    const int e = didx / D, part = didx % D;
    int entry = -1;
    if (e < NCW * SB) {
      const int c = wa + WA * (e / SB), i = wb + WB * (e % SB);
      if (c < k && i < S) entry = i * k + c;
    } else if (e < NE) {
      const int g = NW * (e - NCW * SB) + wave;
      if (g < NG) entry = k * S + g;
    }
    if (entry >= 0) reinterpret_cast<double*>(partial + (int64_t)entry * pnb + blockIdx.x)[part] = v;
  };
  if constexpr (P >= 64) {
#pragma unroll
    for (int j = 0; j < P / 64; ++j) put(j + (P / 64) * lane, f[j]);
  } else {
    if ((lane & (64 / P - 1)) == 0) put(lane / (64 / P), f[0]);
  }
}

// The same measurement for the drift watch of block runs (every first or second restart cycle: no copies around it): the
// coefficients travel as a kernel argument, the last workgroup to arrive writes the two sums to the pinned host words `host2`
// (visible when the kernel ends, i.e. before the publication of H behind it) and clears the accumulators for the next use.
template <class T> struct ProbeCoef { T c[kBlkKMax]; };
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_relation_watch(const T* __restrict__ V, int64_t ld, int nc, const T* __restrict__ w, const ProbeCoef<T> coef, int64_t n, int64_t stride,
                     double* __restrict__ acc /* [sum, rows, ticket] */, double* __restrict__ host2, double seq, const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  __shared__ double red[2][kBlock / 64];
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t row = (t >> 4) * stride * 16 + (t & 15);
  double r2 = 0.0, cnt = 0.0;
  if (row < n) {
    T a = w[row];
    for (int i = 0; i < nc; ++i) a = sub_(a, mul_(V[row + (int64_t)i * ld], coef.c[i]));
    r2 = abs2_(a);
    cnt = 1.0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { r2 += __shfl_xor(r2, off, 64); cnt += __shfl_xor(cnt, off, 64); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = r2; red[1][wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) { a += red[0][i]; b += red[1][i]; }
    atomicAdd(acc, a);
    atomicAdd(acc + 1, b);
    __threadfence();
    unsigned* ticket = reinterpret_cast<unsigned*>(acc + 2);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      __threadfence();
      host2[0] = atomicAdd(acc, 0.0);
      host2[1] = atomicAdd(acc + 1, 0.0);
      acc[0] = 0.0; acc[1] = 0.0; *ticket = 0u;
      __threadfence_system();
      host2[2] = seq;   // (last: the host takes the sums only when it finds its sequence number)
      __threadfence_system();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// z = sigma (y - theta x): the Newton-basis step after a plain operator product (operators without a fused form)
// ---------------------------------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_shift_scale(T* __restrict__ y, const T* __restrict__ x, T theta, double sigma, int64_t ld, const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  constexpr int R = Pack<T>::R;
  int64_t pb, pe;
  block_range(ld / R, blockIdx.x, gridDim.x, pb, pe);
  for (int64_t p = pb + threadIdx.x; p < pe; p += kBlock) {
    const auto yv = ld_pack(y + p * R);
    const auto xv = ld_pack(x + p * R);
    if constexpr (sizeof(T) == 8) {
      st_pack_nt(y + p * R, make_double2(sigma * (yv.x - theta * xv.x), sigma * (yv.y - theta * xv.y)));
    } else {
      const cd tx = mul_(theta, xv);
      st_pack_nt(y + p * R, cd{sigma * (yv.x - tx.x), sigma * (yv.y - tx.y)});
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Relation probe: out[0] += sum over SAMPLED rows of |w[r] - sum_{i < nc} V[r, i] coef[i]|^2, out[1] += sampled rows.
// w = A V[:, c], coef = H[0:nc, c]: the residual of the Arnoldi relation of ONE column, on every `stride`-th chunk of 16 rows
// (a violation of the relation is a multiple of a unit vector that the truncation dropped: a strided sample sees its share).
// Guard only (threshold decisions far from rounding): the order of the atomic additions does not matter.
// ---------------------------------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_relation_probe(const T* __restrict__ V, int64_t ld, int nc, const T* __restrict__ w, const T* __restrict__ coef, int64_t n, int64_t stride,
                     double* __restrict__ out) {
  __shared__ double red[2][kBlock / 64];
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t row = (t >> 4) * stride * 16 + (t & 15);
  double r2 = 0.0, cnt = 0.0;
  if (row < n) {
    T a = w[row];
    for (int i = 0; i < nc; ++i) a = sub_(a, mul_(V[row + (int64_t)i * ld], coef[i]));
    r2 = abs2_(a);
    cnt = 1.0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { r2 += __shfl_xor(r2, off, 64); cnt += __shfl_xor(cnt, off, 64); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = r2; red[1][wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < kBlock / 64; ++i) { a += red[0][i]; b += red[1][i]; }
    atomicAdd(out, a);
    atomicAdd(out + 1, b);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// BDOTS (pass 1):  partial[i*k + c][b] = sum_{rows of b} conj(S[r,c]) Z[r,i],   partial[k*S + g(i,i2)][b] = sum conj(Z[r,i]) Z[r,i2]
// S = V[:, 0:k), Z = V[:, k:k+S).
// ---------------------------------------------------------------------------------------------------------------------------
template <class T, int NCW, int S, int U, bool NT, int NW = 4, int WB = 1>
__global__ void __launch_bounds__(64 * NW, (blk_wps<T, NCW, S, U, NW>()))
    k_bdots(const T* __restrict__ V, int64_t ldv, int k, T* __restrict__ partial, int pnb, const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  constexpr int WA = NW / WB, SB = (S + WB - 1) / WB;
  constexpr int NG = S * (S + 1) / 2, NGW = (NG + NW - 1) / NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wa = wave % WA, wb = wave / WA;
  const T* colp[NCW];
  bool valid[NCW];
  T acc[NCW][SB];
  T gacc[NGW];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii) {
    const int c = wa + WA * ii;
    valid[ii] = c < k;
    colp[ii] = V + (int64_t)(valid[ii] ? c : 0) * ldv;
#pragma unroll
    for (int i = 0; i < SB; ++i) acc[ii][i] = zero_of(T{});
  }
#pragma unroll
  for (int g = 0; g < NGW; ++g) gacc[g] = zero_of(T{});
  const T* Z = V + (int64_t)k * ldv;
  int64_t pb, pe;
  block_range(ldv / R, blockIdx.x, gridDim.x, pb, pe);
  for (int64_t base = pb; base < pe; base += 64 * U) {
    int64_t r[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q = base + u * 64 + lane;
      ok[u] = q < pe;
      r[u] = (ok[u] ? q : pb) * R;
    }
    P v[NCW][U];
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      if (valid[ii]) {  // wave-uniform
#pragma unroll
        for (int u = 0; u < U; ++u) v[ii][u] = ld_v<NT>(colp[ii] + r[u]);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) v[ii][u] = zero_pack(T{});
      }
    }
    P z[S][U];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        z[i][u] = ld_pack(Z + (int64_t)i * ldv + r[u]);
        if (!ok[u]) z[i][u] = zero_pack(T{});
      }
    blk_by_idx<WB>(wb, [&](auto btag) {
      constexpr int B = decltype(btag)::value;
#pragma unroll
      for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
        for (int jj = 0; jj < SB; ++jj)
          if (B + WB * jj < S) {
#pragma unroll
            for (int u = 0; u < U; ++u) dotp(acc[ii][jj], v[ii][u], z[B + WB * jj][u]);
          }
    });
    // Gram entries: entry g = (i, i2), i <= i2, belongs to wave g % NW (static accumulator index g / NW)
    blk_by_idx<NW>(wave, [&](auto wtag) {
      constexpr int W = decltype(wtag)::value;
#pragma unroll
      for (int i2 = 0; i2 < S; ++i2)
#pragma unroll
        for (int i = 0; i <= i2; ++i) {
          const int g = gram_idx(i, i2);
          if ((g % NW) == W) {
#pragma unroll
            for (int u = 0; u < U; ++u) dotp(gacc[g / NW], z[i][u], z[i2][u]);
          }
        }
    });
  }
  constexpr int D = Dpe<T>::value;
  constexpr int NE = NCW * SB + NGW;
  constexpr int PD = next_pow2(NE * D);
  double f[PD];
#pragma unroll
  for (int e = 0; e < PD; ++e) f[e] = 0.0;
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) put_acc(f, ii * SB + i, acc[ii][i]);
#pragma unroll
  for (int g = 0; g < NGW; ++g) put_acc(f, NCW * SB + g, gacc[g]);
  fold_stage<PD, 32>(f, lane);
  store_folded<T, NCW, S, NGW, NW, WB>(f, lane, wave, k, partial, pnb);
}

// ---------------------------------------------------------------------------------------------------------------------------
// BUPDATE (pass 2):  Qt = Z R1inv - S coefp  (in place over Z);  partial[i*k + c] = conj(S[:,c]) . Qt[:,i];  Gram of Qt.
// coefp: k x S column-major (leading dimension ldc), r1inv: S x S upper triangular column-major (leading dimension S).
// Wave (wa, wb) forms, for the block columns i = wb (mod WB), the partial row sums
//   t[i] = sum_{own columns c = wa mod WA} S[r,c] coefp[c,i]  -  sum_{l = wa mod WA} Z[r,l] r1inv[l,i];
// the WA partial sums of every block column meet in LDS,  Qt[r,i] = -(t_0 + ... + t_{WA-1})[i]  (every wave forms all of Qt:
// its inner products need the columns of its class, its share of the Gram matrix arbitrary pairs).  Wave w stores the
// columns i = w (mod NW).
// ---------------------------------------------------------------------------------------------------------------------------
// Exchange buffer: double (one barrier per iteration) while one workgroup has the CU to itself; SINGLE (a second barrier per
// iteration) in the instantiations compiled for two resident workgroups, so that both fit the CU's 160 KiB of LDS.
// (Staging the written block in LDS and writing 16 KiB bursts per column -- what pays in k_axpy_dots_cs, which writes ONE
// column in 96 KiB bursts -- was measured here and is slower: 710 against 651 us at k = 21, s = 5; five columns leave no room
// for bursts of that size.)
template <class T, int NCW, int S, int U, bool NT, int NW = 4, int WB = 1>
__global__ void __launch_bounds__(64 * NW, (blk_wps<T, NCW, S, U, NW>()))
    k_bupdate(T* __restrict__ V, int64_t ldv, int k, const T* __restrict__ coefp, int ldc, const T* __restrict__ r1inv,
              T* __restrict__ partial, int pnb, const DevState* __restrict__ st, int dbg = 0) {
  if (st && st->breakdown >= 0) return;
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  constexpr int WA = NW / WB, SB = (S + WB - 1) / WB, SP = SB * WB;
  constexpr int NG = S * (S + 1) / 2, NGW = (NG + NW - 1) / NW;
  constexpr int NB = (blk_wpe<T, NCW, S, U, NW>() == 2 || 2 * (NW / WB) * U * S * 64 * (int)sizeof(typename Pack<T>::type) > 96 * 1024) ? 1 : 2;
  constexpr int NTHREADS = 64 * NW;
  __shared__ P tbuf[NB][WA][U][S][64];
  __shared__ T cf[WA * NCW][SP];  // coefp rows (columns of the basis) as the waves index them: cf[c][i]; zero beyond S
  __shared__ T ri[S][SP];         // r1inv[l][i], l <= i; zero elsewhere
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wa = wave % WA, wb = wave / WA;
  for (int e = threadIdx.x; e < WA * NCW * SP; e += NTHREADS) {
    const int c = e / SP, i = e % SP;
    cf[c][i] = (c < k && i < S) ? coefp[c + (int64_t)i * ldc] : zero_of(T{});
  }
  for (int e = threadIdx.x; e < S * SP; e += NTHREADS) {
    const int l = e % S, i = e / S;
    ri[l][i] = (l <= i && i < S) ? r1inv[l + i * S] : zero_of(T{});
  }
  __syncthreads();
  const T* colp[NCW];
  bool valid[NCW];
  T acc[NCW][SB];
  T gacc[NGW];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii) {
    const int c = wa + WA * ii;
    valid[ii] = c < k;
    colp[ii] = V + (int64_t)(valid[ii] ? c : 0) * ldv;
#pragma unroll
    for (int i = 0; i < SB; ++i) acc[ii][i] = zero_of(T{});
  }
#pragma unroll
  for (int g = 0; g < NGW; ++g) gacc[g] = zero_of(T{});
  T* Z = V + (int64_t)k * ldv;
  int64_t pb, pe;
  block_range(ldv / R, blockIdx.x, gridDim.x, pb, pe);
  int it = 0;
  for (int64_t base = pb; base < pe; base += 64 * U, ++it) {
    int64_t r[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q = base + u * 64 + lane;
      ok[u] = q < pe;
      r[u] = (ok[u] ? q : pb) * R;
    }
    P v[NCW][U];
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      if (valid[ii]) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[ii][u] = ld_v<NT>(colp[ii] + r[u]);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) v[ii][u] = zero_pack(T{});
      }
    }
    P t[U][SB];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < SB; ++i) t[u][i] = zero_pack(T{});
    // own Z columns (l = wa mod WA): t[i] -= Z_l r1inv[l, i], i >= l (r1inv is stored with zeros below the diagonal)
    blk_by_idx<WA>(wa, [&](auto atag) {
      constexpr int A = decltype(atag)::value;
#pragma unroll
      for (int l = 0; l < S; ++l) {
        if ((l % WA) == A) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const P zl = ld_pack(Z + (int64_t)l * ldv + r[u]);
#pragma unroll
            for (int jj = 0; jj < SB; ++jj)
              if (WB * jj + WB - 1 >= l) axpy_acc(t[u][jj], zl, neg_(ri[l][wb + WB * jj]));
          }
        }
      }
    });
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      if (valid[ii]) {
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) {
          const T g = cf[wa + WA * ii][wb + WB * jj];
#pragma unroll
          for (int u = 0; u < U; ++u) axpy_acc(t[u][jj], v[ii][u], g);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int jj = 0; jj < SB; ++jj)
        if (wb + WB * jj < S) tbuf[it & (NB - 1)][wa][u][wb + WB * jj][lane] = t[u][jj];
    __syncthreads();
    P q[U][S];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < S; ++i) {
        P sum = tbuf[it & (NB - 1)][0][u][i][lane];
#pragma unroll
        for (int a2 = 1; a2 < WA; ++a2) sum = addp(sum, tbuf[it & (NB - 1)][a2][u][i][lane]);
        P qq = sub_pack(zero_pack(T{}), sum);
        if (!ok[u]) qq = zero_pack(T{});
        q[u][i] = qq;
      }
    if (!(dbg & 1)) {  // (dbg & 1: timing probe without the write stream, & 4: cacheable stores)
      blk_by_idx<NW>(wave, [&](auto wtag) {
        constexpr int W = decltype(wtag)::value;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int i = 0; i < S; ++i)
            if ((i % NW) == W && ok[u]) {
              if (dbg & 4) st_pack(Z + (int64_t)i * ldv + r[u], q[u][i]);
              else st_pack_nt(Z + (int64_t)i * ldv + r[u], q[u][i]);
            }
      });
    }
    blk_by_idx<WB>(wb, [&](auto btag) {
      constexpr int B = decltype(btag)::value;
#pragma unroll
      for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
        for (int jj = 0; jj < SB; ++jj)
          if (B + WB * jj < S) {
#pragma unroll
            for (int u = 0; u < U; ++u) dotp(acc[ii][jj], v[ii][u], q[u][B + WB * jj]);
          }
    });
    blk_by_idx<NW>(wave, [&](auto wtag) {
      constexpr int W = decltype(wtag)::value;
#pragma unroll
      for (int i2 = 0; i2 < S; ++i2)
#pragma unroll
        for (int i = 0; i <= i2; ++i) {
          const int g = gram_idx(i, i2);
          if ((g % NW) == W) {
#pragma unroll
            for (int u = 0; u < U; ++u) dotp(gacc[g / NW], q[u][i], q[u][i2]);
          }
        }
    });
    if constexpr (NB == 1) __syncthreads();  // the single exchange buffer may be overwritten from here on
  }
  constexpr int D = Dpe<T>::value;
  constexpr int NE = NCW * SB + NGW;
  constexpr int PD = next_pow2(NE * D);
  double f[PD];
#pragma unroll
  for (int e = 0; e < PD; ++e) f[e] = 0.0;
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) put_acc(f, ii * SB + i, acc[ii][i]);
#pragma unroll
  for (int g = 0; g < NGW; ++g) put_acc(f, NCW * SB + g, gacc[g]);
  fold_stage<PD, 32>(f, lane);
  store_folded<T, NCW, S, NGW, NW, WB>(f, lane, wave, k, partial, pnb);
}

// ---------------------------------------------------------------------------------------------------------------------------
// RING forms (Float64, wide workgroups).  What bounds the register forms above once a wave keeps few columns: every wave
// loads every block column itself, and the L2 serves ~12 TB/s of requests whoever asks -- at s = 10 with eight waves that
// is four requests per byte of the basis (measured: 3.0 TB/s, unchanged by more loads in flight or by the load policy).
// Here every 16-byte pack is fetched from memory ONCE per workgroup, by an asynchronous global -> LDS copy
// (global_load_lds_dwordx4: no staging registers), into a ring of STAGES tiles of 64 packs x (k + s) columns; the waves
// read their operands from the tile (ds_read_b128) while the copies of the next tiles are in flight.  One workgroup
// barrier per tile; the copies are counted with s_waitcnt vmcnt(N) (the compiler does not see them: they are issued from
// inline assembly, and nothing else in the loop uses the vector-memory counter).
// ---------------------------------------------------------------------------------------------------------------------------
// one 1-KiB copy: lane l's 16 bytes at gsrc -> LDS byte address lds_dst + 16 l (M0 carries the wave-uniform base)
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_nt(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n <= 31, then the workgroup barrier
__device__ __forceinline__ void wait_vm_barrier(int n) {
#define KS_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  switch (n) {
    KS_VMW(0) KS_VMW(1) KS_VMW(2) KS_VMW(3) KS_VMW(4) KS_VMW(5) KS_VMW(6) KS_VMW(7) KS_VMW(8) KS_VMW(9) KS_VMW(10) KS_VMW(11)
    KS_VMW(12) KS_VMW(13) KS_VMW(14) KS_VMW(15) KS_VMW(16) KS_VMW(17) KS_VMW(18) KS_VMW(19) KS_VMW(20) KS_VMW(21) KS_VMW(22)
    KS_VMW(23) KS_VMW(24) KS_VMW(25) KS_VMW(26) KS_VMW(27) KS_VMW(28) KS_VMW(29) KS_VMW(30) KS_VMW(31)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef KS_VMW
}

// BDOTS, ring form: same result layout as k_bdots<double, NCW, S, ., ., NW, WB>.  Dynamic LDS: stages * (k + S) KiB.
template <int NCW, int S, int NW, int WB>
__global__ void __launch_bounds__(64 * NW, NW / 4)
    k_bdots_ring(const double* __restrict__ V, int64_t ldv, int k, int stages, double* __restrict__ partial, int pnb,
                 const DevState* __restrict__ st, int dbg, const double* __restrict__ zeros) {
  if (st && st->breakdown >= 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];
  constexpr int WA = NW / WB, SB = (S + WB - 1) / WB;
  constexpr int NG = S * (S + 1) / 2, NGW = (NG + NW - 1) / NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wa = wave % WA, wb = wave / WA;
  const int ncol = k + S;
  const double2* ring = reinterpret_cast<const double2*>(ring_raw);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring_raw;   // LDS byte address (low half of the flat address)
  double acc[NCW][SB];
  double gacc[NGW];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) acc[ii][i] = 0.0;
#pragma unroll
  for (int g = 0; g < NGW; ++g) gacc[g] = 0.0;
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);
  const int niter = (int)((pe - pb + 63) / 64);
  // copies of this wave per tile: columns wave, wave + NW, ... < ncol
  const int nl = (ncol - wave + NW - 1) / NW;
  auto issue = [&](int it, int sl) {  // tile `it` -> ring slot sl = it % stages (rows clamped to the workgroup's range: the tail is masked below)
    const int64_t q = pb + (int64_t)it * 64 + lane;
    const bool in = q < pe;              // packs past the workgroup's range are filled with zeros: nothing to mask afterwards
    const double* src = V + (in ? q : pb) * 2;
    const uint32_t slot = ring_lds + (uint32_t)(sl * ncol) * 1024u;
    if (dbg & 32) return;  // (probe: no copies)
    if (dbg & 64) { for (int j = wave; j < ncol; j += NW) glds16_nt(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
    else { for (int j = wave; j < ncol; j += NW) glds16(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
  };
  for (int it = 0; it < stages - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = stages - 1;   // it % stages, (it + stages - 1) % stages without the division
  for (int it = 0; it < niter; ++it) {
    // tile `it` has landed when at most (stages - 2) later tiles of this wave are outstanding; behind the barrier everybody's
    // part of it is there and everybody has finished tile it - 1, whose slot the next copies overwrite
    wait_vm_barrier((stages - 2) * nl);
    issue(it + stages - 1, sl_new);
    const double2* tile = ring + (size_t)(sl_cur * ncol) * 64;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == stages ? 0 : sl_cur + 1;
    if (dbg & 16) continue;  // (probe: copies only)
    const bool ok = pb + (int64_t)it * 64 + lane < pe;
    double2 v[NCW];
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      const int c = wa + WA * ii;
      v[ii] = c < k ? tile[c * 64 + lane] : make_double2(0.0, 0.0);
    }
    double2 z[S];
#pragma unroll
    for (int i = 0; i < S; ++i) z[i] = tile[(k + i) * 64 + lane];
    blk_by_idx<WB>(wb, [&](auto btag) {
      constexpr int B = decltype(btag)::value;
#pragma unroll
      for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
        for (int jj = 0; jj < SB; ++jj)
          if (B + WB * jj < S) dotp(acc[ii][jj], v[ii], z[B + WB * jj]);
    });
    blk_by_idx<NW>(wave, [&](auto wtag) {
      constexpr int W = decltype(wtag)::value;
#pragma unroll
      for (int i2 = 0; i2 < S; ++i2)
#pragma unroll
        for (int i = 0; i <= i2; ++i) {
          const int g = gram_idx(i, i2);
          if ((g % NW) == W) dotp(gacc[g / NW], z[i], z[i2]);
        }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (copies of tiles past the end are still in flight)
  constexpr int NE = NCW * SB + NGW;
  constexpr int PD = next_pow2(NE);
  double f[PD];
#pragma unroll
  for (int e = 0; e < PD; ++e) f[e] = 0.0;
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) f[ii * SB + i] = acc[ii][i];
#pragma unroll
  for (int g = 0; g < NGW; ++g) f[NCW * SB + g] = gacc[g];
  fold_stage<PD, 32>(f, lane);
  store_folded<double, NCW, S, NGW, NW, WB>(f, lane, wave, k, partial, pnb);
}

// 16-byte streaming store from inline assembly (counted by hand on the vector-memory counter, like the copies)
typedef double blk_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gst16_nt(double* p, double2 v) {
  blk_d2v w;
  w.x = v.x;
  w.y = v.y;
  asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void lgkm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BUPDATE, ring form: same result as k_bupdate<double, NCW, S, ., ., NW, WB>.  Per tile of 64 packs:
//   A  tile landed (counted wait + barrier); copies of tile it + stages - 1 issued
//      partial row sums t over the wave's columns (operands from the tile) -> tbuf            barrier B
//   1  every block column is finished by ONE wave: Qt_i = -(sum of its WA partials), stored to memory and to qbuf
//                                                                                                barrier C
//   2  inner products of the wave's columns with the block columns of its class, its share of the Gram triangle (qbuf)
// Dynamic LDS: [ring: stages x (k + S) KiB | tbuf: WA x S KiB | qbuf: S KiB | coefficients].
template <int NCW, int S, int NW, int WB>
__global__ void __launch_bounds__(64 * NW, NW / 4)
    k_bupdate_ring(double* __restrict__ V, int64_t ldv, int k, int stages, const double* __restrict__ coefp, int ldc,
                   const double* __restrict__ r1inv, double* __restrict__ partial, int pnb, const DevState* __restrict__ st, int dbg,
                   const double* __restrict__ zeros) {
  if (st && st->breakdown >= 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];
  constexpr int WA = NW / WB, SB = (S + WB - 1) / WB, SP = SB * WB;
  constexpr int NG = S * (S + 1) / 2, NGW = (NG + NW - 1) / NW;
  constexpr int NTHREADS = 64 * NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wa = wave % WA, wb = wave / WA;
  const int ncol = k + S;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring_raw;
  double2* ring = reinterpret_cast<double2*>(ring_raw);
  double2* tbuf = ring + (size_t)stages * ncol * 64;          // [WA][S][64]
  double2* qbuf = tbuf + (size_t)WA * S * 64;                 // [S][64]
  double* cf = reinterpret_cast<double*>(qbuf + (size_t)S * 64);  // [WA * NCW][SP]
  double* ri = cf + WA * NCW * SP;                            // [S][SP]
  for (int e = threadIdx.x; e < WA * NCW * SP; e += NTHREADS) {
    const int c = e / SP, i = e % SP;
    cf[e] = (c < k && i < S) ? coefp[c + (int64_t)i * ldc] : 0.0;
  }
  for (int e = threadIdx.x; e < S * SP; e += NTHREADS) {
    const int l = e / SP, i = e % SP;
    ri[e] = (l <= i && i < S) ? r1inv[l + i * S] : 0.0;
  }
  lgkm_barrier();
  double acc[NCW][SB];
  double gacc[NGW];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) acc[ii][i] = 0.0;
#pragma unroll
  for (int g = 0; g < NGW; ++g) gacc[g] = 0.0;
  double* Z = V + (int64_t)k * ldv;
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);
  const int niter = (int)((pe - pb + 63) / 64);
  const int nl = (ncol - wave + NW - 1) / NW;                 // copies of this wave per tile
  // block columns this wave finishes (phase 1): i = wb + WB jj with jj % WA == wa
  int nst = 0;
#pragma unroll
  for (int jj = 0; jj < SB; ++jj)
    if ((jj % WA) == wa && wb + WB * jj < S) ++nst;
  if (dbg & 1) nst = 0;
  auto issue = [&](int it, int sl) {
    const int64_t q = pb + (int64_t)it * 64 + lane;
    const bool in = q < pe;              // packs past the workgroup's range are filled with zeros: nothing to mask afterwards
    const double* src = V + (in ? q : pb) * 2;
    const uint32_t slot = ring_lds + (uint32_t)(sl * ncol) * 1024u;
    if (dbg & 64) { for (int j = wave; j < ncol; j += NW) glds16_nt(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
    else { for (int j = wave; j < ncol; j += NW) glds16(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
  };
  // inner products of tile `it - 1` (its basis packs are still in registers, its block columns in qbuf) run in the same
  // barrier interval as the partial row sums of tile `it`: two independent instruction streams per wave, two barriers per tile
  auto phase2 = [&](const double2* vp) {
    double2 q[S];
#pragma unroll
    for (int i = 0; i < S; ++i) q[i] = qbuf[i * 64 + lane];
    blk_by_idx<WB>(wb, [&](auto btag) {
      constexpr int B = decltype(btag)::value;
#pragma unroll
      for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
        for (int jj = 0; jj < SB; ++jj)
          if (B + WB * jj < S) dotp(acc[ii][jj], vp[ii], q[B + WB * jj]);
    });
    blk_by_idx<NW>(wave, [&](auto wtag) {
      constexpr int W = decltype(wtag)::value;
#pragma unroll
      for (int i2 = 0; i2 < S; ++i2)
#pragma unroll
        for (int i = 0; i <= i2; ++i) {
          const int g = gram_idx(i, i2);
          if ((g % NW) == W) dotp(gacc[g / NW], q[i], q[i2]);
        }
    });
  };
  double2 vprev[NCW];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii) vprev[ii] = make_double2(0.0, 0.0);
  for (int it = 0; it < stages - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = stages - 1;   // it % stages, (it + stages - 1) % stages
  for (int it = 0; it < niter; ++it) {
    // issue order per tile: nl copies, then nst stores; tile `it` was copied (stages - 1) tiles ago, behind it are the
    // copies of stages - 2 tiles and the stores of the last min(it, stages - 1) tiles.  Behind this barrier: tile `it` is in
    // LDS, qbuf holds the block columns of tile it - 1, tbuf and the ring slot of tile it - 1 are free.
    wait_vm_barrier((stages - 2) * nl + (it < stages - 1 ? it : stages - 1) * nst);
    issue(it + stages - 1, sl_new);
    const double2* tile = ring + (size_t)(sl_cur * ncol) * 64;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == stages ? 0 : sl_cur + 1;
    const int64_t qi = pb + (int64_t)it * 64 + lane;
    const bool ok = qi < pe;
    double2 v[NCW];
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      const int c = wa + WA * ii;
      v[ii] = c < k ? tile[c * 64 + lane] : make_double2(0.0, 0.0);
    }
    double2 t[SB];
#pragma unroll
    for (int jj = 0; jj < SB; ++jj) t[jj] = make_double2(0.0, 0.0);
    blk_by_idx<WA>(wa, [&](auto atag) {
      constexpr int A = decltype(atag)::value;
#pragma unroll
      for (int l = 0; l < S; ++l) {
        if ((l % WA) == A) {
          const double2 zl = tile[(k + l) * 64 + lane];
#pragma unroll
          for (int jj = 0; jj < SB; ++jj)
            if (WB * jj + WB - 1 >= l) axpy_acc(t[jj], zl, -ri[l * SP + wb + WB * jj]);
        }
      }
    });
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      if (wa + WA * ii < k) {
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) axpy_acc(t[jj], v[ii], cf[(wa + WA * ii) * SP + wb + WB * jj]);
      }
    }
#pragma unroll
    for (int jj = 0; jj < SB; ++jj)
      if (wb + WB * jj < S) tbuf[(wa * S + wb + WB * jj) * 64 + lane] = t[jj];
    if (it > 0) phase2(vprev);
    lgkm_barrier();
    // phase 1: every block column is finished by one wave
    blk_by_idx<WA>(wa, [&](auto atag) {
      constexpr int A = decltype(atag)::value;
#pragma unroll
      for (int jj = 0; jj < SB; ++jj)
        if ((jj % WA) == A) {
          const int i = wb + WB * jj;
          if (i < S) {
            double2 sum = tbuf[(0 * S + i) * 64 + lane];
#pragma unroll
            for (int a2 = 1; a2 < WA; ++a2) sum = addp(sum, tbuf[(a2 * S + i) * 64 + lane]);
            const double2 qq = make_double2(-sum.x, -sum.y);   // (zero in rows past the end: their operands are)
            qbuf[i * 64 + lane] = qq;
            if (ok && !(dbg & 1)) gst16_nt(Z + (int64_t)i * ldv + qi * 2, qq);
          }
        }
    });
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) vprev[ii] = v[ii];
  }
  lgkm_barrier();
  if (niter > 0) phase2(vprev);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  constexpr int NE = NCW * SB + NGW;
  constexpr int PD = next_pow2(NE);
  double f[PD];
#pragma unroll
  for (int e = 0; e < PD; ++e) f[e] = 0.0;
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) f[ii * SB + i] = acc[ii][i];
#pragma unroll
  for (int g = 0; g < NGW; ++g) f[NCW * SB + g] = gacc[g];
  fold_stage<PD, 32>(f, lane);
  store_folded<double, NCW, S, NGW, NW, WB>(f, lane, wave, k, partial, pnb);
}

// ---------------------------------------------------------------------------------------------------------------------------
// LARGE blocks (s = 20: one block per restart cycle at 20 / 40).  Same ring of tiles, two changes that keep a wave inside 256
// registers: operands are read from the tile when they are used (five block columns at a time) instead of being held for the
// whole tile, and the Gram triangle (210 entries) is dealt in 5 x 5 BLOCKS -- six off-diagonal ones (25 entries, ten columns to
// read) to waves 0-5, the four diagonal ones (15 entries each) in pairs to waves 6 and 7 -- instead of entry by entry.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kBlkL = 20;   // block size of the large form
constexpr int kBlkLG = 5;   // columns per Gram group
// group pair (a, b), a < b, of the off-diagonal block wave w < 6 owns
__host__ __device__ __forceinline__ constexpr int blkl_a(int w) { return w < 3 ? 0 : (w < 5 ? 1 : 2); }
__host__ __device__ __forceinline__ constexpr int blkl_b(int w) { return w < 3 ? w + 1 : (w < 5 ? w - 1 : 3); }
// accumulator e of wave w -> Gram entry (i, i2), i <= i2 (false: padding)
__device__ __forceinline__ bool blkl_gram_entry(int w, int e, int& i, int& i2) {
  if (w < 6) {
    if (e >= 25) return false;
    i = kBlkLG * blkl_a(w) + e / kBlkLG;
    i2 = kBlkLG * blkl_b(w) + e % kBlkLG;
    return true;
  }
  if (e >= 30) return false;
  const int g = 2 * (w - 6) + e / 15, t = e % 15;
  int q = 0;
  while ((q + 1) * (q + 2) / 2 <= t) ++q;
  const int pp = t - q * (q + 1) / 2;
  i = kBlkLG * g + pp;
  i2 = kBlkLG * g + q;
  return true;
}
// per-lane values -> wave totals -> put(index, total): P a power of two
template <int P, class PUT> __device__ __forceinline__ void fold_put(double* f, int lane, PUT&& put) {
  fold_stage<P, 32>(f, lane);
  if constexpr (P >= 64) {
#pragma unroll
    for (int j = 0; j < P / 64; ++j) put(j + (P / 64) * lane, f[j]);
  } else {
    if ((lane & (64 / P - 1)) == 0) put(lane / (64 / P), f[0]);
  }
}
// Gram accumulation of wave W from a column source rd(i) -> double2 (the block's columns of this lane's pack)
template <int W, class RD> __device__ __forceinline__ void blkl_gram(double* gacc, RD&& rd) {
  if constexpr (W < 6) {
    constexpr int A = blkl_a(W), B = blkl_b(W);
    double2 za[kBlkLG], zb[kBlkLG];
#pragma unroll
    for (int p2 = 0; p2 < kBlkLG; ++p2) { za[p2] = rd(kBlkLG * A + p2); zb[p2] = rd(kBlkLG * B + p2); }
#pragma unroll
    for (int p2 = 0; p2 < kBlkLG; ++p2)
#pragma unroll
      for (int q2 = 0; q2 < kBlkLG; ++q2) dotp(gacc[p2 * kBlkLG + q2], za[p2], zb[q2]);
  } else {
#pragma unroll
    for (int d2 = 0; d2 < 2; ++d2) {
      constexpr int G0 = 2 * (W - 6);
      double2 za[kBlkLG];
#pragma unroll
      for (int p2 = 0; p2 < kBlkLG; ++p2) za[p2] = rd(kBlkLG * (G0 + d2) + p2);
#pragma unroll
      for (int q2 = 0; q2 < kBlkLG; ++q2)
#pragma unroll
        for (int p2 = 0; p2 <= q2; ++p2) dotp(gacc[15 * d2 + q2 * (q2 + 1) / 2 + p2], za[p2], za[q2]);
    }
  }
}

// BDOTS, large form: waves as a 4 x 2 grid (NCW = ceil(k / 4) <= 6, ten block columns per class)
template <int NCW>
__global__ void __launch_bounds__(512, 2)
    k_bdots_ringL(const double* __restrict__ V, int64_t ldv, int k, int stages, double* __restrict__ partial, int pnb,
                  const DevState* __restrict__ st, int dbg, const double* __restrict__ zeros) {
  if (st && st->breakdown >= 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];
  constexpr int S = kBlkL, NW = 8, WB = 2, WA = 4, SB = S / WB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wa = wave % WA, wb = wave / WA;
  const int ncol = k + S;
  const double2* ring = reinterpret_cast<const double2*>(ring_raw);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring_raw;
  double acc[NCW][SB];
  double gacc[30];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) acc[ii][i] = 0.0;
#pragma unroll
  for (int g = 0; g < 30; ++g) gacc[g] = 0.0;
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);
  const int niter = (int)((pe - pb + 63) / 64);
  const int nl = (ncol - wave + NW - 1) / NW;
  auto issue = [&](int it, int sl) {
    const int64_t q = pb + (int64_t)it * 64 + lane;
    const bool in = q < pe;              // packs past the workgroup's range are filled with zeros: nothing to mask afterwards
    const double* src = V + (in ? q : pb) * 2;
    const uint32_t slot = ring_lds + (uint32_t)(sl * ncol) * 1024u;
    if (dbg & 64) { for (int j = wave; j < ncol; j += NW) glds16_nt(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
    else { for (int j = wave; j < ncol; j += NW) glds16(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
  };
  for (int it = 0; it < stages - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = stages - 1;
  for (int it = 0; it < niter; ++it) {
    wait_vm_barrier((stages - 2) * nl);
    issue(it + stages - 1, sl_new);
    const double2* tile = ring + (size_t)(sl_cur * ncol) * 64;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == stages ? 0 : sl_cur + 1;
    const bool ok = pb + (int64_t)it * 64 + lane < pe;
    auto rdz = [&](int i) { return tile[(k + i) * 64 + lane]; };
#pragma unroll
    for (int jb = 0; jb < SB / kBlkLG; ++jb) {
      double2 z5[kBlkLG];
#pragma unroll
      for (int j = 0; j < kBlkLG; ++j) z5[j] = rdz(wb + WB * (kBlkLG * jb + j));
#pragma unroll
      for (int ii = 0; ii < NCW; ++ii) {
        const int c = wa + WA * ii;
        if (c < k) {  // uniform
          const double2 v = tile[c * 64 + lane];
#pragma unroll
          for (int j = 0; j < kBlkLG; ++j) dotp(acc[ii][kBlkLG * jb + j], v, z5[j]);
        }
      }
    }
    blk_by_idx<NW>(wave, [&](auto wtag) { blkl_gram<decltype(wtag)::value>(gacc, rdz); });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    constexpr int P = next_pow2(NCW * SB);
    double f[P];
#pragma unroll
    for (int e = 0; e < P; ++e) f[e] = 0.0;
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
      for (int i = 0; i < SB; ++i) f[ii * SB + i] = acc[ii][i];
    fold_put<P>(f, lane, [&](int e, double v) {
      if (e < NCW * SB) {
        const int c = wa + WA * (e / SB), i = wb + WB * (e % SB);
        if (c < k) partial[(int64_t)(i * k + c) * pnb + blockIdx.x] = v;
      }
    });
  }
  {
    double f[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) f[e] = e < 30 ? gacc[e] : 0.0;
    fold_put<32>(f, lane, [&](int e, double v) {
      int i, i2;
      if (blkl_gram_entry(wave, e, i, i2)) partial[(int64_t)(k * S + gram_idx(i, i2)) * pnb + blockIdx.x] = v;
    });
  }
}

// BUPDATE, large form: waves as a 2 x 4 grid (NCW = ceil(k / 2) <= 12, five block columns per class); three barriers per tile
// (partial row sums -> every block column finished by one wave -> inner products); two tiles in the ring.
// Dynamic LDS: [ring: 2 x (k + 20) KiB | tbuf: 2 x 20 KiB | qbuf: 20 KiB | coefficients class-major].
template <int NCW>
__global__ void __launch_bounds__(512, 2)
    k_bupdate_ringL(double* __restrict__ V, int64_t ldv, int k, int stages, const double* __restrict__ coefp, int ldc,
                    const double* __restrict__ r1inv, double* __restrict__ partial, int pnb, const DevState* __restrict__ st, int dbg,
                    const double* __restrict__ zeros) {
  if (st && st->breakdown >= 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];
  constexpr int S = kBlkL, NW = 8, WB = 4, WA = 2, SB = S / WB;   // SB = 5 = kBlkLG
  constexpr int NTHREADS = 64 * NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wa = wave % WA, wb = wave / WA;
  const int ncol = k + S;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring_raw;
  double2* ring = reinterpret_cast<double2*>(ring_raw);
  double2* tbuf = ring + (size_t)stages * ncol * 64;              // [WA][S][64]
  double2* qbuf = tbuf + (size_t)WA * S * 64;                     // [S][64]
  double* cf = reinterpret_cast<double*>(qbuf + (size_t)S * 64);  // [WA * NCW][WB][SB]: coefficient (c, i = wb + WB jj)
  double* ri = cf + WA * NCW * S;                                 // [S][WB][SB]: r1inv[l, i = wb + WB jj], zero below the diagonal
  for (int e = threadIdx.x; e < WA * NCW * S; e += NTHREADS) {
    const int c = e / S, r = e % S, b = r / SB, jj = r % SB, i = b + WB * jj;
    cf[e] = c < k ? coefp[c + (int64_t)i * ldc] : 0.0;
  }
  for (int e = threadIdx.x; e < S * S; e += NTHREADS) {
    const int l = e / S, r = e % S, b = r / SB, jj = r % SB, i = b + WB * jj;
    ri[e] = l <= i ? r1inv[l + i * S] : 0.0;
  }
  lgkm_barrier();
  double acc[NCW][SB];
  double gacc[30];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < SB; ++i) acc[ii][i] = 0.0;
#pragma unroll
  for (int g = 0; g < 30; ++g) gacc[g] = 0.0;
  double* Z = V + (int64_t)k * ldv;
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);
  const int niter = (int)((pe - pb + 63) / 64);
  const int nl = (ncol - wave + NW - 1) / NW;
  // block columns this wave finishes: i = wb + WB jj with jj % WA == wa
  int nst = (SB - wa + WA - 1) / WA;
  if (dbg & 1) nst = 0;
  auto issue = [&](int it, int sl) {
    const int64_t q = pb + (int64_t)it * 64 + lane;
    const bool in = q < pe;              // packs past the workgroup's range are filled with zeros: nothing to mask afterwards
    const double* src = V + (in ? q : pb) * 2;
    const uint32_t slot = ring_lds + (uint32_t)(sl * ncol) * 1024u;
    if (dbg & 64) { for (int j = wave; j < ncol; j += NW) glds16_nt(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
    else { for (int j = wave; j < ncol; j += NW) glds16(in ? src + (int64_t)j * ldv : zeros, slot + (uint32_t)j * 1024u); }
  };
  for (int it = 0; it < stages - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = stages - 1;
  for (int it = 0; it < niter; ++it) {
    wait_vm_barrier((stages - 2) * nl + (it < stages - 1 ? it : stages - 1) * nst);
    issue(it + stages - 1, sl_new);
    const double2* tile = ring + (size_t)(sl_cur * ncol) * 64;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == stages ? 0 : sl_cur + 1;
    const int64_t qi = pb + (int64_t)it * 64 + lane;
    const bool ok = qi < pe;
    // partial row sums over the wave's columns, block columns of its class
    double2 t[SB];
#pragma unroll
    for (int jj = 0; jj < SB; ++jj) t[jj] = make_double2(0.0, 0.0);
    blk_by_idx<WA>(wa, [&](auto atag) {
      constexpr int A = decltype(atag)::value;
#pragma unroll
      for (int l = 0; l < S; ++l) {
        if ((l % WA) == A) {
          const double2 zl = tile[(k + l) * 64 + lane];
          const double* rr = ri + (l * WB + wb) * SB;
#pragma unroll
          for (int jj = 0; jj < SB; ++jj)
            if (WB * jj + WB - 1 >= l) axpy_acc(t[jj], zl, -rr[jj]);
        }
      }
    });
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      const int c = wa + WA * ii;
      if (c < k) {  // uniform
        const double2 v = tile[c * 64 + lane];
        const double* cc = cf + (c * WB + wb) * SB;
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) axpy_acc(t[jj], v, cc[jj]);
      }
    }
#pragma unroll
    for (int jj = 0; jj < SB; ++jj) tbuf[(wa * S + wb + WB * jj) * 64 + lane] = t[jj];
    lgkm_barrier();
    // every block column is finished by one wave
    blk_by_idx<WA>(wa, [&](auto atag) {
      constexpr int A = decltype(atag)::value;
#pragma unroll
      for (int jj = 0; jj < SB; ++jj)
        if ((jj % WA) == A) {
          const int i = wb + WB * jj;
          double2 sum = tbuf[(0 * S + i) * 64 + lane];
#pragma unroll
          for (int a2 = 1; a2 < WA; ++a2) sum = addp(sum, tbuf[(a2 * S + i) * 64 + lane]);
          const double2 qq = make_double2(-sum.x, -sum.y);   // (zero in rows past the end: their operands are)
          qbuf[i * 64 + lane] = qq;
          if (ok && !(dbg & 1)) gst16_nt(Z + (int64_t)i * ldv + qi * 2, qq);
        }
    });
    lgkm_barrier();
    // inner products: own columns x block columns of the class; the wave's block(s) of the Gram triangle
    {
      double2 q5[SB];
#pragma unroll
      for (int jj = 0; jj < SB; ++jj) q5[jj] = qbuf[(wb + WB * jj) * 64 + lane];
#pragma unroll
      for (int ii = 0; ii < NCW; ++ii) {
        const int c = wa + WA * ii;
        if (c < k) {
          const double2 v = tile[c * 64 + lane];
#pragma unroll
          for (int jj = 0; jj < SB; ++jj) dotp(acc[ii][jj], v, q5[jj]);
        }
      }
    }
    auto rdq = [&](int i) { return qbuf[i * 64 + lane]; };
    blk_by_idx<NW>(wave, [&](auto wtag) { blkl_gram<decltype(wtag)::value>(gacc, rdq); });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    constexpr int P = next_pow2(NCW * SB);
    double f[P];
#pragma unroll
    for (int e = 0; e < P; ++e) f[e] = 0.0;
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
      for (int i = 0; i < SB; ++i) f[ii * SB + i] = acc[ii][i];
    fold_put<P>(f, lane, [&](int e, double v) {
      if (e < NCW * SB) {
        const int c = wa + WA * (e / SB), i = wb + WB * (e % SB);
        if (c < k) partial[(int64_t)(i * k + c) * pnb + blockIdx.x] = v;
      }
    });
  }
  {
    double f[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) f[e] = e < 30 ? gacc[e] : 0.0;
    fold_put<32>(f, lane, [&](int e, double v) {
      int i, i2;
      if (blkl_gram_entry(wave, e, i, i2)) partial[(int64_t)(k * S + gram_idx(i, i2)) * pnb + blockIdx.x] = v;
    });
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// FIN_BLK: reduction of the block's partial sums + the small algebra, one launch per stage.  Workgroup e reduces entry e
// (k*s inner products + s(s+1)/2 Gram entries), the LAST workgroup to arrive (device-scope counter, as k_fin_step_t) does
// the algebra with 256 threads.  Everything the algebra touches is O((k + s)^2 s): a few microseconds.
// ---------------------------------------------------------------------------------------------------------------------------
template <class T> struct BlkShifts {
  T theta[kBlkSMax];
  double sigma[kBlkSMax];
};

// device-resident scratch of a batch of blocks (one per workspace)
// IN-CHAIN DEFLATION (round 6; tools/model_deflated_chain.py).  The Newton chain of a block is orthogonalised against the basis only
// at the END of the block.  With locked Schur vectors U = V[:, 0:nd) of DOMINANT eigenvalues (:LM problems with outliers) that is
// too late: A z has components along U through the non-normal coupling even though z is orthogonal to U, every further step
// multiplies them by |lambda_locked| / |lambda_rest|, and a shift AT the locked value puts -theta z_{i-1} (a vector of the basis)
// on top of every later column -- either way the projected columns come out parallel (pivot <= 0 at the 5th-7th column on the
// operator of test/partial_schur.jl:122-138; cond(R_1) 3e14 in the model).  Cure: no shift at locked values, and every chain
// vector is projected against the (few) locked columns as soon as it exists:
//     z_i <- z_i - U c_i,  c_i = U^H z_i        (z_i = sigma_i (A - theta_i) z_{i-1} as the operator kernel wrote it)
// so that  A z_{i-1} = z_i / sigma_i + theta_i z_{i-1} + U c_i / sigma_i  -- the H recovery of k_fin_blk adds c_i / sigma_i to the
// locked rows.  Two launches per step, nd + 1 columns read twice: k_defl_dots (partial sums per workgroup) and k_defl_apply
// (every workgroup sums the partials itself, updates its rows; workgroup 0 stores c_i for k_fin_blk).
constexpr int kDeflMax = 16;
__device__ __forceinline__ double shfl_xor_(double v, int off) { return __shfl_xor(v, off, 64); }
__device__ __forceinline__ cd shfl_xor_(cd v, int off) { return cd{__shfl_xor(v.x, off, 64), __shfl_xor(v.y, off, 64)}; }

template <class T> struct BlkScratch {
  T P[kBlkKMax * kBlkSMax];        // true coordinates of Z in V_k (k x s, column stride k)
  T R1[kBlkSMax * kBlkSMax];       // stage-1 triangular factor (s x s, column stride s)
  T coefp[kBlkKMax * kBlkSMax];    // (T P) R1^-1: what k_bupdate subtracts (k x s, column stride k)
  T r1inv[kBlkSMax * kBlkSMax];
  T u[kBlkKMax + kBlkSMax];        // coordinates of the last stored column in the true basis (valid after a block)
  T cdefl[kBlkSMax * kDeflMax];    // in-chain deflation (k_defl_apply): U^H z_i of chain step i, column stride kDeflMax
};

__device__ __forceinline__ double inv_(double a) { return 1.0 / a; }
__device__ __forceinline__ cd inv_(cd a) {
  const double d = 1.0 / fma(a.x, a.x, a.y * a.y);
  return cd{a.x * d, -a.y * d};
}

__device__ __forceinline__ double bcast_(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ cd bcast_(cd v, int src) { return cd{__shfl(v.x, src, 64), __shfl(v.y, src, 64)}; }
// the same with a COMPILE-TIME source lane: v_readlane_b32 (a scalar register, no LDS round trip)
template <int SRC> __device__ __forceinline__ double rdlane_(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), SRC), hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), SRC);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int SRC> __device__ __forceinline__ cd rdlane_(cd v) { return cd{rdlane_<SRC>(v.x), rdlane_<SRC>(v.y)}; }

// Cholesky R^H R = G of an s x s Hermitian matrix given by its upper triangle in LDS (column stride s) AND the inverse of the
// factor: on exit the upper triangle of G holds R and X = R^-1 (full s x s, zeros below the diagonal).  All threads of the
// workgroup call it; returns the smallest pivot ratio d_i / G_ii (<= 0 when the matrix is not positive definite); the
// factorisation stops (and the caller bails) at the first ratio <= pivmin.
// One ENTRY of the upper triangle per thread (s (s + 1) / 2 <= 210 of the 256), the entry in a register, a compact loop over the
// s steps with ONE barrier each:
//   factorisation (right-looking): the owners of row p + 1 publish their updated, still unscaled entries; after the barrier every
//     thread reads the pivot d and the two entries of the pivot row it needs, forms 1 / sqrt(d) itself (hardware estimate + two
//     Newton steps: full double precision, not correctly rounded -- every thread and every rank runs the same code on the same
//     numbers, so the replicated decisions stay identical; the pivot test by cross-multiplication) and subtracts the outer
//     product from its entry;
//   inverse: the same row operations applied to the identity leave R^-H (the elimination [G | I] -> [R | R^-H]): entry (l, q) of it
//     is touched in steps q .. l, entry (q, l) of G in steps 0 .. q -- the thread that owns the one owns the other, the pivot row
//     of the identity part is published beside the pivot row of G, and the inverse costs no step and no barrier of its own.
// A step is an LDS round trip, ~12 dependent FP64 operations and a barrier: ~260 ns measured, 5 us at s = 20.  (Two variants
// that lost: the back substitution as a second loop of s steps -- 10.5 us for the pair --, and column c of R^-1 as dot products
// with column c of R in the shadow of step c + 1 -- the dot of up to 19 terms is LDS round trips in a row, longer than the step.)
// (What came before, for the record -- profiles/r06_fin_blk_timing.txt: wave 0 alone with a column per lane in registers, the
// rows travelling by v_readlane, the s steps unrolled: 17 us for the pair at s = 20, right-looking or left-looking, one uniform
// branch per entry or none, broadcasts batched ahead of the fmas or not.  ONE wave running ~20 KB of straight-line 8-byte
// instructions once is bound by its instruction fetch -- ~1 byte per cycle measured --, not by the arithmetic or the latencies.)
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * fma(-0.5 * d * y, y, 1.5);
  y = y * fma(-0.5 * d * y, y, 1.5);
  return y;
}
template <class T> __device__ double chol_inv_coop(T* G, T* X, int s, double pivmin) {
  constexpr int SMX = blk_smax<T>();
  static_assert(SMX * (SMX + 1) / 2 <= kBlock, "one entry of the upper triangle per thread");
  __shared__ T urow[2][SMX];        // the pivot row of G, unscaled (double-buffered over the steps)
  __shared__ T yrow[2][SMX];        // the pivot row of the eliminated identity, unscaled
  __shared__ double g0[SMX];
  const int tid = threadIdx.x, ng = s * (s + 1) / 2;
  int l = 0;                        // this thread's entry (q, l), q <= l
  while ((l + 1) * (l + 2) / 2 <= tid) ++l;
  const int q = tid - l * (l + 1) / 2;
  const bool mine = tid < ng;
  T g = mine ? G[q + l * s] : zero_of(T{});
  T y = from_real(mine && q == l ? 1.0 : 0.0, T{});   // entry (l, q) of the eliminated identity
  if (mine && q == l) g0[l] = real_of(g);
  if (mine && q == 0) urow[0][l] = g;
  if (tid == 0) yrow[0][0] = y;
  for (int e = tid; e < s * s; e += kBlock)
    if (e % s > e / s) X[e] = zero_of(T{});
  __syncthreads();
  double wd = 1.0, wg = 1.0;   // the smallest pivot ratio seen so far, as a fraction wd / wg
  for (int p = 0; p < s; ++p) {
    const T* u = urow[p & 1];
    const T* yv = yrow[p & 1];
    // ONE update  t -= conj(a) b / d  per thread and step, the operands picked by address (a wave that ran the two kinds of update
    // as the two sides of a branch paid both latencies in a row: 470 ns per step instead of 260):
    //   q > p:        entry (q, l) of G:         a = G[p, q], b = G[p, l]
    //   q <= p < l:   entry (l, q) of R^-H:      a = G[p, l], b = Y[p, q]
    const bool up = q > p;
    const T a_in = mine ? (up ? u[q] : u[l]) : u[p], b_in = mine ? (up ? u[l] : yv[q <= p ? q : 0]) : u[p];
    const double di = real_of(u[p]), gi = g0[p];
    const bool pos = gi > 0.0 && di > pivmin * gi;   // ratio = di / gi > pivmin
    if (!(gi > 0.0)) { wd = 0.0; wg = 1.0; }
    else if (di * wg < wd * gi) { wd = di; wg = gi; }
    if (!pos) break;   // (uniform: every thread reads the same numbers)
    const double rinv = rsqrt_nr(di);
    const T A = scl(a_in, rinv), B = scl(b_in, rinv);
    const T r = sub_(up ? g : y, mul_(conj_(A), B));
    if (mine) {
      if (up) g = r;
      else if (l > p) y = r;
      if (q == p + 1) urow[(p + 1) & 1][l] = g;
      if (l == p + 1) yrow[(p + 1) & 1][q] = y;
      if (q == p) G[p + l * s] = l == p ? from_real(di * rinv, T{}) : A;   // row p of R is final
      if (l == p) X[q + p * s] = conj_(scl(y, rinv));                      // row p of R^-H is final
    }
    __syncthreads();
  }
  __syncthreads();
  return wd / wg;
}

// The non-trivial columns ntrue..k-1 of T as the algebra reads them: element (l, c) at tb[l + (c - ntrue) * tld] -- staged in
// LDS (coalesced cooperative load) when they fit, else straight from memory (tld = ldt).  A thread of th_times walks DOWN a
// column of T: from memory that is a chain of dependent, uncoalesced L2 round trips (~30 us at k = 37).
// X[c, i] = sum_l conj(T[l, c]) Y[l, i]   (true coordinates from stored-column inner products); k x s, column stride k
template <class T>
__device__ void th_times(const T* __restrict__ tb, int64_t tld, int ntrue, int k, int s, const T* Y, T* X) {
  for (int e = threadIdx.x; e < k * s; e += kBlock) {
    const int c = e % k, i = e / k;
    T a;
    if (c < ntrue) a = Y[i * k + c];
    else {
      T a0 = zero_of(T{}), a1 = zero_of(T{});
      const T* tc = tb + (int64_t)(c - ntrue) * tld;
      const T* y = Y + i * k;
      int l = 0;
      for (; l + 1 <= c; l += 2) {
        a0 = fma_(conj_(tc[l]), y[l], a0);
        a1 = fma_(conj_(tc[l + 1]), y[l + 1], a1);
      }
      if (l <= c) a0 = fma_(conj_(tc[l]), y[l], a0);
      a = add_(a0, a1);
    }
    X[e] = a;
  }
  __syncthreads();
}
// X[r, i] = sum_c T[r, c] Y[c, i]        (coefficients of the stored columns from true coordinates)
template <class T>
__device__ void t_times(const T* __restrict__ tb, int64_t tld, int ntrue, int k, int s, const T* Y, T* X) {
  for (int e = threadIdx.x; e < k * s; e += kBlock) {
    const int r = e % k, i = e / k;
    T a0 = r < ntrue ? Y[i * k + r] : zero_of(T{}), a1 = zero_of(T{});
    const int c0 = r > ntrue ? r : ntrue;
    int c = c0;
    for (; c + 1 < k; c += 2) {
      a0 = fma_(tb[r + (int64_t)(c - ntrue) * tld], Y[i * k + c], a0);
      a1 = fma_(tb[r + (int64_t)(c + 1 - ntrue) * tld], Y[i * k + c + 1], a1);
    }
    if (c < k) a0 = fma_(tb[r + (int64_t)(c - ntrue) * tld], Y[i * k + c], a0);
    X[e] = add_(a0, a1);
  }
  __syncthreads();
}

// stage 1 (after k_bdots) / stage 2 (after k_bupdate).  `first`: first block of the batch (the last stored column is an
// ordinary one: u = e_{k-1}).
// mode 0: single GPU (reduce, elect, algebra);  mode 1: reduce only -> red[] (then the context's all-reduce over the ranks:
// RCCL / peer-to-peer / host-staged);  mode 2: algebra only from red[], ONE workgroup -- every rank holds the same sums and
// takes the same decisions (row-partitioned basis, replicated H / T: SURVEY 8e).
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_defl_dots(const T* __restrict__ U, int64_t ld, int nd, const T* __restrict__ z, int64_t n, T* __restrict__ partial /* [nd][gridDim.x] */) {
  __shared__ T red[kBlock / 64][kDeflMax];
  T acc[kDeflMax];
#pragma unroll
  for (int l = 0; l < kDeflMax; ++l) acc[l] = zero_of(T{});
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < n; row += (int64_t)gridDim.x * kBlock) {
    const T zv = z[row];
#pragma unroll
    for (int l = 0; l < kDeflMax; ++l)
      if (l < nd) acc[l] = fma_(conj_(U[row + (int64_t)l * ld]), zv, acc[l]);
  }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int l = 0; l < kDeflMax; ++l) {
    if (l < nd) {
      T v = acc[l];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v = add_(v, shfl_xor_(v, off));
      if (lane == 0) red[wv][l] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < nd) {
    T v = zero_of(T{});
    for (int w = 0; w < kBlock / 64; ++w) v = add_(v, red[w][threadIdx.x]);
    partial[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = v;
  }
}

// several ranks: the partial sums of this rank -> red[0:nd) (ONE workgroup; then the context's all-reduce, then k_defl_apply
// with red as its only "partial")
template <class T>
__global__ void __launch_bounds__(kBlock) k_defl_reduce(const T* __restrict__ partial, int nbp, int nd, T* __restrict__ red) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int l = wv; l < nd; l += kBlock / 64) {
    T v = zero_of(T{});
    for (int b = lane; b < nbp; b += 64) v = add_(v, partial[(int64_t)l * nbp + b]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = add_(v, shfl_xor_(v, off));
    if (lane == 0) red[l] = v;
  }
}

template <class T>
__global__ void __launch_bounds__(kBlock)
    k_defl_apply(const T* __restrict__ U, int64_t ld, int nd, T* __restrict__ z, int64_t n, const T* __restrict__ partial, int nbp, T* __restrict__ cout) {
  __shared__ T c[kDeflMax];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int l = wv; l < nd; l += kBlock / 64) {   // (fixed order: every workgroup computes the same c)
    T v = zero_of(T{});
    for (int b = lane; b < nbp; b += 64) v = add_(v, partial[(int64_t)l * nbp + b]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = add_(v, shfl_xor_(v, off));
    if (lane == 0) c[l] = v;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < kDeflMax) cout[threadIdx.x] = threadIdx.x < nd ? c[threadIdx.x] : zero_of(T{});
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < n; row += (int64_t)gridDim.x * kBlock) {
    T zv = z[row];
#pragma unroll
    for (int l = 0; l < kDeflMax; ++l)
      if (l < nd) zv = sub_(zv, mul_(U[row + (int64_t)l * ld], c[l]));
    z[row] = zv;
  }
}

template <class T>
__global__ void __launch_bounds__(kBlock)
    k_fin_blk(int stage, int mode, const T* __restrict__ partial, int nb, int pnb, int k, int s, T* __restrict__ red, T* __restrict__ Hd,
              int ldh, T* __restrict__ Tm, int ldt, int ntrue, BlkScratch<T>* __restrict__ bs, BlkShifts<T> sh, int first,
              double pivmin, double gdevmax, DevState* __restrict__ st, unsigned* __restrict__ counter, int ndefl = 0) {
  const int bd_flag = st->breakdown;   // (tested behind the loads of the reduction: one round trip to memory instead of two in a row)
  __shared__ int last_wg;
  __shared__ double redv[kBlock / 64][3];
  constexpr int SMX = blk_smax<T>();
  __shared__ T rfinv[SMX];
  constexpr int KS = (kBlkKMax - SMX) * SMX;   // k s with k + s <= kBlkKMax, s <= SMX (<= kBlkKMax / 2)
  __shared__ T rs[KS + SMX * (SMX + 1) / 2];
  __shared__ T A1[KS];      // stage 1: P;     stage 2: C
  __shared__ T A2[KS];      // stage 1: T P;   stage 2: T C, then PC
  __shared__ T Gm[SMX * SMX], Xi[SMX * SMX], Rf[SMX * SMX];
  __shared__ T R1l[SMX * SMX];   // stage 2: stage 1's factor and P (global memory behind bs: fetched with everything else up front --
  __shared__ T Pl[KS];           // every dependent round trip to memory in the middle of the algebra is ~1 us of this kernel)
  __shared__ T zu[kBlkKMax + SMX], hk[kBlkKMax + SMX];
  __shared__ T Tl[kBlkTLds];
  __shared__ T Hl[kBlkHLds];
  __shared__ T rhs[kBlkKMax * (SMX - 1)];         // (k + s <= kBlkKMax rows, s - 1 right-hand sides)
  const int tid = threadIdx.x;
  const int ng = s * (s + 1) / 2, ne = k * s + ng;
#ifdef KS_FIN_TIMING
  long long tq[16] = {wall_clock64(), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define KS_TQ(i) tq[i] = wall_clock64()
#else
#define KS_TQ(i) ((void)0)
#endif
  if (mode == 2 && bd_flag >= 0) return;
  if (mode != 2) {
    // one WAVE per sum (the producers run one workgroup per CU: 256 partial sums each -- a workgroup per sum was 630 workgroups
    // of one load per thread, two barriers each, and with this kernel's LDS footprint they did not fit on the chip at once)
    const int c = blockIdx.x * (kBlock / 64) + (tid >> 6), lane = tid & 63;
    T v = zero_of(T{});
    if (c < ne) {
      const T* pp = partial + (int64_t)c * pnb;
      T v0 = zero_of(T{}), v1 = zero_of(T{}), v2 = zero_of(T{}), v3 = zero_of(T{});
      int b = lane;
      for (; b + 192 < nb; b += 256) {
        v0 = add_(v0, pp[b]);
        v1 = add_(v1, pp[b + 64]);
        v2 = add_(v2, pp[b + 128]);
        v3 = add_(v3, pp[b + 192]);
      }
      for (; b < nb; b += 64) v0 = add_(v0, pp[b]);
      KS_TQ(11);
      v = wave_sum(add_(add_(v0, v1), add_(v2, v3)));
    }
    if (bd_flag >= 0) return;   // (the same on every workgroup; nothing has been written)
    KS_TQ(12);
    if (c < ne && lane == 0) {
      if (mode == 0) {
        st_agent(red + c, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the sum is where the elected workgroup will read it before this one takes its ticket)
      } else {
        red[c] = v;
      }
    }
    if (mode == 1) return;
    KS_TQ(13);
    __syncthreads();
    if (tid == 0) last_wg = (atomicAdd(counter, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    KS_TQ(14);
    if (!last_wg) return;
    if (tid == 0) *counter = 0u;   // (ordered against the next launch by the end of this kernel; the sums are read with agent-scope loads)
#ifdef KS_FIN_TIMING
    if (tid == 0 && k == 21 && s == 20) printf("[fin_blk reduce] loads issued+landed %.2f | wave sum %.2f | store + fence %.2f | barrier, atomic, barrier %.2f us\n",
                                                (tq[11] - tq[0]) * 0.01, (tq[12] - tq[11]) * 0.01, (tq[13] - tq[12]) * 0.01, (tq[14] - tq[13]) * 0.01);
#endif
  }
  KS_TQ(1);
  for (int e = tid; e < ne; e += kBlock) rs[e] = ld_agent(red + e);
  // columns ntrue..k-1 of T -> LDS (rows 0..k-1; entries below the diagonal are never read)
  const int nlz = k - ntrue;
  const bool t_lds = nlz > 0 && nlz * k <= kBlkTLds;
  if (t_lds)
    for (int e = tid; e < nlz * k; e += kBlock) {
      const int l = e % k, c = ntrue + e / k;
      Tl[e] = l <= c ? Tm[l + (int64_t)c * ldt] : zero_of(T{});
    }
  const T* tb = t_lds ? Tl : Tm + (int64_t)ntrue * ldt;
  const int64_t tld = t_lds ? k : ldt;
  // H[0:k, 0:k-1) -> LDS (coalesced; a thread of the H recovery walks ALONG a row of H: from memory that is a chain of L2 round trips)
  const bool h_lds = stage == 2 && k * (k - 1) <= kBlkHLds;
  if (stage == 2) {
    for (int e = tid; e < s * s; e += kBlock) R1l[e] = bs->R1[e];
    for (int e = tid; e < k * s; e += kBlock) Pl[e] = bs->P[e];
    // zu = u (coordinates of z_0 in V_k)
    for (int r = tid; r < k; r += kBlock) zu[r] = first ? from_real(r == k - 1 ? 1.0 : 0.0, T{}) : bs->u[r];
    if (h_lds)
      for (int e = tid; e < k * (k - 1); e += kBlock) Hl[e] = Hd[(e % k) + (int64_t)(e / k) * ldh];
  }
  __syncthreads();
  const T* Gin = rs + k * s;
  KS_TQ(2);
  // A1 = T^H (raw inner products)
  th_times(tb, tld, ntrue, k, s, rs, A1);
  KS_TQ(3);
  // Gm = Gin - A1^H A1   (upper triangle; one entry per thread: ng <= 210), and in the same pass what the acceptance tests need,
  // reduced over the workgroup in one go (wave butterflies, then the wave results through LDS behind the barrier Gm needs anyway):
  //   stage 2: how far the block that stage 1 wrote is from orthonormal: G_t = I + delta, delta ~ eps cond(R_1)^2 (the cancellation
  //     in G_Z - P^H P amplified by the conditioning of the Newton basis).  The recovered Hessenberg columns carry errors of order
  //     eps cond(R_1), so a block is only accepted while |delta|_max <= gdevmax (default 1e-8: cond <= ~1e4, H at the per-step
  //     path's accuracy; tests/test_sstep_model.py); beyond that it is abandoned like a rank-deficient one and the host lowers
  //     s.  Also |Gm - I|_max: decides below whether Gm needs a factorisation at all.
  //   stage 1: a COLLAPSING chain: z_i = sigma (A z_{i-1} - theta_i z_{i-1}) is computed with an error of eps ||A|| ||z_{i-1}||,
  //     and the H recovery divides by the UNSCALED factor -- its pivots carry the norms of the chain --, so a step that shrinks
  //     the vector by a factor f (a shift next to an eigenvalue whose invariant subspace dominates the vector: a tight cluster)
  //     puts an error of eps / f into H.  The pivot test of the factorisation is relative to each column's own norm and does not
  //     see it.  Below f = 3e-4 (norm^2 ratio 1e-7: errors beyond 1e-12) the block is abandoned like a rank-deficient one; the
  //     per-step path meets the same situation as a near-breakdown and takes the reference's decisions (src/expansion.jl:91-102).
  static_assert(SMX * (SMX + 1) / 2 <= kBlock, "one entry of the Gram matrix per thread");
  {
    double dv1 = 0.0, dv2 = 0.0, bad = 0.0;
    if (tid < ng) {
      int i2 = 0;
      while ((i2 + 1) * (i2 + 2) / 2 <= tid) ++i2;
      const int i = tid - i2 * (i2 + 1) / 2;
      T a = Gin[tid];
      T a0 = zero_of(T{}), a1 = zero_of(T{});
      int c = 0;
      for (; c + 1 < k; c += 2) {
        a0 = fma_(conj_(A1[i * k + c]), A1[i2 * k + c], a0);
        a1 = fma_(conj_(A1[i * k + c + 1]), A1[i2 * k + c + 1], a1);
      }
      if (c < k) a0 = fma_(conj_(A1[i * k + c]), A1[i2 * k + c], a0);
      a = sub_(a, add_(a0, a1));
      if (i == i2) a = from_real(real_of(a), T{});
      Gm[i + i2 * s] = a;
      const T del = from_real(i == i2 ? 1.0 : 0.0, T{});
      if (stage == 2) {
        dv1 = abs2_(sub_(Gin[tid], del));
        dv2 = abs2_(sub_(a, del));
      } else if (i == i2) {
        const double gd = real_of(Gin[tid]), prev = i ? real_of(Gin[gram_idx(i - 1, i - 1)]) : 1.0;   // (z_0 is a basis column)
        if (!(gd > 1e-7 * prev)) bad = 1.0;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      dv1 = fmax(dv1, __shfl_xor(dv1, off, 64));
      dv2 = fmax(dv2, __shfl_xor(dv2, off, 64));
      bad = fmax(bad, __shfl_xor(bad, off, 64));
    }
    if ((tid & 63) == 0) { redv[tid >> 6][0] = dv1; redv[tid >> 6][1] = dv2; redv[tid >> 6][2] = bad; }
  }
  __syncthreads();
  double gdev = 0.0, mdev = 0.0;
  bool collapse = false;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    gdev = fmax(gdev, redv[w][0]);
    mdev = fmax(mdev, redv[w][1]);
    collapse = collapse || redv[w][2] != 0.0;
  }
  gdev = sqrt(gdev);
  mdev = sqrt(mdev);
  const bool too_far = (stage == 2 && !(gdev <= gdevmax)) || collapse;
  KS_TQ(6);
  // stage 2, Gm = I + E with |E|_max <= 1e-9 (the usual case: E ~ eps cond(R_1)^2): R = I + striu(E) + diag(E) / 2 and
  // R^-1 = I - striu(E) - diag(E) / 2 are the factor and its inverse up to terms of order s |E|^2 <= 2e-17 -- below the rounding
  // of the factorisation they replace, whose 2 s dependent steps are ~10 us of this kernel.
  const bool first_order = stage == 2 && !too_far && mdev <= 1e-9;
  double worst;
  if (first_order) {
    for (int e = tid; e < s * s; e += kBlock) {
      const int i = e % s, i2 = e / s;
      if (i <= i2) {
        const T g = Gm[e];
        Gm[e] = i == i2 ? from_real(1.0 + 0.5 * (real_of(g) - 1.0), T{}) : g;
        Xi[e] = i == i2 ? from_real(1.0 - 0.5 * (real_of(g) - 1.0), T{}) : neg_(g);
      } else {
        Xi[e] = zero_of(T{});
      }
    }
    __syncthreads();
    worst = 1.0 - 2.0 * s * mdev;   // (a lower bound of every pivot ratio d_i / G_ii)
  } else {
    worst = too_far ? 0.0 : chol_inv_coop(Gm, Xi, s, stage == 1 ? pivmin : 0.25);
  }
  KS_TQ(7);
  if (too_far || !(worst > (stage == 1 ? pivmin : 0.25))) {
    // the block is (numerically) rank deficient: a breakdown, or a Newton basis too ill-conditioned to trust.  Nothing of
    // this block has been committed to T / H; the host redoes its steps one at a time (the per-step path takes the
    // reference's breakdown decisions, src/expansion.jl:99-102)
    if (tid == 0) {
      st->breakdown = k;   // step index of the block's first step == number of existing columns
      st->blk_bail = k;
      if (stage == 1) st->blk_piv1 = fmin(st->blk_piv1, worst);
      else { st->blk_piv2 = fmin(st->blk_piv2, worst); st->blk_gdev = fmax(st->blk_gdev, gdev); }
    }
    return;
  }
  KS_TQ(4);
  if (stage == 1) {
    t_times(tb, tld, ntrue, k, s, A1, A2);   // T P
    for (int e = tid; e < k * s; e += kBlock) {
      const int r = e % k, i = e / k;
      T a = zero_of(T{});
      for (int l = 0; l <= i; ++l) a = fma_(A2[l * k + r], Xi[l + i * s], a);
      bs->coefp[e] = a;
      bs->P[e] = A1[e];
    }
    for (int e = tid; e < s * s; e += kBlock) {
      const int l = e % s, i = e / s;
      bs->R1[e] = l <= i ? Gm[e] : zero_of(T{});
      bs->r1inv[e] = l <= i ? Xi[e] : zero_of(T{});
    }
    if (tid == 0) st->blk_piv1 = fmin(st->blk_piv1, worst);
#ifdef KS_FIN_TIMING
    if (tid == 0 && (k == 31 || (k == 21 && s == 20))) printf("[fin_blk stage 1 k=%d s=%d] reduce+elect %.2f | fetch %.2f | T^H %.2f | gram+tests %.2f factor+inverse %.2f | rest %.2f us\n", k, s,
                                    (tq[1] - tq[0]) * 0.01, (tq[2] - tq[1]) * 0.01, (tq[3] - tq[2]) * 0.01, (tq[6] - tq[3]) * 0.01, (tq[7] - tq[6]) * 0.01, (wall_clock64() - tq[4]) * 0.01);
#endif
    return;
  }
  // ---------------- stage 2: A1 = C, Gm = R2, Xi = R2^-1 ----------------
  t_times(tb, tld, ntrue, k, s, A1, A2);     // T C
  KS_TQ(12);
  // new columns of T
  for (int e = tid; e < (k + s) * s; e += kBlock) {
    const int r = e % (k + s), i = e / (k + s);
    T a = zero_of(T{});
    if (r < k) {
      for (int l = 0; l <= i; ++l) a = fma_(A2[l * k + r], Xi[l + i * s], a);
      a = neg_(a);
    } else if (r - k <= i) {
      a = Xi[(r - k) + i * s];
    }
    Tm[r + (int64_t)(k + i) * ldt] = a;
  }
  KS_TQ(13);
  // R = R2 R1 (upper x upper),  PC = P + C R1
  for (int e = tid; e < s * s; e += kBlock) {
    const int a_ = e % s, b = e / s;
    T v = zero_of(T{});
    for (int l = a_; l <= b; ++l) v = fma_(Gm[a_ + l * s], R1l[l + b * s], v);
    Rf[e] = v;
  }
  __syncthreads();
  if (tid < s) rfinv[tid] = inv_(Rf[tid + tid * s]);   // (read behind the next barrier: one division per column instead of one per row and column)
  for (int e = tid; e < k * s; e += kBlock) {
    const int r = e % k, i = e / k;
    T v = Pl[e];
    for (int l = 0; l <= i; ++l) v = fma_(A1[l * k + r], R1l[l + i * s], v);
    A2[e] = v;
  }
  __syncthreads();
  // zeta_i[r] for i >= 1:  r < k: A2[(i-1) k + r],  r = k + a: Rf[a + (i-1) s]
  auto zeta = [&](int i, int r) -> T { return r < k ? A2[(i - 1) * k + r] : Rf[(r - k) + (i - 1) * s]; };
  const int m = k + s;
  KS_TQ(5);
  const T* hb = h_lds ? Hl : Hd;
  const int64_t hld = h_lds ? k : ldh;
  KS_TQ(8);
  // H column k-1:  (zeta_1 / sigma_1 + theta_1 zeta_0 - H[:, 0:k-1) u[0:k-1)) / u[k-1]
  for (int r = tid; r < m; r += kBlock) {
    T a = scl(zeta(1, r), 1.0 / sh.sigma[0]);
    if (r < ndefl) a = add_(a, scl(bs->cdefl[r], 1.0 / sh.sigma[0]));   // (in-chain deflation: + U c_1 / sigma_1)
    if (r < k) {
      a = fma_(sh.theta[0], zu[r], a);
      T h0 = zero_of(T{}), h1 = zero_of(T{});
      int c = r > 0 ? r - 1 : 0;  // upper Hessenberg
      for (; c + 1 < k - 1; c += 2) {
        h0 = fma_(hb[r + c * hld], zu[c], h0);
        h1 = fma_(hb[r + (c + 1) * hld], zu[c + 1], h1);
      }
      if (c < k - 1) h0 = fma_(hb[r + c * hld], zu[c], h0);
      a = sub_(a, add_(h0, h1));
    }
    hk[r] = mul_(a, inv_(zu[k - 1]));
  }
  __syncthreads();
  for (int r = tid; r < ldh; r += kBlock) Hd[r + (int64_t)(k - 1) * ldh] = r < m ? hk[r] : zero_of(T{});
  KS_TQ(9);
  // H columns k .. k+s-2:  M R_{s-1} = rhs,  rhs[:, i-1] = zeta_{i+1} / sigma_{i+1} + theta_{i+1} zeta_i - Hext PC[:, i-1]
  // (Hext = [ H[:, 0:k-1) | hk ]);  all (row, column) pairs of rhs in parallel, then a row-parallel forward substitution
  if (s > 1) {
    for (int e = tid; e < m * (s - 1); e += kBlock) {
      const int r = e % m, i = 1 + e / m;
      T a = fma_(sh.theta[i], zeta(i, r), scl(zeta(i + 1, r), 1.0 / sh.sigma[i]));
      if (r < ndefl) a = add_(a, scl(bs->cdefl[i * kDeflMax + r], 1.0 / sh.sigma[i]));
      T h0 = zero_of(T{}), h1 = zero_of(T{});
      const T* pc = A2 + (i - 1) * k;
      if (r < k) {
        int c = r > 0 ? r - 1 : 0;
        for (; c + 3 < k - 1; c += 4) {   // (eight LDS reads in flight per trip: a trip is one LDS latency whatever it carries)
          const T b0 = hb[r + c * hld], b1 = hb[r + (c + 1) * hld], b2 = hb[r + (c + 2) * hld], b3 = hb[r + (c + 3) * hld];
          const T p0 = pc[c], p1 = pc[c + 1], p2 = pc[c + 2], p3 = pc[c + 3];
          h0 = fma_(b0, p0, h0);
          h1 = fma_(b1, p1, h1);
          h0 = fma_(b2, p2, h0);
          h1 = fma_(b3, p3, h1);
        }
        for (; c < k - 1; ++c) h0 = fma_(hb[r + c * hld], pc[c], h0);
      }
      h0 = fma_(hk[r], pc[k - 1], h0);
      rhs[e] = sub_(a, add_(h0, h1));
    }
    __syncthreads();
    KS_TQ(10);
    for (int r = tid; r < m; r += kBlock) {
      // (row of M in registers, statically unrolled and predicated on s: behind run-time indices it lived in scratch memory,
      // ~25 us of the stage)
      T Mrow[SMX];
#pragma unroll
      for (int i = 1; i < SMX; ++i) {
        Mrow[i - 1] = zero_of(T{});
        if (i < s) {
          T a = rhs[r + (i - 1) * m];
#pragma unroll
          for (int l = 0; l < SMX - 2; ++l)
            if (l < i - 1) a = sub_(a, mul_(Mrow[l], Rf[l + (i - 1) * s]));
          Mrow[i - 1] = mul_(a, rfinv[i - 1]);
        }
      }
#pragma unroll
      for (int i = 1; i < SMX; ++i)
        if (i < s) Hd[r + (int64_t)(k - 1 + i) * ldh] = Mrow[i - 1];
    }
    KS_TQ(14);
    for (int e = tid; e < (ldh - m) * (s - 1); e += kBlock) {
      const int r = m + e % (ldh - m), i = 1 + e / (ldh - m);
      Hd[r + (int64_t)(k - 1 + i) * ldh] = zero_of(T{});
    }
  }
  // coordinates of the last stored column (input of the next block)
  for (int r = tid; r < m; r += kBlock) bs->u[r] = r < k ? A1[(s - 1) * k + r] : Gm[(r - k) + (s - 1) * s];
  if (tid == 0) {
    st->n_steps += s;
    st->wnorm = real_of(Rf[(s - 1) + (s - 1) * s]);
    st->n_reorth += s;   // (the second projection is always part of a block)
    st->blk_piv2 = fmin(st->blk_piv2, worst);
    st->blk_gdev = fmax(st->blk_gdev, gdev);
  }
#ifdef KS_FIN_TIMING
  if (tid == 0 && (k == 31 || (k == 21 && s == 20))) printf("[fin_blk stage 2 k=%d s=%d] reduce+elect %.2f | fetch %.2f | T^H %.2f | gram+tests %.2f factor+inverse %.2f | T C %.2f T cols %.2f R, PC %.2f | col k-1 %.2f rhs %.2f subst %.2f rest %.2f us\n", k, s,
                                  (tq[1] - tq[0]) * 0.01, (tq[2] - tq[1]) * 0.01, (tq[3] - tq[2]) * 0.01, (tq[6] - tq[3]) * 0.01, (tq[7] - tq[6]) * 0.01, (tq[12] - tq[4]) * 0.01, (tq[13] - tq[12]) * 0.01, (tq[5] - tq[13]) * 0.01,
                                  (tq[9] - tq[8]) * 0.01, (tq[10] - tq[9]) * 0.01, (tq[14] - tq[10]) * 0.01, (wall_clock64() - tq[14]) * 0.01);
#endif
}

}  // namespace ksd
