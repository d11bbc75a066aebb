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

constexpr int kBlkSMax = 10;                                   // largest block (steps per block)
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

// packs per lane and iteration, and the occupancy the kernels are compiled for: a lane keeps NCW x S accumulators, NCW x U
// basis packs and 2 S U block packs; while that fits 256 registers TWO workgroups share a CU (measured on the 216^3 basis:
// pass 1 6.2-6.5 TB/s with two resident workgroups, 3.0-5.0 with one; pass 2 5.0 against 3.8) -- what counts is the number
// of bytes in flight per CU.
template <class T, int NCW, int S> constexpr int blk_u() { return S <= 5 ? 2 : 1; }
template <class T, int NCW, int S, int U> constexpr int blk_regs() {
  constexpr int D = (int)(sizeof(T) / 8);
  return 2 * (D * NCW * S + 2 * NCW * U + 2 * S * U + D * ((S * (S + 1) / 2 + 3) / 4)) + 44;
}
template <class T, int NCW, int S, int U> constexpr int blk_wpe() { return blk_regs<T, NCW, S, U>() <= 256 ? 2 : 1; }

// upper-triangle index of Gram entry (i, i2), i <= i2
__host__ __device__ __forceinline__ constexpr int gram_idx(int i, int i2) { return i2 * (i2 + 1) / 2 + i; }

// Write the folded accumulators of one wave.  Flattened accumulator index (in elements of T):
//   e < NCW*S        : column c = wave + 4 (e / S), right-hand side i = e % S  -> partial entry  i*k + c
//   e = NCW*S + gi   : Gram entry g = 4 gi + wave (if < ng)                    -> partial entry  k*S + g
template <class T, int NCW, int S, int NGW>
__device__ __forceinline__ void store_folded(const double* f, int lane, int wave, int k, T* __restrict__ partial, int pnb) {
  constexpr int D = Dpe<T>::value;
  constexpr int NE = NCW * S + NGW;            // elements
  constexpr int P = next_pow2(NE * D);         // doubles, padded
  constexpr int NG = S * (S + 1) / 2;
  auto put = [&](int didx, double v) {
    const int e = didx / D, part = didx % D;
    int entry = -1;
    if (e < NCW * S) {
      const int c = wave + 4 * (e / S), i = e % S;
      if (c < k) entry = i * k + c;
    } else if (e < NE) {
      const int g = 4 * (e - NCW * S) + wave;
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
// BDOTS (pass 1):  partial[i*k + c][b] = sum_{rows of b} conj(S[r,c]) Z[r,i],   partial[k*S + g(i,i2)][b] = sum conj(Z[r,i]) Z[r,i2]
// S = V[:, 0:k), Z = V[:, k:k+S).
// ---------------------------------------------------------------------------------------------------------------------------
template <class T, int NCW, int S, int U, bool NT>
__global__ void __launch_bounds__(kBlock, (blk_wpe<T, NCW, S, U>()))
    k_bdots(const T* __restrict__ V, int64_t ldv, int k, T* __restrict__ partial, int pnb, const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  constexpr int NG = S * (S + 1) / 2, NGW = (NG + 3) / 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const T* colp[NCW];
  bool valid[NCW];
  T acc[NCW][S];
  T gacc[NGW];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii) {
    const int c = wave + 4 * ii;
    valid[ii] = c < k;
    colp[ii] = V + (int64_t)(valid[ii] ? c : 0) * ldv;
#pragma unroll
    for (int i = 0; i < S; ++i) acc[ii][i] = zero_of(T{});
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
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int u = 0; u < U; ++u) dotp(acc[ii][i], v[ii][u], z[i][u]);
    // Gram entries: entry g = (i, i2), i <= i2, belongs to wave g % 4 (uniform branch, static accumulator index g / 4)
#pragma unroll
    for (int i2 = 0; i2 < S; ++i2)
#pragma unroll
      for (int i = 0; i <= i2; ++i) {
        const int g = gram_idx(i, i2);
        if ((g & 3) == wave) {
#pragma unroll
          for (int u = 0; u < U; ++u) dotp(gacc[g >> 2], z[i][u], z[i2][u]);
        }
      }
  }
  constexpr int D = Dpe<T>::value;
  constexpr int NE = NCW * S + NGW;
  constexpr int PD = next_pow2(NE * D);
  double f[PD];
#pragma unroll
  for (int e = 0; e < PD; ++e) f[e] = 0.0;
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < S; ++i) put_acc(f, ii * S + i, acc[ii][i]);
#pragma unroll
  for (int g = 0; g < NGW; ++g) put_acc(f, NCW * S + g, gacc[g]);
  fold_stage<PD, 32>(f, lane);
  store_folded<T, NCW, S, NGW>(f, lane, wave, k, partial, pnb);
}

// ---------------------------------------------------------------------------------------------------------------------------
// BUPDATE (pass 2):  Qt = Z R1inv - S coefp  (in place over Z);  partial[i*k + c] = conj(S[:,c]) . Qt[:,i];  Gram of Qt.
// coefp: k x S column-major (leading dimension ldc), r1inv: S x S upper triangular column-major (leading dimension S).
// Every wave forms the partial row sums  t_w[i] = sum_{own columns} S[r,c] coefp[c,i]  -  sum_{own l = w mod 4} Z[r,l] r1inv[l,i],
// the four meet in LDS (double buffered, one barrier per iteration),  Qt[r,i] = -(t_0 + t_1 + t_2 + t_3)[i].
// Wave w stores the columns i = w mod 4.
// ---------------------------------------------------------------------------------------------------------------------------
// Exchange buffer: double (one barrier per iteration) while one workgroup has the CU to itself; SINGLE (a second barrier per
// iteration) in the instantiations compiled for two resident workgroups, so that both fit the CU's 160 KiB of LDS.
// (Staging the written block in LDS and writing 16 KiB bursts per column -- what pays in k_axpy_dots_cs, which writes ONE
// column in 96 KiB bursts -- was measured here and is slower: 710 against 651 us at k = 21, s = 5; five columns leave no room
// for bursts of that size.)
template <class T, int NCW, int S, int U, bool NT>
__global__ void __launch_bounds__(kBlock, (blk_wpe<T, NCW, S, U>()))
    k_bupdate(T* __restrict__ V, int64_t ldv, int k, const T* __restrict__ coefp, int ldc, const T* __restrict__ r1inv,
              T* __restrict__ partial, int pnb, const DevState* __restrict__ st, int dbg = 0) {
  if (st && st->breakdown >= 0) return;
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  constexpr int NG = S * (S + 1) / 2, NGW = (NG + 3) / 4;
  constexpr int NB = blk_wpe<T, NCW, S, U>() == 2 ? 1 : 2;
  __shared__ P tbuf[NB][4][U][S][64];
  __shared__ T cf[4 * NCW][S];   // coefp rows (columns of the basis) as the waves index them: cf[c][i]
  __shared__ T ri[S][S];         // r1inv[l][i], l <= i
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = threadIdx.x; e < 4 * NCW * S; e += kBlock) {
    const int c = e / S, i = e % S;
    cf[c][i] = c < k ? coefp[c + (int64_t)i * ldc] : zero_of(T{});
  }
  for (int e = threadIdx.x; e < S * S; e += kBlock) {
    const int l = e % S, i = e / S;
    ri[l][i] = l <= i ? r1inv[l + i * S] : zero_of(T{});
  }
  __syncthreads();
  const T* colp[NCW];
  bool valid[NCW];
  T acc[NCW][S];
  T gacc[NGW];
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii) {
    const int c = wave + 4 * ii;
    valid[ii] = c < k;
    colp[ii] = V + (int64_t)(valid[ii] ? c : 0) * ldv;
#pragma unroll
    for (int i = 0; i < S; ++i) acc[ii][i] = zero_of(T{});
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
    P t[U][S];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < S; ++i) t[u][i] = zero_pack(T{});
    // own Z columns (l = wave mod 4): t[i] -= Z_l r1inv[l, i], i >= l
#pragma unroll
    for (int l = 0; l < S; ++l) {
      if ((l & 3) == wave) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const P zl = ld_pack(Z + (int64_t)l * ldv + r[u]);
#pragma unroll
          for (int i = l; i < S; ++i) axpy_acc(t[u][i], zl, neg_(ri[l][i]));
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii) {
      if (valid[ii]) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
          const T g = cf[wave + 4 * ii][i];
#pragma unroll
          for (int u = 0; u < U; ++u) axpy_acc(t[u][i], v[ii][u], g);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < S; ++i) tbuf[it & (NB - 1)][wave][u][i][lane] = t[u][i];
    __syncthreads();
    P q[U][S];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < S; ++i) {
        const P t0 = tbuf[it & (NB - 1)][0][u][i][lane], t1 = tbuf[it & (NB - 1)][1][u][i][lane];
        const P t2 = tbuf[it & (NB - 1)][2][u][i][lane], t3 = tbuf[it & (NB - 1)][3][u][i][lane];
        P qq = sub_pack(zero_pack(T{}), addp(addp(t0, t1), addp(t2, t3)));
        if (!ok[u]) qq = zero_pack(T{});
        q[u][i] = qq;
        if ((i & 3) == wave && ok[u] && !(dbg & 1)) {  // (dbg & 1: timing probe without the write stream, & 4: cacheable stores)
          if (dbg & 4) st_pack(Z + (int64_t)i * ldv + r[u], qq);
          else st_pack_nt(Z + (int64_t)i * ldv + r[u], qq);
        }
      }
#pragma unroll
    for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int u = 0; u < U; ++u) dotp(acc[ii][i], v[ii][u], q[u][i]);
#pragma unroll
    for (int i2 = 0; i2 < S; ++i2)
#pragma unroll
      for (int i = 0; i <= i2; ++i) {
        const int g = gram_idx(i, i2);
        if ((g & 3) == wave) {
#pragma unroll
          for (int u = 0; u < U; ++u) dotp(gacc[g >> 2], q[u][i], q[u][i2]);
        }
      }
    if constexpr (NB == 1) __syncthreads();  // the single exchange buffer may be overwritten from here on
  }
  constexpr int D = Dpe<T>::value;
  constexpr int NE = NCW * S + NGW;
  constexpr int PD = next_pow2(NE * D);
  double f[PD];
#pragma unroll
  for (int e = 0; e < PD; ++e) f[e] = 0.0;
#pragma unroll
  for (int ii = 0; ii < NCW; ++ii)
#pragma unroll
    for (int i = 0; i < S; ++i) put_acc(f, ii * S + i, acc[ii][i]);
#pragma unroll
  for (int g = 0; g < NGW; ++g) put_acc(f, NCW * S + g, gacc[g]);
  fold_stage<PD, 32>(f, lane);
  store_folded<T, NCW, S, NGW>(f, lane, wave, k, partial, pnb);
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
template <class T> struct BlkScratch {
  T P[kBlkKMax * kBlkSMax];        // true coordinates of Z in V_k (k x s, column stride k)
  T R1[kBlkSMax * kBlkSMax];       // stage-1 triangular factor (s x s, column stride s)
  T coefp[kBlkKMax * kBlkSMax];    // (T P) R1^-1: what k_bupdate subtracts (k x s, column stride k)
  T r1inv[kBlkSMax * kBlkSMax];
  T u[kBlkKMax + kBlkSMax];        // coordinates of the last stored column in the true basis (valid after a block)
};

__device__ __forceinline__ double inv_(double a) { return 1.0 / a; }
__device__ __forceinline__ cd inv_(cd a) {
  const double d = 1.0 / fma(a.x, a.x, a.y * a.y);
  return cd{a.x * d, -a.y * d};
}

__device__ __forceinline__ double bcast_(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ cd bcast_(cd v, int src) { return cd{__shfl(v.x, src, 64), __shfl(v.y, src, 64)}; }

// Cholesky R^H R = G of an s x s Hermitian matrix given by its upper triangle in LDS (column stride s), in place: on exit
// the upper triangle holds R.  All threads of the workgroup call it; returns the smallest pivot ratio d_i / G_ii (<= 0 when
// the matrix is not positive definite); the factorisation stops (and the caller bails) at the first ratio <= pivmin.
// Done by WAVE 0 in registers: lane l owns column l, column i of R travels by lane broadcasts -- no barrier and no LDS round
// trip inside the s dependent steps (the first version, thread 0 + two barriers per step, cost 10 us of a 28 us kernel).
template <class T> __device__ double chol_upper_lds(T* G, int s, double pivmin, double* sh_ratio) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int lane = tid;
    T r[kBlkSMax];
#pragma unroll
    for (int i = 0; i < kBlkSMax; ++i) r[i] = (lane < s && i <= lane && i < s) ? G[i + lane * s] : zero_of(T{});
    double worst = 1.0;
    bool okay = true;
#pragma unroll
    for (int i = 0; i < kBlkSMax; ++i) {
      if (i < s && okay) {  // (uniform)
        double d = real_of(r[i]);
        const double gii = d;
#pragma unroll
        for (int p = 0; p < i; ++p) d -= abs2_(r[p]);
        const double di = __shfl(d, i, 64), gi = __shfl(gii, i, 64);
        const double ratio = gi > 0.0 ? di / gi : 0.0;
        worst = fmin(worst, ratio);
        if (!(ratio > pivmin)) {
          okay = false;
        } else {
          const double rii = sqrt(di), rinv = 1.0 / rii;
          T a = r[i];
#pragma unroll
          for (int p = 0; p < i; ++p) a = sub_(a, mul_(conj_(bcast_(r[p], i)), r[p]));
          r[i] = lane == i ? from_real(rii, T{}) : scl(a, rinv);
        }
      }
    }
    if (okay && lane < s) {
#pragma unroll
      for (int i = 0; i < kBlkSMax; ++i)
        if (i <= lane && i < s) G[i + lane * s] = r[i];
    }
    if (lane == 0) *sh_ratio = worst;
  }
  __syncthreads();
  const double w = *sh_ratio;
  __syncthreads();
  return w;
}
// X = R^-1 (upper triangular, column stride s): thread c < s computes column c by back substitution
template <class T> __device__ void tri_inv_lds(const T* Rm, T* X, int s) {
  const int c = threadIdx.x;
  if (c < s) {
    for (int i = 0; i < s; ++i) X[i + c * s] = zero_of(T{});
    X[c + c * s] = inv_(Rm[c + c * s]);
    for (int i = c - 1; i >= 0; --i) {
      T a = zero_of(T{});
      for (int l = i + 1; l <= c; ++l) a = fma_(Rm[i + l * s], X[l + c * s], a);
      X[i + c * s] = mul_(neg_(a), inv_(Rm[i + i * s]));
    }
  }
  __syncthreads();
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
    k_fin_blk(int stage, int mode, const T* __restrict__ partial, int nb, int pnb, int k, int s, T* __restrict__ red, T* __restrict__ Hd,
              int ldh, T* __restrict__ Tm, int ldt, int ntrue, BlkScratch<T>* __restrict__ bs, BlkShifts<T> sh, int first,
              double pivmin, double gdevmax, DevState* __restrict__ st, unsigned* __restrict__ counter) {
  if (st->breakdown >= 0) return;
  __shared__ T sm[kBlock];
  __shared__ int last_wg;
  __shared__ double sh_ratio;
  __shared__ T rs[kBlkKMax * kBlkSMax + kBlkGram];
  __shared__ T A1[kBlkKMax * kBlkSMax];      // stage 1: P;     stage 2: C
  __shared__ T A2[kBlkKMax * kBlkSMax];      // stage 1: T P;   stage 2: T C, then PC
  __shared__ T Gm[kBlkSMax * kBlkSMax], Xi[kBlkSMax * kBlkSMax], Rf[kBlkSMax * kBlkSMax];
  __shared__ T zu[kBlkKMax + kBlkSMax], hk[kBlkKMax + kBlkSMax];
  __shared__ T Tl[kBlkTLds];
  __shared__ T Hl[kBlkHLds];
  __shared__ T rhs[(kBlkKMax + kBlkSMax) * (kBlkSMax - 1)];
  const int tid = threadIdx.x;
  const int ng = s * (s + 1) / 2, ne = k * s + ng;
#ifdef KS_FIN_TIMING
  long long tq[8] = {wall_clock64(), 0, 0, 0, 0, 0, 0, 0};
#define KS_TQ(i) tq[i] = wall_clock64()
#else
#define KS_TQ(i) ((void)0)
#endif
  if (mode != 2) {
    const int c = blockIdx.x;
    const T v = block_sum(partial + (int64_t)c * pnb, nb, sm);
    if (tid == 0) {
      red[c] = v;
      if (mode == 0) {
        __threadfence();
        last_wg = (atomicAdd(counter, 1u) == gridDim.x - 1) ? 1 : 0;
      }
    }
    if (mode == 1) return;
    __syncthreads();
    if (!last_wg) return;
    if (tid == 0) *counter = 0u;
    __threadfence();
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
  __syncthreads();
  const T* Gin = rs + k * s;
  KS_TQ(2);
  // A1 = T^H (raw inner products)
  th_times(tb, tld, ntrue, k, s, rs, A1);
  KS_TQ(3);
  // Gm = Gin - A1^H A1   (upper triangle)
  for (int g = tid; g < ng; g += kBlock) {
    int i2 = 0;
    while ((i2 + 1) * (i2 + 2) / 2 <= g) ++i2;
    const int i = g - i2 * (i2 + 1) / 2;
    T a = Gin[g];
    for (int c = 0; c < k; ++c) a = sub_(a, mul_(conj_(A1[i * k + c]), A1[i2 * k + c]));
    if (i == i2) a = from_real(real_of(a), T{});
    Gm[i + i2 * s] = a;
  }
  __syncthreads();
  // How far the block that stage 1 wrote is from orthonormal: G_t = I + delta, delta ~ eps cond(R_1)^2 (the cancellation in
  // G_Z - P^H P amplified by the conditioning of the Newton basis).  The recovered Hessenberg columns carry errors of order
  // eps cond(R_1), so a block is only accepted while delta <= gdevmax (default 1e-8: cond <= ~1e4, H at the per-step path's
  // accuracy; tests/test_sstep_model.py); beyond that it is abandoned like a rank-deficient one and the host lowers s.
  double gdev = 0.0;
  if (stage == 2) {
    if (tid < 64) {  // ng <= 55 entries: one wave, |entry - delta|^2, wave maximum
      double dv = 0.0;
      if (tid < ng) {
        int i2 = 0;
        while ((i2 + 1) * (i2 + 2) / 2 <= tid) ++i2;
        const int i = tid - i2 * (i2 + 1) / 2;
        dv = abs2_(sub_(Gin[tid], from_real(i == i2 ? 1.0 : 0.0, T{})));
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) dv = fmax(dv, __shfl_xor(dv, off, 64));
      if (tid == 0) sh_ratio = sqrt(dv);
    }
    __syncthreads();
    gdev = sh_ratio;
    __syncthreads();
  }
  const bool too_far = stage == 2 && !(gdev <= gdevmax);
  const double worst = too_far ? 0.0 : chol_upper_lds(Gm, s, stage == 1 ? pivmin : 0.25, &sh_ratio);
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
  tri_inv_lds(Gm, Xi, s);
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
    if (tid == 0 && k == 29) printf("[fin_blk stage 1 k=%d s=%d] reduce+elect %.2f | fetch %.2f | T^H %.2f | gram+chol+inv %.2f | rest %.2f us\n", k, s,
                                    (tq[1] - tq[0]) * 0.01, (tq[2] - tq[1]) * 0.01, (tq[3] - tq[2]) * 0.01, (tq[4] - tq[3]) * 0.01, (wall_clock64() - tq[4]) * 0.01);
#endif
    return;
  }
  // ---------------- stage 2: A1 = C, Gm = R2, Xi = R2^-1 ----------------
  t_times(tb, tld, ntrue, k, s, A1, A2);     // T C
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
  // R = R2 R1 (upper x upper),  PC = P + C R1
  for (int e = tid; e < s * s; e += kBlock) {
    const int a_ = e % s, b = e / s;
    T v = zero_of(T{});
    for (int l = a_; l <= b; ++l) v = fma_(Gm[a_ + l * s], bs->R1[l + b * s], v);
    Rf[e] = v;
  }
  __syncthreads();
  for (int e = tid; e < k * s; e += kBlock) {
    const int r = e % k, i = e / k;
    T v = bs->P[e];
    for (int l = 0; l <= i; ++l) v = fma_(A1[l * k + r], bs->R1[l + i * s], v);
    A2[e] = v;
  }
  // zu = u (coordinates of z_0 in V_k)
  for (int r = tid; r < k; r += kBlock) zu[r] = first ? from_real(r == k - 1 ? 1.0 : 0.0, T{}) : bs->u[r];
  __syncthreads();
  // zeta_i[r] for i >= 1:  r < k: A2[(i-1) k + r],  r = k + a: Rf[a + (i-1) s]
  auto zeta = [&](int i, int r) -> T { return r < k ? A2[(i - 1) * k + r] : Rf[(r - k) + (i - 1) * s]; };
  const int m = k + s;
  KS_TQ(5);
  // H[0:k, 0:k-1) -> LDS (coalesced; a thread below walks ALONG a row of H: from memory that is a chain of L2 round trips)
  const bool h_lds = k * (k - 1) <= kBlkHLds;
  if (h_lds)
    for (int e = tid; e < k * (k - 1); e += kBlock) Hl[e] = Hd[(e % k) + (int64_t)(e / k) * ldh];
  const T* hb = h_lds ? Hl : Hd;
  const int64_t hld = h_lds ? k : ldh;
  __syncthreads();
  // H column k-1:  (zeta_1 / sigma_1 + theta_1 zeta_0 - H[:, 0:k-1) u[0:k-1)) / u[k-1]
  for (int r = tid; r < m; r += kBlock) {
    T a = scl(zeta(1, r), 1.0 / sh.sigma[0]);
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
  // H columns k .. k+s-2:  M R_{s-1} = rhs,  rhs[:, i-1] = zeta_{i+1} / sigma_{i+1} + theta_{i+1} zeta_i - Hext PC[:, i-1]
  // (Hext = [ H[:, 0:k-1) | hk ]);  all (row, column) pairs of rhs in parallel, then a row-parallel forward substitution
  if (s > 1) {
    for (int e = tid; e < m * (s - 1); e += kBlock) {
      const int r = e % m, i = 1 + e / m;
      T a = fma_(sh.theta[i], zeta(i, r), scl(zeta(i + 1, r), 1.0 / sh.sigma[i]));
      T h0 = zero_of(T{}), h1 = zero_of(T{});
      const T* pc = A2 + (i - 1) * k;
      if (r < k) {
        int c = r > 0 ? r - 1 : 0;
        for (; c + 1 < k - 1; c += 2) {
          h0 = fma_(hb[r + c * hld], pc[c], h0);
          h1 = fma_(hb[r + (c + 1) * hld], pc[c + 1], h1);
        }
        if (c < k - 1) h0 = fma_(hb[r + c * hld], pc[c], h0);
      }
      h0 = fma_(hk[r], pc[k - 1], h0);
      rhs[e] = sub_(a, add_(h0, h1));
    }
    __syncthreads();
    for (int r = tid; r < m; r += kBlock) {
      T Mrow[kBlkSMax];
      for (int i = 1; i < s; ++i) {
        T a = rhs[r + (i - 1) * m];
        for (int l = 0; l < i - 1; ++l) a = sub_(a, mul_(Mrow[l], Rf[l + (i - 1) * s]));
        Mrow[i - 1] = mul_(a, inv_(Rf[(i - 1) + (i - 1) * s]));
      }
      for (int i = 1; i < s; ++i) Hd[r + (int64_t)(k - 1 + i) * ldh] = Mrow[i - 1];
    }
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
  if (tid == 0 && k == 29) printf("[fin_blk stage 2 k=%d s=%d] reduce+elect %.2f | fetch %.2f | T^H %.2f | gram+chol+inv %.2f | T cols, R, PC %.2f | H %.2f us\n", k, s,
                                  (tq[1] - tq[0]) * 0.01, (tq[2] - tq[1]) * 0.01, (tq[3] - tq[2]) * 0.01, (tq[4] - tq[3]) * 0.01, (tq[5] - tq[4]) * 0.01, (wall_clock64() - tq[5]) * 0.01);
#endif
}

}  // namespace ksd
