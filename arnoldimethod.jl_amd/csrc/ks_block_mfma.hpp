// gfx950 streaming kernels of the s-step expansion on the FP64 matrix instruction (v_mfma_f64_4x4x4_4b_f64), Float64 blocks
// of up to 4 NT steps on up to 4 NGS existing columns.  Same results (entry layout of `partial`) as k_bdots / k_bupdate of
// ks_block_kernels.hpp; the sums are taken in a different order (tests compare the two at rounding level).
//
// Why.  The vector forms keep one partial sum PER LANE for every inner product: 20 k + 210 accumulators at s = 20 do not fit
// one wave, so the accumulators are dealt over the eight waves of a workgroup and EVERY wave reads every row of the tile from
// LDS, in lock-step through three barriers per tile (k_bupdate_ringL: 0.42 of the HBM rate; memory, LDS and arithmetic phases
// never overlap).  A 4x4x4 matrix instruction sums over the rows inside the instruction: a 4 x 4 tile of inner products is ONE
// accumulator register pair per lane, the whole (k + s) x s result of a pass is 45 of them, and a wave can own a SLAB OF ROWS
// with everything that belongs to it: no data is shared between waves, no barrier inside the loop, every wave runs its own ring
// of asynchronous global -> LDS copies and drifts freely against the others (that is what overlaps the three pipes).
// The FP64 matrix pipe is not extra arithmetic (tools/fp64_pipes.hip, profiles/r05_fp64_pipes.txt: 16.7 cycles per
// instruction = the vector rate, and the two do not add up); what it buys is 256 multiply-adds per issued instruction with
// 2 operand registers, and the 64-fold smaller accumulator footprint.
//
// Lane layout of D = A B + C, four independent 4x4x4 blocks b (tools/mfma4_layout.hip):
//   A_b[i][k] in lane 16 k + 4 b + i,   B_b[k][j] in lane 16 k + 4 b + j,   D_b[i][j] in lane 16 i + 4 b + j
// hence a result register read as the B operand is the same 4 x 4 matrix, read as the A operand it is its TRANSPOSE.
//
// Rows: a workgroup walks its row range in tiles of 64 packs (128 rows); wave w owns the 16 rows 16 w .. 16 w + 15 of every
// tile (a SLAB), block b of an instruction the rows 4 b .. 4 b + 3 of the slab.
// LDS slab: the k existing columns in 1-KiB groups of eight (column c at (c / 8) KiB + (c % 8) * 128 B, 16 rows of 8 B), then
// the block's columns in the same way starting at a fresh group.  One global_load_lds_dwordx4 fills one group: lane l copies
// pack l % 8 of column l / 8 (128 contiguous bytes per column).  Two read patterns:
//   linear  (lane l: row l % 16, column l / 16 of a 4-column group)  = byte 8 l of the group's half          -> A of Z R - S c
//   gather  (lane l: row 4 ((l / 4) % 4) + l / 16, column l % 4)                                              -> A / B of the inner products
#pragma once

#include "ks_block_kernels.hpp"

namespace ksd {

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void wait_vm(int n) {
#define KS_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); break;
  switch (n) {
    KS_VMW(0) KS_VMW(1) KS_VMW(2) KS_VMW(3) KS_VMW(4) KS_VMW(5) KS_VMW(6) KS_VMW(7) KS_VMW(8) KS_VMW(9) KS_VMW(10) KS_VMW(11)
    KS_VMW(12) KS_VMW(13) KS_VMW(14) KS_VMW(15) KS_VMW(16) KS_VMW(17) KS_VMW(18) KS_VMW(19) KS_VMW(20) KS_VMW(21) KS_VMW(22)
    KS_VMW(23) KS_VMW(24) KS_VMW(25) KS_VMW(26) KS_VMW(27) KS_VMW(28) KS_VMW(29) KS_VMW(30) KS_VMW(31) KS_VMW(32) KS_VMW(33)
    KS_VMW(34) KS_VMW(35) KS_VMW(36) KS_VMW(37) KS_VMW(38) KS_VMW(39) KS_VMW(40)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
#undef KS_VMW
}
// 8-byte streaming store from inline assembly (counted by hand on the vector-memory counter)
__device__ __forceinline__ void gst8_nt(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory"); }

template <int NGS, int NT> struct BlkMfma {
  static constexpr int NJS = (NGS + 1) / 2, NJZ = (NT + 1) / 2, NJ = NJS + NJZ;   // 1-KiB groups (= copies) per slab
  static constexpr int SLAB = NJ * 1024;                                          // bytes
  static constexpr int NTS = NGS * NT, NTG = NT * (NT + 1) / 2, NTILE = NTS + NTG; // result tiles: S^H X, upper triangle of X^H X
  static constexpr int MBYTES = NTILE * 128;                                       // coefficient tiles of the second pass
  __host__ __device__ static constexpr int s_off(int g) { return (g / 2) * 1024 + (g % 2) * 512; }
  __host__ __device__ static constexpr int z_off(int t) { return NJS * 1024 + (t / 2) * 1024 + (t % 2) * 512; }
  __host__ __device__ static constexpr int gt(int a, int t) { return t * (t + 1) / 2 + a; }   // tile (a, t), a <= t, of the triangle
  static constexpr size_t lds_bytes(int ring, bool cx = false) { return (size_t)8 * ring * SLAB + (cx ? 2 : 1) * (size_t)MBYTES; }
};

// per-wave accumulators -> partial[entry][workgroup].  acc: NTILE registers in the D layout (one 4 x 4 tile per block b);
// CX: NTILE more with the imaginary parts, `partial` holds complex entries (two doubles each).
template <int NGS, int NT, bool CX = false>
__device__ __forceinline__ void blkm_finish(double* acc, unsigned char* lds, int lane, int wave, int k, int s, double* __restrict__ partial, int pnb) {
  using C = BlkMfma<NGS, NT>;
  constexpr int NA = CX ? 2 * C::NTILE : C::NTILE;
  double* red = reinterpret_cast<double*>(lds);   // [wave][acc][16]
  __syncthreads();                                // every wave is done with its ring
#pragma unroll
  for (int e = 0; e < NA; ++e) {
    double v = acc[e];
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if ((lane & 12) == 0) red[(wave * NA + e) * 16 + (lane >> 4) * 4 + (lane & 3)] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NA * 16; e += 512) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[w * NA * 16 + e];
    const int part = (e >> 4) / C::NTILE;          // 0: real parts, 1: imaginary parts
    const int tile = (e >> 4) % C::NTILE, ii = (e >> 2) & 3, jj = e & 3;
    int entry = -1;
    if (tile < C::NTS) {
      const int g = tile / NT, t = tile % NT, c = 4 * g + ii, i = 4 * t + jj;
      if (c < k && i < s) entry = i * k + c;
    } else {
      int t = 0, rem = tile - C::NTS;
      while (rem > t) { rem -= t + 1; ++t; }
      const int a = rem, i = 4 * a + ii, i2 = 4 * t + jj;
      if (i <= i2 && i2 < s) entry = k * s + gram_idx(i, i2);
    }
    if (entry >= 0) partial[((int64_t)entry * pnb + blockIdx.x) * (CX ? 2 : 1) + part] = v;
  }
}


// the copies of one slab: S columns (zeros beyond k), block columns Zb[:, 0:s) (zeros beyond s; in place: Zb = V + k ldv);
// rows past the range: zeros
template <int NGS, int NT>
__device__ __forceinline__ void blkm_issue(const double* __restrict__ V, int64_t ldv, int k, const double* __restrict__ Zb, int64_t ldz, int s,
                                           int64_t pack0, int64_t pe, int lane, uint32_t slab_lds, const double* __restrict__ zeros, bool nt) {
  using C = BlkMfma<NGS, NT>;
  const int64_t p = pack0 + (lane & 7);
  const bool in = p < pe;
  const int cl = lane >> 3;
#pragma unroll
  for (int j = 0; j < C::NJ; ++j) {
    const int c = j < C::NJS ? 8 * j + cl : 8 * (j - C::NJS) + cl;     // column inside its region
    const bool have = in && (j < C::NJS ? c < k : c < s);
    const double* src = have ? (j < C::NJS ? V + (int64_t)c * ldv : Zb + (int64_t)c * ldz) + p * 2 : zeros;
    if (nt) glds16_nt(src, slab_lds + (uint32_t)j * 1024u);
    else glds16(src, slab_lds + (uint32_t)j * 1024u);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// pass 1:  partial[i k + c] = S[:, c] . Z[:, i],  partial[k s + g(i, i2)] = Z[:, i] . Z[:, i2]        (k <= 4 NGS, s <= 4 NT)
// Z = Zb[:, 0:s): the columns V[:, k : k + s) (in place), or scratch columns the Newton chain was written to.
// Dynamic LDS: 8 waves x ring x slab.
// ---------------------------------------------------------------------------------------------------------------------------
// ComplexF64 (CX): the SAME kernel on the real view of the basis -- a complex column of n rows is a real column of 2 n rows
// (re, im interleaved; a 16-byte pack is one complex element, a slab 8 complex rows) -- plus the imaginary parts:
//   x^H y = sum_rows x_row y_row  +  i sum_rows x_row (P y)_row,     (P y)_re = y_im,  (P y)_im = -y_re
// P y is the operand read at the partner row (byte offset ^ 8) with the sign of the row's parity: one more LDS read and one more
// matrix instruction per tile.  `partial` then holds complex entries.
template <int NGS, int NT, bool CX = false>
__global__ void __launch_bounds__(512, 2)
    k_bdots_mfma(const double* __restrict__ V, int64_t ldv, int k, const double* __restrict__ Zb, int64_t ldz, int s, int ring,
                 double* __restrict__ partial, int pnb, const DevState* __restrict__ st, int dbg, const double* __restrict__ zeros) {
  if (st && st->breakdown >= 0) return;
  using C = BlkMfma<NGS, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* myring = lds_raw + (size_t)wave * ring * C::SLAB;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)myring;
  const int gat = (lane & 3) * 128 + (4 * ((lane >> 2) & 3) + (lane >> 4)) * 8;
  const double psign = ((lane >> 4) & 1) ? -1.0 : 1.0;   // (P y) of this lane's row: + y[partner] on real rows, - on imaginary ones
  constexpr int NA = CX ? 2 * C::NTILE : C::NTILE;
  double acc[NA];
#pragma unroll
  for (int e = 0; e < NA; ++e) acc[e] = 0.0;
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);   // (pack-granular on purpose: ranges aligned to whole tiles ran 4 % slower)
  const int niter = (int)((pe - pb + 63) / 64);
  const bool nt = (dbg & 64) != 0;
  auto issue = [&](int it, int sl) {
    if (dbg & 32) return;
    blkm_issue<NGS, NT>(V, ldv, k, Zb, ldz, s, pb + (int64_t)it * 64 + wave * 8, pe, lane, ring_lds + (uint32_t)(sl * C::SLAB), zeros, nt);
  };
  for (int it = 0; it < ring - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = ring - 1;
  for (int it = 0; it < niter; ++it) {
    wait_vm((ring - 2) * C::NJ);           // slab `it` has landed (copies of this wave complete in order)
    issue(it + ring - 1, sl_new);          // into the slot of slab it - 1 (its reads have returned: lgkmcnt(0) above)
    const unsigned char* slab = myring + (size_t)sl_cur * C::SLAB;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == ring ? 0 : sl_cur + 1;
    if (dbg & 16) continue;
    double a[NGS], z[NT], zp[CX ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) z[t] = *reinterpret_cast<const double*>(slab + C::z_off(t) + gat);
    if constexpr (CX) {
#pragma unroll
      for (int t = 0; t < NT; ++t) zp[t] = psign * *reinterpret_cast<const double*>(slab + C::z_off(t) + (gat ^ 8));
    }
#pragma unroll
    for (int g = 0; g < NGS; ++g) a[g] = *reinterpret_cast<const double*>(slab + C::s_off(g) + gat);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q <= t; ++q) {
        acc[C::NTS + C::gt(q, t)] = mfma4(z[q], z[t], acc[C::NTS + C::gt(q, t)]);
        if constexpr (CX) acc[C::NTILE + C::NTS + C::gt(q, t)] = mfma4(z[q], zp[t], acc[C::NTILE + C::NTS + C::gt(q, t)]);
      }
#pragma unroll
    for (int g = 0; g < NGS; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[g * NT + t] = mfma4(a[g], z[t], acc[g * NT + t]);
        if constexpr (CX) acc[C::NTILE + g * NT + t] = mfma4(a[g], zp[t], acc[C::NTILE + g * NT + t]);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  blkm_finish<NGS, NT, CX>(acc, lds_raw, lane, wave, k, s, partial, pnb);
}

// ---------------------------------------------------------------------------------------------------------------------------
// pass 2:  Qt = Z R1inv - S coefp -> V[:, k : k + s)  (Z = Zb[:, 0:s): the same columns when in place, or scratch columns the
// Newton chain was written to);  partial[i k + c] = S[:, c] . Qt[:, i];  Gram of Qt.
// Per slab: Qt tile t = sum_g S_g M_{g,t} + sum_{u <= t} Z_u M'_{u,t}  (M = -coefp, M' = R1inv: upper triangular, so the
// tiles below the diagonal are skipped), 4 x 4 coefficient tiles read from LDS in the B layout; the result registers are
// stored, and ARE the operands of the inner products (as B: Qt, as A: Qt^T).
// Dynamic LDS: 8 waves x ring x slab | coefficient tiles.
// ---------------------------------------------------------------------------------------------------------------------------
// ComplexF64 (CX), on the real view as in pass 1: a complex coefficient m acts on a column as  m_re x + m_im (J x),
// (J x)_re = -x_im, (J x)_im = x_re  (the operand read at the partner row with a sign); the inner products take P Qt from the
// result registers of the partner rows (lane ^ 16).  coefp / r1inv are complex then (ldc, s in complex elements).
template <int NGS, int NT, bool CX = false>
__global__ void __launch_bounds__(512, 2)
    k_bupdate_mfma(double* __restrict__ V, int64_t ldv, int k, const double* __restrict__ Zb, int64_t ldz, int s, int ring,
                   const double* __restrict__ coefp, int ldc, const double* __restrict__ r1inv, double* __restrict__ partial, int pnb,
                   const DevState* __restrict__ st, int dbg, const double* __restrict__ zeros) {
  if (st && st->breakdown >= 0) return;
  using C = BlkMfma<NGS, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* myring = lds_raw + (size_t)wave * ring * C::SLAB;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)myring;
  constexpr int CD = CX ? 2 : 1;   // doubles per coefficient
  double* mt = reinterpret_cast<double*>(lds_raw + (size_t)8 * ring * C::SLAB);   // [part][tile][k][j]  (part: real, imaginary)
  for (int e = threadIdx.x; e < CD * C::NTILE * 16; e += 512) {
    const int part = (e >> 4) / C::NTILE;
    const int tile = (e >> 4) % C::NTILE, kk = (e >> 2) & 3, jj = e & 3;
    double v = 0.0;
    if (tile < C::NTS) {
      const int g = tile / NT, t = tile % NT, c = 4 * g + kk, i = 4 * t + jj;
      if (c < k && i < s) v = -coefp[(c + (int64_t)i * ldc) * CD + part];
    } else {
      int t = 0, rem = tile - C::NTS;
      while (rem > t) { rem -= t + 1; ++t; }
      const int l = 4 * rem + kk, i = 4 * t + jj;
      if (l <= i && i < s) v = r1inv[(l + i * s) * CD + part];
    }
    mt[e] = v;
  }
  __syncthreads();
  const int lin = lane * 8;
  const int gat = (lane & 3) * 128 + (4 * ((lane >> 2) & 3) + (lane >> 4)) * 8;
  const double jsign = (lane & 1) ? 1.0 : -1.0;           // (J x) of the linear pattern's row (row = lane % 16)
  const double psign = ((lane >> 4) & 1) ? -1.0 : 1.0;    // (P y) of the result layout's row (row = 4 b + lane / 16)
  const double* mrd = mt + (lane >> 4) * 4 + (lane & 3);
  const double* mri = mrd + C::NTILE * 16;
  constexpr int NA = CX ? 2 * C::NTILE : C::NTILE;
  double acc[NA];
#pragma unroll
  for (int e = 0; e < NA; ++e) acc[e] = 0.0;
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);   // (pack-granular on purpose: ranges aligned to whole tiles ran 4 % slower)
  const int niter = (int)((pe - pb + 63) / 64);
  const bool nt = (dbg & 64) != 0;
  // stores: 8 bytes per lane straight from the result registers (16 lanes cover 128 contiguous bytes of a column).  Staging the
  // tiles in LDS for 16-byte stores was measured and is not faster (1 005 against 995 us at k = 21, s = 20), neither are
  // cacheable stores (1 050) nor writing out of place (1 000-1 030): tools/rw_streams.hip shows the memory system itself
  // at 960 us for this mix of 41 column streams in and 20 out IN PLACE, whatever the kernel does in between.
  // (exactly the tiles that have a column < s: a tile whose lanes are all predicated off issues NO store and would let the wait
  // below count one copy too few -- s = 13..16 on the 5-tile kernels, ComplexF64 s <= 4 on the 2-tile ones)
  const int nst = (dbg & 1) ? 0 : (s + 3) / 4;
  auto issue = [&](int it, int sl) {
    blkm_issue<NGS, NT>(V, ldv, k, Zb, ldz, s, pb + (int64_t)it * 64 + wave * 8, pe, lane, ring_lds + (uint32_t)(sl * C::SLAB), zeros, nt);
  };
  // this lane's element of a result tile: row 4 b + i of the slab, column j of the tile
  const int row = 4 * ((lane >> 2) & 3) + (lane >> 4), cj = lane & 3;
  double* zst = V + (int64_t)(k + cj) * ldv + row;
  for (int it = 0; it < ring - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = ring - 1;
  for (int it = 0; it < niter; ++it) {
    // queue of this wave, oldest first: copies(it) stores(it - ring + 1) copies(it + 1) ... copies(it + ring - 2) stores(it - 1)
    wait_vm((ring - 2) * C::NJ + (it < ring - 1 ? it : ring - 1) * nst);
    issue(it + ring - 1, sl_new);
    const unsigned char* slab = myring + (size_t)sl_cur * C::SLAB;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == ring ? 0 : sl_cur + 1;
    const int64_t pack0 = pb + (int64_t)it * 64 + wave * 8;
    double x[NGS], zx[NT], d[NT];
#pragma unroll
    for (int g = 0; g < NGS; ++g) x[g] = *reinterpret_cast<const double*>(slab + C::s_off(g) + lin);
#pragma unroll
    for (int u = 0; u < NT; ++u) zx[u] = *reinterpret_cast<const double*>(slab + C::z_off(u) + lin);
#pragma unroll
    for (int t = 0; t < NT; ++t) d[t] = 0.0;
#pragma unroll
    for (int g = 0; g < NGS; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t) d[t] = mfma4(x[g], mrd[(g * NT + t) * 16], d[t]);
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int t = u; t < NT; ++t) d[t] = mfma4(zx[u], mrd[(C::NTS + C::gt(u, t)) * 16], d[t]);
    if constexpr (CX) {
#pragma unroll
      for (int g = 0; g < NGS; ++g) {
        const double xj = jsign * *reinterpret_cast<const double*>(slab + C::s_off(g) + (lin ^ 8));
#pragma unroll
        for (int t = 0; t < NT; ++t) d[t] = mfma4(xj, mri[(g * NT + t) * 16], d[t]);
      }
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const double zj = jsign * *reinterpret_cast<const double*>(slab + C::z_off(u) + (lin ^ 8));
#pragma unroll
        for (int t = u; t < NT; ++t) d[t] = mfma4(zj, mri[(C::NTS + C::gt(u, t)) * 16], d[t]);
      }
    }
    if (nst) {
      const bool ok = pack0 + (row >> 1) < pe;
      double* dst = zst + pack0 * 2;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (4 * t < s) {   // wave-uniform: tile t issues a store (lane cj = 0 of a row in range is active) or none at all
          if (ok && 4 * t + cj < s) gst8_nt(dst + (int64_t)(4 * t) * ldv, d[t]);
        }
      }
    }
    double a[NGS], dp[CX ? NT : 1];
#pragma unroll
    for (int g = 0; g < NGS; ++g) a[g] = *reinterpret_cast<const double*>(slab + C::s_off(g) + gat);
    if constexpr (CX) {
#pragma unroll
      for (int t = 0; t < NT; ++t) dp[t] = psign * __shfl_xor(d[t], 16, 64);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q <= t; ++q) {
        acc[C::NTS + C::gt(q, t)] = mfma4(d[q], d[t], acc[C::NTS + C::gt(q, t)]);
        if constexpr (CX) acc[C::NTILE + C::NTS + C::gt(q, t)] = mfma4(d[q], dp[t], acc[C::NTILE + C::NTS + C::gt(q, t)]);
      }
#pragma unroll
    for (int g = 0; g < NGS; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[g * NT + t] = mfma4(a[g], d[t], acc[g * NT + t]);
        if constexpr (CX) acc[C::NTILE + g * NT + t] = mfma4(a[g], dp[t], acc[C::NTILE + g * NT + t]);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  blkm_finish<NGS, NT, CX>(acc, lds_raw, lane, wave, k, s, partial, pnb);
}


// ---------------------------------------------------------------------------------------------------------------------------
// RESTART ROTATION + pass 1 in ONE sweep (src/run.jl:363-365 followed by the first block of src/expansion.jl:116-133).
//   V[:, out0 : knew) <- V[:, 0:cin) M          M = T Q of the restart, cin x (knew - out0), leading dimension cin
//   partial           <- S_new' Z, Z' Z          S_new = V[:, 0:knew) after the rotation (columns below out0 are untouched),
//                                                Z = Zb[:, 0:s): the Newton chain, written to scratch columns BEFORE this kernel
// Separately the two kernels read cin + (knew + s) columns and write knew - out0; fused, the rotated columns never travel back
// in: cin + s read, knew - out0 written (headline: 82 column passes instead of 102).  The rotated tile of a slab is the result
// register of its matrix instructions -- stored, and, read as the A operand (= its transpose), multiplied with the block's
// columns.  Columns below out0 take an identity block of M (they are inputs of the inner products too).
// Dynamic LDS: 8 waves x ring x slab(cin + s columns) | coefficient tiles.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NGX, int NTK, int NT> struct BlkRot {
  static constexpr int NJX = (NGX + 1) / 2, NJZ = (NT + 1) / 2, NJ = NJX + NJZ;
  static constexpr int SLAB = NJ * 1024;
  static constexpr int NTM = NGX * NTK;                                             // coefficient tiles
  static constexpr int NTS = NTK * NT, NTG = NT * (NT + 1) / 2, NTILE = NTS + NTG;  // result tiles (as BlkMfma<NTK, NT>)
  __host__ __device__ static constexpr int x_off(int g) { return (g / 2) * 1024 + (g % 2) * 512; }
  __host__ __device__ static constexpr int z_off(int t) { return NJX * 1024 + (t / 2) * 1024 + (t % 2) * 512; }
  __host__ __device__ static constexpr int gt(int a, int t) { return t * (t + 1) / 2 + a; }
  static constexpr size_t lds_bytes(int ring, bool cx = false) { return (size_t)8 * ring * SLAB + (cx ? 2 : 1) * (size_t)NTM * 128; }
};

// ComplexF64 (CX), on the real view as in the two passes: M is complex (interleaved), a coefficient m acts on a column as
// m_re x + m_im (J x) -- the operand read at the partner row with a sign --, the inner products take P z from the partner rows of
// the block's columns (one more LDS read and one more matrix instruction per tile); `partial` holds complex entries.
template <int NGX, int NTK, int NT, bool CX = false>
__global__ void __launch_bounds__(512, 2)
    k_brotdots_mfma(double* __restrict__ V, int64_t ldv, int cin, const double* __restrict__ M, int out0, int knew,
                    const double* __restrict__ Zb, int64_t ldz, int s, int ring, double* __restrict__ partial, int pnb, int dbg,
                    const double* __restrict__ zeros) {
  using C = BlkRot<NGX, NTK, NT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* myring = lds_raw + (size_t)wave * ring * C::SLAB;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)myring;
  constexpr int CD = CX ? 2 : 1;   // doubles per coefficient
  double* mt = reinterpret_cast<double*>(lds_raw + (size_t)8 * ring * C::SLAB);   // [part][g][t][kk][jj]  (part: real, imaginary)
  for (int e = threadIdx.x; e < CD * C::NTM * 16; e += 512) {
    const int part = (e >> 4) / C::NTM;
    const int tile = (e >> 4) % C::NTM, kk = (e >> 2) & 3, jj = e & 3;
    const int g = tile / NTK, t = tile % NTK, c = 4 * g + kk, j = 4 * t + jj;
    double v = 0.0;
    if (c < cin && j < knew) v = j < out0 ? ((c == j && part == 0) ? 1.0 : 0.0) : M[(c + (int64_t)(j - out0) * cin) * CD + part];
    mt[e] = v;
  }
  __syncthreads();
  const int lin = lane * 8;
  const int gat = (lane & 3) * 128 + (4 * ((lane >> 2) & 3) + (lane >> 4)) * 8;
  const double jsign = (lane & 1) ? 1.0 : -1.0;           // (J x) of the linear pattern's row (row = lane % 16)
  const double psign = ((lane >> 4) & 1) ? -1.0 : 1.0;    // (P y) of the gather pattern's row
  const double* mrd = mt + (lane >> 4) * 4 + (lane & 3);
  const double* mri = mrd + C::NTM * 16;
  constexpr int NA = CX ? 2 * C::NTILE : C::NTILE;
  double acc[NA];
#pragma unroll
  for (int e = 0; e < NA; ++e) acc[e] = 0.0;
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);
  const int niter = (int)((pe - pb + 63) / 64);
  const bool nt = (dbg & 64) != 0;
  // store instructions per slab: the 4-column tiles that hold an output column (wave-uniform: the count must be exact)
  const int t_lo = out0 >> 2, t_hi = (knew - 1) >> 2;
  const int nst = (dbg & 1) ? 0 : (t_hi - t_lo + 1);
  auto issue = [&](int it, int sl) {
    const int64_t p = pb + (int64_t)it * 64 + wave * 8 + (lane & 7);
    const bool in = p < pe;
    const int cl = lane >> 3;
    const uint32_t slab_lds = ring_lds + (uint32_t)(sl * C::SLAB);
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
      const int c = j < C::NJX ? 8 * j + cl : 8 * (j - C::NJX) + cl;
      const bool have = in && (j < C::NJX ? c < cin : c < s);
      const double* src = have ? (j < C::NJX ? V + (int64_t)c * ldv : Zb + (int64_t)c * ldz) + p * 2 : zeros;
      if (nt) glds16_nt(src, slab_lds + (uint32_t)j * 1024u);
      else glds16(src, slab_lds + (uint32_t)j * 1024u);
    }
  };
  const int row = 4 * ((lane >> 2) & 3) + (lane >> 4), cj = lane & 3;
  double* vst = V + (int64_t)cj * ldv + row;
  for (int it = 0; it < ring - 1; ++it) issue(it, it);
  int sl_cur = 0, sl_new = ring - 1;
  for (int it = 0; it < niter; ++it) {
    wait_vm((ring - 2) * C::NJ + (it < ring - 1 ? it : ring - 1) * nst);
    issue(it + ring - 1, sl_new);
    const unsigned char* slab = myring + (size_t)sl_cur * C::SLAB;
    sl_new = sl_cur;
    sl_cur = sl_cur + 1 == ring ? 0 : sl_cur + 1;
    const int64_t pack0 = pb + (int64_t)it * 64 + wave * 8;
    double x[NGX], d[NTK], z[NT], zp[CX ? NT : 1];
#pragma unroll
    for (int g = 0; g < NGX; ++g) x[g] = *reinterpret_cast<const double*>(slab + C::x_off(g) + lin);
#pragma unroll
    for (int u = 0; u < NT; ++u) z[u] = *reinterpret_cast<const double*>(slab + C::z_off(u) + gat);
    if constexpr (CX) {
#pragma unroll
      for (int u = 0; u < NT; ++u) zp[u] = psign * *reinterpret_cast<const double*>(slab + C::z_off(u) + (gat ^ 8));
    }
#pragma unroll
    for (int t = 0; t < NTK; ++t) d[t] = 0.0;
#pragma unroll
    for (int g = 0; g < NGX; ++g)
#pragma unroll
      for (int t = 0; t < NTK; ++t) d[t] = mfma4(x[g], mrd[(g * NTK + t) * 16], d[t]);
    if constexpr (CX) {
#pragma unroll
      for (int g = 0; g < NGX; ++g) {
        const double xj = jsign * *reinterpret_cast<const double*>(slab + C::x_off(g) + (lin ^ 8));
#pragma unroll
        for (int t = 0; t < NTK; ++t) d[t] = mfma4(xj, mri[(g * NTK + t) * 16], d[t]);
      }
    }
    if (nst) {
      const bool ok = pack0 + (row >> 1) < pe;
      double* dst = vst + pack0 * 2;
#pragma unroll
      for (int t = 0; t < NTK; ++t) {
        if (t >= t_lo && t <= t_hi) {   // (uniform)
          const int j = 4 * t + cj;
          if (ok && j >= out0 && j < knew) gst8_nt(dst + (int64_t)(4 * t) * ldv, d[t]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int q = 0; q <= u; ++q) {
        acc[C::NTS + C::gt(q, u)] = mfma4(z[q], z[u], acc[C::NTS + C::gt(q, u)]);
        if constexpr (CX) acc[C::NTILE + C::NTS + C::gt(q, u)] = mfma4(z[q], zp[u], acc[C::NTILE + C::NTS + C::gt(q, u)]);
      }
#pragma unroll
    for (int t = 0; t < NTK; ++t)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        acc[t * NT + u] = mfma4(d[t], z[u], acc[t * NT + u]);
        if constexpr (CX) acc[C::NTILE + t * NT + u] = mfma4(d[t], zp[u], acc[C::NTILE + t * NT + u]);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  blkm_finish<NTK, NT, CX>(acc, lds_raw, lane, wave, knew, s, partial, pnb);
}

}  // namespace ksd
