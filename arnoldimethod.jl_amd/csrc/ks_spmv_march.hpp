// SpMV, stencil-mask layout, MARCHING form (gfx950):  y = A x  or the Newton step  y = sigma (A x - theta x)  of mul!(y, A, x),
// src/expansion.jl:121 -- the paired kernel k_spmv_stencil2 (ks_kernels.hpp) turned into a persistent, software-pipelined loop
// with a lean instruction stream.
//
// What the counters say about k_spmv_stencil2 (profiles/r06_spmv_probe.txt): 22.2 M vector instructions per launch (282 per wave
// of 128 rows: 64-bit clamped addresses and "which half of the clamped pair" selects for every slot) keep the vector ALUs busy
// for about half of its 44 us, and a wave lives for one load round trip + its arithmetic + one store round trip (s_endpgm
// waits for the stores), so what a CU holds in flight is bounded by 32 waves x 128 rows.  Here
//   * a workgroup walks tiles t, t + G, t + 2 G, ... (G workgroups; per XCD a contiguous range of tiles, so the rows in flight
//     on one XCD are a band that sweeps through its planes and the z - 1 / z + 1 taps hit the same L2);
//   * the loads of the NEXT tile (NS pairs + the rows' own pair + the masks) are issued before the arithmetic of the current
//     one: two tiles per wave in flight, stores never waited for;
//   * interior tiles (every slot of every row inside [0, n)) need no clamping at all: the address of slot k is a SCALAR base
//     (x + 512 t + delta_k) plus the lane's constant byte offset, the loaded pair IS the wanted pair; the few tiles next to the
//     ends of the vector take the clamped path of k_spmv_stencil2;
//   * products are masked with and-masks from v_bfe_i32 (+0.0 for an absent slot: s + 0.0 == s bit for bit, s is never -0.0)
//     instead of compare + two selects.
// Products rounded separately, added in slot order: y is bit-identical to every other layout.
#pragma once

#include <algorithm>
#include <type_traits>

#include "ks_kernels.hpp"

namespace ksd {

typedef double f64x2m __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double and_mask(double p, int bm) {
  const uint64_t u = __builtin_bit_cast(uint64_t, p);
  const uint32_t lo = (uint32_t)u & (uint32_t)bm, hi = (uint32_t)(u >> 32) & (uint32_t)bm;
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// tile of workgroup slot `slot` (0 .. S - 1) of XCD `xcd` in round `it`: XCD j owns tiles [T_j, T_{j+1})
__device__ __forceinline__ int march_tile(int xcd, int slot, int it, int S, int ntiles, int& tend) {
  const int q = ntiles >> 3, r = ntiles & 7;
  const int t0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  tend = t0 + (xcd < r ? q + 1 : q);
  return t0 + slot + it * S;
}

// loads and stores of the loop: scalar base + the lane's constant byte offset, issued from inline assembly and counted by hand
__device__ __forceinline__ void march_ld16(f64x2m& dst, uint32_t voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void march_ld_u16(uint32_t& dst, uint32_t voff, const void* sbase) {
  asm volatile("global_load_ushort %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void march_st16(uint32_t voff, f64x2m v, void* sbase, bool plain) {
  if (plain) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}

__device__ __forceinline__ void march_ld8(double& dst, uint32_t voff, const void* sbase) {
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// lane i <- lane i - 1 (lane 0 keeps `edge`) / lane i <- lane i + 1 (lane 63 keeps `edge`): DPP wave shifts, no LDS, no memory
__device__ __forceinline__ double wave_shr1(double edge, double src) {
  const uint64_t e = __builtin_bit_cast(uint64_t, edge), v = __builtin_bit_cast(uint64_t, src);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)e, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(e >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double wave_shl1(double edge, double src) {
  const uint64_t e = __builtin_bit_cast(uint64_t, edge), v = __builtin_bit_cast(uint64_t, src);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)e, (int)(uint32_t)v, 0x130, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(e >> 32), (int)(uint32_t)(v >> 32), 0x130, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// KOWN: the slot with delta 0 (its pair is the rows' own pair of the Newton step), -1: none -- the own pair is one more load.
// KM1 / KP1 (both or neither, and only with KOWN >= 0): the slots with delta -1 / +1.  Their pairs (x[r-1], x[r]) and (x[r+1], x[r+2])
// are NOT loaded: x[r], x[r+1] are the own pair, x[r-1] / x[r+2] the neighbouring lanes' (wave shifts), and the two values a wave
// lacks at its ends come with one 8-byte load in which the lanes read only two distinct addresses.  Two 16-byte loads of nine
// vector-memory operations per tile less: in the solver's chain the address/L1 path is what the taps cost (one +- pair of taps
// = 4-6 us of a 50-us product, profiles/r06_spmv_columns.txt).
template <int NSLOT, int KOWN, int KM1 = -1, int KP1 = -1>
__global__ void __launch_bounds__(kBlock)
    k_spmv_stencil_march(const uint16_t* __restrict__ mask2, const StencilDict<double> d, const double* __restrict__ x,
                         double* __restrict__ y, int64_t n, int ntiles, const DevState* __restrict__ st, int shifted, double theta,
                         double sigma) {
  if (st && st->breakdown >= 0) return;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;   // (the host launches a multiple of 8 workgroups)
  int64_t dmin = 0, dmax = 0;
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    dmin = d.delta[k] < dmin ? d.delta[k] : dmin;
    dmax = d.delta[k] > dmax ? d.delta[k] : dmax;
  }
  const uint32_t lane_b = threadIdx.x * 16u, lane_m = threadIdx.x * 2u;   // byte offsets of this lane's pair / mask inside a tile
  const bool plain_st = (shifted & 2) != 0;
  int tend;
  int t = march_tile(xcd, slot, 0, S, ntiles, tend);
  constexpr bool NEAR = KM1 >= 0 && KP1 >= 0 && KOWN >= 0;
  static_assert((KM1 >= 0) == (KP1 >= 0) && (KM1 < 0 || KOWN >= 0), "delta -1 and +1 come from the own pair: all three slots or none");
  constexpr int NL = NSLOT + (KOWN < 0 ? 1 : 0) + 1 - (NEAR ? 1 : 0);   // loads of a tile (NEAR: two pairs less, one edge load more)
  // the wave's two missing neighbours: lanes 0-31 read x[first row - 1], lanes 32-63 x[last row + 1] (two distinct addresses)
  // (byte offsets from x + first row of the tile - 1: the 32-bit offset of a scalar-base load is unsigned)
  const uint32_t lane_e = ((threadIdx.x >> 6) * 128u + ((threadIdx.x & 32u) ? 129u : 0u)) * 8u;

  struct Tile {
    f64x2m v[NSLOT], own;
    double edge;
    uint32_t m;
  };
  auto issue = [&](int tt, Tile& T) {
    const int64_t r0 = (int64_t)tt * 512;
    const double* b[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      b[k] = x + (r0 + d.delta[k]);
      asm volatile("" : "+s"(b[k]));
    }
    const double* bo = x + r0;
    const uint16_t* bm = mask2 + (r0 >> 1);
    const double* be = x + (r0 - 1);
    asm volatile("" : "+s"(bo));
    asm volatile("" : "+s"(bm));
    asm volatile("" : "+s"(be));
    asm volatile("s_nop 4" ::: "memory");   // (a base that came through v_readfirstlane must not be read by the next 5 instructions)
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
      if (!(NEAR && (k == KM1 || k == KP1))) march_ld16(T.v[k], lane_b, b[k]);
    if constexpr (KOWN < 0) march_ld16(T.own, lane_b, bo);
    if constexpr (NEAR) march_ld8(T.edge, lane_e, be);
    march_ld_u16(T.m, lane_m, bm);
  };
  // nwait: vector-memory operations that may still be outstanding once this tile's loads have landed (wave-uniform; one of
  // 0, 1, NL, NL + 1).  ONE consumer site per register set: with two sites hipcc ties the set to different registers per site
  // and copies it in front of the wait (seen in the ISA), which reads registers whose loads have not landed.
  auto finish = [&](int tt, Tile& T, int nwait) {
    // a wait-only statement, THEN empty statements that redefine the tile's registers: consumers depend on the redefinitions
    // and cannot be hoisted above them (hipcc hoisted the v_bfe of the masks right behind the load without them), the
    // redefinitions cannot move above the wait (volatile order), and any register copy hipcc makes for a tie lands behind it
    if (nwait == NL + 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NL + 1) : "memory");
    else if (nwait == NL) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NL) : "memory");
    else if (nwait == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
      if (!(NEAR && (k == KM1 || k == KP1))) asm volatile("" : "+v"(T.v[k]));
    if constexpr (KOWN < 0) asm volatile("" : "+v"(T.own));
    if constexpr (NEAR) {
      asm volatile("" : "+v"(T.edge));
      const f64x2m o = T.v[KOWN < 0 ? 0 : KOWN];
      T.v[KM1 < 0 ? 0 : KM1].x = wave_shr1(T.edge, o.y);   // x[r - 1]: the lane below's second row
      T.v[KM1 < 0 ? 0 : KM1].y = o.x;
      T.v[KP1 < 0 ? 0 : KP1].x = o.y;
      T.v[KP1 < 0 ? 0 : KP1].y = wave_shl1(T.edge, o.x);   // x[r + 2]: the lane above's first row
    }
    asm volatile("" : "+v"(T.m));
    const int m0 = (int)(T.m & 0xffu), m1 = (int)(T.m >> 8);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const double p0 = mul_nc(d.val[k], T.v[k].x), p1 = mul_nc(d.val[k], T.v[k].y);
      s0 = add_(s0, and_mask(p0, __builtin_amdgcn_sbfe(m0, k, 1)));
      s1 = add_(s1, and_mask(p1, __builtin_amdgcn_sbfe(m1, k, 1)));
    }
    if (shifted) {
      const f64x2m o = KOWN < 0 ? T.own : T.v[KOWN < 0 ? 0 : KOWN];
      s0 = scl(sub_s(s0, mul_(theta, o.x)), sigma);
      s1 = scl(sub_s(s1, mul_(theta, o.y)), sigma);
    }
    f64x2m o2;
    o2.x = s0;
    o2.y = s1;
    double* by = y + (int64_t)tt * 512;
    asm volatile("" : "+s"(by));
    march_st16(lane_b, o2, by, plain_st);
  };
  // the clamped path of k_spmv_stencil2 for the tiles next to the ends of the vector (no pipelining)
  auto edge = [&](int tt) {
    const int64_t r = (int64_t)tt * 512 + 2 * (int64_t)threadIdx.x;
    if (r >= n) return;
    const bool two = r + 1 < n;
    const uint32_t m = mask2[r >> 1];
    const uint32_t m0 = m & 0xffu, m1 = m >> 8;
    const int64_t cmax = n - 2;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int64_t c = r + d.delta[k];
      int64_t lo = c < 0 ? 0 : (c > cmax ? cmax : c);
      if (cmax < 0) lo = 0;
      const int sh = (int)(c - lo);
      double xa, xb;
      if (n >= 2) ld_pair_u(x + lo, xa, xb);
      else { xa = x[0]; xb = x[0]; }
      const double v0 = sh == 1 ? xb : xa, v1 = sh == -1 ? xa : xb;
      const double p0 = mul_nc(d.val[k], v0), p1 = mul_nc(d.val[k], v1);
      s0 = ((m0 >> k) & 1u) ? add_(s0, p0) : s0;
      s1 = ((m1 >> k) & 1u) ? add_(s1, p1) : s1;
    }
    if (shifted) {
      double x0, x1;
      if (two) ld_pair_u(x + r, x0, x1);
      else { x0 = x[r]; x1 = x0; }
      s0 = scl(sub_s(s0, mul_(theta, x0)), sigma);
      s1 = scl(sub_s(s1, mul_(theta, x1)), sigma);
    }
    if (two) {
      if (plain_st) st_pack(y + r, make_double2(s0, s1));
      else st_pack_nt(y + r, make_double2(s0, s1));
    } else {
      if (plain_st) y[r] = s0;
      else st_elem_nt(y + r, s0);
    }
  };

  // the interior tiles (rows [512 t, 512 t + 512) with every slot inside [0, n) and the last lane's pairs inside x) are one
  // range [t_lo, t_hi) of tile indices; the others sit at the two ends of the vector
  int64_t lo64 = (-dmin + 511) / 512, hi64 = (n - dmax - 1) / 512;   // r0 + dmin >= 0;  r0 + 512 + dmax + 1 <= n
  if (n - dmax - 1 < 0) hi64 = 0;
  if (hi64 > ntiles) hi64 = ntiles;
  if (lo64 > hi64) lo64 = hi64;
  const int t_lo = (int)lo64, t_hi = (int)hi64;
  // edge tiles of this workgroup's sequence first (a handful per launch, on the first and last XCD only)
  if (t < t_lo || tend > t_hi) {
    for (int te = t; te < tend; te += S)
      if (te < t_lo || te >= t_hi) edge(te);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  while (t < t_lo) t += S;
  const int hi = tend < t_hi ? tend : t_hi;
  if (t >= hi) return;
  // software pipeline over the interior tiles, unrolled by two so that the two register sets never move.  Queue of a wave,
  // oldest first:  loads(t) [store(t - S)] loads(t + S) | store(t) ...: the loads of tile t have landed when at most
  // NL (first tile) / NL + 1 (the store of the tile before) later operations are outstanding, 1 / 0 when nothing was issued behind
  Tile A, B;
  issue(t, A);
  bool first = true;
  for (;;) {
    const int tn = t + S;
    const bool has_b = tn < hi;
    if (has_b) issue(tn, B);
    finish(t, A, has_b ? (first ? NL : NL + 1) : (first ? 0 : 1));
    first = false;
    if (!has_b) break;
    t = tn + S;
    const bool has_a = t < hi;
    if (has_a) issue(t, A);
    finish(tn, B, has_a ? NL + 1 : 1);
    if (!has_a) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------------------
// WINDOW form of the marching kernel: the NEAR taps (|delta| <= 256: -nx, -1, 0, +1, +nx of a 3-D grid) of a tile come from ONE
// copy of the rows [512 t - 256, 512 t + 768) in LDS -- 8 KiB per tile, filled by asynchronous global -> LDS copies (two 1-KiB
// pieces per wave, no registers), shared by the four waves -- instead of five 16-byte gathers per lane.  In the chain the solver
// runs, every pair of taps that goes through the address / L1 path costs 5-8 us of a 50-us product whether or not its lines are
// in L2 (profiles/r06_spmv_columns.txt: 1 slot 36 us, +-1 38, +-nx 45, +-P 50): per tile the vector-memory path moves 8 (window) +
// 8 (two far pairs) + 4 (store) + 0.5 KiB instead of 36.5.  Far taps and masks as in k_spmv_stencil_march (registers, one tile
// ahead); two window buffers, ONE workgroup barrier per tile: behind it every wave's pieces of tile t have landed and everybody
// has finished reading the buffer of tile t - 1, which the copies of tile t + 1 then overwrite.
//   NEARM: bit k = slot k is a near tap (the host checks |delta_k| <= 256 for those, > 256 for the others)
//   ODDM:  bit k = near slot k is read as two 8-byte elements (delta odd or unknown); clear: one aligned 16-byte read (delta even)
//   KOWN:  the near slot with delta 0 (the rows' own pair of the Newton step; required)
__device__ __forceinline__ void march_glds16(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int NSLOT, unsigned NEARM, unsigned ODDM, int KOWN>
__global__ void __launch_bounds__(kBlock)
    k_spmv_stencil_marchw(const uint16_t* __restrict__ mask2, const StencilDict<double> d, const double* __restrict__ x,
                          double* __restrict__ y, int64_t n, int ntiles, const DevState* __restrict__ st, int shifted, double theta,
                          double sigma) {
  static_assert(KOWN >= 0 && ((NEARM >> KOWN) & 1u), "the own pair is a near tap");
  if (st && st->breakdown >= 0) return;
  __shared__ __attribute__((aligned(16))) unsigned char win[2][8192];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  int64_t dmin = -256, dmax = 256;
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    dmin = d.delta[k] < dmin ? d.delta[k] : dmin;
    dmax = d.delta[k] > dmax ? d.delta[k] : dmax;
  }
  const uint32_t lane_b = threadIdx.x * 16u, lane_m = threadIdx.x * 2u;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane_w = (threadIdx.x & 63u) * 16u;          // this lane's 16 bytes inside a 1-KiB piece
  const uint32_t win0 = (uint32_t)(uintptr_t)&win[0][0];
  const bool plain_st = (shifted & 2) != 0;
  int tend;
  int t = march_tile(xcd, slot, 0, S, ntiles, tend);
  constexpr int NFAR = NSLOT - __builtin_popcount(NEARM & ((1u << NSLOT) - 1u));
  constexpr int NL = 2 + NFAR + 1;   // vector-memory loads of a tile and wave: two window pieces, the far pairs, the masks

  struct Tile {
    f64x2m v[NSLOT];   // (far slots only)
    uint32_t m;
  };
  auto issue = [&](int tt, Tile& T, int buf) {
    const int64_t r0 = (int64_t)tt * 512;
    const double* b[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      b[k] = x + (r0 + d.delta[k]);
      asm volatile("" : "+s"(b[k]));
    }
    const double* bw = x + (r0 - 256 + (int64_t)wave * 256);   // this wave's two pieces: window elements [256 w, 256 w + 256)
    const uint16_t* bm = mask2 + (r0 >> 1);
    uint32_t ldst = win0 + (uint32_t)buf * 8192u + (uint32_t)wave * 2048u;
    asm volatile("" : "+s"(bw));
    asm volatile("" : "+s"(bm));
    asm volatile("" : "+s"(ldst));
    asm volatile("s_nop 4" ::: "memory");
    march_glds16(lane_w, bw, ldst);
    march_glds16(lane_w + 1024u, bw, ldst + 1024u);
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
      if (!((NEARM >> k) & 1u)) march_ld16(T.v[k], lane_b, b[k]);
    march_ld_u16(T.m, lane_m, bm);
  };
  // behind the wait and the barrier: near pairs from the window, far pairs from the registers
  auto finish = [&](int tt, Tile& T, int buf, bool first) {
    if (first) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");   // (the store of the tile before may still be on its way)
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
      if (!((NEARM >> k) & 1u)) asm volatile("" : "+v"(T.v[k]));
    asm volatile("" : "+v"(T.m));
  };
  auto compute = [&](int tt, Tile& T, int buf) {
    const unsigned char* w = &win[0][0] + buf * 8192 + (256 * 8) + threadIdx.x * 16;   // element 256 + 2 T: this lane's own pair
    f64x2m v[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      if ((NEARM >> k) & 1u) {
        const int off = (int)d.delta[k] * 8;   // scalar
        if (((ODDM >> k) & 1u) == 0) {   // (a constant once the loop is unrolled)
          v[k] = *reinterpret_cast<const f64x2m*>(w + off);
        } else {
          typedef double f64x2u8 __attribute__((ext_vector_type(2), aligned(8)));
          const f64x2u8 p = *reinterpret_cast<const f64x2u8*>(w + off);
          v[k].x = p.x;
          v[k].y = p.y;
        }
      } else {
        v[k] = T.v[k];
      }
    }
    const int m0 = (int)(T.m & 0xffu), m1 = (int)(T.m >> 8);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const double p0 = mul_nc(d.val[k], v[k].x), p1 = mul_nc(d.val[k], v[k].y);
      s0 = add_(s0, and_mask(p0, __builtin_amdgcn_sbfe(m0, k, 1)));
      s1 = add_(s1, and_mask(p1, __builtin_amdgcn_sbfe(m1, k, 1)));
    }
    if (shifted) {
      s0 = scl(sub_s(s0, mul_(theta, v[KOWN].x)), sigma);
      s1 = scl(sub_s(s1, mul_(theta, v[KOWN].y)), sigma);
    }
    f64x2m o2;
    o2.x = s0;
    o2.y = s1;
    double* by = y + (int64_t)tt * 512;
    asm volatile("" : "+s"(by));
    // (the window reads above have returned before the store is issued: their values are its operands; the next barrier is
    // therefore behind every read of this buffer)
    march_st16(lane_b, o2, by, plain_st);
  };
  // the clamped path of k_spmv_stencil2 for the tiles next to the ends of the vector
  auto edge = [&](int tt) {
    const int64_t r = (int64_t)tt * 512 + 2 * (int64_t)threadIdx.x;
    if (r >= n) return;
    const bool two = r + 1 < n;
    const uint32_t m = mask2[r >> 1];
    const uint32_t m0 = m & 0xffu, m1 = m >> 8;
    const int64_t cmax = n - 2;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int64_t c = r + d.delta[k];
      int64_t lo = c < 0 ? 0 : (c > cmax ? cmax : c);
      if (cmax < 0) lo = 0;
      const int sh = (int)(c - lo);
      double xa, xb;
      if (n >= 2) ld_pair_u(x + lo, xa, xb);
      else { xa = x[0]; xb = x[0]; }
      const double v0 = sh == 1 ? xb : xa, v1 = sh == -1 ? xa : xb;
      const double p0 = mul_nc(d.val[k], v0), p1 = mul_nc(d.val[k], v1);
      s0 = ((m0 >> k) & 1u) ? add_(s0, p0) : s0;
      s1 = ((m1 >> k) & 1u) ? add_(s1, p1) : s1;
    }
    if (shifted) {
      double x0, x1;
      if (two) ld_pair_u(x + r, x0, x1);
      else { x0 = x[r]; x1 = x0; }
      s0 = scl(sub_s(s0, mul_(theta, x0)), sigma);
      s1 = scl(sub_s(s1, mul_(theta, x1)), sigma);
    }
    if (two) {
      if (plain_st) st_pack(y + r, make_double2(s0, s1));
      else st_pack_nt(y + r, make_double2(s0, s1));
    } else {
      if (plain_st) y[r] = s0;
      else st_elem_nt(y + r, s0);
    }
  };
  // interior tiles: every slot of every row AND the whole window inside [0, n)
  int64_t lo64 = (-dmin + 511) / 512, hi64 = (n - (dmax > 768 ? dmax + 1 : 768)) / 512;   // r0 + dmin >= 0;  r0 + max(512 + dmax + 1, 768) <= n
  if (n - 512 - dmax - 1 < 0 || n < 768) hi64 = 0;
  else hi64 = std::min<int64_t>((n - 512 - dmax - 1) / 512, (n - 768) / 512) + 1;
  if (hi64 > ntiles) hi64 = ntiles;
  if (lo64 > hi64) lo64 = hi64;
  const int t_lo = (int)lo64, t_hi = (int)hi64;
  if (t < t_lo || tend > t_hi) {
    for (int te = t; te < tend; te += S)
      if (te < t_lo || te >= t_hi) edge(te);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  while (t < t_lo) t += S;
  const int hi = tend < t_hi ? tend : t_hi;
  // (every wave of a workgroup runs the same number of rounds: the barriers match)
  if (t >= hi) return;
  Tile A, B;
  issue(t, A, 0);
  bool first = true;
  for (;;) {
    const int tn = t + S;
    const bool has_b = tn < hi;
    finish(t, A, 0, first);
    if (has_b) issue(tn, B, 1);
    compute(t, A, 0);
    first = false;
    if (!has_b) break;
    t = tn + S;
    const bool has_a = t < hi;
    finish(tn, B, 1, false);
    if (has_a) issue(t, A, 0);
    compute(tn, B, 1);
    if (!has_a) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------------------
// Z-MARCHING form: a workgroup keeps ONE tile of 512 in-plane offsets and walks UP the planes of a z-range, row r -> r + P per
// step.  The far taps then never touch the memory path: x[r - P] is the own pair of the step before (kept in registers),
// x[r + P] the own pair of the NEXT plane's window, which is copied two steps ahead anyway -- per step and workgroup the
// vector-memory path moves one window (8 KiB), the masks and the store (4 KiB) instead of 36.5 KiB (k_spmv_stencil_march) or
// 20.5 (the window form).  Three window buffers (planes z, z + 1 landed, z + 2 in flight), one barrier per step.
// Work items: (in-plane tile, z-range); XCD j owns a contiguous range of the ceil(P / 512) in-plane tiles for ALL z-ranges, so
// that the overlapping windows of neighbouring tiles meet in one L2.  Steps whose windows leave [0, n) (first tile of the first
// plane, last tiles of the last planes) take the clamped path of k_spmv_stencil2.
//   KFM / KFP: the slots with delta -P / +P (the only far taps);  NEARM, ODDM, KOWN as in the window form;  nzr: z-ranges
//   TW: threads of a workgroup (256 / 512): the tile is 2 TW in-plane offsets, the window 2 TW + 512 rows (2x / 1.5x its tile)
template <int NSLOT, unsigned NEARM, unsigned ODDM, int KOWN, int KFM, int KFP, int NB = 3, int TW = 256>
__global__ void __launch_bounds__(TW)
    k_spmv_stencil_marchz(const uint16_t* __restrict__ mask2, const StencilDict<double> d, const double* __restrict__ x,
                          double* __restrict__ y, int64_t n, int nzr, const DevState* __restrict__ st, int shifted, double theta,
                          double sigma) {
  static_assert(KOWN >= 0 && ((NEARM >> KOWN) & 1u) && !((NEARM >> KFM) & 1u) && !((NEARM >> KFP) & 1u), "own pair near, -P / +P far");
  if (st && st->breakdown >= 0) return;
  constexpr int kTile = 2 * TW, kWinB = (kTile + 512) * 8, kPieces = (kTile + 512) / 128, kWaves = TW / 64;
  static_assert(TW == 256 || TW == 512, "four or eight waves");
  static_assert(NB == 3 || TW == 256, "the wait counts of the four-buffer form assume two window pieces per wave");
  static_assert(NB == 3 || NB == 4, "planes z, z + 1 resident; one or two more in flight");
  __shared__ __attribute__((aligned(16))) unsigned char win[NB][kWinB];
  const int64_t P = d.delta[KFP];
  const int nz = (int)((n + P - 1) / P), ntp = (int)((P + kTile - 1) / kTile);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int cnt;
  const int tj0 = march_tile(xcd, 0, 0, 0, ntp, cnt);   // XCD j owns in-plane tiles [tj0, cnt) (march_tile returns the end in `cnt`)
  cnt -= tj0;
  if (cnt <= 0) return;
  const int zr = slot / cnt, tile = tj0 + slot % cnt;
  if (zr >= nzr) return;
  const int za = (int)((int64_t)zr * nz / nzr), zb = (int)((int64_t)(zr + 1) * nz / nzr);
  const int64_t off0 = (int64_t)tile * kTile;
  const uint32_t lane_b = threadIdx.x * 16u, lane_m = threadIdx.x * 2u;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane_w = (threadIdx.x & 63u) * 16u;
  const uint32_t win0 = (uint32_t)(uintptr_t)&win[0][0];
  const bool plain_st = (shifted & 2) != 0;
  const bool lane_in = off0 + 2 * (int64_t)threadIdx.x + 1 < P;                 // this lane's pair belongs to the plane (P is even)
  const int nst = (off0 + 128 * (int64_t)wave + 1 < P) ? 1 : 0;                  // does this wave store anything at all (uniform)
  auto r0_of = [&](int z) { return (int64_t)z * P + off0; };
  auto win_ok = [&](int z) { const int64_t r0 = r0_of(z); return z >= 0 && r0 - 256 >= 0 && r0 + kTile + 256 <= n; };
  // a step is interior when its window is loadable, its rows all exist, and so is the next plane's window -- or nothing of the tile exists there
  auto interior = [&](int z) { return win_ok(z) && r0_of(z) + kTile <= n && (win_ok(z + 1) || r0_of(z + 1) >= n); };

  auto edge = [&](int z) {   // the clamped path of k_spmv_stencil2 on the rows of this tile in plane z
    const int64_t r = r0_of(z) + 2 * (int64_t)threadIdx.x;
    if (!lane_in || r >= n) return;
    const bool two = r + 1 < n;
    const uint32_t m = mask2[r >> 1];
    const uint32_t m0 = m & 0xffu, m1 = m >> 8;
    const int64_t cmax = n - 2;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int64_t c = r + d.delta[k];
      int64_t lo = c < 0 ? 0 : (c > cmax ? cmax : c);
      if (cmax < 0) lo = 0;
      const int sh = (int)(c - lo);
      double xa, xb;
      if (n >= 2) ld_pair_u(x + lo, xa, xb);
      else { xa = x[0]; xb = x[0]; }
      const double v0 = sh == 1 ? xb : xa, v1 = sh == -1 ? xa : xb;
      const double p0 = mul_nc(d.val[k], v0), p1 = mul_nc(d.val[k], v1);
      s0 = ((m0 >> k) & 1u) ? add_(s0, p0) : s0;
      s1 = ((m1 >> k) & 1u) ? add_(s1, p1) : s1;
    }
    if (shifted) {
      double x0, x1;
      if (two) ld_pair_u(x + r, x0, x1);
      else { x0 = x[r]; x1 = x0; }
      s0 = scl(sub_s(s0, mul_(theta, x0)), sigma);
      s1 = scl(sub_s(s1, mul_(theta, x1)), sigma);
    }
    if (two) {
      if (plain_st) st_pack(y + r, make_double2(s0, s1));
      else st_pack_nt(y + r, make_double2(s0, s1));
    } else {
      if (plain_st) y[r] = s0;
      else st_elem_nt(y + r, s0);
    }
  };
  int zi0 = za, zi1 = zb;
  while (zi0 < zi1 && !interior(zi0)) ++zi0;
  while (zi1 > zi0 && !interior(zi1 - 1)) --zi1;
  if (zi0 > za || zi1 < zb) {
    for (int z = za; z < zb; ++z)
      if (z < zi0 || z >= zi1) edge(z);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (zi0 >= zi1) return;   // (uniform over the workgroup: the barriers below match)

  // window of plane zz into buffer `buf`; a plane whose window leaves the vector is replaced by plane `zsafe` (never read then)
  auto issue_w = [&](int zz, int zsafe, int buf) {
    const int zu = win_ok(zz) ? zz : zsafe;
    // pieces of 128 rows: wave w copies piece w and, while there are that many, piece w + kWaves
    const double* bw = x + (r0_of(zu) - 256 + (int64_t)wave * 128);
    uint32_t ldst = win0 + (uint32_t)buf * (uint32_t)kWinB + (uint32_t)wave * 1024u;
    asm volatile("" : "+s"(bw));
    asm volatile("" : "+s"(ldst));
    asm volatile("s_nop 4" ::: "memory");
    march_glds16(lane_w, bw, ldst);
    if (wave + kWaves < kPieces) march_glds16(lane_w + (uint32_t)kWaves * 1024u, bw, ldst + (uint32_t)kWaves * 1024u);
  };
  auto issue_m = [&](int zz, uint32_t& m) {
    const uint16_t* bm = mask2 + (r0_of(zz) >> 1);
    asm volatile("" : "+s"(bm));
    asm volatile("s_nop 4" ::: "memory");
    march_ld_u16(m, lane_m, bm);
  };
  f64x2m prev;   // the own pair of the plane below (x[r - P], x[r + 1 - P])
  {
    const double* bp = x + (zi0 >= 1 ? r0_of(zi0 - 1) : r0_of(zi0));
    asm volatile("" : "+s"(bp));
    asm volatile("s_nop 4" ::: "memory");
    march_ld16(prev, lane_b, bp);
  }
  uint32_t mA, mB;
  issue_w(zi0, zi0, 0);
  issue_w(zi0 + 1, zi0, 1);
  if constexpr (NB == 4) issue_w(zi0 + 2, zi0, 2);
  issue_m(zi0, mA);
  int bc = 0;   // buffer of plane z (plane z + i: (bc + i) % NB)
  auto step = [&](int z, uint32_t& m, uint32_t& mnext, bool first, bool has_next) {
    // landed: window z + 1 (and everything older) and the masks of this step.  Queue of the wave behind the masks of this step
    // (issued in the step before): [NB == 4: the two window pieces issued right behind them] [the store of the step before]
    constexpr int NW_LATE = NB == 4 ? 2 : 0;
    if (first) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else if (nst == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NW_LATE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NW_LATE + 1) : "memory");
    asm volatile("" : "+v"(m));
    if (first) asm volatile("" : "+v"(prev));
    const int bn = bc + 1 == NB ? 0 : bc + 1, b2 = bc == 0 ? NB - 1 : bc - 1;   // plane z + 1; the slot of plane z - 1 = of plane z + NB - 1
    if constexpr (NB == 3) {
      issue_w(z + 2, z, b2);               // (overwrites the window of plane z - 1: everybody is past step z - 1)
      if (has_next) issue_m(z + 1, mnext);
    } else {
      if (has_next) issue_m(z + 1, mnext); // masks FIRST: the wait for them must not drag the window behind them along
      issue_w(z + 3, z, b2);
    }
    const unsigned char* w = &win[0][0] + bc * kWinB + (256 * 8) + threadIdx.x * 16;
    f64x2m v[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      if ((NEARM >> k) & 1u) {
        const int off = (int)d.delta[k] * 8;
        if (((ODDM >> k) & 1u) == 0) {
          v[k] = *reinterpret_cast<const f64x2m*>(w + off);
        } else {
          typedef double f64x2u8 __attribute__((ext_vector_type(2), aligned(8)));
          const f64x2u8 p = *reinterpret_cast<const f64x2u8*>(w + off);
          v[k].x = p.x;
          v[k].y = p.y;
        }
      }
    }
    v[KFM] = prev;
    v[KFP] = *reinterpret_cast<const f64x2m*>(&win[0][0] + bn * kWinB + (256 * 8) + threadIdx.x * 16);
    const int m0 = (int)(m & 0xffu), m1 = (int)(m >> 8);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const double p0 = mul_nc(d.val[k], v[k].x), p1 = mul_nc(d.val[k], v[k].y);
      s0 = add_(s0, and_mask(p0, __builtin_amdgcn_sbfe(m0, k, 1)));
      s1 = add_(s1, and_mask(p1, __builtin_amdgcn_sbfe(m1, k, 1)));
    }
    if (shifted) {
      s0 = scl(sub_s(s0, mul_(theta, v[KOWN].x)), sigma);
      s1 = scl(sub_s(s1, mul_(theta, v[KOWN].y)), sigma);
    }
    prev = v[KOWN];
    f64x2m o2;
    o2.x = s0;
    o2.y = s1;
    double* by = y + r0_of(z);
    asm volatile("" : "+s"(by));
    if (lane_in) march_st16(lane_b, o2, by, plain_st);   // (a wave with no lane in the plane issues nothing: nst == 0 above)
    bc = bn;
  };
  int z = zi0;
  bool first = true;
  for (;;) {
    const bool h1 = z + 1 < zi1;
    step(z, mA, mB, first, h1);
    first = false;
    if (!h1) break;
    ++z;
    const bool h2 = z + 1 < zi1;
    step(z, mB, mA, false, h2);
    if (!h2) break;
    ++z;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace ksd
