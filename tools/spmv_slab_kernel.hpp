// EXPERIMENT (round 6, measured and NOT adopted: profiles/r06_spmv_slab.txt -- 45-54 us against 43-45 us of k_spmv_stencil2 and 34 us
// of the marching kernel that went into the library, csrc/ks_spmv_march.hpp).  Correct (bit-identical on the first run) but bound by
// its own instruction stream and by the barrier that couples the waves: with 8-16 waves per CU nothing overlaps the ~200 vector
// instructions per 128 rows, and deeper rings (fewer waves) are slower, not faster.  Kept as the record of the attempt.
//
// SpMV, stencil-mask layout, SLAB form (gfx950):  y = A x  or the Newton step  y = sigma (A x - theta x)  of mul!(y, A, x),
// src/expansion.jl:121, for a matrix in the one-bit-per-slot-and-row layout of k_spmv_stencil (ks_kernels.hpp) whose slots are
//     delta_k = dv_k P + dp_k,   dv_k in {-1, 0, +1},  |dp_k| <= 256,
// with ONE far stride P (structured grids: P = one xy-plane; the 3-D 7-point Laplacian has dv = -1, 0, 0, 0, 0, 0, +1 and
// dp = 0, -nx, -1, 0, 1, nx, 0; 19- and 27-point stencils fit as well).
//
// Why.  k_spmv_stencil2 is bound by latency x occupancy, not by traffic (profiles/r04_spmv_counters.txt: the waves wait 95 % of
// their life, HBM 1.03x the algorithmic bytes, L2 at 40 % of its rate): a lane fetches seven pairs for one pair of NEW values,
// a wave's life is one load round trip plus one store round trip, and what 2048 threads per CU hold in flight is mostly
// redundant.  Here the bytes in flight are set by a ring depth instead, and every x element enters the CU once:
//   * a workgroup of NW waves owns NW consecutive planes z0 .. z0 + NW - 1 and one SEGMENT [ta, tb) of in-plane offsets; wave v
//     streams rows (z0 + v) P + [ta, tb) in blocks of 256 rows through its own RING of NS 2-KiB slots in LDS, filled by
//     asynchronous global -> LDS copies (global_load_lds_dwordx4, no registers) that run NS - 3 blocks ahead;
//   * the near taps (dv = 0) are read from the wave's own ring (blocks b - 1, b, b + 1 are resident at step b), the far taps
//     (dv = -1 / +1) from the ring of the wave below / above -- the same offsets one plane away, which that wave needs anyway;
//     the planes z0 - 1 and z0 + NW come through two boundary rings (filled by the first / last wave);
//   * one workgroup barrier per block keeps the waves in step (copies in flight are not affected by it); per block and wave:
//     2 copies (+2 on the boundary waves), 1 mask copy, ~14 LDS reads and 2 streaming stores per lane.
// HBM traffic: x once + the boundary planes ((NW + 2) / NW when they miss the L2; neighbouring slabs of one segment are given to
// the same XCD and march at the same pace, so mostly they hit) + 512 halo rows per segment; y once; 1 mask byte per row.
// Every vector-memory operation of the loop is issued from inline assembly and counted by hand (s_waitcnt vmcnt(N)): hipcc
// sees no loads and inserts no waits of its own.
// Products are rounded separately and added in slot order under the row's mask: y is bit-identical to every other layout.
#pragma once

#include <utility>

#include "../arnoldimethod.jl_amd/csrc/ks_block_kernels.hpp"  // glds16

namespace ksd {

typedef double f64x2a __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gst16(double* p, f64x2a v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void gst16_nt(double* p, f64x2a v) { asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

// s_waitcnt vmcnt(n) lgkmcnt(0) with a run-time wave-uniform n <= 63, then the workgroup barrier
__device__ __forceinline__ void slab_wait_barrier(int n) {
#define KS_SW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
#define KS_SW8(B) KS_SW(B) KS_SW(B + 1) KS_SW(B + 2) KS_SW(B + 3) KS_SW(B + 4) KS_SW(B + 5) KS_SW(B + 6) KS_SW(B + 7)
  switch (n) {
    KS_SW(0) KS_SW(1) KS_SW(2) KS_SW(3) KS_SW(4) KS_SW(5) KS_SW(6) KS_SW(7) KS_SW(8) KS_SW(9) KS_SW(10) KS_SW(11) KS_SW(12) KS_SW(13) KS_SW(14)
    KS_SW(15) KS_SW(16) KS_SW(17) KS_SW(18) KS_SW(19) KS_SW(20) KS_SW(21) KS_SW(22) KS_SW(23) KS_SW(24) KS_SW(25) KS_SW(26) KS_SW(27) KS_SW(28)
    KS_SW(29) KS_SW(30) KS_SW(31) KS_SW(32) KS_SW(33) KS_SW(34) KS_SW(35) KS_SW(36) KS_SW(37) KS_SW(38) KS_SW(39) KS_SW(40) KS_SW(41) KS_SW(42)
    KS_SW(43) KS_SW(44) KS_SW(45) KS_SW(46) KS_SW(47) KS_SW(48) KS_SW(49) KS_SW(50) KS_SW(51) KS_SW(52) KS_SW(53) KS_SW(54) KS_SW(55) KS_SW(56)
    KS_SW(57) KS_SW(58) KS_SW(59) KS_SW(60) KS_SW(61) KS_SW(62) KS_SW(63)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef KS_SW8
#undef KS_SW
}

template <int... K, class F> __device__ __forceinline__ void slab_static_for(std::integer_sequence<int, K...>, F&& f) {
  (f(std::integral_constant<int, K>{}), ...);
}

template <int NW, int NS> struct SlabGeom {
  static_assert(NS >= 4, "a step reads three blocks of a ring; at least one more must be in flight");
  static constexpr int kBlk = 256;                 // rows of a block = two 128-row pieces = two 1-KiB copies
  static constexpr int kHalo = 256;                // |dp| the near window covers
  static constexpr int kRingEl = NS * kBlk;        // elements of a ring
  static constexpr int kRingBytes = kRingEl * 8;
  static constexpr int kMaskBytes = NS * 256;      // per wave: one 256-byte slot per block (128 two-row masks of 16 bits)
  static constexpr int kDepth = NS - 3;            // blocks in flight beyond the three a step reads
  static constexpr size_t lds_bytes = (size_t)(NW + 2) * kRingBytes + (size_t)NW * kMaskBytes;
  static_assert(lds_bytes <= 160 * 1024, "LDS of one CU");
};

// the host's check: which far stride the dictionary has, and whether the slab form takes it (0: it does not)
inline int64_t slab_far_stride(const int32_t* delta, int nslots, int64_t n) {
  if (nslots < 1 || nslots > 8 || (n & 1)) return 0;
  int64_t P = 0;
  for (int k = 0; k < nslots; ++k) {
    const int64_t a = delta[k] < 0 ? -(int64_t)delta[k] : delta[k];
    if (a > P) P = a;
  }
  if (P <= 512 || (P & 3)) return 0;  // (no far taps: the one-launch-per-tile kernels are the right tool; P % 4: mask copies are 4-byte aligned)
  for (int k = 0; k < nslots; ++k) {
    const int64_t dl = delta[k];
    const int64_t dv = dl > P / 2 ? 1 : (dl < -(P / 2) ? -1 : 0);
    const int64_t dp = dl - dv * P;
    if (dp < -256 || dp > 256) return 0;
  }
  return P;
}
// bit k set: dp of slot k is odd
inline unsigned slab_odd_mask(const int32_t* delta, int nslots, int64_t P) {
  unsigned m = 0;
  for (int k = 0; k < nslots; ++k) {
    const int64_t dl = delta[k];
    const int64_t dv = dl > P / 2 ? 1 : (dl < -(P / 2) ? -1 : 0);
    if ((dl - dv * P) & 1) m |= 1u << k;
  }
  return m;
}

// grid: nslab * nseg workgroups of NW waves; dynamic LDS SlabGeom<NW, NS>::lds_bytes.
//   mask2 / nmask: 16-bit masks of row pairs (low byte: row 2 i) and how many of them may be read
//   nxr: elements of x that may be read (even, >= n);  P: far stride;  nz = ceil(n / P) planes;  seg_len % 4 == 0
//   order 0: consecutive workgroups are the slabs of one segment (the z-neighbours share an XCD), 1: the segments of one slab
//   ODD: bit k set = slot k is read as two 8-byte elements (any dp); clear = as one aligned pair (dp even, the host checks)
template <int NW, int NS, unsigned ODD>
__global__ void __launch_bounds__(NW * 64)
    k_spmv_stencil_slab(const uint16_t* __restrict__ mask2, int64_t nmask, const StencilDict<double> d, int nslots,
                        const double* __restrict__ x, int64_t nxr, double* __restrict__ y, int64_t n, int64_t P, int nz, int seg_len,
                        int nseg, int nslab, int order, const DevState* __restrict__ st, int shifted, double theta, double sigma) {
  if (st && st->breakdown >= 0) return;
  using G = SlabGeom<NW, NS>;
  constexpr int D = G::kDepth;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  int seg, slab;
  const int dbg = order >> 1;   // (probes of tools/spmv_slab.hip: 1 no arithmetic, 2 no stores, 4 no x copies, 8 no mask copies)
  order &= 1;
  if (order == 0) { seg = w / nslab; slab = w - seg * nslab; }
  else { slab = w / nseg; seg = w - slab * nseg; }
  const int z = slab * NW + v;
  const int64_t ta = (int64_t)seg * seg_len;
  const int64_t tb = ta + seg_len < P ? ta + seg_len : P;
  const int nb_wg = (int)((tb - ta + 255) >> 8);            // steps of the workgroup (every wave takes part in every barrier)
  const int64_t row0 = (int64_t)z * P + ta;                 // first row of this wave
  int64_t len64 = z < nz ? tb - ta : 0;
  if (row0 + len64 > n) len64 = n - row0;
  const int len = len64 > 0 ? (int)len64 : 0;               // rows of this wave
  const int nb_v = (len + 255) >> 8;
  const bool active = nb_v > 0;
  const bool lo = active && v == 0 && z >= 1;               // this wave also fills the ring of plane z - 1 / z + 1
  const bool hi = active && v == NW - 1 && z + 1 < nz;
  const int nld = 1 + (lo ? 1 : 0) + (hi ? 1 : 0);

  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  const uint32_t ring_own = lds0 + (uint32_t)(v + 1) * G::kRingBytes;
  const uint32_t ring_lo = lds0, ring_hi = lds0 + (uint32_t)(NW + 1) * G::kRingBytes;
  const uint32_t mring = lds0 + (uint32_t)(NW + 2) * G::kRingBytes + (uint32_t)v * G::kMaskBytes;

  // block j (-1 .. nb_wg) of the stream that starts at row `base`, into slot `sl` of ring `ring`; beyond nb_wg: dummy copies
  // (the counts of the waits stay the same to the end)
  auto copy_x = [&](uint32_t ring, int64_t base, int j, int sl) {
    const bool live = j <= nb_wg;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t r = base + (int64_t)j * 256 + p * 128 + 2 * lane;
      const bool ok = live && r >= 0 && r + 2 <= nxr && !(dbg & 4);
      glds16(ok ? x + r : x, ring + (uint32_t)sl * 2048u + (uint32_t)p * 1024u);
    }
  };
  auto issue_x = [&](int j, int sl) {
    copy_x(ring_own, row0, j, sl);
    if (lo) copy_x(ring_lo, row0 - P, j, sl);
    if (hi) copy_x(ring_hi, row0 + P, j, sl);
  };
  auto issue_m = [&](int j, int sl) {
    const int64_t i = ((row0 + (int64_t)j * 256) >> 1) + 2 * lane;   // two 16-bit masks per lane
    const bool ok = j < nb_wg && i + 2 <= nmask && !(dbg & 8);
    glds4(ok ? mask2 + i : mask2, mring + (uint32_t)sl * 256u);
  };

  int sx = 0, sm = 0;   // slots of the next x block / mask block to issue: (j + 1) % NS, j % NS
  if (active) {
    issue_x(-1, 0);
    sx = 1;
    for (int j = 0; j <= D; ++j) {
      issue_x(j, sx);
      sx = sx + 1 == NS ? 0 : sx + 1;
      issue_m(j, sm);
      sm = sm + 1 == NS ? 0 : sm + 1;
    }
  }
  const int nstore = (dbg & 2) ? 0 : 2;
  const int ops_p = 2 * nld + 1, ops_s = 2 * nld + 1 + nstore;   // vector-memory operations of a prologue step / of a loop step
  int sb = 1, smb = 0;                                   // slots of block b
  const bool plain_st = (shifted & 2) != 0;
  for (int b = 0; b < nb_wg; ++b) {
    // block b + 1 of the x streams and mask block b have landed when at most this many LATER operations are outstanding
    // (operations of one wave complete in order): the rest of the step that issued block b + 1, then D - 1 whole steps
    int nwait = 0;
    if (b < nb_v) nwait = b < D ? 1 + (D - b - 1) * ops_p + b * ops_s : 1 + nstore + (D - 1) * ops_s;
    slab_wait_barrier(nwait);
    if (active) {
      issue_x(b + 1 + D, sx);   // into the slot of block b - 2: everybody is past step b - 1
      sx = sx + 1 == NS ? 0 : sx + 1;
      issue_m(b + 1 + D, sm);
      sm = sm + 1 == NS ? 0 : sm + 1;
    }
    if (b < nb_v) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int t0 = b * 256 + p * 128;
        if (t0 >= len) break;   // (only in the last step)
        const int t = t0 + 2 * lane;
        const int pos0 = sb * 256 + p * 128 + 2 * lane;   // this lane's pair in ring coordinates
        const uint32_t mm = *reinterpret_cast<const uint16_t*>(lds + (NW + 2) * G::kRingBytes + v * G::kMaskBytes + smb * 256 + p * 128 + 2 * lane);
        const uint32_t m0 = mm & 0xffu, m1 = mm >> 8;
        // phase 1: every LDS read of the piece, no branch in between (slots beyond nslots have delta 0 and no mask bit: they read
        // the lane's own pair)
        double v0[8], v1[8];
        slab_static_for(std::make_integer_sequence<int, 8>{}, [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const int64_t dl = d.delta[k];
          const int dv = dl > P / 2 ? 1 : (dl < -(P / 2) ? -1 : 0);
          const int dp = (int)(dl - dv * P);
          const unsigned char* ring = lds + (v + 1 + dv) * G::kRingBytes;
          int q = pos0 + dp;
          q = q < 0 ? q + G::kRingEl : (q >= G::kRingEl ? q - G::kRingEl : q);
          if constexpr (((ODD >> k) & 1u) == 0) {   // dp even (the host's promise): one aligned 16-byte read
            const f64x2a pr = *reinterpret_cast<const f64x2a*>(ring + q * 8);
            v0[k] = pr.x;
            v1[k] = pr.y;
          } else {
            const int q1 = q + 1 == G::kRingEl ? 0 : q + 1;
            v0[k] = *reinterpret_cast<const double*>(ring + q * 8);
            v1[k] = *reinterpret_cast<const double*>(ring + q1 * 8);
          }
        });
        f64x2a own;
        own.x = 0.0;
        own.y = 0.0;
        if (shifted) own = *reinterpret_cast<const f64x2a*>(lds + (v + 1) * G::kRingBytes + pos0 * 8);
        // phase 2: products rounded separately, added in slot order under the row's mask (as k_spmv_stencil2)
        double s0 = 0.0, s1 = 0.0;
        if (dbg & 1) { s0 = v0[3]; s1 = v1[3]; }
        else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double p0 = mul_nc(d.val[k], v0[k]), p1 = mul_nc(d.val[k], v1[k]);
          s0 = ((m0 >> k) & 1u) ? add_(s0, p0) : s0;
          s1 = ((m1 >> k) & 1u) ? add_(s1, p1) : s1;
        }
        if (shifted) {
          s0 = scl(sub_s(s0, mul_(theta, own.x)), sigma);
          s1 = scl(sub_s(s1, mul_(theta, own.y)), sigma);
        }
        double* yp = y + row0 + t;
        if ((dbg & 2) && s0 != 1.2345e-300) {
        } else if (t + 1 < len) {
          f64x2a o;
          o.x = s0;
          o.y = s1;
          if (plain_st) gst16(yp, o);
          else gst16_nt(yp, o);
        } else if (t < len) {
          *yp = s0;
        }
      }
    }
    sb = sb + 1 == NS ? 0 : sb + 1;
    smb = smb + 1 == NS ? 0 : smb + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace ksd
