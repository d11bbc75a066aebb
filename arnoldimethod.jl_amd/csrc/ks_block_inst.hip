// Instantiations + launch dispatch of the s-step streaming kernels (see ks_block_launch.hpp).  Compiled in PARTS
// (-DKS_BLK_PART=0: Float64 block sizes 1-4, 1: Float64 5 / 8 / 10, 2: ComplexF64) so that build.py can run them in parallel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "ks_block_kernels.hpp"
#include "ks_block_mfma.hpp"
#include "ks_block_launch.hpp"

#ifndef KS_BLK_PART
#error "compile with -DKS_BLK_PART=0|1|2"
#endif

namespace {
using ksd::cd;
using ksd::kBlock;

inline void hipcheck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
// streaming kernels partition the rows into one contiguous range per workgroup: all workgroups must be co-resident
template <class K> int resident(const BlkLaunchArgs& a, K kernel, int& cache, int threads = kBlock) {
  if (cache < 0) {
    int occ = 0;
    hipcheck(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, 0), "occupancy query");
    cache = std::max(1, std::min(occ, a.bpc));
  }
  return a.num_cu * cache;
}
inline int cap(const BlkLaunchArgs& a, int nb, int packs_per_iter, size_t esz) {
  const int64_t npacks = a.ld * (int64_t)esz / 16;
  const int64_t want = std::max<int64_t>(1, npacks / (2 * (int64_t)packs_per_iter));
  return (int)std::min<int64_t>(nb, want);
}

// ring form of pass 1 (Float64 wide instantiations; KS_BLK_RING=0: the register form)
inline int ring_env() { static const int v = [] { const char* e = std::getenv("KS_BLK_RING"); return e ? std::atoi(e) : 3; }(); return v; }
template <int NCW, int S, int NW, int WB> int go_ring0(const BlkLaunchArgs& a) {
  const int ncol = a.k + S;
  int stages = std::min(4, (150 * 1024) / (ncol * 1024));
  static const int st_env = [] { const char* e = std::getenv("KS_BLK_RING_STAGES"); return e ? std::atoi(e) : 0; }();
  if (st_env >= 2) stages = std::min(st_env, (158 * 1024) / (ncol * 1024));
  if (stages < 2) throw std::runtime_error("block kernels: ring does not fit the LDS");
  const size_t smem = (size_t)stages * ncol * 1024;
  auto kern = ksd::k_bdots_ring<NCW, S, NW, WB>;
  static bool attr = false;
  if (!attr) {
    hipcheck(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "dynamic LDS size");
    attr = true;
  }
  const int nb = cap(a, a.num_cu, 64, sizeof(double));   // one workgroup per CU
  kern<<<nb, 64 * NW, smem, a.stream>>>(static_cast<const double*>(a.V), a.ld, a.k, stages, static_cast<double*>(a.partial), a.pnb,
                                        static_cast<const ksd::DevState*>(a.st), a.dbg | (a.nt ? 64 : 0), static_cast<const double*>(a.zeros));
  return nb;
}

template <int NCW, int S, int NW, int WB> int go_ring1(const BlkLaunchArgs& a) {
  constexpr int WA = NW / WB, SP = ((S + WB - 1) / WB) * WB;
  const int ncol = a.k + S;
  const size_t fixed = (size_t)(WA * S + S) * 1024 + (size_t)(WA * NCW * SP + S * SP) * 8;
  int stages = (int)std::min<size_t>(3, (158 * 1024 - fixed) / ((size_t)ncol * 1024));
  static const int st_env = [] { const char* e = std::getenv("KS_BLK_RING_STAGES"); return e ? std::atoi(e) : 0; }();
  if (st_env >= 2) stages = (int)std::min<size_t>(st_env, (158 * 1024 - fixed) / ((size_t)ncol * 1024));
  if (stages < 2) throw std::runtime_error("block kernels: ring does not fit the LDS");
  const size_t smem = (size_t)stages * ncol * 1024 + fixed;
  auto kern = ksd::k_bupdate_ring<NCW, S, NW, WB>;
  static bool attr = false;
  if (!attr) {
    hipcheck(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "dynamic LDS size");
    attr = true;
  }
  const int nb = cap(a, a.num_cu, 64, sizeof(double));
  kern<<<nb, 64 * NW, smem, a.stream>>>(static_cast<double*>(a.V), a.ld, a.k, stages, static_cast<const double*>(a.coefp), a.k,
                                        static_cast<const double*>(a.r1inv), static_cast<double*>(a.partial), a.pnb,
                                        static_cast<const ksd::DevState*>(a.st), a.dbg | (a.nt ? 64 : 0), static_cast<const double*>(a.zeros));
  return nb;
}

// large blocks (s = 20): ring forms of their own (k_bdots_ringL: 4 x 2 waves, k_bupdate_ringL: 2 x 4)
template <int NCW4> int go_ringL0(const BlkLaunchArgs& a) {
  const int ncol = a.k + ksd::kBlkL;
  const int stages = std::min(3, (150 * 1024) / (ncol * 1024));
  if (stages < 2) throw std::runtime_error("block kernels: ring does not fit the LDS");
  const size_t smem = (size_t)stages * ncol * 1024;
  auto kern = ksd::k_bdots_ringL<NCW4>;
  static bool attr = false;
  if (!attr) {
    hipcheck(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "dynamic LDS size");
    attr = true;
  }
  const int nb = cap(a, a.num_cu, 64, sizeof(double));
  kern<<<nb, 512, smem, a.stream>>>(static_cast<const double*>(a.V), a.ld, a.k, stages, static_cast<double*>(a.partial), a.pnb,
                                   static_cast<const ksd::DevState*>(a.st), a.dbg | (a.nt ? 64 : 0), static_cast<const double*>(a.zeros));
  return nb;
}
template <int NCW2> int go_ringL1(const BlkLaunchArgs& a) {
  constexpr int S = ksd::kBlkL;
  const int ncol = a.k + S;
  const size_t fixed = (size_t)(2 * S + S) * 1024 + (size_t)(2 * NCW2 * S + S * S) * 8;
  const int stages = 2;
  const size_t smem = (size_t)stages * ncol * 1024 + fixed;
  if (smem > 160 * 1024) throw std::runtime_error("block kernels: ring does not fit the LDS");
  auto kern = ksd::k_bupdate_ringL<NCW2>;
  static bool attr = false;
  if (!attr) {
    hipcheck(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "dynamic LDS size");
    attr = true;
  }
  const int nb = cap(a, a.num_cu, 64, sizeof(double));
  kern<<<nb, 512, smem, a.stream>>>(static_cast<double*>(a.V), a.ld, a.k, stages, static_cast<const double*>(a.coefp), a.k,
                                   static_cast<const double*>(a.r1inv), static_cast<double*>(a.partial), a.pnb,
                                   static_cast<const ksd::DevState*>(a.st), a.dbg | (a.nt ? 64 : 0), static_cast<const double*>(a.zeros));
  return nb;
}


// matrix-instruction forms (ks_block_mfma.hpp): row slabs per wave, no barrier in the loop.  KS_BLK_MFMA=0: the ring forms.
inline int mfma_env() { static const int v = [] { const char* e = std::getenv("KS_BLK_MFMA"); return e ? std::atoi(e) : 255; }(); return v; }
inline int mfma_ring_env() { static const int v = [] { const char* e = std::getenv("KS_BLK_MFMA_RING"); return e ? std::atoi(e) : 3; }(); return v; }
template <int NGS, int NT, bool CX = false> int go_mfma(int which, const BlkLaunchArgs& a) {
  using C = ksd::BlkMfma<NGS, NT>;
  int ring = mfma_ring_env();
  while (ring > 2 && C::lds_bytes(ring, CX) > 160 * 1024) --ring;
  const size_t smem = C::lds_bytes(ring, CX);
  if (ring < 2 || smem > 160 * 1024) throw std::runtime_error("block kernels: slab ring does not fit the LDS");
  constexpr int RV = CX ? 2 : 1;                       // ComplexF64: the real view of the basis (2 n rows, leading dimension 2 ld)
  const int nb = cap(a, a.num_cu, 64, CX ? 16 : 8);
  const int dbg = a.dbg | (a.nt ? 64 : 0);
  double* Vr = static_cast<double*>(a.V);
  const int64_t ldr = a.ld * RV;
  if (which == 0) {
    auto kern = ksd::k_bdots_mfma<NGS, NT, CX>;
    static bool attr = false;
    if (!attr) {
      hipcheck(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "dynamic LDS size");
      attr = true;
    }
    const double* zb = a.zsrc ? static_cast<const double*>(a.zsrc) : Vr + (int64_t)a.k * ldr;
    kern<<<nb, 512, smem, a.stream>>>(Vr, ldr, a.k, zb, a.zsrc ? a.ldz * RV : ldr, a.s, ring, static_cast<double*>(a.partial), a.pnb,
                                     static_cast<const ksd::DevState*>(a.st), dbg, static_cast<const double*>(a.zeros));
  } else {
    auto kern = ksd::k_bupdate_mfma<NGS, NT, CX>;
    static bool attr = false;
    if (!attr) {
      hipcheck(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "dynamic LDS size");
      attr = true;
    }
    const double* zb = a.zsrc ? static_cast<const double*>(a.zsrc) : Vr + (int64_t)a.k * ldr;
    kern<<<nb, 512, smem, a.stream>>>(Vr, ldr, a.k, zb, a.zsrc ? a.ldz * RV : ldr, a.s, ring, static_cast<const double*>(a.coefp), a.k,
                                     static_cast<const double*>(a.r1inv), static_cast<double*>(a.partial), a.pnb,
                                     static_cast<const ksd::DevState*>(a.st), dbg, static_cast<const double*>(a.zeros));
  }
  return nb;
}
template <int NT, int MAXG, int G = 1, bool CX = false> int go_mfma_by_k(int which, const BlkLaunchArgs& a) {
  if constexpr (G > MAXG) {
    throw std::runtime_error("block kernels: no matrix-instruction form for this many columns");
  } else {
    if ((a.k + 3) / 4 == G) return go_mfma<G, NT, CX>(which, a);
    return go_mfma_by_k<NT, MAXG, G + 1, CX>(which, a);
  }
}

#if KS_BLK_PART >= 1
// restart rotation fused with the first pass (k_brotdots_mfma): the instantiated (column groups in, column tiles out, block tiles)
// -- Float64 in part 1, ComplexF64 (real view of the basis) in part 2
template <int NGX, int NTK, int NT, bool CX> int go_rot(const BlkLaunchArgs& a) {
  using C = ksd::BlkRot<NGX, NTK, NT>;
  int ring = mfma_ring_env();
  while (ring > 2 && C::lds_bytes(ring, CX) > 160 * 1024) --ring;
  const size_t smem = C::lds_bytes(ring, CX);
  if (ring < 2 || smem > 160 * 1024) throw std::runtime_error("block kernels: slab ring of the fused rotation does not fit the LDS");
  constexpr int RV = CX ? 2 : 1;
  const int nb = cap(a, a.num_cu, 64, CX ? 16 : 8);
  auto kern = ksd::k_brotdots_mfma<NGX, NTK, NT, CX>;
  static bool attr = false;
  if (!attr) {
    hipcheck(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "dynamic LDS size");
    attr = true;
  }
  kern<<<nb, 512, smem, a.stream>>>(static_cast<double*>(a.V), a.ld * RV, a.cin, static_cast<const double*>(a.rotm), a.out0, a.k,
                                   static_cast<const double*>(a.zsrc), a.ldz * RV, a.s, ring, static_cast<double*>(a.partial), a.pnb,
                                   a.dbg | (a.nt ? 64 : 0), static_cast<const double*>(a.zeros));
  return nb;
}
#if KS_BLK_PART == 1
#define KS_ROT_SHAPES(X) X(11, 6, 5) X(11, 7, 4) X(11, 7, 3) X(11, 8, 3) X(11, 8, 2) X(6, 3, 3) X(6, 4, 3) X(6, 4, 2)
constexpr bool kRotCX = false;
#else
#define KS_ROT_SHAPES(X) X(6, 3, 3) X(6, 4, 3) X(6, 4, 2) X(11, 6, 3)
constexpr bool kRotCX = true;
#endif
// (the block's tile count is rounded up to an instantiated one: the kernel takes s at run time, missing columns are zeros)
inline bool rot_shape(int cin, int k, int s, int& ngx, int& ntk, int& nt) {
  ngx = (cin + 3) / 4; ntk = (k + 3) / 4;
  const int need = (s + 3) / 4;
  nt = 0;
#define KS_ROT_HAS(A, B, C_) if (ngx == A && ntk == B && need <= C_ && (nt == 0 || C_ < nt)) nt = C_;
  KS_ROT_SHAPES(KS_ROT_HAS)
#undef KS_ROT_HAS
  return nt != 0;
}
inline int go_rot_by_shape(const BlkLaunchArgs& a) {
  int ngx, ntk, nt;
  if (!a.zsrc || !a.rotm || !rot_shape(a.cin, a.k, a.s, ngx, ntk, nt)) throw std::runtime_error("block kernels: no fused rotation for this shape");
#define KS_ROT_GO(A, B, C_) if (ngx == A && ntk == B && nt == C_) return go_rot<A, B, C_, kRotCX>(a);
  KS_ROT_SHAPES(KS_ROT_GO)
#undef KS_ROT_GO
  throw std::runtime_error("block kernels: no fused rotation for this shape");
}
#endif

template <class D, int NCW, int S, int NW = 4, int WB = 1> int go(int which, const BlkLaunchArgs& a) {
  if constexpr (sizeof(D) == 8 && S == 20) {
    if ((mfma_env() >> which) & 1) return go_mfma_by_k<5, 6>(which, a);
  }
  if constexpr (sizeof(D) == 8 && S == 10) {
    if ((mfma_env() >> (which + 2)) & 1) return go_mfma_by_k<3, 8>(which, a);
  }
  if constexpr (sizeof(D) == 8 && S == 8) {
    if ((mfma_env() >> (which + 4)) & 1) return go_mfma_by_k<2, 12>(which, a);
  }
  if constexpr (sizeof(D) == 8 && NW == 8 && S == 20) {
    if (which == 0 && (ring_env() & 1)) return go_ringL0<NCW>(a);
    if (which == 1 && (ring_env() & 2)) {
      if (a.k <= 20) return go_ringL1<10>(a);
      if (a.k <= 22) return go_ringL1<11>(a);
      return go_ringL1<12>(a);
    }
  }
  if constexpr (sizeof(D) == 8 && NW == 8 && S <= 10) {
    if (which == 0 && (ring_env() & 1)) return go_ring0<NCW, S, NW, WB>(a);
    if (which == 1 && (ring_env() & 2)) return go_ring1<NCW, S, NW, WB>(a);
  }
  constexpr int U = ksd::blk_u<D, NCW, S, NW>();
  constexpr int NT_ = 64 * NW;  // threads per workgroup
  const ksd::DevState* st = static_cast<const ksd::DevState*>(a.st);
  D* V = static_cast<D*>(a.V);
  D* part = static_cast<D*>(a.partial);
  if (which == 0) {
    static int cache = -1;
    const int nb = cap(a, resident(a, ksd::k_bdots<D, NCW, S, U, true, NW, WB>, cache, NT_), 64 * U, sizeof(D));
    if (a.nt) ksd::k_bdots<D, NCW, S, U, true, NW, WB><<<nb, NT_, 0, a.stream>>>(V, a.ld, a.k, part, a.pnb, st);
    else ksd::k_bdots<D, NCW, S, U, false, NW, WB><<<nb, NT_, 0, a.stream>>>(V, a.ld, a.k, part, a.pnb, st);
    return nb;
  }
  static int cache = -1;
  const int nb = cap(a, resident(a, ksd::k_bupdate<D, NCW, S, U, true, NW, WB>, cache, NT_), 64 * U, sizeof(D));
  const D* cp = static_cast<const D*>(a.coefp);
  const D* ri = static_cast<const D*>(a.r1inv);
  if (a.nt) ksd::k_bupdate<D, NCW, S, U, true, NW, WB><<<nb, NT_, 0, a.stream>>>(V, a.ld, a.k, cp, a.k, ri, part, a.pnb, st, a.dbg);
  else ksd::k_bupdate<D, NCW, S, U, false, NW, WB><<<nb, NT_, 0, a.stream>>>(V, a.ld, a.k, cp, a.k, ri, part, a.pnb, st, a.dbg);
  return nb;
}

// columns per wave: ceil(k / 4), rounded up to an instantiated width
template <class D, int S> int by_ncw(int which, const BlkLaunchArgs& a) {
  const int need = (a.k + 3) / 4;
  if constexpr (sizeof(D) == 8 && S == 20) {
    // (register forms as the fall-back of the large-block ring kernels: correct, register bound)
    if (need <= 6) return go<D, 6, S, 8, 2>(which, a);
  } else if constexpr (sizeof(D) == 8 && S == 10) {
    // wide form: eight waves, the block columns split two ways (ks_block_kernels.hpp)
    if (need <= 4) return go<D, 4, S, 8, 2>(which, a);
    if (need <= 5) return go<D, 5, S, 8, 2>(which, a);
    if (need <= 6) return go<D, 6, S, 8, 2>(which, a);
    if (need <= 7) return go<D, 7, S, 8, 2>(which, a);
    if (need <= 8) return go<D, 8, S, 8, 2>(which, a);
  } else if constexpr (sizeof(D) == 8) {
    if (need <= 4) return go<D, 4, S>(which, a);
    if (need <= 5) return go<D, 5, S>(which, a);
    if (need <= 6) return go<D, 6, S>(which, a);
    if (need <= 7) return go<D, 7, S>(which, a);
    if (need <= 8) return go<D, 8, S>(which, a);
    if (need <= 9) return go<D, 9, S>(which, a);
    if (need <= 10) return go<D, 10, S>(which, a);
    if (need <= 12) return go<D, 12, S>(which, a);
    if constexpr (S <= 5) return go<D, 16, S>(which, a);
  } else {
    if (need <= 4) return go<D, 4, S>(which, a);
    if (need <= 6) return go<D, 6, S>(which, a);
    if (need <= 8) return go<D, 8, S>(which, a);
  }
  throw std::runtime_error("block kernels: no instantiation for this many columns");
}
}  // namespace

#if KS_BLK_PART == 0
int ks_blk_launch_part0(int which, const BlkLaunchArgs& a) {
  switch (a.s) {
    case 1: return by_ncw<double, 1>(which, a);
    case 2: return by_ncw<double, 2>(which, a);
    case 3: return by_ncw<double, 3>(which, a);
    case 4: return by_ncw<double, 4>(which, a);
    default: throw std::runtime_error("block kernels: block size not in this part");
  }
}
#elif KS_BLK_PART == 1
int ks_blk_mfma_nt_f64(int k, int s) {
  const int m = mfma_env();
  if (s < 1 || k < 1) return 0;
  // (the class follows from s alone -- the same one the instantiated sizes 8 / 10 / 20 dispatch to)
  if (s <= 8) return (k <= 64 && ((m >> 4) & 3) == 3) ? 2 : 0;
  if (s <= 12) return (k <= 48 && ((m >> 2) & 3) == 3) ? 3 : 0;
  if (s <= 16 && k <= 28 && k > 24) return (m & 3) == 3 ? 4 : 0;   // (up to 24 columns the 5-tile kernels take these sizes too)
  return (s <= 20 && k <= 24 && (m & 3) == 3) ? 5 : 0;
}
// second pass of a block on the matrix instruction (only that form reads the block from scratch columns)?
static bool pass2_mfma(int k, int s) {
  const int bit2 = s == 20 ? 1 : (s == 10 ? 3 : (s == 8 ? 5 : -1));
  if (bit2 >= 0 && (s == 20 ? k <= 24 : (s == 10 ? k <= 32 : k <= 48))) return ((mfma_env() >> bit2) & 1) != 0;
  return s > 5 && ks_blk_mfma_nt_f64(k, s) > 0;   // (also 8 / 10 beyond the widths of their register / ring forms)
}
bool ks_blk_rot_ok_f64(int cin, int k, int s) {
  int a, b, c;
  if (!pass2_mfma(k, s) || !((mfma_env() >> 6) & 1)) return false;
  return k >= 1 && cin >= k && rot_shape(cin, k, s, a, b, c);
}
int ks_blk_launch_part1(int which, const BlkLaunchArgs& a) {
  if (which == 2) return go_rot_by_shape(a);
  // (8 / 10 beyond the widths of their register / ring forms exist on the matrix instruction only: the default branch)
  const bool wide = (a.s == 8 && a.k > 48) || (a.s == 10 && a.k > 32);
  switch (wide ? 0 : a.s) {
    case 5: return by_ncw<double, 5>(which, a);
    case 8: return by_ncw<double, 8>(which, a);
    case 10: return by_ncw<double, 10>(which, a);
    case 20: return by_ncw<double, 20>(which, a);
    default:
      // sizes without register / ring forms: the matrix-instruction kernel of the next tile count (run-time s)
      switch (ks_blk_mfma_nt_f64(a.k, a.s)) {
        case 2: return go_mfma_by_k<2, 16>(which, a);
        case 3: return go_mfma_by_k<3, 12>(which, a);
        case 4: return go_mfma_by_k<4, 7, 7>(which, a);   // (25-28 columns only: seven column groups)
        case 5: return go_mfma_by_k<5, 6>(which, a);
        default: throw std::runtime_error("block kernels: block size not in this part");
      }
  }
}
#else
int ks_blk_mfma_nt_c64(int k, int s) {
  if (!((mfma_env() >> 7) & 1) || k < 1 || k > 48 || s < 1) return 0;
  if (s <= 8) return 2;                       // (up to 48 columns)
  return (s <= 10 && k <= 32) ? 3 : 0;        // (k_fin_blk<cd> holds factors of up to 10 x 10: blk_smax)
}
bool ks_blk_rot_ok_c64(int cin, int k, int s) {
  int a, b, c;
  if (!(s > 5 && ks_blk_mfma_nt_c64(k, s) > 0) || !((mfma_env() >> 6) & 1)) return false;
  return k >= 1 && cin >= k && rot_shape(cin, k, s, a, b, c);
}
int ks_blk_launch_part2(int which, const BlkLaunchArgs& a) {
  if (which == 2) return go_rot_by_shape(a);
  // ComplexF64 blocks beyond 5 run on the matrix instruction only (real view of the basis, ks_block_mfma.hpp); bit 7 of KS_BLK_MFMA
  if (a.s > 5 || a.k > 32) {
    const int nt = ks_blk_mfma_nt_c64(a.k, a.s);
    if (nt == 2) return go_mfma_by_k<2, 12, 1, true>(which, a);
    if (nt == 3) return go_mfma_by_k<3, 8, 1, true>(which, a);
  }
  switch (a.s) {
    case 1: return by_ncw<cd, 1>(which, a);
    case 2: return by_ncw<cd, 2>(which, a);
    case 3: return by_ncw<cd, 3>(which, a);
    case 4: return by_ncw<cd, 4>(which, a);
    case 5: return by_ncw<cd, 5>(which, a);
    default: throw std::runtime_error("block kernels: block size not instantiated for ComplexF64");
  }
}
#endif
