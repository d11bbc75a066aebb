// gfx950 (MI355X / CDNA4) kernels of the Krylov-Schur hot path.  Written for wave64, 256 CUs in
// 8 XCDs, HBM3E-bound streaming: 16-byte per-lane accesses, contiguous per-workgroup row ranges,
// deterministic two-stage reductions (no float atomics), decisions of the DGKS test taken on the
// device so a whole expansion can be enqueued without host round trips.
//
// Data layout in HBM
//   V   : column-major n x (maxdim+1), leading dimension ldv (multiple of 64 elements, so every
//         column starts 512-byte aligned); rows n..ldv-1 of every column are ZERO (kernels stream
//         whole 16-byte packs without tail masks and the pad contributes nothing to dots/norms).
//   CSR : rowptr int32[n+1], colidx int32[nnz], val T[nnz]  (12 B / nnz + 4 B / row).
//
// Element types: double (Float64) and cd (ComplexF64, interleaved).  A "pack" is 16 bytes:
// two consecutive rows of a real column or one row of a complex column.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ks_p2p.hpp"

namespace ksd {

struct cd {
  double x, y;
};

constexpr int kBlock = 256;           // threads per workgroup (4 waves)
constexpr double kEta = 0.70710678118654752440;  // sqrt(2)/2, src/expansion.jl:32,74

// Per-workspace device state shared by the kernels of one expansion batch.
struct DevState {
  double rnorm;       // norm before the (current) projection     src/expansion.jl:81,92
  double wnorm;       // norm after the projection                src/expansion.jl:88,96
  double inv_norm;    // 1 / wnorm of the step just finished (consumed by k_scale)
  double rnorm2;      // rnorm of the second pass (= wnorm of the first, src/expansion.jl:92)
  int32_t reorth;     // 1 while the DGKS second pass of the current step is pending
  int32_t breakdown;  // step index j at which orthogonalize! returned false, else -1
  int32_t n_reorth;   // number of second passes taken in this batch
  int32_t n_steps;    // steps completed in this batch
  // deferred normalisation (fused Float64 path): the column produced by the previous step of the batch is
  // still unnormalised when the next step starts; its norm is folded into that step's first reduction.
  int32_t pend;        // 1: the input column of the current step is unnormalised
  int32_t pend_reorth; // 1: its squared norm is the (block-partial) result of the second-pass update,
                       // 0: it is `wnorm` (already reduced) of the first pass
  double rnorm_p;      // reference norm of the pending breakdown test (src/expansion.jl:99)
  double invb;         // 1 / norm of the pending column (1 when none)
  // two-pass expansion: power of two ~ 1 / ||A v|| of the previous step (an upper bound of the norm of the newest stored
  // column).  k_dots multiplies y' = A * (stored column)
  // by it before squaring / multiplying, so that the partial sums stay ~ ||A||^2 instead of ||A||^4 (the stored column is
  // not normalised): the representable range of ||A|| is 1e+-150 as in the reference instead of 1e+-75.  Exact (power of 2).
  double sigma;
  // two-pass expansion: the largest ratio ||c|| / beta (second-pass correction against what is left of the vector) an
  // implicit second pass may carry.  A step whose correction is larger is NOT settled by the triangular factor: the
  // reduction kernel stops the batch there (bail = that step) and the host redoes the step with the correction applied
  // to the vector (the explicit three-pass form).  <= 0: no limit.
  double max_ratio;
  int32_t bail;       // step handed back for the explicit second pass, else -1
  int32_t blk_bail;   // s-step expansion (ks_block_kernels.hpp): first step of a block that was abandoned (rank-deficient Gram
                      // matrix: breakdown or an ill-conditioned Newton basis), else -1; the host redoes it step by step
  double blk_piv1, blk_piv2;  // ... smallest Cholesky pivot ratio of the batch so far, stage 1 / stage 2 (1 = orthogonal block)
  double blk_gdev;            // ... largest entry of |G_t - I|: how far the written blocks were from orthonormal
};
static_assert(sizeof(DevState) <= 128, "DevState must fit its slot in the control block (kCtlStateSlot)");

// ------------------------------------------------------------------------------------------------
// element helpers
// ------------------------------------------------------------------------------------------------
template <class T> struct Pack;
template <> struct Pack<double> {
  using type = double2;
  static constexpr int R = 2;  // rows per 16-byte pack
};
template <> struct Pack<cd> {
  using type = cd;
  static constexpr int R = 1;
};

__device__ __forceinline__ double2 ld_pack(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ cd ld_pack(const cd* p) {
  const double2 t = *reinterpret_cast<const double2*>(p);
  return cd{t.x, t.y};
}
__device__ __forceinline__ void st_pack(double* p, double2 v) { *reinterpret_cast<double2*>(p) = v; }
__device__ __forceinline__ void st_pack(cd* p, cd v) { *reinterpret_cast<double2*>(p) = make_double2(v.x, v.y); }

// Non-temporal (streaming) forms: V columns, matrix values/indices and the updated vector are touched
// once per kernel; keeping them out of L2 is worth ~5 % on reads and removes most of the read/write
// interference of the update kernels on gfx950 (tools/streambench.hip).
__device__ __forceinline__ double2 ld_pack_nt(const double* p) {
  double2 v;
  v.x = __builtin_nontemporal_load(p);
  v.y = __builtin_nontemporal_load(p + 1);
  return v;
}
__device__ __forceinline__ cd ld_pack_nt(const cd* p) {
  const double* q = reinterpret_cast<const double*>(p);
  return cd{__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1)};
}
// Loads of the basis V.  NT = true (default): streaming -- the basis does not fit any cache, every line is used once per
// kernel (+5 % on pure reads, tools/streambench.hip).  NT = false: the basis FITS the 256 MiB memory-side cache (BASELINE
// configs 2-4: 168-328 MB), cacheable loads let it stay there from kernel to kernel: k_dots 34.3 -> 31.9 us, projection
// 29.0 -> 26.8, rotation 57 -> 49 at n = 1e6 (config 3 +3.8 %, config 2 +2.2 %); at 413 MB it is a wash and beyond that
// a loss (n = 2.7e6: -9 %, headline -10 %).  Chosen per workspace from its size (ks_workspace::v_nt).
template <bool NT, class T> __device__ __forceinline__ typename Pack<T>::type ld_v(const T* p) {
  if constexpr (NT) return ld_pack_nt(p);
  else return ld_pack(p);
}

__device__ __forceinline__ void st_pack_nt(double* p, double2 v) {
  __builtin_nontemporal_store(v.x, p);
  __builtin_nontemporal_store(v.y, p + 1);
}
__device__ __forceinline__ void st_pack_nt(cd* p, cd v) {
  double* q = reinterpret_cast<double*>(p);
  __builtin_nontemporal_store(v.x, q);
  __builtin_nontemporal_store(v.y, q + 1);
}

// one element, streaming: y of the SpMV is consumed by the next kernel from HBM anyway; a plain store parks
// the 8n bytes dirty in the memory-side cache and their write-back then lands in the NEXT kernel's read
// stream (measured: +25 us on the k_dots that follows a fast SpMV)
__device__ __forceinline__ void st_elem_nt(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_elem_nt(cd* p, cd v) { st_pack_nt(p, v); }

__device__ __forceinline__ double zero_of(double) { return 0.0; }
__device__ __forceinline__ cd zero_of(cd) { return cd{0.0, 0.0}; }
__device__ __forceinline__ double2 zero_pack(double) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ cd zero_pack(cd) { return cd{0.0, 0.0}; }

// acc += sum over the pack of conj(v) * w
__device__ __forceinline__ void dot_acc(double& acc, double2 v, double2 w) {
  acc = fma(v.x, w.x, acc);
  acc = fma(v.y, w.y, acc);
}
__device__ __forceinline__ void dot_acc(cd& acc, cd v, cd w) {
  acc.x = fma(v.x, w.x, fma(v.y, w.y, acc.x));
  acc.y = fma(v.x, w.y, fma(-v.y, w.x, acc.y));
}
// s += v * g   (g one coefficient)
__device__ __forceinline__ void axpy_acc(double2& s, double2 v, double g) {
  s.x = fma(v.x, g, s.x);
  s.y = fma(v.y, g, s.y);
}
__device__ __forceinline__ void axpy_acc(cd& s, cd v, cd g) {
  s.x = fma(v.x, g.x, fma(-v.y, g.y, s.x));
  s.y = fma(v.x, g.y, fma(v.y, g.x, s.y));
}
__device__ __forceinline__ double2 sub_pack(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cd sub_pack(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ double nrm2_pack(double2 a) { return fma(a.x, a.x, a.y * a.y); }
__device__ __forceinline__ double nrm2_pack(cd a) { return fma(a.x, a.x, a.y * a.y); }
__device__ __forceinline__ double2 scale_pack(double2 a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ cd scale_pack(cd a, double s) { return cd{a.x * s, a.y * s}; }

__device__ __forceinline__ double add_(double a, double b) { return a + b; }
__device__ __forceinline__ cd add_(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ double mul_(double a, double b) { return a * b; }
__device__ __forceinline__ cd mul_(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ cd fma_(cd a, cd b, cd c) {
  return cd{fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y))};
}

// wave64 sum (all lanes end up with the total)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ cd wave_sum(cd v) { return cd{wave_sum(v.x), wave_sum(v.y)}; }

// Contiguous pack range of workgroup b out of nb: [begin, end)
__device__ __forceinline__ void block_range(int64_t npacks, int b, int nb, int64_t& begin, int64_t& end) {
  const int64_t per = (npacks + nb - 1) / nb;
  begin = (int64_t)b * per;
  end = begin + per;
  if (end > npacks) end = npacks;
  if (begin > npacks) begin = npacks;
}

// splitmix64 finaliser: the portable counter-based RNG of SURVEY.md section 8d.
__device__ __host__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __host__ __forceinline__ double uniform_hash(uint64_t seed, uint64_t idx) {
  return (double)(splitmix64(seed ^ idx) >> 11) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------------------------------------
// rand!(v)  (src/expansion.jl:15,21) -- pure function of (seed, global row)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock) k_fill_uniform(T* __restrict__ v, int64_t n, int64_t ld, uint64_t seed,
                                                         uint64_t row_begin) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ld; i += (int64_t)gridDim.x * kBlock) {
    if constexpr (sizeof(T) == 8) {
      v[i] = i < n ? uniform_hash(seed, row_begin + i) : 0.0;
    } else {
      v[i] = i < n ? cd{uniform_hash(seed, 2 * (row_begin + i)), uniform_hash(seed, 2 * (row_begin + i) + 1)}
                   : cd{0.0, 0.0};
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CSR SpMV  y = A x   (mul!(y, A, x), src/expansion.jl:121)
//
// "CSR-stream": a workgroup owns 256 consecutive rows.  Phase 1 streams the tile's column indices
// and values with fully coalesced loads (lane p reads non-zero p), gathers x and writes the products
// to LDS; phase 2: one thread per row sums its LDS segment and writes y coalesced.  Tiles whose
// non-zeros do not fit the LDS buffer fall back to one-wave-per-row.  Tile -> workgroup mapping is
// XCD-aware (each XCD's private L2 sees a contiguous range of rows, so the +-plane neighbours of the
// 3-D stencil hit in the same L2).
//
// Distributed mode: column indices >= n_local address the ghost buffer `xg` (filled by the halo
// exchange: RCCL send/recv, or remote stores of the neighbours in peer-to-peer mode) instead of the local column.
// ------------------------------------------------------------------------------------------------
// Newton-basis step of the s-step expansion (ks_block_kernels.hpp) fused into a product's store: y = sigma (A x - theta x).
// on == 0: the plain product, bit-identical to what the kernels have always stored.
template <class T> struct ShiftArg {
  int on = 0;
  T theta{};
  double sigma = 1.0;
};
__device__ __forceinline__ double shift_apply(const ShiftArg<double>& sh, double s, const double* __restrict__ x, int64_t row) {
  return sh.on ? (s - sh.theta * x[row]) * sh.sigma : s;
}
__device__ __forceinline__ cd shift_apply(const ShiftArg<cd>& sh, cd s, const cd* __restrict__ x, int64_t row) {
  if (!sh.on) return s;
  const cd xv = x[row];
  const cd tx = cd{sh.theta.x * xv.x - sh.theta.y * xv.y, sh.theta.x * xv.y + sh.theta.y * xv.x};
  return cd{(s.x - tx.x) * sh.sigma, (s.y - tx.y) * sh.sigma};
}

constexpr int kSpmvRows = 256;       // rows per block at most
constexpr int kSpmvCapBytes = 32768;  // LDS for the products of a block: NI * 256 * sizeof(T) <= 32 KiB

__device__ __forceinline__ int xcd_remap(int b, int nt) {
  const int q = nt >> 3, r = nt & 7, xcd = b & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ int32_t ld_i32(const int32_t* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ double ld_val(const double* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ cd ld_val(const cd* p, bool nt) { return nt ? ld_pack_nt(p) : *p; }

// The product a * x of one stored entry with every partial result rounded on its own (no fused
// multiply-add across the complex cross terms): all SpMV layouts compute exactly this, which is what makes
// their results bit-identical to each other.
__device__ __forceinline__ void keep(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ double mul_nc(double a, double b) {
  double p = a * b;
  keep(p);
  return p;
}
__device__ __forceinline__ cd mul_nc(cd a, cd b) {
  double p1 = a.x * b.x, p2 = a.y * b.y, p3 = a.x * b.y, p4 = a.y * b.x;
  keep(p1); keep(p2); keep(p3); keep(p4);
  cd r{p1 - p2, p3 + p4};
  keep(r.x); keep(r.y);
  return r;
}

// Row blocks ("CSR-adaptive" tiling, built once at upload by make_csr): block b owns rows [blkrow[b], blkrow[b+1]) and
// non-zeros [blkptr[b], blkptr[b+1]); a block has at most 256 rows and at most CAP = NI * 256 non-zeros.  A row with
// more than CAP entries is cut into CHUNK blocks of <= CAP entries (blkpart[b] >= 0: index of the chunk's partial sum).
// Regular matrices get 256-row blocks (the 7-point stencil: 1792 non-zeros); skewed matrices get short blocks around
// their heavy rows instead of dragging a whole tile onto a slow path, and a very long row is spread over many workgroups.
// IP = type of the non-zero offsets (rowptr / blkptr): int32_t, or int64_t when nnz >= 2^31 (int64-nnz CSR).
//
// VI = true: value-indexed CSR ("CSR-VI").  Matrices with at most 256 distinct stored values (stencils,
// unweighted graphs, uniform finite-element meshes) and fewer than 2^24 local-extended columns are
// uploaded as ONE 32-bit word per non-zero, (dictionary index << 24) | column, plus the dictionary:
// 4 B instead of 12 B per non-zero on the dominant stream of this HBM-bound kernel.  The products and their
// summation order are exactly those of the plain layout, so y is bit-identical.
// The block's non-zeros are loaded by a fully unrolled, predicated loop of NI iterations: all index loads issue
// back to back, then all gathers -- no serial remainder loop.
template <class T, class IP, bool VI, int NI>
__global__ void __launch_bounds__(kBlock)
    k_spmv_csr(const IP* __restrict__ blkptr, const int32_t* __restrict__ blkrow, const IP* __restrict__ rowptr,
               const int32_t* __restrict__ colidx, const T* __restrict__ val, const T* __restrict__ x,
               const T* __restrict__ xg, T* __restrict__ y, int64_t n, int nblk, const DevState* __restrict__ st,
               const uint32_t* __restrict__ hseq, int64_t gstride, int ndict, const int32_t* __restrict__ blkpart,
               T* __restrict__ lpart, const T* __restrict__ yacc = nullptr, int plain_store = 0, HaloFused hf = HaloFused{},
               HaloArgs ha = HaloArgs{}, P2pDev pd = P2pDev{}, bool nt_loads = true, bool row_gather = true,
               ShiftArg<T> sh = ShiftArg<T>{}) {
  // yacc != nullptr: this launch handles ONE COLUMN BLOCK of the matrix (column-blocked layout) and continues the row sums
  // an earlier launch left in yacc -- entries of a row are visited in CSR order across the launches, so y is bit-identical
  if (st && st->breakdown >= 0) return;
  // peer-to-peer halo (ks_p2p.hpp): the ghost vector is double-buffered, the parity of the halo sequence
  // number the push kernel just published selects the slot
  if (hseq) xg += (int64_t)(*hseq & 1u) * gstride;
  constexpr int CAP = NI * kBlock;
  __shared__ T prod[CAP];
  __shared__ int32_t lcol[CAP];
  // VI: the dictionary is staged in LDS (gathering it from global memory per non-zero measured 10 % slower);
  // the index loads are issued BEFORE the staging barrier so the two latencies overlap
  __shared__ T dict[VI ? 256 : 1];
  const int tid = threadIdx.x;
  T dmine = zero_of(T{});
  if (VI && tid < ndict) dmine = val[tid];
  const int b = xcd_remap(blockIdx.x, nblk);
  const int32_t r0 = blkrow[b], r1 = blkrow[b + 1];
  const IP p0 = blkptr[b], p1 = blkptr[b + 1];
  const int32_t cnt = (int32_t)(p1 - p0);  // <= CAP by construction
  const int32_t part = blkpart ? blkpart[b] : -1;
  // peer-to-peer mode: the ghost exchange is part of this launch (ks_p2p.hpp; xg is then the slot the host picked)
  if (hf.enabled) halo_fused_prologue<T>(x, hf, ha, pd, (long long)r0, (long long)r1);
  if (part < 0) {
    // this thread's row bounds for phase 2: issued now, so their latency hides behind phase 1
    const int32_t rmine = (r0 + tid < r1) ? r0 + tid : r1 - 1;
    const int32_t ra = (int32_t)(rowptr[rmine] - p0), rb = (int32_t)(rowptr[rmine + 1] - p0);
    const int32_t* ci = colidx + p0;
    const T* va = val + (VI ? (IP)0 : p0);
    int32_t c[NI];
    T a[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int32_t p = tid + k * kBlock;
      // (the matrix is touched once per application: streaming loads keep it from evicting the x every row re-reads
      // through L2 -- KS_SPMV_CSR_NT=0 restores plain loads)
      c[k] = (p < cnt) ? ld_i32(ci + p, nt_loads) : 0;
      if (!VI) a[k] = (p < cnt) ? ld_val(va + p, nt_loads) : zero_of(T{});
    }
    if (VI) {
      dict[tid] = dmine;
      __syncthreads();
    }
    if (row_gather) {
      // ROW-GATHER form (default): the coalesced loads above only STAGE the block's (column, value) pairs in LDS; then one
      // thread per row walks its segment and gathers x itself.  lane = row, so the gathers of one instruction read
      // x[r + delta], x[r + 1 + delta], ... -- consecutive addresses for a banded matrix, a handful of cache lines per
      // wave instruction -- where the non-zero-parallel gathers below touch one line per band per ~9 lanes; one barrier
      // instead of two, no product round trip through LDS.  Same products, same order of additions: y is bit-identical.
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int32_t p = tid + k * kBlock;
        if (VI) {
          a[k] = dict[(uint32_t)c[k] >> 24];
          c[k] &= 0xffffff;
        }
        if (p < cnt) {
          prod[p] = a[k];
          lcol[p] = c[k];
        }
      }
      __syncthreads();
      if (r0 + tid < r1) {
        T s = yacc ? yacc[r0 + tid] : zero_of(T{});
        for (int32_t p = ra; p < rb; p += 8) {
          int32_t cc[8];
          T aa[8], xx[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool ok = p + u < rb;
            cc[u] = ok ? lcol[p + u] : 0;
            aa[u] = ok ? prod[p + u] : zero_of(T{});
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) xx[u] = (cc[u] < n) ? x[cc[u]] : xg[cc[u] - n];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (p + u < rb) s = add_(s, mul_nc(aa[u], xx[u]));
        }
        if (plain_store) y[r0 + tid] = s;  // (an intermediate column block: the next launch reads it back)
        else st_elem_nt(y + r0 + tid, shift_apply(sh, s, x, r0 + tid));
      }
      return;
    }
    T xv[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (VI) {
        a[k] = dict[(uint32_t)c[k] >> 24];
        c[k] &= 0xffffff;
      }
      xv[k] = (c[k] < n) ? x[c[k]] : xg[c[k] - n];  // (ghost columns only occur in rows of boundary blocks, which waited)
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int32_t p = tid + k * kBlock;
      if (p < cnt) prod[p] = mul_nc(a[k], xv[k]);
    }
    __syncthreads();
    if (r0 + tid < r1) {
      T s = yacc ? yacc[r0 + tid] : zero_of(T{});
      for (int32_t p = ra; p < rb; ++p) s = add_(s, prod[p]);
      if (plain_store) y[r0 + tid] = s;  // (an intermediate column block: the next launch reads it back)
      else st_elem_nt(y + r0 + tid, shift_apply(sh, s, x, r0 + tid));
    }
  } else {
    // one CHUNK (<= CAP entries) of a row too long for a block: the same coalesced loads, then every thread adds up
    // its NI products and a fixed-shape LDS tree leaves the chunk's partial sum in lpart[part]; k_spmv_longfix adds
    // the partials of a row in chunk order.  (A long row processed by ONE workgroup was a latency chain of
    // L / 1024 dependent round trips: 64 rows of 40 000 entries cost 100 us.)
    const int32_t* ci = colidx + p0;
    const T* va = val + (VI ? (IP)0 : p0);
    int32_t c[NI];
    T a[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int32_t p = tid + k * kBlock;
      // (the matrix is touched once per application: streaming loads keep it from evicting the x every row re-reads
      // through L2 -- KS_SPMV_CSR_NT=0 restores plain loads)
      c[k] = (p < cnt) ? ld_i32(ci + p, nt_loads) : 0;
      if (!VI) a[k] = (p < cnt) ? ld_val(va + p, nt_loads) : zero_of(T{});
    }
    if (VI) {
      dict[tid] = dmine;
      __syncthreads();
    }
    T xv[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (VI) {
        a[k] = dict[(uint32_t)c[k] >> 24];
        c[k] &= 0xffffff;
      }
      xv[k] = (c[k] < n) ? x[c[k]] : xg[c[k] - n];  // (ghost columns only occur in rows of boundary blocks, which waited)
    }
    T s = zero_of(T{});
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (tid + k * kBlock < cnt) s = add_(s, mul_nc(a[k], xv[k]));
    prod[tid] = s;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
      if (tid < off) prod[tid] = add_(prod[tid], prod[tid + off]);
      __syncthreads();
    }
    if (tid == 0) lpart[part] = prod[0];
  }
}

// ------------------------------------------------------------------------------------------------
// Column-blocked CSR in ONE launch (KS_LAYOUT_CSR_CB; matrices with scattered columns whose x does not fit one XCD's L2:
// BASELINE config 3).  The matrix is stored as NB column blocks, each a CSR over all rows restricted to ~4 MiB of x.  A
// workgroup owns a tile of RPT x 256 rows for the WHOLE product and walks the column blocks in the outer loop: per block
// it streams the tile's entries of that block (coalesced, non-temporal), gathers x -- every workgroup of the launch is
// gathering from the same block at roughly the same time, so the block stays resident in each XCD's L2 -- leaves the
// rounded products in LDS, and every thread continues the sums of its RPT rows IN REGISTERS.  Entries of a row are sorted
// by column, so the additions happen in CSR order: y is bit-identical to the plain layout.
// RPT is chosen at upload so that all tiles of the matrix are resident at once (one "round" of workgroups: they start
// together and stay in step without any global synchronisation).  Replaces one launch per block with the row sums
// written to and read back from y in between (16 n bytes per block boundary, and a launch each: 46.5 us at n = 1e6).
// ------------------------------------------------------------------------------------------------
constexpr int kCbMaxBlocks = 8;
template <class T> struct CbArgs {
  int nb;
  const int32_t* rowptr[kCbMaxBlocks];
  const int32_t* colidx[kCbMaxBlocks];
  const T* val[kCbMaxBlocks];
  const T* xb[kCbMaxBlocks];  // gather base of the block; null: x (distributed operators: the ghost vector for ghost-column blocks)
};

template <class T, int NI, int RPT>
__global__ void __launch_bounds__(kBlock)
    k_spmv_csr_cb(const CbArgs<T> A, const T* __restrict__ x, T* __restrict__ y, int64_t n, int ntiles,
                  const DevState* __restrict__ st, ShiftArg<T> sh = ShiftArg<T>{}) {
  if (st && st->breakdown >= 0) return;
  constexpr int CAP = NI * kBlock;
  __shared__ T prod[CAP];
  const int tid = threadIdx.x;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int64_t row0 = (int64_t)tile * (kBlock * RPT);
  const int64_t rowe = (row0 + kBlock * RPT < n) ? row0 + kBlock * RPT : n;
  T s[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) s[q] = zero_of(T{});
  for (int b = 0; b < A.nb; ++b) {
    const int32_t* __restrict__ rp = A.rowptr[b];
    const int32_t p0 = rp[row0], p1 = rp[rowe];
    const int32_t cnt = p1 - p0;  // <= CAP by construction (RPT is chosen at upload)
    // this thread's row bounds in the block: issued now, their latency hides behind the staging loads
    int32_t ra[RPT], rb[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      int64_t r = row0 + (int64_t)q * kBlock + tid;
      if (r >= rowe) r = rowe - 1;
      ra[q] = rp[r] - p0;
      rb[q] = rp[r + 1] - p0;
    }
    const int32_t* ci = A.colidx[b] + p0;
    const T* va = A.val[b] + p0;
    const T* __restrict__ xs = A.xb[b] ? A.xb[b] : x;
    int32_t c[NI];
    T a[NI], xv[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int32_t p = tid + k * kBlock;
      c[k] = (p < cnt) ? ld_i32(ci + p, true) : 0;
      a[k] = (p < cnt) ? ld_val(va + p, true) : zero_of(T{});
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) xv[k] = xs[c[k]];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int32_t p = tid + k * kBlock;
      if (p < cnt) prod[p] = mul_nc(a[k], xv[k]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      if (row0 + (int64_t)q * kBlock + tid < rowe)
        for (int32_t p = ra[q]; p < rb[q]; ++p) s[q] = add_(s[q], prod[p]);
    }
    __syncthreads();  // (prod is reused by the next block)
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int64_t r = row0 + (int64_t)q * kBlock + tid;
    if (r < rowe) st_elem_nt(y + r, shift_apply(sh, s[q], x, r));
  }
}

// y[lrow[i]] = sum of the chunk partials of long row i, in chunk order (deterministic)
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_spmv_longfix(const int32_t* __restrict__ lrow, const int32_t* __restrict__ lfirst, const T* __restrict__ lpart,
                   T* __restrict__ y, int nlong, const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nlong) return;
  T s = zero_of(T{});
  for (int32_t q = lfirst[i]; q < lfirst[i + 1]; ++q) s = add_(s, lpart[q]);
  y[lrow[i]] = s;
}

// ------------------------------------------------------------------------------------------------
// SpMV, sliced-ELLPACK layout ("SELL-64-sigma").  The CSR-stream kernel above assigns consecutive LANES to consecutive
// NON-ZEROS: its x gathers touch ~64 different cache lines per wave instruction even for a banded matrix, and the
// texture-address path (one line per cycle and CU) caps it at ~4.8 TB/s on the 7-point Laplacian.  Here a slice of 64
// consecutive rows is stored column-major -- entry k of all 64 rows contiguous -- so lane = row: index and value loads
// are 256/512-byte coalesced and, for banded / stencil matrices, so are the gathers x[r + delta].  A slice is as wide
// as its longest row; padding entries carry column -1 and are skipped (never multiplied: 0 * Inf must not appear).
// Entries are visited in CSR order with a separately rounded product and add, so y is bit-identical to the CSR kernels.
// sigma > 1: rows are sorted by length inside windows of sigma rows before slicing (less padding for ragged
// matrices); `perm` maps slice position -> row.  Chosen at upload when the padding is small (make_csr).
// VI as for k_spmv_csr: one 32-bit word per entry (dictionary index << 24 | column), padding = 0xFFFFFFFF.
// ------------------------------------------------------------------------------------------------
template <class T, class IP, bool VI, int UN, bool NTL = true>
__global__ void __launch_bounds__(kBlock)
    k_spmv_sell(const IP* __restrict__ sliceptr, const int32_t* __restrict__ scol, const T* __restrict__ sval,
                const int32_t* __restrict__ perm, const T* __restrict__ x, const T* __restrict__ xg, T* __restrict__ y,
                int64_t n, int nslices, int ngroups, const DevState* __restrict__ st, const uint32_t* __restrict__ hseq,
                int64_t gstride, int ndict, ShiftArg<T> sh = ShiftArg<T>{}) {
  if (st && st->breakdown >= 0) return;
  if (hseq) xg += (int64_t)(*hseq & 1u) * gstride;
  __shared__ T dict[VI ? 256 : 1];
  if (VI) {
    if ((int)threadIdx.x < ndict) dict[threadIdx.x] = sval[threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int slice = xcd_remap(blockIdx.x, ngroups) * (kBlock / 64) + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  const IP p0 = sliceptr[slice], p1 = sliceptr[slice + 1];
  const int32_t w = (int32_t)((p1 - p0) >> 6);
  const int64_t pos = (int64_t)slice * 64 + lane;
  const int32_t* ci = scol + p0 + lane;
  const T* va = sval + (VI ? (IP)0 : p0) + (VI ? 0 : lane);
  T s = zero_of(T{});
  for (int32_t k0 = 0; k0 < w; k0 += UN) {
    int32_t c[UN];
    T a[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool in = k0 + u < w;  // wave-uniform
      c[u] = in ? ld_i32(ci + (int64_t)(k0 + u) * 64, NTL) : -1;  // the matrix is touched once per application: streaming loads
      if (!VI) a[u] = in ? ld_val(va + (int64_t)(k0 + u) * 64, NTL) : zero_of(T{});
    }
    T xv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      int32_t cc = c[u];
      if (VI) {
        a[u] = dict[(uint32_t)cc >> 24 & 0xffu];
        cc = (cc == -1) ? -1 : (cc & 0xffffff);
        c[u] = cc;
      }
      const int32_t cl = cc < 0 ? 0 : cc;
      xv[u] = (cl < n) ? x[cl] : xg[cl - n];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (c[u] >= 0) s = add_(s, mul_nc(a[u], xv[u]));
  }
  if (pos < n) {
    const int64_t row = perm ? perm[pos] : pos;
    st_elem_nt(y + row, shift_apply(sh, s, x, row));
  }
}

// ------------------------------------------------------------------------------------------------
// SpMV, delta-value-indexed layout ("CSR-DVI").  When the set of distinct (column - row, value) pairs of a
// matrix has at most 256 members -- stencils on structured grids, banded Toeplitz-like operators, regular
// graph Laplacians -- every stored entry is ONE byte indexing that dictionary: the matrix stream shrinks
// from 12 to 1 byte per non-zero and, because lane = row, the x gathers of one instruction read
// consecutive addresses x[r + delta] (coalesced, unlike the CSR-stream gather).  Row entries are visited in CSR
// order with a separate multiply and add, so y is bit-identical to the plain layout.
//
// A workgroup owns 256 * RPT consecutive rows; thread t handles rows base + k*256 + t, k < RPT (RPT independent
// dependency chains per thread: rowptr -> codes -> x).  At one row per thread the kernel was LATENCY-bound (three
// dependent memory round trips per 27 bytes of a row: 3.1-3.3 TB/s); RPT = 4 puts four times the bytes in flight.
// The tile's code bytes are contiguous in memory: they are staged into LDS with 16-byte coalesced loads (a row's
// 7 single-byte global loads become 7/16 of one wide load); tiles whose codes do not fit kDviLds take them from
// global memory directly.
// ------------------------------------------------------------------------------------------------
constexpr int kDviLds = 16384;  // bytes of staged codes per tile (1024 rows x up to 16 entries)

template <class T, class IP, int UN, int RPT>
__global__ void __launch_bounds__(kBlock)
    k_spmv_dvi(const IP* __restrict__ rowptr, const uint8_t* __restrict__ codes, const int32_t* __restrict__ ddelta,
               const T* __restrict__ dval, const T* __restrict__ x, const T* __restrict__ xg, T* __restrict__ y,
               int64_t n, int ntiles, int ndict, const DevState* __restrict__ st, const uint32_t* __restrict__ hseq,
               int64_t gstride, ShiftArg<T> sh = ShiftArg<T>{}) {
  if (st && st->breakdown >= 0) return;
  if (hseq) xg += (int64_t)(*hseq & 1u) * gstride;
  __shared__ int32_t sd[256];
  __shared__ T sv[256];
  __shared__ __attribute__((aligned(16))) uint8_t sc[kDviLds + 16];
  const int tid = threadIdx.x;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int64_t rbase = (int64_t)tile * (kBlock * RPT);
  const int64_t rend = (rbase + kBlock * RPT < n) ? rbase + kBlock * RPT : n;
  const IP p0 = rowptr[rbase], p1 = rowptr[rend];
  int64_t r[RPT];
  IP a[RPT];
  int32_t len[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    r[k] = rbase + k * kBlock + tid;
    const bool live = r[k] < n;
    a[k] = live ? rowptr[r[k]] : p0;
    len[k] = live ? (int32_t)(rowptr[r[k] + 1] - a[k]) : 0;
  }
  // stage the tile's codes (16-byte loads from the 16-byte aligned address at or below p0; the buffer is padded)
  const IP q0 = p0 & ~(IP)15;
  const int64_t nbytes = (int64_t)(p1 - q0);
  const bool staged = nbytes <= kDviLds;
  if (staged) {
    for (int32_t i = tid * 16; i < nbytes; i += kBlock * 16)
      *reinterpret_cast<uint4*>(sc + i) = *reinterpret_cast<const uint4*>(codes + q0 + i);
  }
  if (tid < ndict) {
    sd[tid] = ddelta[tid];
    sv[tid] = dval[tid];
  }
  __syncthreads();
  T s[RPT];
  int32_t maxlen = 0;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    s[k] = zero_of(T{});
    maxlen = len[k] > maxlen ? len[k] : maxlen;
  }
  for (int32_t e0 = 0; e0 < maxlen; e0 += UN) {
    uint8_t code[RPT][UN];
#pragma unroll
    for (int k = 0; k < RPT; ++k)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const bool in = e0 + u < len[k];
        code[k][u] = in ? (staged ? sc[(int32_t)(a[k] - q0) + e0 + u] : codes[a[k] + e0 + u]) : (uint8_t)0;
      }
    T xv[RPT][UN];
#pragma unroll
    for (int k = 0; k < RPT; ++k)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t c = (e0 + u < len[k]) ? r[k] + sd[code[k][u]] : (r[k] < n ? r[k] : 0);
        xv[k][u] = (c < n) ? x[c] : xg[c - n];
      }
#pragma unroll
    for (int k = 0; k < RPT; ++k)
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (e0 + u < len[k]) s[k] = add_(s[k], mul_nc(sv[code[k][u]], xv[k][u]));  // rounded product first, then the add: as the LDS-staged kernel
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k)
    if (r[k] < n) st_elem_nt(y + r[k], shift_apply(sh, s[k], x, r[k]));
}

// ------------------------------------------------------------------------------------------------
// SpMV, stencil-mask layout.  A matrix whose delta-value dictionary has at most 32 entries and whose rows are all
// SUB-SEQUENCES of one ordered list of those entries (constant-coefficient stencils on structured grids with truncated
// boundary rows: the 3-D 7-point Laplacian has 7 slots, interior rows use all of them, boundary rows a subset) is stored
// as ONE BIT PER SLOT AND ROW: bit k of mask[r] says whether row r has slot k = (delta_k, value_k).  The dictionary
// travels in the kernel arguments (scalar registers), so the matrix stream is 1 byte per ROW (7 slots: 17 bytes per row
// in total with x and y, against 27 for the delta-value-indexed and 104 for the plain CSR layout) and the inner loop is a
// handful of vector instructions per entry with no LDS at all -- k_spmv_dvi is bound by instruction issue (byte decode +
// two dictionary reads per entry), this kernel by the memory system.  lane = row, so the gathers x[r + delta_k] are
// coalesced.  Slots are visited in the common order (= CSR order of every row), products rounded separately, absent
// slots skipped (never multiplied): y is bit-identical to the other layouts.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sub_s(double a, double b) { return a - b; }
__device__ __forceinline__ cd sub_s(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ double scl(double a, double s);
__device__ __forceinline__ cd scl(cd a, double s);
constexpr int kStencilSlots = 32;
template <class T> struct StencilDict {
  int32_t delta[kStencilSlots];
  T val[kStencilSlots];
};

// The x loads do NOT wait for the mask: every lane loads x at the clamped address r + delta_k for every slot and the
// mask only selects which products are added (one memory round trip per row instead of two; measured 54 -> see
// profiles/).  RPT rows per thread (rows r, r + 256, ...) put more independent loads in flight.
template <class T, class MT, int RPT>
__global__ void __launch_bounds__(kBlock)
    k_spmv_stencil(const MT* __restrict__ mask, const StencilDict<T> d, int nslots, const T* __restrict__ x,
                   const T* __restrict__ xg, T* __restrict__ y, int64_t n, int64_t nghost, int ntiles,
                   const DevState* __restrict__ st, HaloFused hf, HaloArgs ha, P2pDev pd, int shifted = 0, T theta = T{},
                   double sigma = 1.0, int bt_nlow = -1, int bt_skip = 0) {
  // (xg: the ghost vector of THIS exchange -- the host picks the slot of the double buffer, ks_p2p.hpp)
  if (st && st->breakdown >= 0) return;
  int tile = xcd_remap(blockIdx.x, ntiles);
  // BOUNDARY launch (round 6c, collective transports): the grid covers only the tiles whose rows reference ghost columns -- the
  // first bt_nlow tiles and the last ones, bt_skip tiles further up; the interior rows were written by the paired kernel
  // (k_spmv_stencil2) on the same stream just before (ks_operators.hpp: the split product)
  if (bt_nlow >= 0 && tile >= bt_nlow) tile += bt_skip;
  // peer-to-peer mode: the ghost exchange is part of this launch (ks_p2p.hpp).  The leading boundary rows (a slab's first
  // plane) go to the END of the dispatch order, next to the trailing ones: by the time those workgroups start, the
  // neighbours' entries have long arrived -- nobody sits on a CU spinning while the interior tiles want its slots.
  if (hf.enabled && hf.tile_shift) { tile += hf.tile_shift; if (tile >= ntiles) tile -= ntiles; }
  // Only a BOUNDARY tile (rows outside [ghost_lo_end, ghost_hi_begin)) may touch the ghost vector, and only after it waited
  // for the exchange: every other tile folds ghost-range addresses (slots its rows do not have -- the loads below do not
  // wait for the mask) back onto local rows.  So no tile can pull a ghost line into this CU's vector L1 before the data
  // arrived, and nobody needs an acquire fence or cache-bypassing loads (both were tried: an acquire per waiting workgroup
  // invalidates what the other tiles stream through the caches, 15 -> 35 us; bypassing loads break up the back-to-back
  // issue of the gathers, 15 -> 22 us).
  bool btile = true;
  if (hf.enabled) {
    const long long row_lo = (long long)tile * (kBlock * RPT), row_hi = row_lo + kBlock * RPT;
    btile = row_lo < hf.ghost_lo_end || row_hi > hf.ghost_hi_begin;
    halo_fused_prologue<T>(x, hf, ha, pd, row_lo, row_hi);
  }
  const int64_t cmax = btile ? n + nghost - 1 : n - 1;
  int64_t r[RPT];
  uint32_t m[RPT];
  T s[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    r[q] = (int64_t)tile * (kBlock * RPT) + q * kBlock + threadIdx.x;
    m[q] = r[q] < n ? (uint32_t)mask[r[q]] : 0u;
    s[q] = zero_of(T{});
  }
  constexpr int UN = 8;
#pragma unroll
  for (int k0 = 0; k0 < kStencilSlots; k0 += UN) {
    if (k0 < nslots) {  // uniform
      T xv[RPT][UN];
#pragma unroll
      for (int q = 0; q < RPT; ++q)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          // clamped, mask-independent address (slots beyond nslots have delta 0).  A column beyond the local-extended
          // range is folded back onto the row itself, NOT onto the last ghost slot: in peer-to-peer mode ghost loads bypass
          // the caches, and every interior row hitting the same uncached word serialised the whole launch on one channel
          int64_t c = r[q] + d.delta[k0 + u];
          c = c < 0 ? 0 : (c > cmax ? (r[q] < n ? r[q] : n - 1) : c);
          xv[q][u] = (c < n) ? x[c] : xg[c - n];
        }
#pragma unroll
      for (int q = 0; q < RPT; ++q)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const bool on = (m[q] >> (k0 + u)) & 1u;  // bits of slots >= nslots are never set
          const T p = mul_nc(d.val[k0 + u], xv[q][u]);
          s[q] = on ? add_(s[q], p) : s[q];
        }
    }
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q)
    if (r[q] < n) {
      // shifted: Newton-basis step of the s-step expansion, y = sigma (A x - theta x) (as in k_spmv_stencil2)
      if (shifted) s[q] = scl(sub_s(s[q], mul_(theta, x[r[q]])), sigma);
      st_elem_nt(y + r[q], s[q]);
    }
}

// Paired form (single GPU, no ghost columns): a lane owns the two consecutive rows 2t, 2t+1 and fetches x[r + delta_k],
// x[r + 1 + delta_k] with ONE 16-byte load (8-byte aligned: global loads only need dword alignment).  The 8-byte form
// above is bound by the vector-memory path -- seven 8-byte loads per row through an L1 that moves 8-byte accesses at
// about half the 16-byte rate (MI355X_MICROARCH.md) -- not by HBM: 171 MB took 54 us, a copy of the same bytes takes 28.
// Addresses are clamped so that the pair stays inside x; `sh` says where the wanted elements sit in a clamped pair.
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void ld_pair_u(const double* p, double& a, double& b) {
  const f64x2u v = *reinterpret_cast<const f64x2u*>(p);
  a = v.x;
  b = v.y;
}
__device__ __forceinline__ void ld_pair_u(const cd* p, cd& a, cd& b) {  // complex: two 16-byte elements
  a = p[0];
  b = p[1];
}

template <class T, class MT2>
__global__ void __launch_bounds__(kBlock)
    k_spmv_stencil2(const MT2* __restrict__ mask2, const StencilDict<T> d, int nslots, const T* __restrict__ x,
                    T* __restrict__ y, int64_t n, int ntiles, const DevState* __restrict__ st, int shifted = 0,
                    T theta = T{}, double sigma = 1.0) {
  if (st && st->breakdown >= 0) return;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int64_t r = 2 * ((int64_t)tile * kBlock + threadIdx.x);  // rows r, r + 1
  if (r >= n) return;
  const bool two = r + 1 < n;
  constexpr int MB = (int)sizeof(MT2) * 4;  // mask bits per row
  const MT2 mm = mask2[r >> 1];             // masks of both rows (the mask array is padded to an even row count)
  const uint32_t m0 = (uint32_t)(mm & (MT2)((((uint64_t)1) << MB) - 1)), m1 = (uint32_t)((uint64_t)mm >> MB);
  const int64_t cmax = n - 2;               // last admissible pair start
  T s0 = zero_of(T{}), s1 = zero_of(T{});
  constexpr int UN = 8;
#pragma unroll
  for (int k0 = 0; k0 < kStencilSlots; k0 += UN) {
    if (k0 < nslots) {  // uniform
      T xa[UN], xb[UN];
      int sh[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t c = r + d.delta[k0 + u];
        int64_t lo = c < 0 ? 0 : (c > cmax ? cmax : c);
        if (cmax < 0) lo = 0;
        sh[u] = (int)(c - lo);
        if (n >= 2) ld_pair_u(x + lo, xa[u], xb[u]);
        else { xa[u] = x[0]; xb[u] = x[0]; }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        // wanted: row r -> x[c] = pair[sh], row r+1 -> x[c+1] = pair[sh+1]; out-of-pair positions belong to absent slots
        const T v0 = sh[u] == 1 ? xb[u] : xa[u];
        const T v1 = sh[u] == -1 ? xa[u] : xb[u];
        const T p0 = mul_nc(d.val[k0 + u], v0), p1 = mul_nc(d.val[k0 + u], v1);
        s0 = ((m0 >> (k0 + u)) & 1u) ? add_(s0, p0) : s0;
        s1 = ((m1 >> (k0 + u)) & 1u) ? add_(s1, p1) : s1;
      }
    }
  }
  if (shifted) {
    // Newton-basis step of the s-step expansion (ks_block_kernels.hpp): y = sigma (A x - theta x), fused -- the rows' own x
    // entries are one more (cached) pair load instead of a 24 n-byte pass of their own.  shifted == 0: the plain product,
    // bit-identical to every other layout.
    T x0, x1;
    if (two) ld_pair_u(x + r, x0, x1);
    else { x0 = x[r]; x1 = x0; }
    s0 = scl(sub_s(s0, mul_(theta, x0)), sigma);
    s1 = scl(sub_s(s1, mul_(theta, x1)), sigma);
    if (shifted & 2) {  // cacheable stores: the next launch is the next product of the chain and gathers exactly this vector
      if (two) { if constexpr (sizeof(T) == 8) st_pack(y + r, make_double2(s0, s1)); else { y[r] = s0; y[r + 1] = s1; } }
      else y[r] = s0;
      return;
    }
  }
  if (two) {
    if constexpr (sizeof(T) == 8) st_pack_nt(y + r, make_double2(s0, s1));
    else { st_elem_nt(y + r, s0); st_elem_nt(y + r + 1, s1); }
  } else {
    st_elem_nt(y + r, s0);
  }
}

// ------------------------------------------------------------------------------------------------
// Dense operator  y = A x  (mul!(y, A::Matrix, x), src/expansion.jl:121 with a dense A): A row-major with a
// padded leading dimension (multiple of 2 elements, so every row starts 16-byte aligned).  One wave per row
// and trip: lanes stride over the row with 16-byte non-temporal loads (the matrix is the HBM stream, x stays
// in cache), fixed-order wave reduction -> deterministic.  HBM-bound at 8 (16) bytes per entry.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_gemv_rows(const T* __restrict__ A, int64_t lda, int64_t nrows, int64_t ncols, const T* __restrict__ x,
                T* __restrict__ y, const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t npack = ncols / R;  // full packs per row; a real row of odd length has one scalar tail entry
  for (int64_t r = wave0; r < nrows; r += nwaves) {
    const T* row = A + r * lda;
    T acc0 = zero_of(T{}), acc1 = zero_of(T{});
    int64_t p = lane;
    for (; p + 64 < npack; p += 128) {  // two independent accumulators, two loads in flight per lane
      const P a0 = ld_pack_nt(row + p * R), a1 = ld_pack_nt(row + (p + 64) * R);
      const P x0 = ld_pack(x + p * R), x1 = ld_pack(x + (p + 64) * R);
      if constexpr (R == 2) {
        acc0 = fma(a0.x, x0.x, fma(a0.y, x0.y, acc0));
        acc1 = fma(a1.x, x1.x, fma(a1.y, x1.y, acc1));
      } else {
        acc0 = fma_(a0, x0, acc0);
        acc1 = fma_(a1, x1, acc1);
      }
    }
    for (; p < npack; p += 64) {
      const P a0 = ld_pack_nt(row + p * R);
      const P x0 = ld_pack(x + p * R);
      if constexpr (R == 2) acc0 = fma(a0.x, x0.x, fma(a0.y, x0.y, acc0));
      else acc0 = fma_(a0, x0, acc0);
    }
    if (R == 2 && (ncols & 1) && lane == 0) acc1 = fma_(row[ncols - 1], x[ncols - 1], acc1);
    const T s = wave_sum(add_(acc0, acc1));
    if (lane == 0) st_elem_nt(y + r, s);
  }
}

// gather x[idx[i]] into a contiguous send buffer (halo exchange pack)
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_gather(const T* __restrict__ x, const int32_t* __restrict__ idx, T* __restrict__ out, int64_t cnt,
             const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * kBlock)
    out[i] = x[idx[i]];
}

// ------------------------------------------------------------------------------------------------
// DOTS:  partial[b][c] = sum_{rows of b} conj(V[r,c]) w[r]  (c < j),  partial[b][norm_slot] = sum |w[r]|^2
// (j = columns of this launch; > 40 columns are covered by several launches on column chunks)
// First half of  mul!(h, Vprev', v)  + norm(v)   (src/expansion.jl:81,84 / :93).
//
// NC4 = ceil(j/4): accumulators live in registers with static indexing; the ragged last chunk
// re-reads column j-1 (an L1/L2 hit) instead of branching so all loads of a pack issue back to back.
// `pass` 1: first projection; 2: DGKS correction (skipped unless st->reorth).
// ------------------------------------------------------------------------------------------------
template <class T, int NC4, int U = 1, bool NT = true>
__global__ void __launch_bounds__(kBlock)
    k_dots(const T* __restrict__ V, int64_t ldv, int j, const T* __restrict__ w, T* __restrict__ partial,
           int pnb, int norm_slot, int pass, const DevState* __restrict__ st) {
  if (st) {
    if (st->breakdown >= 0) return;
    if (pass == 2 && !st->reorth) return;
  }
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  constexpr int NC = 4 * NC4;
  T acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = zero_of(T{});
  double nrm = 0.0;
  const double sig = st ? st->sigma : 1.0;  // power of two (1 outside the two-pass expansion): see DevState::sigma

  int64_t pb, pe;
  block_range(ldv / R, blockIdx.x, gridDim.x, pb, pe);
  int64_t p = pb + threadIdx.x;
  // U packs per lane per iteration: U x 4 KiB contiguous per column and workgroup
  for (; p + (int64_t)(U - 1) * kBlock < pe; p += (int64_t)kBlock * U) {
    P wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      wv[u] = scale_pack(ld_pack(w + (p + (int64_t)u * kBlock) * R), sig);
      nrm += nrm2_pack(wv[u]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // only the last 3 columns of the 4-wide template granule can be absent; `c < j` is wave-uniform,
      // so the ragged tail costs a scalar branch and NO extra memory traffic (non-temporal loads of a
      // clamped duplicate column would go back to HBM).
      if (c < NC - 3 || c < j) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const P v = ld_v<NT>(V + (int64_t)c * ldv + (p + (int64_t)u * kBlock) * R);
          dot_acc(acc[c], v, wv[u]);
        }
      }
    }
  }
  for (; p < pe; p += kBlock) {  // remainder, one pack at a time
    const int64_t r = p * R;
    const P wv = scale_pack(ld_pack(w + r), sig);
    nrm += nrm2_pack(wv);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < NC - 3 || c < j) {
        const P v = ld_v<NT>(V + (int64_t)c * ldv + r);
        dot_acc(acc[c], v, wv);
      }
    }
  }

  __shared__ T red[kBlock / 64][NC + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const T s = wave_sum(acc[c]);
    if (lane == 0) red[wave][c] = s;
  }
  {
    const double s = wave_sum(nrm);
    if (lane == 0) {
      if constexpr (sizeof(T) == 8) red[wave][NC] = s; else red[wave][NC] = cd{s, 0.0};
    }
  }
  __syncthreads();
  // thread c < j writes column c of this chunk; thread j writes |w|^2 into `norm_slot` (if any)
  for (int c = threadIdx.x; c <= j; c += kBlock) {
    const int src = c < j ? c : NC;
    if (c == j && norm_slot < 0) continue;
    T s = red[0][src];
#pragma unroll
    for (int wv = 1; wv < kBlock / 64; ++wv) s = add_(s, red[wv][src]);
    partial[(int64_t)(c < j ? c : norm_slot) * pnb + blockIdx.x] = s;  // [column][workgroup]
  }
}

// ------------------------------------------------------------------------------------------------
// FIN_DOTS: h[c] = sum_b partial[c][b], ONE WORKGROUP PER COLUMN c = 0..j (column j = |w|^2), every
// thread a few independent loads + a fixed-shape LDS tree (deterministic).
//   mode 0: reduce + post (single GPU);  mode 1: reduce only -> red[c] (then RCCL all-reduce);
//   mode 2: post only from red (launched with one workgroup per column as well).
// post: pass 1: H[c, j-1] = h[c], coef[c] = h[c], rnorm = sqrt(h[j])      (src/expansion.jl:81,84)
//       pass 2: H[c, j-1] += h[c], coef[c] = h[c]                          (src/expansion.jl:93,95)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_fin_dots(const T* __restrict__ partial, int nb, int pnb, int j, T* __restrict__ red, T* __restrict__ Hcol,
               T* __restrict__ coef, int pass, int mode, DevState* __restrict__ st) {
  if (st->breakdown >= 0) return;
  if (pass == 2 && !st->reorth) return;
  __shared__ T sm[kBlock];
  const int tid = threadIdx.x;
  const int c = blockIdx.x;  // 0..j
  T s = zero_of(T{});
  if (mode != 2) {
    const T* pc = partial + (int64_t)c * pnb;
    for (int b = tid; b < nb; b += kBlock) s = add_(s, pc[b]);
    sm[tid] = s;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
      if (tid < off) sm[tid] = add_(sm[tid], sm[tid + off]);
      __syncthreads();
    }
    s = sm[0];
    if (mode == 1) {
      if (tid == 0) red[c] = s;
      return;
    }
  } else {
    s = red[c];
  }
  if (tid != 0) return;
  if (c < j) {
    coef[c] = s;
    Hcol[c] = (pass == 1) ? s : add_(Hcol[c], s);
  } else if (pass == 1) {
    double v;
    if constexpr (sizeof(T) == 8) v = s; else v = s.x;
    st->rnorm = sqrt(v);
  }
}

// ------------------------------------------------------------------------------------------------
// AXPY:  w -= V[:, 0:j) coef;  partial2[b] = sum |w|^2      (mul!(v, Vprev, h, -1, 1) + norm(v),
// src/expansion.jl:85,88 / :94,96)
// ------------------------------------------------------------------------------------------------
template <class T, int U, int REM>
__device__ __forceinline__ void axpy_tail(typename Pack<T>::type (&s)[U], const T* __restrict__ Vc, int64_t ldv,
                                          const int64_t (&r)[U], const T* g) {
  typename Pack<T>::type tv[REM][U];
#pragma unroll
  for (int t = 0; t < REM; ++t)
#pragma unroll
    for (int u = 0; u < U; ++u) tv[t][u] = ld_pack_nt(Vc + (int64_t)t * ldv + r[u]);
#pragma unroll
  for (int t = 0; t < REM; ++t) {
    const T gc = g[t];
#pragma unroll
    for (int u = 0; u < U; ++u) axpy_acc(s[u], tv[t][u], gc);
  }
}

// U packs (rows r[0..U)) of  w -= V[:, jb:jb+jc) g ; returns sum |w_new|^2 of those packs
template <class T, int U, bool PLAIN = false>
__device__ __forceinline__ double axpy_body(const T* __restrict__ V, int64_t ldv, int jb, int jc, T* __restrict__ w,
                                            const T* g, const int64_t (&r)[U], const T* __restrict__ wsrc) {
  using P = typename Pack<T>::type;
  P s[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = zero_pack(T{});
  int c0 = 0;
  for (; c0 + 4 <= jc; c0 += 4) {  // full groups of 4 columns: 4*U independent loads in flight
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const T gc = g[c0 + t];
      const T* colp = V + (int64_t)(jb + c0 + t) * ldv;
#pragma unroll
      for (int u = 0; u < U; ++u) axpy_acc(s[u], ld_pack_nt(colp + r[u]), gc);
    }
  }
  // ragged tail: 1..3 columns, straight-line code per case so all loads issue together; no duplicate
  // reads (non-temporal loads of a clamped column would go back to HBM).
  switch (jc - c0) {
    case 1: axpy_tail<T, U, 1>(s, V + (int64_t)(jb + c0) * ldv, ldv, r, g + c0); break;
    case 2: axpy_tail<T, U, 2>(s, V + (int64_t)(jb + c0) * ldv, ldv, r, g + c0); break;
    case 3: axpy_tail<T, U, 3>(s, V + (int64_t)(jb + c0) * ldv, ldv, r, g + c0); break;
    default: break;
  }
  double nrm = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    P wv = ld_pack(wsrc + r[u]);
    wv = sub_pack(wv, s[u]);
    if (PLAIN) st_pack(w + r[u], wv); else st_pack_nt(w + r[u], wv);
    nrm += nrm2_pack(wv);
  }
  return nrm;
}

template <class T, int U = 4, bool PLAIN = false>
__global__ void __launch_bounds__(kBlock)
    k_axpy(const T* __restrict__ V, int64_t ldv, int j, T* __restrict__ w, const T* __restrict__ coef,
           double* __restrict__ partial2, int pass, const DevState* __restrict__ st, const T* __restrict__ wsrc0 = nullptr) {
  constexpr int R = Pack<T>::R;
  if (st) {
    if (st->breakdown >= 0) return;
    if (pass == 2 && !st->reorth) {
      if (wsrc0) {  // out-of-place mode: the first projection left w' in the scratch vector -- move it home
        int64_t pb0, pe0;
        block_range(ldv / R, blockIdx.x, gridDim.x, pb0, pe0);
        for (int64_t p = pb0 + threadIdx.x; p < pe0; p += kBlock) st_pack_nt(w + p * R, ld_pack(wsrc0 + p * R));
      }
      return;
    }
  }
  // U packs per lane per iteration: U x 4 KiB contiguous per column and workgroup
  __shared__ T g[128];
  __shared__ double red[kBlock / 64];
  int64_t pb, pe;
  block_range(ldv / R, blockIdx.x, gridDim.x, pb, pe);
  double nrm = 0.0;
  for (int jb = 0; jb < j; jb += 128) {  // chunks of <= 128 columns (maxdim > 128 loops)
    const int jc = (j - jb) < 128 ? (j - jb) : 128;
    __syncthreads();
    if (threadIdx.x < 128) g[threadIdx.x] = threadIdx.x < jc ? coef[jb + threadIdx.x] : zero_of(T{});
    __syncthreads();
    const bool last = jb + 128 >= j;
    int64_t p = pb + threadIdx.x;
    for (; p + (U - 1) * kBlock < pe; p += kBlock * U) {  // all U slots of this lane are in range
      int64_t r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = (p + (int64_t)u * kBlock) * R;
      const double t = axpy_body<T, U, PLAIN>(V, ldv, jb, jc, w, g, r, (wsrc0 && jb == 0) ? wsrc0 : w);
      if (last) nrm += t;
    }
    for (; p < pe; p += kBlock) {  // remainder, one pack at a time
      const int64_t r1[1] = {p * R};
      const double t = axpy_body<T, 1, PLAIN>(V, ldv, jb, jc, w, g, r1, (wsrc0 && jb == 0) ? wsrc0 : w);
      if (last) nrm += t;
    }
  }
  const double s = wave_sum(nrm);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial2[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------------
// AXPY+DOTS: the first DGKS projection fused with the inner products of the second one.
//     w' = w - V[:,0:j) h                       (mul!(v, Vprev, h, -1, 1),  src/expansion.jl:85)
//     partial2[b]  = sum |w'|^2                  (norm(v),                   src/expansion.jl:88)
//     partial[c][b] = sum conj(V[r,c]) w'[r]     (correction = Vprev' * v,   src/expansion.jl:93)
// The correction inner products are row-local once w' is known, so each lane keeps its slice of V in
// registers between the two uses and V is streamed from HBM ONCE for both -- the step reads V three
// times instead of the reference's four whenever the DGKS test asks for the second pass (it always
// does for operators with a dominant diagonal such as the Laplacian).  The correction is speculative:
// k_fin_mid_def decides afterwards whether it is used (src/expansion.jl:91).
//
// Column-split form: the j columns are dealt round-robin to the 4 waves of the workgroup (wave q owns
// columns q, q+4, ...), all waves walk the SAME rows.  Each lane therefore keeps only NCW = ceil(j/4)
// column slices (16-byte packs: 2 rows of a Float64 column or 1 row of a ComplexF64 column) instead of j,
// which lifts occupancy from 2 to 4-6 waves per SIMD.  The per-row partial sums of the projection are
// exchanged through a double-buffered 8 KiB LDS array (one barrier per iteration); the second-pass inner
// products then need no cross-wave reduction at all, because every column belongs to exactly one wave.
// Summation order is fixed (wave 0..3), so results are run-to-run deterministic.
// (A software-pipelined form -- two register sets, loads of tile t+1 issued before the barrier of tile t --
// measured the same 5.2 TB/s: the kernel is not stalled on its barrier.)
// The updated rows are staged in LDS and written back every WB tiles by all four waves (WB x U x 1 KiB bursts
// instead of one wave's U KiB per tile).  NCW <= 10 runs U = 4 packs per lane, NCW 11..16 (40 < j <= 64) U = 2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 addp(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd addp(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
// acc += sum over the pack of conj(v) * w, in the association the Float64 kernel has always used
__device__ __forceinline__ void dotp(double& acc, double2 v, double2 w) { acc = fma(v.x, w.x, fma(v.y, w.y, acc)); }
__device__ __forceinline__ void dotp(cd& acc, cd v, cd w) { dot_acc(acc, v, w); }
__device__ __forceinline__ double real_of(double a) { return a; }
__device__ __forceinline__ double real_of(cd a) { return a.x; }
__device__ __forceinline__ double scl(double a, double s) { return a * s; }
__device__ __forceinline__ cd scl(cd a, double s) { return cd{a.x * s, a.y * s}; }
__device__ __forceinline__ double from_real(double v, double) { return v; }
__device__ __forceinline__ cd from_real(double v, cd) { return cd{v, 0.0}; }

template <class T, int NCW, int U, int WB, bool NT = true>
__global__ void __launch_bounds__(kBlock)
    k_axpy_dots_cs(const T* __restrict__ V, int64_t ldv, int j, T* __restrict__ w, const T* __restrict__ coef,
                   T* __restrict__ partial, int pnb, double* __restrict__ partial2, const DevState* __restrict__ st,
                   int defer, T* __restrict__ wdst0 = nullptr, int plain_store = 0) {
  if (st && st->breakdown >= 0) return;
  T* __restrict__ wdst = wdst0 ? wdst0 : w;
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  // lazy normalisation: y (= w on entry) = A * (unnormalised column j-1) carries the factor beta_{j-1}
  const double invb = (defer && st) ? st->invb : 1.0;
  __shared__ P tbuf[2][4][U][64];
  __shared__ P wout[WB > 1 ? WB * U * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const T* colp[NCW];
  T g[NCW];
  T acc[NCW];
#pragma unroll
  for (int i = 0; i < NCW; ++i) {
    const int c = wave + 4 * i;
    const bool valid = c < j;
    colp[i] = V + (int64_t)(valid ? c : j - 1) * ldv;
    g[i] = valid ? coef[c] : zero_of(T{});
    acc[i] = zero_of(T{});
  }
  const bool last_valid = (wave + 4 * (NCW - 1)) < j;
  double nrm = 0.0;
  int64_t pb, pe;
  block_range(ldv / R, blockIdx.x, gridDim.x, pb, pe);
  int it = 0;
  for (int64_t base = pb; base < pe; base += 64 * U, ++it) {
    int64_t r[U];
    bool ok[U];
    P v[NCW][U];
    P wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q = base + u * 64 + lane;
      ok[u] = q < pe;
      r[u] = (ok[u] ? q : pb) * R;
    }
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      if (i < NCW - 1 || last_valid) {  // wave-uniform: only this wave's last column can be absent
#pragma unroll
        for (int u = 0; u < U; ++u) v[i][u] = ld_v<NT>(colp[i] + r[u]);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) v[i][u] = zero_pack(T{});
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) wv[u] = scale_pack(ld_pack(w + r[u]), invb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      P t = zero_pack(T{});
#pragma unroll
      for (int i = 0; i < NCW; ++i) axpy_acc(t, v[i][u], g[i]);
      tbuf[it & 1][wave][u][lane] = t;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const P t0 = tbuf[it & 1][0][u][lane], t1 = tbuf[it & 1][1][u][lane];
      const P t2 = tbuf[it & 1][2][u][lane], t3 = tbuf[it & 1][3][u][lane];
      P wn = sub_pack(wv[u], addp(addp(t0, t1), addp(t2, t3)));
      if (!ok[u]) wn = zero_pack(T{});
      if (wave == 0 && ok[u]) {
        if constexpr (WB > 1) wout[((it % WB) * U + u) * 64 + lane] = wn;
        else st_pack_nt(wdst + r[u], wn);
        nrm += nrm2_pack(wn);
      }
#pragma unroll
      for (int i = 0; i < NCW; ++i) dotp(acc[i], v[i][u], wn);
    }
    if constexpr (WB > 1) {
      const bool lastt = base + 64 * U >= pe;
      if ((it % WB) == WB - 1 || lastt) {  // workgroup-uniform
        __syncthreads();
        const int64_t fb = base - (int64_t)(it % WB) * 64 * U;  // first pack staged
        const int64_t fe = (base + 64 * U < pe) ? base + 64 * U : pe;
        // plain_store: the new column is the NEXT kernel's gather source (two-pass expansion: the operator is applied to it
        // right away) -- a cacheable store leaves it in the memory-side cache instead of streaming it past
        if (plain_store == 1) for (int64_t o = fb + threadIdx.x; o < fe; o += kBlock) st_pack(wdst + o * R, wout[o - fb]);
        else if (plain_store == 0) for (int64_t o = fb + threadIdx.x; o < fe; o += kBlock) st_pack_nt(wdst + o * R, wout[o - fb]);
        else if (plain_store == 3) for (int64_t o = fb + threadIdx.x; o < fe; o += kBlock) st_pack_nt(wdst + (o & 4095) * R, wout[o - fb]);
        // (plain_store == 2: no store at all, 3: every store into one 64 KiB window -- tools/fused_probe.hip measures what
        // the write stream costs and whether it is the memory or the issue path that pays)
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NCW; ++i) {
    const T sred = wave_sum(acc[i]);
    const int c = wave + 4 * i;
    if (lane == 0 && c < j) partial[(int64_t)c * pnb + blockIdx.x] = sred;
  }
  if (wave == 0) {
    const double sred = wave_sum(nrm);
    if (lane == 0) partial2[blockIdx.x] = sred;
  }
}

// Sum of nb values by one 256-thread workgroup: strided partial sums, a wave-level butterfly, then the four wave totals
// through LDS -- two barriers instead of the nine of a plain LDS tree (these kernels are a few microseconds long and sit on
// the critical path of every step: at n = 1e6 they are a tenth of it).  Fixed shape -> deterministic.
template <class T>
__device__ __forceinline__ T block_sum(const T* __restrict__ p, int nb, T* sm) {
  const int tid = threadIdx.x;
  T s = zero_of(T{});
  for (int b = tid; b < nb; b += kBlock) s = add_(s, p[b]);
  s = wave_sum(s);
  if ((tid & 63) == 0) sm[tid >> 6] = s;
  __syncthreads();
  const T r = add_(add_(sm[0], sm[1]), add_(sm[2], sm[3]));
  __syncthreads();
  return r;
}

// peer-to-peer exchange of one element of type T plus one real: the sums over ranks of (x, extra).
// Elements of the LL window: NV = sizeof(T)/8 + 1 consecutive doubles per workgroup (ks_p2p.hpp).
template <class T> struct P2pNV { static constexpr int value = (int)(sizeof(T) / 8) + 1; };
__device__ __forceinline__ void p2p_pair(const P2pDev& p, int c, double x, double extra, double& gx, double& gextra) {
  const double in[2] = {x, extra};
  double out[2];
  p2p_sum_wave<2>(p, 2 * c, in, out);
  gx = out[0];
  gextra = out[1];
}
__device__ __forceinline__ void p2p_pair(const P2pDev& p, int c, cd x, double extra, cd& gx, double& gextra) {
  const double in[3] = {x.x, x.y, extra};
  double out[3];
  p2p_sum_wave<3>(p, 3 * c, in, out);
  gx = cd{out[0], out[1]};
  gextra = out[2];
}

// ------------------------------------------------------------------------------------------------
// Lazy normalisation (fused path, Float64 and ComplexF64, maxdim <= 64).  The column a step produces is NOT
// normalised by a streaming pass (v ./= wnorm, src/expansion.jl:106): it stays in HBM as w~ = beta * v together
// with a per-column REAL factor cs[c] = 1/beta (device array `colscale`, 1 for ordinary columns), and every
// consumer folds the factor into the small quantities instead of touching n-sized data:
//     y = A w~                      carries beta_{j-1}:      y/beta is formed on the fly in k_axpy_dots_cs
//     h[c]    = cs[c] * (w~_c^H y) / beta_{j-1}              (k_fin_dots_def)
//     w -= sum_c w~_c * (cs[c] h[c])                          (coefficient vector handed to the kernels)
//     V Q  ->  rows of Q scaled by cs[c] on the host          (restart rotation)
// The norm of the newest column is not even reduced by its own stage: its block partials are folded into
// the NEXT step's first reduction (multi-GPU: one all-reduce and two launches fewer per step).  The
// breakdown test of the previous step (src/expansion.jl:99) is therefore evaluated one step late, which
// the host cannot observe: it only sees the state after the batch.  Columns are materialised (scaled
// once) before anything outside the expansion/rotation pair reads them.
// FIN_DOTS_DEF: workgroup c <= j: column c of the partials (column j = |y|^2); in reduce-only mode an extra
// workgroup j+1 delivers the pending norm so that ONE all-reduce of j+2 elements serves everything.
//   mode 0: reduce + post (single GPU);  mode 1: reduce only -> red[c] (then all-reduce over the ranks);
//   mode 2: post only from red;  mode 3: peer-to-peer -- reduce, exchange and post in ONE kernel.
//
// Entry test.  Workgroup j of THIS launch may set st->breakdown = j-1 (pending breakdown of the previous step)
// while other workgroups of the same launch have not started yet; in mode 3 a workgroup that then left early
// would withhold its contribution from the peers, which spin for it until the transport times out.  A value
// written by an EARLIER launch is always < j-1 in this batch, so "breakdown >= 0 and != j-1" is exactly
// "a previous launch broke down" and only that ends the workgroup before the exchange.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_fin_dots_def(const T* __restrict__ partial, int nb, int pnb, const double* __restrict__ partial2, int nb2,
                   int j, T* __restrict__ red, T* __restrict__ Hcol, T* __restrict__ Hsub_prev,
                   T* __restrict__ coef, double* __restrict__ colscale, int mode, DevState* __restrict__ st,
                   P2pDev p2p) {
  {
    const int bd = st->breakdown;
    if (bd >= 0 && bd != j - 1) return;
  }
  __shared__ T sm[kBlock];
  double* smd = reinterpret_cast<double*>(sm);
  const int c = blockIdx.x;  // 0..j (mode 1: ..j+1)
  const bool pend = st->pend != 0;
  const bool pre = st->pend_reorth != 0;
  T s = zero_of(T{});
  double b2 = 1.0;
  if (mode == 3) {
    // workgroup c sums its column over the ranks itself (its column and the pending norm every workgroup needs)
    const T own = block_sum(partial + (int64_t)c * pnb, nb, sm);
    const double pn = (pend && pre) ? block_sum(partial2, nb2, smd) : 0.0;
    if (threadIdx.x >= 64) return;
    double gp;
    p2p_pair(p2p, c, own, pn, s, gp);
    if (pend) b2 = pre ? gp : st->wnorm * st->wnorm;
  } else if (mode != 2) {
    if (c <= j) s = block_sum(partial + (int64_t)c * pnb, nb, sm);
    if (mode == 1) {
      if (c == j + 1) s = from_real((pend && pre) ? block_sum(partial2, nb2, smd) : 0.0, T{});
      if (threadIdx.x == 0) red[c] = s;
      return;
    }
    if (pend) b2 = pre ? block_sum(partial2, nb2, smd) : st->wnorm * st->wnorm;
  } else {
    s = red[c];
    if (pend) b2 = pre ? real_of(red[j + 1]) : st->wnorm * st->wnorm;
  }
  if (threadIdx.x != 0) return;
  // factor of the input column: settled earlier (1 if the column is normalised) or established right now
  double invb = pend ? 0.0 : colscale[j - 1];
  if (pend) {
    const double beta = pre ? sqrt(b2) : st->wnorm;
    if (beta <= kEta * st->rnorm_p) {  // breakdown of the PREVIOUS step, src/expansion.jl:99-102
      if (c == j) {
        *Hsub_prev = zero_of(T{});
        st->breakdown = j - 1;
        st->inv_norm = 0.0;
      }
      return;
    }
    invb = 1.0 / beta;
    if (c == j) {
      *Hsub_prev = from_real(beta, T{});  // H[j, j-1] of the previous step, src/expansion.jl:105
      st->n_steps += 1;
    }
  }
  if (c < j) {
    const double cs = (c == j - 1) ? invb : colscale[c];
    const T h = scl(scl(s, invb), cs);  // true coefficient w.r.t. the NORMALISED column c
    Hcol[c] = h;
    coef[c] = scl(h, cs);               // what multiplies the stored (possibly unnormalised) column
  } else {
    st->rnorm = sqrt(real_of(s)) * invb;
    st->invb = invb;
    if (pend) colscale[j - 1] = invb;  // only workgroup j writes; the others derived the same value themselves
  }
}

// ------------------------------------------------------------------------------------------------
// FIN_MID_DEF: after k_axpy_dots_cs both ||w'||^2 and the speculative second-pass inner products c are
// known, so ONE reduction stage (and, multi-GPU, ONE all-reduce of j+1 elements) serves the DGKS test and the
// correction:   workgroup c < j reduces column c, workgroup j reduces |w'|^2.
//   wnorm = sqrt(sum |w'|^2);   wnorm < eta * rnorm  ->  second pass: H[0:j, j-1] += c, coef = c (src/expansion.jl:91-95)
// When no second pass is needed it does not finalise the step (no H[j+1,j], no 1/wnorm): it records what the
// next k_fin_dots_def / k_fin_pend needs.  Every workgroup evaluates the (identical) test itself from read-only
// inputs; only workgroup j writes state.  (This kernel never writes st->breakdown, so its entry test is safe.)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_fin_mid_def(const T* __restrict__ partial, const double* __restrict__ partial2, int nb, int pnb, int j,
                  T* __restrict__ red, T* __restrict__ Hcol, T* __restrict__ coef,
                  const double* __restrict__ colscale, int mode, DevState* __restrict__ st, P2pDev p2p) {
  if (st->breakdown >= 0) return;
  __shared__ T sm[kBlock];
  double* smd = reinterpret_cast<double*>(sm);
  const int c = blockIdx.x;  // 0..j
  T s = zero_of(T{});
  double nrm2 = 0.0;
  if (mode == 3) {  // peer-to-peer: see k_fin_dots_def
    const double n2 = block_sum(partial2, nb, smd);
    const T own = (c < j) ? block_sum(partial + (int64_t)c * pnb, nb, sm) : from_real(n2, T{});
    if (threadIdx.x >= 64) return;
    p2p_pair(p2p, c, own, n2, s, nrm2);
  } else if (mode != 2) {
    const double n2 = (c == j || mode == 0) ? block_sum(partial2, nb, smd) : 0.0;
    s = (c < j) ? block_sum(partial + (int64_t)c * pnb, nb, sm) : from_real(n2, T{});
    if (mode == 1) {
      if (threadIdx.x == 0) red[c] = s;
      return;
    }
    nrm2 = n2;
  } else {
    s = red[c];
    nrm2 = real_of(red[j]);
  }
  if (threadIdx.x != 0) return;
  const double wnorm = sqrt(nrm2);
  const double rnorm = st->rnorm;
  const bool reorth = wnorm < kEta * rnorm;  // src/expansion.jl:91
  if (c < j) {
    if (reorth) {
      const double cs = colscale[c];
      Hcol[c] = add_(Hcol[c], scl(s, cs));  // h .+= correction, :95
      coef[c] = scl(scl(s, cs), cs);
    }
    return;
  }
  st->wnorm = wnorm;
  st->pend = 1;
  if (reorth) {
    st->reorth = 1;
    st->pend_reorth = 1;
    st->rnorm_p = wnorm;  // :92  rnorm <- wnorm; the final norm comes out of the second-pass update
    st->n_reorth += 1;
  } else {
    st->reorth = 0;
    st->pend_reorth = 0;
    st->rnorm_p = rnorm;
  }
}

// FIN_PEND (end of a batch, ONE workgroup): settle the pending normalisation of the last column: beta, breakdown
// test, H[j+1,j] and the column's factor.
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_fin_pend(const double* __restrict__ partial2, int nb2, double* __restrict__ red, T* __restrict__ Hsub, int j,
               double* __restrict__ colscale, int mode, DevState* __restrict__ st, P2pDev p2p) {
  if (st->breakdown >= 0) return;
  if (!st->pend) return;
  __shared__ double sm[kBlock];
  const bool pre = st->pend_reorth != 0;
  double b2 = 0.0;
  if (mode == 3) {  // peer-to-peer: see k_fin_dots_def
    const double x[1] = {pre ? block_sum(partial2, nb2, sm) : 0.0};
    if (threadIdx.x >= 64) return;
    double g[1];
    p2p_sum_wave<1>(p2p, 0, x, g);
    b2 = g[0];
  } else if (mode != 2) {
    b2 = pre ? block_sum(partial2, nb2, sm) : 0.0;
    if (mode == 1) {
      if (threadIdx.x == 0) red[0] = b2;
      return;
    }
  } else {
    b2 = red[0];
  }
  if (threadIdx.x != 0) return;
  const double beta = pre ? sqrt(b2) : st->wnorm;
  st->pend = 0;
  st->reorth = 0;
  if (beta <= kEta * st->rnorm_p) {
    *Hsub = zero_of(T{});
    st->breakdown = j;
    st->inv_norm = 0.0;
  } else {
    *Hsub = from_real(beta, T{});
    st->inv_norm = 1.0 / beta;
    colscale[j] = 1.0 / beta;  // the column stays unnormalised in HBM
    st->wnorm = beta;
    st->n_steps += 1;
  }
}

// ------------------------------------------------------------------------------------------------
// IMPLICIT SECOND PASS (two reads of the basis per step instead of three; Float64 and ComplexF64, maxdim <= 64).
//
// The DGKS second pass  v <- v - V c,  c = V^H v  (src/expansion.jl:93-94) is never applied to the n-vector.  The column
// stays in HBM as the FIRST projection w' and the basis is carried in factored form
//         V_true = S * T          S: the stored columns,  T: (maxdim+1)^2 upper triangular, device resident,
// column j of T = ( -(T c)/beta ; 1/beta ) with c the second-pass coefficients (0 when the DGKS test did not ask for
// the pass) and beta = ||w' - V c|| = sqrt(||w'||^2 - ||c||^2)  (V orthonormal).  Everything the expansion needs from
// V_true is a small triangular transform of what the two streaming kernels deliver for S:
//     y' = A S[:,j-1]                          the operator is applied to the STORED column; by linearity and the Arnoldi
//                                              relation  A v_true = (y' - V_true g) / beta,   g = H[0:j, 0:j-1] c
//     s = S^H y'  (k_dots)                     t = T^H s = V_true^H y';   h = (t - g) / beta          -> H[0:j, j-1]
//                                              ||A v_true||^2 = (|y'|^2 - 2 Re g^H t + |g|^2) / beta^2  -> rnorm (:81)
//     w' = y'/beta - S (T t / beta)            (k_axpy_dots_cs, unchanged: coefficient vector + the factor 1/beta)
//     c_raw = S^H w',  ||w'||^2                c = T^H c_raw;  DGKS test (:91), h += c (:95), beta, breakdown test (:99),
//                                              H[j, j-1] = beta (:105), new column of T, g for the next step
// The restart rotation folds T into Q (V_true Q = S (T Q)), after which every column is an ordinary one again.  Same
// decisions, same quantities as the reference up to rounding: validated against the oracle (same restart trail, products
// and residuals, Ritz values to 1e-13).  Columns < ntrue are ordinary (T = identity there, never read).
//
// FIN_STEP_T (below): workgroup c reduces one column of the partial sums (the last one of each set the squared norm),
// publishes it in red[], and the LAST workgroup to arrive (device-scope counter) does the small algebra with 256 threads.
//   mode 0: single GPU;  mode 1: reduce only -> red (then all-reduce);  mode 2: algebra only from red (one workgroup);
//   mode 3: peer-to-peer -- every workgroup exchanges its own element, the last one does the algebra.
// ------------------------------------------------------------------------------------------------
constexpr int kTMax = 65;  // kFusedMaxJ + 1
constexpr int kTLdsBytes = 24576;  // LDS staging area for the non-trivial columns of T
constexpr int kHLdsBytes = 24576;  // ... and for the block of H that g = H c reads
__device__ __forceinline__ double conj_(double a) { return a; }
__device__ __forceinline__ cd conj_(cd a) { return cd{a.x, -a.y}; }
__device__ __forceinline__ double sub_(double a, double b) { return a - b; }
__device__ __forceinline__ cd sub_(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ double abs2_(double a) { return a * a; }
__device__ __forceinline__ double abs2_(cd a) { return fma(a.x, a.x, a.y * a.y); }
__device__ __forceinline__ double neg_(double a) { return -a; }
__device__ __forceinline__ cd neg_(cd a) { return cd{-a.x, -a.y}; }
// Re(conj(a) * b)
__device__ __forceinline__ double redot_(double a, double b) { return a * b; }
__device__ __forceinline__ double redot_(cd a, cd b) { return fma(a.x, b.x, a.y * b.y); }
// loads that must see what OTHER workgroups of the same launch wrote (bypass the per-CU vector cache)
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ cd ld_agent(const cd* p) {
  const double* q = reinterpret_cast<const double*>(p);
  return cd{ld_agent(q), ld_agent(q + 1)};
}

// ... and the store that goes with them: written THROUGH to where the other workgroups' agent-scope loads read, complete once
// the wave's vmcnt reaches zero.  (A plain store + __threadfence() makes the same promise by writing back EVERY dirty line of
// this XCD's L2 -- behind a kernel that streamed gigabytes through it that is 4 - 6 us: profiles/r06_fin_blk_timing.txt.)
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(cd* p, cd v) {
  double* q = reinterpret_cast<double*>(p);
  st_agent(q, v.x);
  st_agent(q + 1, v.y);
}

// FIN_STEP_T: ONE reduction kernel per Arnoldi step.  The second reduction of step jm (c_raw, ||w'||^2 from the projection
// kernel) is not needed before the NEXT step's first reduction (s = S^H y', |y'|^2 from k_dots): the operator and k_dots of
// step jm+1 work on stored columns only.  So both are reduced by the same launch -- 2j+3 workgroups, one election, one
// exchange in the multi-GPU modes -- and the last workgroup runs the algebra of both in order:
//   MID(jm):  c = T^H c_raw; DGKS test; h += c; beta; breakdown test; H[jm, jm-1] = beta; column jm of T; g
//   DOTS(jd): t = T^H s; h = (t - g)/beta -> H[0:jd, jd-1]; rnorm; coefficient vector T t / beta       (jd = jm + 1)
// jm = 0: first step of a batch (nothing pending);  jd = 0: after the last step (settle it).  The breakdown of step jm
// is thus detected after the operator and k_dots of step jm+1 already ran (on a column that is then discarded).
// red[0 .. jm] = MID sums (last = ||w'||^2), red[nm .. nm+jd] = DOTS sums (last = |y'|^2), nm = jm ? jm+1 : 0.
// sum_{k < n} conj(a[k]) x[k]  and  sum_{k0 <= k < n} a[k * stride] x[k]: loads of four terms issued together, two
// accumulators (a serial fma chain over un-batched LDS loads cost 90 cycles per term: 6 us for the algebra of one step)
template <class T> __device__ __forceinline__ T dotc_contig(const T* __restrict__ a, const T* __restrict__ x, int n) {
  T s0 = zero_of(T{}), s1 = zero_of(T{});
  int k = 0;
  for (; k + 3 < n; k += 4) {
    const T a0 = a[k], a1 = a[k + 1], a2 = a[k + 2], a3 = a[k + 3];
    const T x0 = x[k], x1 = x[k + 1], x2 = x[k + 2], x3 = x[k + 3];
    s0 = fma_(conj_(a0), x0, s0);
    s1 = fma_(conj_(a1), x1, s1);
    s0 = fma_(conj_(a2), x2, s0);
    s1 = fma_(conj_(a3), x3, s1);
  }
  for (; k < n; ++k) s0 = fma_(conj_(a[k]), x[k], s0);
  return add_(s0, s1);
}
template <class T> __device__ __forceinline__ T dot_strided(const T* __restrict__ a, int64_t stride, const T* __restrict__ x, int k0, int n) {
  T s0 = zero_of(T{}), s1 = zero_of(T{});
  int k = k0;
  for (; k + 3 < n; k += 4) {
    const T a0 = a[k * stride], a1 = a[(k + 1) * stride], a2 = a[(k + 2) * stride], a3 = a[(k + 3) * stride];
    const T x0 = x[k], x1 = x[k + 1], x2 = x[k + 2], x3 = x[k + 3];
    s0 = fma_(a0, x0, s0);
    s1 = fma_(a1, x1, s1);
    s0 = fma_(a2, x2, s0);
    s1 = fma_(a3, x3, s1);
  }
  for (; k < n; ++k) s0 = fma_(a[k * stride], x[k], s0);
  return add_(s0, s1);
}

template <class T>
__global__ void __launch_bounds__(kBlock)
    k_fin_step_t(const T* __restrict__ part_s, int nb_s, const T* __restrict__ part_c, const double* __restrict__ partial2,
                 int nb_c, int pnb, int jm, int jd, T* __restrict__ red, T* __restrict__ Hd, int ldh, T* __restrict__ Tm,
                 int ldt, int ntrue, T* __restrict__ gvec, T* __restrict__ coef, int mode, DevState* __restrict__ st,
                 P2pDev p2p, unsigned* __restrict__ counter) {
  if (st->breakdown >= 0) return;
#ifdef KS_FIN_TIMING
  const long long tk0 = wall_clock64();
  long long tk1 = 0, tk2 = 0, tk3 = 0;
#endif
  __shared__ T sm[kBlock];
  __shared__ T r_s[2 * kTMax + 2];          // every reduced value: [MID sums | DOTS sums]
  __shared__ T c_s[kTMax], t_s[kTMax];
  __shared__ int last_wg;
  __shared__ __attribute__((aligned(16))) unsigned char tl_raw[kTLdsBytes];
  __shared__ __attribute__((aligned(16))) unsigned char hl_raw[kHLdsBytes];
  T* Tl = reinterpret_cast<T*>(tl_raw);
  T* Hl = reinterpret_cast<T*>(hl_raw);
  const int tid = threadIdx.x;
  const int nm = jm ? jm + 1 : 0, nd = jd ? jd + 1 : 0;
  const int jt = jd ? jd : jm;       // T[0:jt, 0:jt) is what the algebra touches (column jm is produced here when jd = jm+1)
  // ---- prefetch INTO REGISTERS (every workgroup; only the last one to arrive uses it, but nobody knows who that is, and
  // this way the loads are in flight together with those of the reduction instead of adding round trips after it) ----
  constexpr int PF = (int)(kTLdsBytes / sizeof(T)) / kBlock;  // elements per thread: 12 (Float64) / 6 (ComplexF64)
  const int nl = jt - ntrue;         // columns ntrue..jt-1 of T are not unit vectors
  const bool use_lds = nl > 0 && nl * jt <= (int)(kTLdsBytes / sizeof(T));
  const int have = jm ? jm : jt;     // columns < have exist in memory already
  const int hr = jm + 1, hc = jm - 1;
  const bool h_lds = jm && hc > 0 && hr * hc <= (int)(kHLdsBytes / sizeof(T));
  const int nT = use_lds ? nl * jt : 0, nH = h_lds ? hr * hc : 0;
  T treg[PF], hreg[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int e = tid + u * kBlock;
    treg[u] = zero_of(T{});
    hreg[u] = zero_of(T{});
    if (e < nT) {
      const int k = e % jt, i = ntrue + e / jt;
      if (k <= i && i < have) treg[u] = Tm[k + (int64_t)i * ldt];
    }
    if (e < nH) hreg[u] = Hd[(e % hr) + (int64_t)(e / hr) * ldh];
  }
  T h0 = zero_of(T{});
  if (jm && (tid & 63) < jm) h0 = Hd[(tid & 63) + (int64_t)(jm - 1) * ldh];  // h of step jm as its DOTS half left it (one copy per wave)
  T gi = zero_of(T{});
  if (!jm && jd && (jd - 1) >= ntrue && (tid & 63) < jd) gi = gvec[tid & 63];  // (only when a batch continues on factored columns)
  const double rnorm = st->rnorm, rnorm2 = st->rnorm2, mr = st->max_ratio;
  const double sig = st->sigma;  // what k_dots scaled y' by (set by the PREVIOUS launch of this kernel)
  // ---- reduction of this workgroup's column + election ----
  if (mode != 2) {
    const int c = blockIdx.x;
    T s;
    if (c < nm) {
      if (c == jm) s = from_real(block_sum(partial2, nb_c, reinterpret_cast<double*>(sm)), T{});
      else s = block_sum(part_c + (int64_t)c * pnb, nb_c, sm);
    } else {
      s = block_sum(part_s + (int64_t)(c - nm) * pnb, nb_s, sm);
    }
    if (mode == 3 && tid < 64) {
      T g;
      double dummy;
      p2p_pair(p2p, c, s, 0.0, g, dummy);
      s = g;
    }
#ifdef KS_FIN_TIMING
    tk1 = wall_clock64();
#endif
    if (tid == 0) {
      if (mode != 1) {
        st_agent(red + c, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the sum has landed where the elected workgroup reads it: no L2 write-back -- st_agent)
        last_wg = (atomicAdd(counter, 1u) == gridDim.x - 1) ? 1 : 0;
      } else {
        red[c] = s;
      }
    }
    if (mode == 1) return;
    __syncthreads();
    if (!last_wg) return;
    if (tid == 0) *counter = 0u;  // armed for the next launch (stream order)
#ifdef KS_FIN_TIMING
    tk2 = wall_clock64();
#endif
  }
  // ---- the last workgroup: ONE round trip for every reduced value, the prefetched blocks go to LDS, then ALL FOUR waves do
  // the algebra: every vector has <= 64 elements (lane li = tid & 63 owns element li in every wave) and every small
  // matrix-vector product is split over the waves by SUMMATION RANGE (wave q sums the q-th quarter of the terms of
  // element li, the four partial sums meet in LDS in fixed order -> deterministic).  One wave alone spent ~6 us at j = 36
  // in four dependent products of up to j terms each (LDS round trips in a serial chain); a quarter of the terms per
  // wave and two barriers per product is ~2 us.  Scalars (norms, the DGKS decisions) are formed redundantly and
  // identically in every wave, so control flow stays uniform; only wave 0 writes results. ----
  if (tid < nm + nd) r_s[tid] = ld_agent(red + tid);
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int e = tid + u * kBlock;
    if (e < nT) Tl[e] = treg[u];
    if (e < nH) Hl[e] = hreg[u];
  }
  __syncthreads();
  const int li = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ T ps[4][64];
  auto combine = [&](T part) -> T {
    ps[q][li] = part;
    __syncthreads();
    const T tot = add_(add_(ps[0][li], ps[1][li]), add_(ps[2][li], ps[3][li]));
    __syncthreads();
    return tot;
  };
  // this wave's quarter [a, b) of the summation range [k0, n) (multiples of four terms: dot_* batch their loads by four)
  auto quarter = [&](int k0, int n, int& a, int& b) {
    const int len = n > k0 ? n - k0 : 0;
    const int per = ((len + 15) >> 4) << 2;
    a = k0 + q * per;
    b = a + per;
    if (a > n) a = n;
    if (b > n) b = n;
  };
  // T as the algebra sees it: element (k, i), i >= ntrue, at tb[k + (i - ntrue) * tld]
  const T* tb = use_lds ? Tl : Tm + (int64_t)ntrue * ldt;
  const int64_t tld = use_lds ? jt : ldt;
  // sum_{k in this wave's quarter of [0, i]} conj(T[k, i]) x[k]   (i >= ntrue)
  auto tcol_dot = [&](const T* x, int i) -> T {
    int a, b;
    quarter(0, i + 1, a, b);
    return dotc_contig(tb + (i - ntrue) * tld + a, x + a, b - a);
  };
  // sum_{k in this wave's quarter of [max(i, ntrue), n)} T[i, k] x[k]
  auto trow_dot = [&](const T* x, int i, int n) -> T {
    int a, b;
    quarter(i > ntrue ? i : ntrue, n, a, b);
    return dot_strided(tb + i - ntrue * tld, tld, x, a, b);
  };
  double binv_in = 1.0;  // 1 / beta of the column the DOTS half works on (column jd-1)
  // ================================ MID(jm) ================================
  if (jm) {
    const int j = jm;
    T* Hcol = Hd + (int64_t)(j - 1) * ldh;
    const T* a_s = r_s;  // c_raw[0..j-1], ||w'||^2 at [j]
    T ci;
    {  // c = T^H c_raw = V_true^H w'
      const T part = (li < j && li >= ntrue) ? tcol_dot(a_s, li) : zero_of(T{});
      ci = combine(part);
      if (li < j && li < ntrue) ci = a_s[li];
      if (li >= j) ci = zero_of(T{});
    }
    const double wn2 = real_of(a_s[j]);
    const double wnorm = sqrt(wn2);
    // DGKS test, src/expansion.jl:91 -- against the larger of ||A v_true|| (the reference's rnorm) and ||y'|| / beta, the
    // norm of the vector the projection was really applied to.  The two differ only when the input column carries a
    // second-pass correction that is NOT small against it AND A v_true nearly cancels (a basis vector in the null space
    // of A right after a near-breakdown: test/partial_schur.jl:6-27); there the first projection loses as many digits as
    // y' - V g did, and the (implicit) second pass is what restores orthogonality.  Everywhere else the maximum is rnorm.
    // (The maximum can only make the test fire MORE often than the reference's, never less.)
    const bool reorth = wnorm < kEta * fmax(rnorm, rnorm2);
    const double c2 = wave_sum((reorth && li < j) ? abs2_(ci) : 0.0);
    double beta, rnorm_p;
    if (reorth) {
      const double b2 = wn2 - c2;  // ||w' - V c||^2 with V orthonormal
      beta = sqrt(b2 > 0.0 ? b2 : 0.0);
      rnorm_p = wnorm;             // :92
    } else {
      beta = wnorm;
      rnorm_p = rnorm;
      ci = zero_of(T{});
    }
    // A correction that is not small against what remains of the vector is not carried implicitly: g = H c leans on the
    // Arnoldi relation of the earlier columns, which locked columns satisfy only to tol * |lambda| (src/run.jl:360), so
    // its error is bounded by (residual of the relation) * ||c|| / beta.  Rounding-level corrections (every step of an
    // operator with a dominant diagonal) stay implicit; a genuine one (near-breakdown) goes back to the host, which
    // redoes this step with the second projection applied to the vector.  Checked before the breakdown test: the explicit
    // form takes that decision itself.
    if (reorth && mr > 0.0 && c2 > mr * mr * beta * beta) {
      if (tid == 0) {
        st->breakdown = j;  // every later kernel of the batch exits at once
        st->bail = j;
      }
      return;
    }
    if (beta <= kEta * rnorm_p) {  // src/expansion.jl:99-102
      if (tid == 0) {
        Hcol[j] = zero_of(T{});
        st->breakdown = j;
        st->inv_norm = 0.0;
        if (reorth) st->n_reorth += 1;
      }
      if (reorth && q == 0 && li < j) Hcol[li] = add_(h0, ci);  // h .+= correction happens before the test, :95
      return;
    }
    const double binv = 1.0 / beta;
    T hi = h0;
    if (li < j && reorth) {
      hi = add_(hi, ci);  // :95
      if (q == 0) Hcol[li] = hi;
    }
    if (q == 0 && li < j) c_s[li] = ci;
    __syncthreads();
    {  // new column of T:  -(T c) / beta
      const T part = (reorth && li < j) ? trow_dot(c_s, li, j) : zero_of(T{});
      T a = combine(part);
      if (reorth && li < ntrue && li < j) a = add_(a, c_s[li]);
      const T tv = scl(neg_(a), binv);
      if (q == 0 && li < j) {
        Tm[li + (int64_t)j * ldt] = tv;
        if (use_lds && jd) Tl[li + (j - ntrue) * jt] = tv;
      }
    }
    {  // g for the next step: H[0:j+1, 0:j] c   (upper Hessenberg: H[i,k] = 0 for k < i-1); entries 0..j-1 here, entry j below
      T part = zero_of(T{});
      if (reorth && li < j) {
        const T* __restrict__ hb = h_lds ? Hl : Hd;
        int a, b;
        quarter(li > 0 ? li - 1 : 0, j - 1, a, b);
        part = dot_strided(hb + li, h_lds ? (int64_t)hr : (int64_t)ldh, c_s, a, b);
      }
      T g = combine(part);
      if (reorth && li < j) g = fma_(hi, c_s[j - 1], g);  // column j-1 of H as this step leaves it
      if (q == 0 && li < j) gvec[li] = g;
      gi = g;
      if (tid == 0) {
        const T glast = reorth ? scl(c_s[j - 1], beta) : zero_of(T{});  // H[j, j-1] c[j-1]
        gvec[j] = glast;
        c_s[kTMax - 1] = glast;
      }
    }
    if (tid == 0) {
      Hcol[j] = from_real(beta, T{});  // :105
      Tm[j + (int64_t)j * ldt] = from_real(binv, T{});
      if (use_lds && jd) Tl[j + (j - ntrue) * jt] = from_real(binv, T{});
      st->wnorm = beta;
      st->inv_norm = binv;
      st->reorth = 0;
      st->pend = 0;
      st->n_steps += 1;
      if (reorth) st->n_reorth += 1;
    }
    binv_in = binv;
    __syncthreads();  // the new column of T (LDS or memory), c_s[kTMax-1] are settled before the DOTS half reads them
    if (jd && li == j && j < 64) gi = c_s[kTMax - 1];  // lane j owns entry j of g in the DOTS half (jd = j+1 lanes)
#ifdef KS_FIN_TIMING
    tk3 = wall_clock64();
#endif
  } else if (jd && (jd - 1) >= ntrue) {
    binv_in = real_of(tb[(jd - 1) + (jd - 1 - ntrue) * tld]);
  }
  // ================================ DOTS(jd) ================================
  if (jd) {
    const int j = jd;
    T* Hcol = Hd + (int64_t)(j - 1) * ldh;
    // k_dots delivered the sums for sigma * y' (sigma a power of two): b_s = sigma s, sigma^2 |y'|^2; the algebra below runs on
    // the scaled quantities (t'' = sigma t, g'' = sigma g) and divides the power of two out at the end -- exact
    const T* b_s = r_s + nm;  // sigma s[0..j-1], sigma^2 |y'|^2 at [j]
    const double isg = 1.0 / sig;
    const T gs = scl(gi, sig);
    T ti;
    {  // t''[i] = sum_{k <= i} conj(T[k,i]) s''[k]
      const T part = (li < j && li >= ntrue) ? tcol_dot(b_s, li) : zero_of(T{});
      ti = combine(part);
      if (li < j && li < ntrue) ti = b_s[li];
      if (li >= j) ti = zero_of(T{});
    }
    if (q == 0 && li < j) {
      t_s[li] = ti;
      Hcol[li] = scl(scl(sub_(ti, gs), isg), binv_in);  // h = V_true^H (A v_true) = (t - g) / beta
    }
    // ||A v_true||^2 = (|y'|^2 - 2 Re g^H t + |g|^2) / beta^2
    const double tot = wave_sum((li < j) ? fma(-2.0, redot_(gs, ti), abs2_(gs)) : 0.0);
    if (tid == 0) {
      const double rn2 = real_of(b_s[j]) + tot;
      const double rn = sqrt(rn2 > 0.0 ? rn2 : 0.0) * isg * binv_in, rp = sqrt(real_of(b_s[j])) * isg * binv_in;
      st->rnorm = rn;
      st->rnorm2 = rp;  // norm of what the projection kernel actually works on: y' / beta
      st->invb = binv_in;
      // scale for the NEXT step's k_dots: the column this step stores, w', has norm <= max(rn, rp) ~ ||A||
      const double mag = fmax(rn, rp);
      st->sigma = (mag > 0.0 && mag < 1.7e308) ? ldexp(1.0, -ilogb(mag)) : 1.0;
    }
    __syncthreads();
    {  // coefficients of the STORED columns: T t / beta
      const T part = (li < j) ? trow_dot(t_s, li, j) : zero_of(T{});
      T a = combine(part);
      if (li < ntrue && li < j) a = add_(a, t_s[li]);
      if (q == 0 && li < j) coef[li] = scl(scl(a, isg), binv_in);
    }
  }
#ifdef KS_FIN_TIMING
  if (tid == 0 && jm == 35)
    printf("[fin jm=%d jd=%d wg=%d] reduce %.2f us | election %.2f us | MID %.2f us | DOTS %.2f us\n", jm, jd, (int)blockIdx.x,
           (tk1 - tk0) * 0.01, (tk2 - tk1) * 0.01, (tk3 - tk2) * 0.01, (wall_clock64() - tk3) * 0.01);
#endif
}

// ------------------------------------------------------------------------------------------------
// FIN_NORM (one workgroup): wnorm = sqrt(sum_b partial2[b]) and the DGKS decisions
// (src/expansion.jl:88-108).  mode as in k_fin_dots (red[0] carries the all-reduced sum).
//   pass 1:  wnorm <  eta*rnorm -> request second pass (rnorm <- wnorm), else finalize
//   pass 2:  (only if requested) finalize
//   finalize: wnorm <= eta*rnorm -> H[j,j-1] = 0, breakdown = j;  else H[j,j-1] = wnorm, inv = 1/wnorm
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_fin_norm(const double* __restrict__ partial2, int nb, double* __restrict__ red, T* __restrict__ Hsub, int j,
               int pass, int mode, DevState* __restrict__ st) {
  if (st->breakdown >= 0) return;
  if (pass == 2 && !st->reorth) return;
  __shared__ double sm[kBlock];
  const int tid = threadIdx.x;
  double s = 0.0;
  if (mode != 2) {
    for (int b = tid; b < nb; b += kBlock) s += partial2[b];
    sm[tid] = s;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
      if (tid < off) sm[tid] += sm[tid + off];
      __syncthreads();
    }
    s = sm[0];
    if (mode == 1) {
      if (tid == 0) red[0] = s;
      return;
    }
  } else {
    s = red[0];
  }
  if (tid != 0) return;
  const double wnorm = sqrt(s);
  st->wnorm = wnorm;
  if (pass == 1) {
    if (wnorm < kEta * st->rnorm) {  // src/expansion.jl:91
      st->reorth = 1;
      st->rnorm2 = wnorm;            // :92
      st->n_reorth += 1;
      return;
    }
    st->reorth = 0;
  } else {
    st->reorth = 0;
  }
  const double rn = (pass == 1) ? st->rnorm : st->rnorm2;
  if (wnorm <= kEta * rn) {          // :99
    if constexpr (sizeof(T) == 8) *Hsub = 0.0; else *Hsub = cd{0.0, 0.0};
    st->breakdown = j;
    st->inv_norm = 0.0;
  } else {
    if constexpr (sizeof(T) == 8) *Hsub = wnorm; else *Hsub = cd{wnorm, 0.0};
    st->inv_norm = 1.0 / wnorm;      // :106  v ./= wnorm
    st->n_steps += 1;
  }
}

// v *= st->inv_norm  (v ./= wnorm, src/expansion.jl:106) -- or by an immediate factor when st == null
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_scale(T* __restrict__ v, int64_t ld, double factor, const DevState* __restrict__ st) {
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  double f = factor;
  if (st) {
    if (st->breakdown >= 0) return;
    f = st->inv_norm;
  }
  int64_t pb, pe;
  block_range(ld / R, blockIdx.x, gridDim.x, pb, pe);
  for (int64_t p = pb + threadIdx.x; p < pe; p += kBlock) {
    const int64_t r = p * R;
    st_pack(v + r, scale_pack(ld_pack(v + r), f));
  }
}

// partial2[b] = sum |v|^2   (norm(v))
template <class T>
__global__ void __launch_bounds__(kBlock) k_norm2(const T* __restrict__ v, int64_t ld, double* __restrict__ partial2) {
  constexpr int R = Pack<T>::R;
  __shared__ double red[kBlock / 64];
  int64_t pb, pe;
  block_range(ld / R, blockIdx.x, gridDim.x, pb, pe);
  double nrm = 0.0;
  for (int64_t p = pb + threadIdx.x; p < pe; p += kBlock) nrm += nrm2_pack(ld_pack(v + p * R));
  const double s = wave_sum(nrm);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial2[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = sum_b partial2[b]   (plain reduction used by the synchronous verbs)
static __global__ void __launch_bounds__(kBlock) k_sum(const double* __restrict__ partial2, int nb, double* __restrict__ out) {
  __shared__ double sm[kBlock];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int b = tid; b < nb; b += kBlock) s += partial2[b];
  sm[tid] = s;
  __syncthreads();
  for (int off = kBlock / 2; off >= 1; off >>= 1) {
    if (tid < off) sm[tid] += sm[tid + off];
    __syncthreads();
  }
  if (tid == 0) out[0] = sm[0];
}

// ------------------------------------------------------------------------------------------------
// Generic tall-skinny product (fallback + checker for the MFMA kernel, and the complex path):
//   OUT[:, 0:r) = V[:, 0:c) * Qd[0:c, 0:r)     Qd device, column-major ldq.
// In place (OUT == V, or OUT = any columns of the same array) is allowed when c <= CT: every thread first reads its whole
// row slice.
// ------------------------------------------------------------------------------------------------
template <class T, int CT>
__global__ void __launch_bounds__(kBlock)
    k_rotate_valu(const T* __restrict__ Vin, int64_t ldv, int c, int r, const T* __restrict__ Qd, int ldq,
                  T* __restrict__ Vout, int64_t ldo, int extra_out = -1) {
  using P = typename Pack<T>::type;
  constexpr int R = Pack<T>::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* qs = reinterpret_cast<T*>(smem_raw);  // c x r, column-major, ld = c
  for (int i = threadIdx.x; i < c * r; i += kBlock) qs[i] = Qd[(i % c) + (int64_t)(i / c) * ldq];
  __syncthreads();
  int64_t pb, pe;
  block_range(ldv / R, blockIdx.x, gridDim.x, pb, pe);
  for (int64_t p = pb + threadIdx.x; p < pe; p += kBlock) {
    const int64_t row = p * R;
    P in[CT];
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
      const int c2 = cc < c ? cc : c - 1;
      in[cc] = ld_pack(Vin + (int64_t)c2 * ldv + row);
    }
    for (int rr = 0; rr < r; ++rr) {
      P s = zero_pack(T{});
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        if (cc < c) axpy_acc(s, in[cc], qs[cc + rr * c]);
      }
      // (extra_out >= 0: the LAST output goes to column extra_out of Vout instead of column r-1)
      const int oc = (extra_out >= 0 && rr == r - 1) ? extra_out : rr;
      st_pack(Vout + (int64_t)oc * ldo + row, s);
    }
  }
}

// Out-of-place general product for shapes the in-place kernels do not cover (c > 64) and for
// real-basis x complex-coefficient products (partialeigen, src/eigvals.jl:94):
//   OUT[:, rr] = sum_cc V[:, cc] * Y[cc, rr],  TV basis type, TY coefficient/result type.
template <class TV, class TY>
__global__ void __launch_bounds__(kBlock)
    k_gemm_tall(const TV* __restrict__ V, int64_t ldv, int64_t n, int c, int r, const TY* __restrict__ Y, int ldy,
                TY* __restrict__ out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  TY* ys = reinterpret_cast<TY*>(smem_raw);
  for (int i = threadIdx.x; i < c * r; i += kBlock) ys[i] = Y[(i % c) + (int64_t)(i / c) * ldy];
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < n; row += (int64_t)gridDim.x * kBlock) {
    for (int r0 = 0; r0 < r; r0 += 8) {
      TY acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = zero_of(TY{});
      for (int cc = 0; cc < c; ++cc) {
        const TV v = V[(int64_t)cc * ldv + row];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (r0 + u < r) {
            const TY y = ys[cc + (r0 + u) * c];
            if constexpr (sizeof(TV) == 8 && sizeof(TY) == 16) {
              acc[u].x = fma(v, y.x, acc[u].x);
              acc[u].y = fma(v, y.y, acc[u].y);
            } else {
              acc[u] = fma_(v, y, acc[u]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u < r) out[(int64_t)(r0 + u) * ldo + row] = acc[u];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA restart rotation (Float64):  V[:, 0:r) <- V[:, 0:c) * Q[0:c, 0:r)  IN PLACE.
// (mul!(V_tmp, V, Q) + copyto!(V, V_tmp), src/run.jl:363-364, :382-383 -- the one dense contraction.)
//
// v_mfma_f64_16x16x4_f64 computes D(16x16) = A(16x4) B(4x16) + C.  We compute the TRANSPOSE of the
// output tile so that both the loads of V and the stores of the result are 16-byte per lane and
// 256-byte contiguous per 16-lane group:
//     D^T[i][jn] = sum_k  Q[k][n0+i] * V[row0+jn][k]
//   A operand: lane l holds Q[kb + (l>>4)][n0 + (l&15)]            (from LDS)
//   B operand: lane l holds V[row0 + rowsel(l&15)][kb + (l>>4)]     (one double2 load = 2 row tiles)
//   D:         lane l, reg v holds out[row0 + rowsel(l&15)][n0 + (l>>4) + 4v]
// A wave owns 32 consecutive rows: row tile 0 = even rows, row tile 1 = odd rows, so lane l's
// double2 at rows (2*(l&15), 2*(l&15)+1) feeds both tiles and the results pair up again into one
// double2 store.  Every output row depends only on the same input row, and a wave reads all its
// c <= 4*KC input columns into registers before storing anything, so the update is in place and the
// reference's V_tmp + copy-back (2 extra passes over n x r) disappear.
// ------------------------------------------------------------------------------------------------
typedef double double4v __attribute__((ext_vector_type(4)));

// RT = 32-row tiles a wave handles per trip (RT x KC 16-byte loads in flight per lane; RT = 2: 512 B instead of 256 B
// contiguous per column and wave).
template <int KC, int RT>
__global__ void __launch_bounds__(kBlock)
    k_rotate_mfma(double* __restrict__ V, int64_t ldv, int c, int r, const double* __restrict__ Qd, int ldq, int out0 = 0,
                  int extra_out = -1) {
  // inputs: columns 0..c-1 of V; outputs: columns out0..out0+r-1 (extra_out >= 0: the LAST output goes to column
  // extra_out instead).  out0 = 0, extra_out = -1 is the plain in-place rotation.
  // Q^T tile in LDS: qs[n][k] with k < 4*KC (zero padded), n < ntile*16 (zero padded)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* qs = reinterpret_cast<double*>(smem_raw);  // [ncol16][4*KC] : qs[n * KP + k]
  constexpr int KP = 4 * KC + 1;                     // +1 pad: lanes of a 16-group read stride-KP -> conflict free
  const int ntile = (r + 15) >> 4;
  const int ncol = ntile * 16;
  for (int i = threadIdx.x; i < ncol * 4 * KC; i += kBlock) {
    const int k = i % (4 * KC), n = i / (4 * KC);
    qs[n * KP + k] = (k < c && n < r) ? Qd[k + (int64_t)n * ldq] : 0.0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int64_t nwt = ldv / 32;  // 32-row wave tiles (ldv is a multiple of 64)
  const int64_t stride = (int64_t)gridDim.x * (kBlock / 64) * RT;
  for (int64_t wt0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * RT; wt0 < nwt; wt0 += stride) {
    double2 b[RT][KC];
    int64_t row[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int64_t wt = wt0 + t < nwt ? wt0 + t : nwt - 1;  // (ldv / 32 is even: with RT = 2 a trip never straddles the end)
      row[t] = wt * 32 + 2 * l15;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int col = 4 * kc + l4;
        const int c2 = col < c ? col : c - 1;  // padded k rows of Q are zero, value irrelevant
        b[t][kc] = ld_pack_nt(V + (int64_t)c2 * ldv + row[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      if (wt0 + t >= nwt) break;
      for (int nt = 0; nt < ntile; ++nt) {
        double4v acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double* qrow = qs + (nt * 16 + l15) * KP + l4;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const double a = qrow[4 * kc];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t][kc].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t][kc].y, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int col = nt * 16 + l4 + 4 * v;
          const int oc = (extra_out >= 0 && col == r - 1) ? extra_out : out0 + col;
          if (col < r) st_pack_nt(V + (int64_t)oc * ldv + row[t], make_double2(acc0[v], acc1[v]));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// VECTOR-ALU form of the Float64 rotation -- the DEFAULT on gfx950 for c <= 64.  On this part the FP64 matrix cores run
// at HALF the rate of the FP64 vector ALUs (v_mfma_f64_16x16x4_f64 issues every ~128 cycles per SIMD: measured, the MFMA
// kernels above read V at 4.8 TB/s with their stores removed and at 6.7 TB/s with the MFMAs removed as well), and the
// 16-wide MFMA tiles pad r = 21 outputs to 32.  A plain FMA loop needs 2 c r flops per row and nothing else:
//   lane = one 16-byte pack (two rows); all c <= CT input columns of the pack in registers (so the update is in place);
//   per output column a dot product over the inputs against Q^T from LDS (broadcast 16-byte reads: two coefficients each),
//   four independent accumulation chains; the result is stored at once -- 4 KiB contiguous per column and workgroup.
// 2 c r / (8 (c + r)) ~ 3.5 flop/B at c = 41, r = 21: 0.22 ms of vector-ALU time at n = 1e7 against 0.8 ms of memory time
// -- HBM-bound, which the MFMA forms are not on this part (0.72 ms of matrix-core time).  Measured at n = 1e7, c = 41,
// r = 21 (profiles/r03_rotation.txt): 4.97 TB/s against 4.86 of k_rotate_mfma.  What does NOT lift it further, all tried:
// staging the output in LDS and writing 6 KiB bursts per column (4.5-4.8), writing out of place (4.92), one to three
// workgroups per CU (4.97-5.01).  With its stores removed the kernel reads at 6.7 TB/s: what is left is what this memory
// system makes of 41 read streams next to 21 write streams (a third of the traffic is writes; the expansion kernels
// write 3 %).
// ------------------------------------------------------------------------------------------------
// REVERSE MAILBOX.  The restart's rotation is enqueued BEFORE the host's Schur step, behind a one-workgroup
// gate that waits for a word in pinned host memory: when the host has Q it writes the product T Q and the shape of the
// rotation into pinned memory and releases the word -- the gate copies both into device memory and the rotation, already
// queued behind it, starts at once.  What leaves the critical path: a stream synchronisation, a hipMemcpyAsync of Q and a
// kernel launch issued only after the host step (20-30 us per restart cycle).  The wait is bounded by wall clock; a gate
// that is cancelled (breakdown in the last batch, an exception in the host step) or times out marks the parameters
// cancelled and the rotation behind it returns immediately.
struct RotGate {
  // host -> device (pinned, polled by the gate): (sequence number << 1) | cancel.  Release and cancellation are ONE word tied to
  // the sequence number (round-4 advice: a `cancel` field of its own could be re-armed to 0 before a gate kernel that had not
  // started yet read it -- which happens as soon as other kernels sit in the stream in front of the gate -- and the stale
  // rotation behind it then ran); a gate that finds a LATER sequence number was superseded and counts as cancelled.
  uint64_t flag;
  int32_t c, r, out0, extra_out, ldq, cancel, nq, pad_;
  uint64_t timed_out;     // device -> host: the gate gave up waiting (sequence number)
  // device -> host: (sequence number << 1) | (0: took the release, 1: gave up / was cancelled), written as soon as the gate has
  // decided.  The host reads it back after releasing: a gate that gave up between the host's last look at `timed_out` and its
  // release would otherwise leave the host believing in a rotation that never ran (round-4 advice).
  uint64_t taken;
};
static __global__ void __launch_bounds__(kBlock) k_rot_gate(RotGate* __restrict__ host, uint64_t seq, RotGate* __restrict__ dev,
                                                            const double* __restrict__ q_host, double* __restrict__ q_dev,
                                                            long long timeout_ticks) {
  __shared__ int state;   // 1: released, 2: gave up, 3: cancelled
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int st = 0;
    while (st == 0) {
      const uint64_t f = __hip_atomic_load(&host->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((f >> 1) == seq) st = (f & 1u) ? 3 : 1;
      else if ((f >> 1) > seq) st = 3;           // superseded: cancelled
      else if (wall_clock64() - t0 > timeout_ticks) st = 2;
      else __builtin_amdgcn_s_sleep(8);
    }
    state = st;
    __hip_atomic_store(&host->taken, (seq << 1) | (st == 1 ? 0u : 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  if (state != 1) {
    if (threadIdx.x == 0) {
      dev->cancel = 1;
      if (state == 2) __hip_atomic_store(&host->timed_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const int nq = host->nq;
  for (int i = threadIdx.x; i < nq; i += kBlock) q_dev[i] = q_host[i];
  if (threadIdx.x == 0) {
    dev->c = host->c; dev->r = host->r; dev->out0 = host->out0; dev->extra_out = host->extra_out; dev->ldq = host->ldq;
    dev->cancel = 0;
  }
}

template <int CT, bool NT = true>
__global__ void __launch_bounds__(kBlock)
    k_rotate_fma(double* __restrict__ V, int64_t ldv, int c, int r, const double* __restrict__ Qd, int ldq, int out0, int extra_out,
                 const RotGate* __restrict__ gate = nullptr) {
  static_assert(CT % 4 == 0, "CT must be a multiple of four");
  if (gate) {  // shape from the gate in front of this launch (reverse mailbox)
    if (gate->cancel) return;
    c = gate->c; r = gate->r; ldq = gate->ldq; out0 = gate->out0; extra_out = gate->extra_out;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* qs = reinterpret_cast<double*>(smem_raw);  // [r][CT]: qs[n * CT + k] = Q[k, n], zero for k >= c
  for (int i = threadIdx.x; i < r * CT; i += kBlock) {
    const int k = i % CT, n = i / CT;
    qs[i] = k < c ? Qd[k + (int64_t)n * ldq] : 0.0;
  }
  __syncthreads();
  int64_t pb, pe;
  block_range(ldv / 2, blockIdx.x, gridDim.x, pb, pe);
  for (int64_t p = pb + threadIdx.x; p < pe; p += kBlock) {
    const int64_t row = p * 2;
    double2 in[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      const int k2 = k < c ? k : c - 1;  // (coefficient zero: value irrelevant, the load is an L1 / L2 hit)
      in[k] = ld_v<NT>(V + (int64_t)k2 * ldv + row);
    }
    for (int n = 0; n < r; ++n) {
      const double* qn = qs + n * CT;
      double2 s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < CT; k += 4) {
        const double2 qa = *reinterpret_cast<const double2*>(qn + k), qb = *reinterpret_cast<const double2*>(qn + k + 2);
        s0.x = fma(in[k].x, qa.x, s0.x);
        s0.y = fma(in[k].y, qa.x, s0.y);
        s1.x = fma(in[k + 1].x, qa.y, s1.x);
        s1.y = fma(in[k + 1].y, qa.y, s1.y);
        s0.x = fma(in[k + 2].x, qb.x, s0.x);
        s0.y = fma(in[k + 2].y, qb.x, s0.y);
        s1.x = fma(in[k + 3].x, qb.y, s1.x);
        s1.y = fma(in[k + 3].y, qb.y, s1.y);
      }
      const int oc = (extra_out >= 0 && n == r - 1) ? extra_out : out0 + n;
      st_pack_nt(V + (int64_t)oc * ldv + row, make_double2(s0.x + s1.x, s0.y + s1.y));
    }
  }
}

// Gram / cross-Gram for the on-device residual checks:
//   partial[b][i + j*nc] = sum_rows conj(A[r,i]) * B[r,j]     (i < nc_a, j < nc_b, both <= 8 wide tiles)
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_gram_tile(const T* __restrict__ A, int64_t lda, int na, const T* __restrict__ B, int64_t ldb, int nbcols,
                int64_t n, T* __restrict__ partial) {
  T acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = zero_of(T{});
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < n; row += (int64_t)gridDim.x * kBlock) {
    T a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = i < na ? A[(int64_t)i * lda + row] : zero_of(T{});
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = j < nbcols ? B[(int64_t)j * ldb + row] : zero_of(T{});
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (sizeof(T) == 8) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        else {
          acc[i][j].x = fma(a[i].x, b[j].x, fma(a[i].y, b[j].y, acc[i][j].x));
          acc[i][j].y = fma(a[i].x, b[j].y, fma(-a[i].y, b[j].x, acc[i][j].y));
        }
      }
  }
  __shared__ T red[kBlock / 64][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const T s = wave_sum(acc[i][j]);
      if (lane == 0) red[wave][i + 8 * j] = s;
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    T s = red[0][threadIdx.x];
    for (int wv = 1; wv < kBlock / 64; ++wv) s = add_(s, red[wv][threadIdx.x]);
    partial[(int64_t)blockIdx.x * 64 + threadIdx.x] = s;
  }
}

// out[c] = sum_b partial[b*stride + c], c < cnt  (one workgroup)
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_reduce_cols(const T* __restrict__ partial, int nb, int stride, int cnt, T* __restrict__ out) {
  __shared__ T sm[kBlock];
  const int tid = threadIdx.x;
  for (int c0 = 0; c0 < cnt; c0 += 64) {
    const int c = c0 + (tid & 63);
    T s = zero_of(T{});
    if (c < cnt)
      for (int b = tid >> 6; b < nb; b += kBlock / 64) s = add_(s, partial[(int64_t)b * stride + c]);
    sm[tid] = s;
    __syncthreads();
    if (tid < 64 && c < cnt) out[c] = add_(add_(sm[tid], sm[tid + 64]), add_(sm[tid + 128], sm[tid + 192]));
    __syncthreads();
  }
}

// y = f * x (column copy with an optional factor), pads included
// ------------------------------------------------------------------------------------------------
// PUBLISH: hand a piece of the control block (H columns + DevState) to the host WITHOUT the runtime's copy machinery:
// one workgroup stores it into host-pinned (device-mapped, coherent) memory and then releases a sequence number the host
// spins on.  A hipMemcpyAsync in its place costs the host ~100 us of enqueue time and stalls the stream's submission;
// a kernel is 3 us in line.  No early exit: the host must always be woken (the state it reads says what happened).
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(kBlock) k_publish(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst_host, int nwords,
                                                    uint64_t* flag_host, uint64_t seq) {
  for (int i = threadIdx.x; i < nwords; i += kBlock) dst_host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <class T>
__global__ void __launch_bounds__(kBlock) k_copy(const T* __restrict__ x, T* __restrict__ y, int64_t ld, double f) {
  constexpr int R = Pack<T>::R;
  int64_t pb, pe;
  block_range(ld / R, blockIdx.x, gridDim.x, pb, pe);
  for (int64_t p = pb + threadIdx.x; p < pe; p += kBlock) st_pack(y + p * R, scale_pack(ld_pack(x + p * R), f));
}

// w = A*x - (combination) helpers for the residual checks:  y -= sum_c X[:,c] * coef[c]
template <class T>
__global__ void __launch_bounds__(kBlock)
    k_sub_lincomb(T* __restrict__ y, const T* __restrict__ X, int64_t ldx, int nc, const T* __restrict__ coef,
                  int64_t n) {
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < n; row += (int64_t)gridDim.x * kBlock) {
    T s = y[row];
    for (int c = 0; c < nc; ++c) {
      const T g = coef[c];
      const T v = X[(int64_t)c * ldx + row];
      if constexpr (sizeof(T) == 8) s = fma(-v, g, s);
      else {
        s.x -= v.x * g.x - v.y * g.y;
        s.y -= v.x * g.y + v.y * g.x;
      }
    }
    y[row] = s;
  }
}

}  // namespace ksd
