// Shift-invert operator from caller-supplied triangular factors:  y = P_out U^-1 L^-1 P_in (s o x)  with both sparse
// triangular solves on the device.  This is what a user of the reference wraps as
//     LinearMap{ComplexF64}((y, x) -> ldiv!(y, F, x), n; ismutating = true),   F = lu(A - sigma I)      docs/src/index.md:246-249
// and hands to partialschur as the operator `A` of mul!(y, A, x) (src/expansion.jl:121).  The factorisation stays the
// caller's (SuiteSparse / SuperLU on the host, exactly as in the reference); only its APPLICATION is on the hot path
// (once per Arnoldi step), and here it never leaves HBM.
// Part of the ONE translation unit of libkschur_hip.so (included by ks_hip.hip after ks_operators.hpp).
//
// Algorithm: synchronisation-free sparse triangular solve in CSR row form.  One wavefront per row; the lanes hold the
// row's off-diagonal entries and wait for the solution entries they need to appear, then the wave reduces and publishes
// the row's entry.  No launch per dependency level: one or two launches per triangular factor.
// Host side, once (upload_factor):
//   * The LAST K pivots (the separators next to the root of the elimination tree) are the "top" part.  Without them the
//     factor falls apart into independent parts (connected components of its dependency graph), which are packed by work
//     into up to 8 groups: ONE launch runs all groups, each on its own XCD (every hand-over inside that XCD's L2), a
//     second launch runs the top part on one XCD -- after the groups for the lower factor, before them for the upper one.
//     The top part is split the same way again (up to four layers of groups); the entries of a launch that refer to rows
//     of earlier launches are summed beforehand by a dependency-free pre-pass over the whole device.
//   * Inside each of these segments the rows are RENUMBERED by dependency level (stable: ties keep the elimination
//     order).  Any topological order keeps a triangular factor triangular; this one puts what can run concurrently next to
//     each other (in elimination order the backward solve walks the elimination tree depth-first and the window of rows
//     in flight holds one dependent path: measured 5x slower than the forward solve).  Both factors become LOWER
//     triangular in their own numbering; the permutations between the caller's x, the two numberings and the caller's y
//     are folded into three index arrays.
//   * Where the levels are narrow -- the dense triangles of the separators: a chain -- runs of up to 256 consecutive rows
//     are INVERTED (x_run = T^-1 s_run), which turns 256 dependent hand-overs into two (see "dense runs" below).
// Device side (k_sptrsv):
//   * Solution entries are published as "LL" words -- 32 bits of payload + the 32-bit sequence number of this solve in
//     one 8-byte atomic store -- so a reader that sees the sequence number has the data (no flag + fence pair, nothing
//     to reset between solves).  Float64: 2 words per entry, ComplexF64: 4.
//   * Rows are handed out through a ticket counter in numbering order (one workgroup = 16 consecutive rows, one per wave),
//     so every row a resident wave waits for belongs to a workgroup that already runs or has finished: forward progress
//     does not depend on the dispatch order of workgroups.
//   * Every wait is bounded by a wall-clock budget (KS_LU_TIMEOUT_S, default 20 s): reports KS_ERR_OPERATOR at the next
//     synchronisation point of the context instead of hanging the device (the host rejects malformed factors up front;
//     this guards what it cannot see, e.g. a device shared with a process that starves the producers).
// The product is bound by dependency chains and memory round trips, not by bytes (profiles/r03_shift_invert.txt has the
// path from 605 ms to 2.0 ms per product at n = 5e5); ks_operator_lu_info / _layout report levels, fill, groups and runs so
// a caller can judge an ordering (fill-reducing with a short, bushy elimination tree).
#pragma once

namespace ksd {

constexpr int kLuMaxLayers = 4;  // layers per factor the control words (LuOp::kWords) provide for

struct TrsvArgs {
  const DevState* st;      // expansion batch state (nullptr outside a batch): after a breakdown / bail of the batch every
                           // remaining product is skipped, like every other operator kernel does
  int64_t row0, n;         // this launch solves rows [row0, n) (everything before is complete: earlier launch)
  const int64_t* rowptr;   // strictly lower triangular part in the factor's own (level) numbering, CSR, columns ascending
  const int64_t* rowbegin; // first entry of row r this launch has to sum itself (== rowptr, or past the entries a pre-pass summed)
  const void* pre;         // pre-summed part of every row (nullptr: none)
  const int32_t* colind;
  const void* val;
  const void* diag;        // n INVERSE diagonal entries; nullptr: unit diagonal
  uint64_t* sol;           // LL words of this factor's solution (W per row)
  const void* rhs;         // first solve: x;  second solve: nullptr
  const uint64_t* rhs_ll;  // second solve: the first solve's LL words (complete: previous launch)
  const int32_t* src;      // row i takes rhs[src[i]] (first) / rhs_ll[src[i]] (second)
  const double* scale;     // first solve: ... times scale[i] (already in row order; nullptr: none)
  void* out;               // second solve: y[dst[i]] = entry i
  const int32_t* dst;
  int* ticket;             // this launch's control words: [0] row counter, [2] owning XCD + 1 (zeroed by the host side before the product);
                           // grouped launch: four words per group
  int64_t gbeg[8], gend[8];  // grouped launch (LOCAL = 5): rows of group g; a group belongs to ONE XCD
  signed char xcc_group[16]; //   XCC id -> group (-1: none)
  int* err;                // pinned host word
  long long timeout_ticks;
  uint32_t seq;
  int backoff;             // nap between two polls of the awaited entry, in units of 128 clocks
  int nap_lds;             // nap between two polls of an LDS flag, in units of 64 clocks
  int64_t stall_row;       // fault injection (KS_LU_INJECT_STALL=<row>): this row is never published; -1: none
  unsigned long long* timeline;  // KS_LU_TIMELINE=<prefix>: per row, wall clock (10 ns ticks): ticket obtained, far part summed, published; entries from its own chunk
  unsigned long long* stats;  // KS_LU_STATS=1: [0] ticks in ticket + barriers [1] rows [2] ticks of rows [3] ticks at the gate [4] ticks on LDS only
                              // [5] gate polls [6] attempts [7] cached hits [8] cached misses [9] coherent tries (per lane)
};

template <class D> struct LLWords { static constexpr int W = (int)(sizeof(D) / 4); };

// LOCAL (3, 4, 5): every producer and consumer of a launch (of a group, 5) runs on ONE XCD (see k_sptrsv).  Measured before
// the factor was cut into groups and runs, 200 x 250 / 500 x 1000 grid, ms per product: all XCDs 2.7 / 13.5 (and 15-65 once
// more than ~64 workgroups wait: every poll crosses the fabric); one XCD 3.1 / -- with device-scope stores (4); 2.2 / 13.9
// with stores that stop at that XCD's L2 (3 and 5: the workgroup-scope bits, sc0) and device-scope loads, which that L2
// serves.  Loads with the workgroup-scope bits never see another CU's store (they are served by the CU's own cache, which
// neither snoops nor is dropped by `buffer_inv sc0`): the solve stalls until its time budget ends.  Whatever a load
// returns, a word carrying the current sequence number is valid data.
template <int LOCAL> __device__ __forceinline__ uint64_t ll_word(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class D, int LOCAL> __device__ __forceinline__ void ll_put(uint64_t* p, D v, uint32_t seq) {
  constexpr int W = LLWords<D>::W;
  uint32_t w[W];
  __builtin_memcpy(w, &v, sizeof(D));
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const uint64_t word = ((uint64_t)seq << 32) | w[k];
    if (LOCAL == 3 || LOCAL == 5) __hip_atomic_store(p + k, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p + k, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// CACHED: an ordinary load that may be served by this CU's / this XCD's caches.  A word that carries the current
// sequence number is valid wherever it was found (payload and number are one 8-byte store); a stale line just reads as
// "not there yet" and the caller asks again with the coherent form.
template <class D, int LOCAL, bool CACHED = false> __device__ __forceinline__ bool ll_get(const uint64_t* p, D& v, uint32_t seq) {
  constexpr int W = LLWords<D>::W;
  uint64_t u[W];
  if (CACHED) {
    // 16 bytes per request: the cache's request rate, not its bandwidth, is what a dense triangle's gathers run into
    // (every 8-byte word is still consistent in itself: payload and number never straddle a naturally aligned dword pair)
#pragma unroll
    for (int k = 0; k < W; k += 2) {
      const ulonglong2 q = *reinterpret_cast<const ulonglong2*>(p + k);
      u[k] = q.x;
      u[k + 1] = q.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) u[k] = ll_word<LOCAL>(p + k);
  }
  bool ok = true;
  uint32_t w[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    ok = ok && (uint32_t)(u[k] >> 32) == seq;
    w[k] = (uint32_t)u[k];
  }
  if (ok) __builtin_memcpy(&v, w, sizeof(D));
  return ok;
}
// payload of words known to be complete (written by an earlier launch)
template <class D> __device__ __forceinline__ D ll_payload(const uint64_t* p) {
  constexpr int W = LLWords<D>::W;
  uint32_t w[W];
#pragma unroll
  for (int k = 0; k < W; ++k) w[k] = (uint32_t)p[k];
  D v;
  __builtin_memcpy(&v, w, sizeof(D));
  return v;
}

__device__ __forceinline__ bool trsv_expired(int* err, long long t0, long long budget, long spins) {
  if ((spins & 255) != 255) return false;
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return true;
  if (wall_clock64() - t0 > budget) {
    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
  return false;
}

// Forward substitution on a factor in level numbering.  A workgroup of C waves takes C consecutive rows per ticket, ONE
// ROW PER WAVE: every row in flight has fetched its entries and consumed every solution entry that already exists, and
// sits waiting for the first missing one -- so the time along a dependency chain is hand-over latency only, not the
// row's start-up (offsets -> columns -> solution words are three dependent memory round trips).
//   * entries produced by the SAME workgroup travel through LDS (value + flag: ~0.2 us per hand-over): consecutive
//     single-row levels (the dense triangle of a separator) are where a factor's chain is longest;
//   * entries of earlier chunks come from the LL words in memory: first through the caches (entries finished long ago
//     -- nearly all of a long row), then coherently.  A waiting wave polls ONE entry -- the latest missing dependency
//     of its current batch -- with W lanes, and only when that has arrived do its lanes fetch theirs again: thousands
//     of waiting waves cost a few loads per microsecond each instead of 64 x W (which saturates the fabric and slows
//     the producers: measured 25 us per level).
// (one 1024-thread workgroup per CU: with registers for two -- 64 VGPRs at unroll 2 with spills, or unroll 1 without -- a
// product is slower, 4.7 against 2.9 ms at the time and 2.2-2.3 against 2.0 in the final form: more rows in flight do not
// pay, shorter batches cost; what a chain needs is a quick hand-over, not occupancy)
constexpr int kTrsvWaves = 16, kTrsvUnroll = 4;

// LOCAL: the launch is 8x oversubscribed and only the workgroups that landed on ONE XCD work -- the XCD of whichever
// workgroup asks first (an election, not "XCD 0": in a partitioned device there is one XCD and it need not be number 0);
// the others leave at once.  Tickets make it irrelevant which workgroups those are.
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf); }  // HW_REG_XCC_ID[3:0]

// wave64 sum without LDS traffic: four DPP steps inside each row of 16 lanes, then the four row sums through scalar
// registers (the butterfly of ks_kernels.hpp's wave_sum is 12 ds_bpermute per double: ~0.5 us, too slow for a chain)
template <int CTRL> __device__ __forceinline__ double dpp_perm(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int l) {  // l uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ cd lane_value(cd v, int l) { return cd{lane_value(v.x, l), lane_value(v.y, l)}; }
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_perm<0xB1>(v);   // quad_perm(1,0,3,2)
  v += dpp_perm<0x4E>(v);   // quad_perm(2,3,0,1)
  v += dpp_perm<0x141>(v);  // row_half_mirror
  v += dpp_perm<0x140>(v);  // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ cd wave_sum_dpp(cd v) { return cd{wave_sum_dpp(v.x), wave_sum_dpp(v.y)}; }

template <class D, int LOCAL, bool PROBE>
__global__ void __launch_bounds__(kTrsvWaves * 64) k_sptrsv(const TrsvArgs a) {
  if (a.st && a.st->breakdown >= 0) return;
  // (the probes -- ten per-lane counters, clock reads -- cost a third of the register budget: compiled out unless asked for)
  unsigned long long* const stats_ = PROBE ? a.stats : nullptr;
  unsigned long long* const timeline_ = PROBE ? a.timeline : nullptr;
  constexpr int W = LLWords<D>::W, C = kTrsvWaves, U = kTrsvUnroll;
  __shared__ D xs[C];
  __shared__ int ready[C];
  __shared__ int s_ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const D* __restrict__ val = static_cast<const D*>(a.val);
  const D* __restrict__ dinv = static_cast<const D*>(a.diag);
  int64_t row0 = a.row0, nend = a.n;
  int* ticket = a.ticket;
  if (LOCAL == 5) {
    // independent parts of the factor, one per XCD: every hand-over stays inside that XCD's L2
    const int g = a.xcc_group[xcc_id()];
    if (g < 0) return;
    row0 = a.gbeg[g]; nend = a.gend[g]; ticket = a.ticket + 4 * g;
  } else if (LOCAL) {
    if (threadIdx.x == 0) {
      const int mine = xcc_id() + 1;
      const int seen = atomicCAS(a.ticket + 2, 0, mine);  // [2]: owner of this launch (0: none yet)
      s_ticket = (seen == 0 || seen == mine) ? 1 : 0;
    }
    __syncthreads();
    if (s_ticket == 0) return;
  }
  unsigned long long st_ticket = 0, st_rows = 0, st_row = 0, st_gate = 0, st_near = 0, st_polls = 0, st_att = 0, st_hit = 0, st_miss = 0, st_coh = 0;
  for (;;) {
    const long long tk0 = stats_ ? wall_clock64() : 0;
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1);
    if (threadIdx.x < C) ready[threadIdx.x] = 0;
    __syncthreads();
    const long long tk1 = (stats_ || timeline_) ? wall_clock64() : 0;
    st_ticket += (unsigned long long)(tk1 - tk0);
    const int64_t base = row0 + (int64_t)s_ticket * C;
    if (base >= nend) break;
    const int64_t r = base + wave;
    if (r >= nend) continue;
    const int64_t p0 = a.rowbegin[r], p1 = a.rowptr[r + 1];
    // Entries produced by THIS workgroup: the last m of the row (columns ascend).  Lane l < 16 looks at entry p1-1-l;
    // they are consumed one by one at the end, in column order, by the whole wave in step (no reduction on the chain).
    int li_reg = -1;
    D av_reg = zero_of(D{});
    if (lane < C && p1 - 1 - lane >= p0) {
      const int32_t c = a.colind[p1 - 1 - lane];
      if (c >= base) { li_reg = (int)(c - base); av_reg = val[p1 - 1 - lane]; }
    }
    const int m = __popcll(__ballot(li_reg >= 0));
    const int64_t pf = p1 - m;  // entries of earlier chunks: [p0, pf)
    // right-hand side and inverse pivot: fetched now, needed last (same address in every lane: one transaction)
    D b;
    {
      const int64_t s = a.src[r];
      if (s < 0) {
        b = zero_of(D{});  // solution row of a dense run: x = T^-1 s, no right-hand side of its own
      } else if (a.rhs) {
        b = static_cast<const D*>(a.rhs)[s];
        if (a.scale) b = scl(b, a.scale[r]);
      } else {
        b = ll_payload<D>(a.rhs_ll + (size_t)s * W);
      }
      if (a.pre) b = sub_(b, static_cast<const D*>(a.pre)[r]);
    }
    const D piv = dinv ? dinv[r] : zero_of(D{});
    D acc = zero_of(D{});
    const long long t0 = wall_clock64();
    long spins = 0;
    bool dead = false;
    // Batches of 64 x U entries, the SHORT one first: the last batch then holds the 64 x U latest dependencies, and a row
    // of a dense triangle reaches it while the front is still hundreds of rows away.  (Short batch last: its gate opens
    // a few rows before the row's turn, and fetching that batch -- columns, values, solution words: three round trips --
    // then stalls the chain; with 16 rows per chunk some row nearly always hit it: 8 us per chunk instead of 2.)
    const int64_t first = (pf - p0) % (64 * U) == 0 ? 64 * U : (pf - p0) % (64 * U);
    for (int64_t q0 = p0, qe = p0 + first; q0 < pf && !dead; q0 = qe, qe += 64 * U) {
      int32_t c[U];
      D av[U], xv[U];
      bool need[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = q0 + u * 64 + lane;
        need[u] = p < qe;
        c[u] = need[u] ? a.colind[p] : 0;
        av[u] = need[u] ? val[p] : zero_of(D{});
        xv[u] = zero_of(D{});
      }
      // first attempt: through the caches
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (need[u]) {
          if (ll_get<D, LOCAL, true>(a.sol + (size_t)c[u] * W, xv[u], a.seq)) { need[u] = false; ++st_hit; }
          else ++st_miss;
        }
      for (;;) {
        // one coherent attempt at everything still missing; `far` = this lane's latest missing entry
        int far = -1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!need[u]) continue;
          if (++st_coh, ll_get<D, LOCAL>(a.sol + (size_t)c[u] * W, xv[u], a.seq)) need[u] = false;
          else far = c[u] > far ? c[u] : far;
        }
        int gate = far;  // latest missing entry of the wave
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(gate, off, 64); gate = o > gate ? o : gate; }
        ++st_att;
        if (gate < 0) break;  // batch complete
        const long long tg0 = stats_ ? wall_clock64() : 0;
        const uint64_t* g = a.sol + (size_t)gate * W;
        for (;;) {
          bool ok = true;
          if (lane < W) ok = (uint32_t)(ll_word<LOCAL>(g + lane) >> 32) == a.seq;
          ++st_polls;
          if (__all(ok)) break;
          for (int q = 0; q < a.backoff; ++q) __builtin_amdgcn_s_sleep(2);
          if (trsv_expired(a.err, t0, a.timeout_ticks, spins++)) { dead = true; break; }
        }
        st_gate += stats_ ? (unsigned long long)(wall_clock64() - tg0) : 0;
        if (dead) break;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc = fma_(av[u], xv[u], acc);
    }
    acc = wave_sum_dpp(acc);
    const unsigned long long tl_far = timeline_ ? (unsigned long long)wall_clock64() : 0;
    // the chain: entries of this chunk, oldest first, every lane in step
    const long long tn0 = stats_ ? wall_clock64() : 0;
    for (int k = m - 1; k >= 0 && !dead; --k) {
      const int li = __builtin_amdgcn_readlane(li_reg, k);
      const D av = lane_value(av_reg, k);
      while (__hip_atomic_load(&ready[li], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
        for (int q = 0; q < a.nap_lds; ++q) __builtin_amdgcn_s_sleep(1);  // the producer shares this CU's issue slots
        if (trsv_expired(a.err, t0, a.timeout_ticks, spins++)) { dead = true; break; }
      }
      acc = fma_(av, xs[li], acc);
    }
    st_near += stats_ ? (unsigned long long)(wall_clock64() - tn0) : 0;
    D x = sub_(b, acc);
    if (dinv) x = mul_(x, piv);
    if (lane == 0 && r != a.stall_row) {
      xs[wave] = x;
      __hip_atomic_store(&ready[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      ll_put<D, LOCAL>(a.sol + (size_t)r * W, x, a.seq);
      if (a.out) {
        const int64_t d = a.dst[r];
        if (d >= 0) static_cast<D*>(a.out)[d] = x;
      }
      if (timeline_) {
        unsigned long long* tl = timeline_ + 4 * (size_t)r;
        tl[0] = (unsigned long long)tk1; tl[1] = tl_far; tl[2] = (unsigned long long)wall_clock64(); tl[3] = (unsigned long long)m;
      }
    }
    if (stats_) { ++st_rows; st_row += (unsigned long long)(wall_clock64() - tk1); }
  }
  if (stats_) {
    unsigned long long v[3] = {st_hit, st_miss, st_coh};  // per-lane counters -> wave totals
    for (int k = 0; k < 3; ++k)
      for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    if (lane == 0) {
      const unsigned long long w[10] = {st_ticket, st_rows, st_row, st_gate, st_near, st_polls, st_att, v[0], v[1], v[2]};
      for (int k = 0; k < 10; ++k) atomicAdd(stats_ + k, w[k]);
      if (wave == 0) { atomicOr(stats_ + 10, 1ull << xcc_id()); atomicAdd(stats_ + 11, 1ull); }
    }
  }
}

// Pre-pass of a factor's SECOND launch: the entries of its rows that refer to rows of the first launch need no waiting
// (that launch is complete) and no particular XCD -- one wave per row over the whole device, plain cached loads.  What is
// left for the solve kernel of the part next to the root (one XCD: bandwidth of one XCD) is the part that really is a chain.
template <class D>
__global__ void __launch_bounds__(256) k_trsv_pre(int64_t row0, int64_t row1, const int64_t* __restrict__ rowptr, const int64_t* __restrict__ rowmid,
                                                  const int32_t* __restrict__ colind, const D* __restrict__ val, const uint64_t* __restrict__ sol, D* __restrict__ pre,
                                                  const DevState* __restrict__ st) {
  if (st && st->breakdown >= 0) return;
  constexpr int W = LLWords<D>::W;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = row0 + wave; r < row1; r += nwaves) {
    D acc = zero_of(D{});
    for (int64_t p = rowptr[r] + lane; p < rowmid[r]; p += 64) acc = fma_(val[p], ll_payload<D>(sol + (size_t)colind[p] * W), acc);
    acc = wave_sum_dpp(acc);
    if (lane == 0) pre[r] = acc;
  }
}

// which XCC ids exist on this device (bit mask)
__global__ void k_xcc_probe(unsigned* mask) {
  if (threadIdx.x == 0) atomicOr(mask, 1u << xcc_id());
}
// after a grouped launch: every group's tickets must have been handed out (a group whose XCD got no workgroup -- a device
// shared with something that occupies a whole XCD -- would otherwise go unnoticed)
__global__ void k_trsv_check(const int* words, const int* needed, int ngroups, int* err, const DevState* st) {
  if (st && st->breakdown >= 0) return;  // (the solve kernels skipped themselves: nothing was handed out)
  if (threadIdx.x == 0)
    for (int g = 0; g < ngroups; ++g)
      if (words[4 * g] < needed[g]) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace ksd

namespace {

inline double from_real_host(double v, double) { return v; }
inline ksd::cd from_real_host(double v, ksd::cd) { return ksd::cd{v, 0.0}; }
inline double inverse_host(double v) { return 1.0 / v; }
inline ksd::cd inverse_host(ksd::cd v) {
  const std::complex<double> r = 1.0 / std::complex<double>(v.x, v.y);
  return ksd::cd{r.real(), r.imag()};
}
inline std::complex<double> to_complex_host(double v) { return {v, 0.0}; }
inline std::complex<double> to_complex_host(ksd::cd v) { return {v.x, v.y}; }
inline double from_complex_host(std::complex<double> v, double) { return v.real(); }
inline ksd::cd from_complex_host(std::complex<double> v, ksd::cd) { return ksd::cd{v.real(), v.imag()}; }
inline bool is_zero_host(double v) { return v == 0.0; }
inline bool is_zero_host(ksd::cd v) { return v.x == 0.0 && v.y == 0.0; }

template <class D> struct TriFactor {
  int64_t nnz = 0;       // strictly triangular entries
  int64_t levels = 0;    // length of the longest dependency chain
  int64_t rows = 0;      // rows of the system the kernel solves (n + the rows of the dense runs)
  int64_t run_rows = 0;  // rows that belong to inverted dense runs
  int64_t stored = 0;    // entries stored for the kernel
  std::vector<int32_t> order, pos;  // host: kernel's row -> caller's row;  caller's row -> kernel's row that holds its solution entry
  std::vector<char> kind;           // host: 0 ordinary row, 1 right-hand-side row of a run, 2 solution row of a run
  // Launch structure.  The rows next to the root of the elimination tree ("top": narrow levels, dense runs) are one
  // launch on one XCD; what is left falls apart into independent parts (connected components of the dependency graph),
  // packed into `ngroups` groups of similar work, one XCD each, ONE launch for all of them.  Lower factor: groups, then
  // top; upper factor (solved from the root): top, then groups.
  struct Layer {                    // one launch of independent groups
    int ngroups = 0;
    int64_t gbeg[8] = {}, gend[8] = {};   // rows of the groups (kernel numbering)
    int64_t begin = 0, end = 0;           // all of them
    int* needed_d = nullptr;              // tickets each group has to hand out (k_trsv_check)
  };
  std::vector<Layer> layers;        // in LAUNCH order; empty: everything is "top" (one launch)
  int ngroups = 0;                  // groups of the outermost layer (reported)
  bool top_first = false;
  int64_t top_begin = 0, top_end = 0;   // rows of the top launch (kernel numbering)
  int64_t* rowmid = nullptr;        // per row: first entry that refers to a row of the SAME launch (before it: pre-pass)
  D* pre = nullptr;                 // pre-pass sums
  int64_t* rowptr = nullptr;
  int32_t* colind = nullptr;
  D* val = nullptr;
  D* diag = nullptr;     // inverse diagonal entries; nullptr: unit diagonal
  uint64_t* sol = nullptr;
  void release() {
    (void)hipFree(rowptr); (void)hipFree(colind); (void)hipFree(val); (void)hipFree(diag); (void)hipFree(sol); (void)hipFree(rowmid); (void)hipFree(pre);
    for (auto& l : layers) (void)hipFree(l.needed_d);
  }
};

// Split a triangular CSR factor into its strict part and its diagonal (checking the triangle), renumber rows and columns
// by dependency level, upload.
template <class D>
void upload_factor(TriFactor<D>& f, int64_t n, const int64_t* rp, const int32_t* ci, const D* vv, bool lower, const char* name, int want_groups) {
  std::vector<int64_t> srp((size_t)n + 1, 0);
  std::vector<int32_t> sci;
  std::vector<D> sv, dg((size_t)n, D{});
  std::vector<char> has((size_t)n, 0);
  sci.reserve((size_t)rp[n]);
  sv.reserve((size_t)rp[n]);
  bool any_diag = false;
  for (int64_t r = 0; r < n; ++r) {
    KS_REQUIRE(rp[r + 1] >= rp[r], KS_ERR_ARGUMENT, std::string(name) + ": row offsets must not decrease");
    for (int64_t p = rp[r]; p < rp[r + 1]; ++p) {
      const int64_t c = ci[p];
      KS_REQUIRE(c >= 0 && c < n, KS_ERR_ARGUMENT, std::string(name) + ": column index out of range");
      if (c == r) {
        KS_REQUIRE(!has[r], KS_ERR_ARGUMENT, std::string(name) + ": duplicate diagonal entry");
        has[r] = 1;
        dg[r] = vv[p];
        any_diag = true;
        continue;
      }
      KS_REQUIRE(lower ? c < r : c > r, KS_ERR_ARGUMENT, std::string(name) + (lower ? ": entry above the diagonal of the lower factor" : ": entry below the diagonal of the upper factor"));
      sci.push_back((int32_t)c);
      sv.push_back(vv[p]);
    }
    srp[r + 1] = (int64_t)sci.size();
  }
  const bool with_diag = any_diag || !lower;
  if (with_diag) {
    for (int64_t r = 0; r < n; ++r) {
      if (!has[r]) {
        KS_REQUIRE(lower, KS_ERR_ARGUMENT, std::string(name) + ": the upper factor needs every diagonal entry");
        dg[r] = from_real_host(1.0, D{});
      }
      KS_REQUIRE(!is_zero_host(dg[r]), KS_ERR_ARGUMENT, std::string(name) + ": zero on the diagonal (singular factor)");
    }
  }
  // dependency levels in elimination order (ascending rows for L, descending for U), then a stable counting sort
  std::vector<int32_t> lev((size_t)n, 0);
  int64_t top = 0;
  for (int64_t t = 0; t < n; ++t) {
    const int64_t r = lower ? t : n - 1 - t;
    int32_t l = 0;
    for (int64_t p = srp[r]; p < srp[r + 1]; ++p) l = std::max(l, lev[sci[p]] + 1);
    lev[r] = l;
    top = std::max<int64_t>(top, l + 1);
  }
  f.levels = n > 0 ? top : 0;
  // "top": the LAST K pivots (the separators next to the root of the elimination tree; the same rows in both factors).
  // Without them the rest of the factor falls apart into independent parts -- connected components of its dependency
  // graph -- which are packed by work into groups, one XCD each.  K: the smallest of 128, 256, 512, ... for which the
  // heaviest group stays under 1.75 / groups of the work (found by adding rows to a union-find from the largest K down).
  // The same is then done to the top part itself (its own last K' pivots stay, the rest falls apart again), up to four
  // layers: what is finally left for ONE XCD is the root separator and its nearest descendants.
  struct LayerPlan { int64_t lo, cut; int bins; std::vector<int32_t> bin; };  // pivots [lo, cut) in `bins` groups
  std::vector<LayerPlan> plan;
  if (want_groups > 1 && n >= 4096) {
    // edges keyed by their LARGER endpoint: the rows of the lower factor as they are, the columns of the upper one
    std::vector<int64_t> ep;
    std::vector<int32_t> ei;
    if (lower) { ep = srp; ei = sci; }
    else {
      ep.assign((size_t)n + 1, 0);
      for (int32_t c : sci) ep[(size_t)c + 1]++;
      for (int64_t r = 0; r < n; ++r) ep[r + 1] += ep[r];
      ei.resize(sci.size());
      std::vector<int64_t> fill(ep.begin(), ep.end() - 1);
      for (int64_t r = 0; r < n; ++r)
        for (int64_t p = srp[r]; p < srp[r + 1]; ++p) ei[fill[sci[p]]++] = (int32_t)r;
    }
    std::vector<int32_t> parent((size_t)n);
    auto find = [&](int32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    std::vector<double> work((size_t)n), load;
    std::vector<int32_t> comps, bin_of((size_t)n, 0);
    // (LuOp::kWords reserves ticket words for at most kLuMaxLayers layers per factor: ADVICE r3)
    const int max_layers = std::min(ksd::kLuMaxLayers, std::max(1, env_int("KS_LU_LAYERS", ksd::kLuMaxLayers)));
    int64_t lo = 0;
    for (int depth = 0; depth < max_layers && n - lo >= (depth == 0 ? 4096 : 1024); ++depth) {
      for (int64_t r = lo; r < n; ++r) parent[r] = (int32_t)r;
      std::vector<int64_t> cand;
      for (int64_t K = depth == 0 ? 128 : 64; K <= (n - lo) / 2; K *= 2) cand.push_back(K);
      int64_t added = lo, best_K = -1;  // rows [lo, added) are in the union-find
      int best_bins = 0;
      std::vector<int32_t> best_bin;
      for (size_t k = cand.size(); k-- > 0;) {
        const int64_t K = cand[k], nb = n - K;
        for (int64_t r = added; r < nb; ++r)
          for (int64_t p = ep[r]; p < ep[r + 1]; ++p)
            if (ei[p] >= lo) { const int32_t a = find((int32_t)r), b = find(ei[p]); if (a != b) parent[a] = b; }
        added = nb;
        std::fill(work.begin() + lo, work.begin() + nb, 0.0);
        double total = 0.0;
        for (int64_t r = lo; r < nb; ++r) { const double w = (double)(srp[r + 1] - srp[r]) + 8.0; work[find((int32_t)r)] += w; total += w; }
        comps.clear();
        for (int64_t r = lo; r < nb; ++r) if (parent[r] == r) comps.push_back((int32_t)r);
        std::sort(comps.begin(), comps.end(), [&](int32_t a, int32_t b) { return work[a] != work[b] ? work[a] > work[b] : a < b; });
        const int bins = std::min<int>(want_groups, (int)comps.size());
        load.assign((size_t)bins, 0.0);
        for (int32_t c : comps) {
          int best = 0;
          for (int b = 1; b < bins; ++b) if (load[b] < load[best]) best = b;
          bin_of[c] = best;
          load[best] += work[c];
        }
        const double worst = *std::max_element(load.begin(), load.end());
        // the outermost layer must fill the device; further in, the tree next to the root is binary: two parts are a gain
        const double limit = depth == 0 ? 1.75 / want_groups : std::max(1.75 / want_groups, 0.6);
        if (bins < 2 || worst > limit * total) break;  // (smaller K only merges parts)
        best_K = K;
        best_bins = bins;
        best_bin.assign((size_t)(nb - lo), 0);
        for (int64_t r = lo; r < nb; ++r) best_bin[r - lo] = bin_of[find((int32_t)r)];
      }
      if (best_K <= 0) break;
      plan.push_back(LayerPlan{lo, n - best_K, best_bins, std::move(best_bin)});
      lo = n - best_K;
    }
  }
  // segments in LAUNCH order: lower factor = layers outside in, then the top; upper factor = the top, then layers inside out
  std::vector<int32_t> seg((size_t)n, 0);
  int nseg = 1, top_seg = 0;
  std::vector<int> layer_first_seg(plan.size(), 0);  // per plan entry: id of its group 0
  if (!plan.empty()) {
    int total_groups = 0;
    for (auto& l : plan) total_groups += l.bins;
    nseg = total_groups + 1;
    if (lower) {
      int base = 0;
      for (size_t d = 0; d < plan.size(); ++d) { layer_first_seg[d] = base; base += plan[d].bins; }
      top_seg = total_groups;
    } else {
      int base = 1;
      for (size_t d = plan.size(); d-- > 0;) { layer_first_seg[d] = base; base += plan[d].bins; }
      top_seg = 0;
    }
    const int64_t top_lo = plan.back().cut;
    for (int64_t r = top_lo; r < n; ++r) seg[r] = top_seg;
    for (size_t d = 0; d < plan.size(); ++d)
      for (int64_t r = plan[d].lo; r < plan[d].cut; ++r) seg[r] = layer_first_seg[d] + plan[d].bin[r - plan[d].lo];
  }
  // numbering: by segment, inside a segment by level, ties in elimination order (stable sort of the elimination order)
  f.order.assign((size_t)n, 0);
  f.pos.assign((size_t)n, 0);
  for (int64_t t = 0; t < n; ++t) f.order[t] = (int32_t)(lower ? t : n - 1 - t);
  std::stable_sort(f.order.begin(), f.order.end(), [&](int32_t a, int32_t b) { return seg[a] != seg[b] ? seg[a] < seg[b] : lev[a] < lev[b]; });
  for (int64_t i = 0; i < n; ++i) f.pos[f.order[i]] = (int32_t)i;
  std::vector<int64_t> segb((size_t)nseg + 1, 0);  // segment boundaries in this numbering
  for (int64_t r = 0; r < n; ++r) segb[seg[r] + 1]++;
  for (int k = 0; k < nseg; ++k) segb[k + 1] += segb[k];
  // the factor in its own numbering: row i = caller's row order[i], columns pos[c], ascending
  std::vector<int64_t> nrp((size_t)n + 1, 0);
  std::vector<int32_t> nci(sci.size());
  std::vector<D> nv(sv.size());
  std::vector<std::pair<int32_t, int64_t>> tmp;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = f.order[i];
    tmp.clear();
    for (int64_t p = srp[r]; p < srp[r + 1]; ++p) tmp.emplace_back(f.pos[sci[p]], p);
    std::sort(tmp.begin(), tmp.end());
    int64_t q = nrp[i];
    for (auto& e : tmp) {
      nci[q] = e.first;
      nv[q] = sv[e.second];
      ++q;
    }
    nrp[i + 1] = q;
  }
  // ---- dense runs -------------------------------------------------------------------------------------------------
  // Where the levels are narrow (the dense triangles of the top separators: one or two rows per level, every row depending
  // on all rows before it) substitution is a chain: measured 0.36 us per row through LDS plus 2.3 us per 16-row chunk
  // through memory, i.e. 0.48 us per row, and nearly half of a product.  There the rows are cut into runs of up to R
  // consecutive rows and the triangle T of each run (its diagonal and the entries between its rows) is INVERTED here, once:
  //     x_run = T^-1 s_run,     s_run = b_run - (entries outside the run) x
  // which is again a sparse triangular system, in twice the unknowns of the run: the rows of s depend on earlier x only,
  // the rows of x on the s of their run only (entries -T^-1).  The kernel solves the augmented system as it stands: the
  // chain through a run of R rows is two hand-overs through memory instead of R.  (Inverted diagonal blocks are what
  // dense triangular solves do on GPUs; the factors of a pivoted LU keep the inverse of such a block tame.)
  // A run is only taken if its inverse is tame: kappa = max|T^-1| * max|T| <= KS_LU_RUN_COND (default 1e4).  The factors
  // of an UNPIVOTED factorisation of an indefinite matrix (shift-invert with diagonal pivots) can have triangles whose
  // inverses grow; multiplying by such an inverse costs digits that substitution keeps (measured on a real, nearly
  // singular 2-D problem: residual 9e-9 without runs, 6e-7 with runs of 256, the host solve 7e-8).  A run that fails is
  // halved, and halved again; what is left under 16 rows stays a chain.
  const int64_t R = std::max(0, env_int("KS_LU_RUN", 256));
  const int64_t narrow = std::max(1, env_int("KS_LU_NARROW", 64));
  const double kappa_max = (double)std::max(1, env_int("KS_LU_RUN_COND", 10000));
  using cplx_t = std::complex<double>;
  auto to_c = [](const D& v) { return to_complex_host(v); };
  struct Run { int64_t a, b; std::vector<cplx_t> Ti; };
  std::vector<Run> runs;                        // in increasing row order
  std::vector<char> in_run((size_t)n, 0);       // level-numbered row is part of a run
  double kappa_seen = 0.0;
  int64_t rejected = 0;
  auto invert = [&](int64_t a, int64_t b, std::vector<cplx_t>& Ti) {
    const int64_t r = b - a;
    std::vector<cplx_t> T((size_t)(r * r), cplx_t(0.0, 0.0));
    double tmax = 0.0, imax = 0.0;
    for (int64_t q = a; q < b; ++q) {
      T[(size_t)((q - a) * r + (q - a))] = with_diag ? to_c(dg[f.order[q]]) : cplx_t(1.0, 0.0);
      for (int64_t p = nrp[q]; p < nrp[q + 1]; ++p)
        if (nci[p] >= a) T[(size_t)((q - a) * r + (nci[p] - a))] = to_c(nv[p]);
    }
    for (const cplx_t& v : T) tmax = std::max(tmax, std::abs(v));
    // Ti = T^-1, column by column (forward substitution on the identity)
    Ti.assign((size_t)(r * r), cplx_t(0.0, 0.0));
    for (int64_t c = 0; c < r; ++c) {
      Ti[(size_t)(c * r + c)] = cplx_t(1.0, 0.0) / T[(size_t)(c * r + c)];
      for (int64_t q = c + 1; q < r; ++q) {
        cplx_t acc(0.0, 0.0);
        const cplx_t* trow = &T[(size_t)(q * r)];
        for (int64_t t = c; t < q; ++t)
          if (trow[t] != cplx_t(0.0, 0.0)) acc += trow[t] * Ti[(size_t)(t * r + c)];
        Ti[(size_t)(q * r + c)] = -acc / trow[q];
      }
    }
    for (const cplx_t& v : Ti) imax = std::max(imax, std::abs(v));
    return tmax * imax;
  };
  std::function<void(int64_t, int64_t)> take = [&](int64_t a, int64_t b) {
    if (b - a < 16) return;
    std::vector<cplx_t> Ti;
    const double kappa = invert(a, b, Ti);
    if (kappa <= kappa_max) {
      kappa_seen = std::max(kappa_seen, kappa);
      for (int64_t q = a; q < b; ++q) in_run[q] = 1;
      runs.push_back(Run{a, b, std::move(Ti)});
      return;
    }
    ++rejected;
    const int64_t mid = (a + b) / 2;
    take(a, mid);
    take(mid, b);
  };
  if (R >= 2) {
    // in every segment (the top part and each group has its own narrow end: its sub-separators)
    for (int sg = 0; sg < nseg; ++sg) {
      const int64_t ta = segb[sg], tb = segb[sg + 1];
      auto level_end = [&](int64_t i) { int64_t e = i; const int32_t l = lev[f.order[i]]; while (e < tb && lev[f.order[e]] == l) ++e; return e; };
      int64_t i = ta;
      while (i < tb) {
        int64_t j = i;  // a stretch of narrow levels starting at row i (the first row of its level)
        while (j < tb) {
          const int64_t e = level_end(j);
          if (e - j > narrow) break;
          j = e;
        }
        if (j - i >= 32) {
          for (int64_t a = i; a < j; a += R) take(a, std::min(a + R, j));
          i = j;
        } else {
          i = j > i ? j : level_end(i);
        }
      }
    }
  }
  if (env_int("KS_LU_STATS", 0))
    std::fprintf(stderr, "[lu %s] %zu dense runs inverted (largest max|T^-1| max|T| = %.3g), %lld pieces refused and halved (limit %.3g)\n", lower ? "L" : "U", runs.size(), kappa_seen, (long long)rejected, kappa_max);
  // augmented numbering: a row outside the runs keeps one unknown; a run of r rows becomes r rows of s, then r rows of x
  std::vector<int32_t> xidx((size_t)n, 0);      // level-numbered row -> augmented index of its x
  std::vector<int32_t> aug_owner;               // augmented row -> level-numbered row
  std::vector<char> aug_kind;                   // 0: ordinary row, 1: s row, 2: x row
  std::vector<int32_t> aug_run;                 // augmented row -> index of its run (-1: none)
  std::vector<int64_t> seg_aug((size_t)nseg + 1, 0);  // segment boundaries in the augmented numbering
  {
    size_t rb = 0;
    int64_t i = 0;
    int sgi = 0;
    while (i < n) {
      while (sgi <= nseg && segb[sgi] <= i) seg_aug[sgi++] = (int64_t)aug_owner.size();
      if (!in_run[i]) {
        xidx[i] = (int32_t)aug_owner.size();
        aug_owner.push_back((int32_t)i);
        aug_kind.push_back(0);
        aug_run.push_back(-1);
        ++i;
        continue;
      }
      while (runs[rb].a < i) ++rb;
      const int64_t a = runs[rb].a, b = runs[rb].b;
      for (int64_t q = a; q < b; ++q) { aug_owner.push_back((int32_t)q); aug_kind.push_back(1); aug_run.push_back((int32_t)rb); }
      for (int64_t q = a; q < b; ++q) { xidx[q] = (int32_t)aug_owner.size(); aug_owner.push_back((int32_t)q); aug_kind.push_back(2); aug_run.push_back((int32_t)rb); }
      i = b;
      ++rb;
    }
    while (sgi <= nseg) seg_aug[sgi++] = (int64_t)aug_owner.size();
  }
  const int64_t N = (int64_t)aug_owner.size();
  KS_REQUIRE(N < (int64_t)2147483647, KS_ERR_ARGUMENT, std::string(name) + ": too many rows");
  std::vector<int64_t> arp((size_t)N + 1, 0);
  std::vector<int32_t> aci;
  std::vector<D> av, adg((size_t)N, from_real_host(1.0, D{}));
  aci.reserve(nci.size() + (size_t)(N - n) * 8);
  av.reserve(nci.size() + (size_t)(N - n) * 8);
  for (int64_t k = 0; k < N; ++k) {
    const int64_t i = aug_owner[k];
    if (aug_kind[k] == 0) {
      for (int64_t p = nrp[i]; p < nrp[i + 1]; ++p) { aci.push_back(xidx[nci[p]]); av.push_back(nv[p]); }
      if (with_diag) adg[k] = inverse_host(dg[f.order[i]]);
    } else if (aug_kind[k] == 1) {
      // s row: the entries outside the run, unit pivot
      const int64_t run_a = runs[aug_run[k]].a;
      for (int64_t p = nrp[i]; p < nrp[i + 1]; ++p)
        if (nci[p] < run_a) { aci.push_back(xidx[nci[p]]); av.push_back(nv[p]); }
    } else {
      // x row: x_i = sum_j Ti[i][j] s_j  ->  entries -Ti on the s rows of the run (they sit r rows before the x rows)
      const Run& rn = runs[aug_run[k]];
      const int64_t r = rn.b - rn.a, q = i - rn.a;
      const int32_t s0 = xidx[rn.a] - (int32_t)r;
      for (int64_t t = 0; t <= q; ++t) {
        const cplx_t w = rn.Ti[(size_t)(q * r + t)];
        if (w != cplx_t(0.0, 0.0)) { aci.push_back(s0 + (int32_t)t); av.push_back(from_complex_host(-w, D{})); }
      }
    }
    arp[k + 1] = (int64_t)aci.size();
  }
  // order/pos now describe the augmented numbering: order[k] = caller's row, kind[k], pos[caller's row] = index of its x
  {
    std::vector<int32_t> o2((size_t)N), p2((size_t)n);
    for (int64_t k = 0; k < N; ++k) o2[k] = f.order[aug_owner[k]];
    for (int64_t i = 0; i < n; ++i) p2[f.order[i]] = xidx[i];
    f.order.swap(o2);
    f.pos.swap(p2);
    f.kind.assign(aug_kind.begin(), aug_kind.end());
  }
  f.rows = N;
  f.run_rows = N - n;
  f.ngroups = plan.empty() ? 0 : plan.front().bins;
  f.top_first = !lower;
  f.top_begin = seg_aug[top_seg];
  f.top_end = seg_aug[top_seg + 1];
  if (!plan.empty()) {
    // layers in launch order
    std::vector<size_t> order_d;
    if (lower) for (size_t d = 0; d < plan.size(); ++d) order_d.push_back(d);
    else for (size_t d = plan.size(); d-- > 0;) order_d.push_back(d);
    for (size_t d : order_d) {
      typename TriFactor<D>::Layer L;
      L.ngroups = plan[d].bins;
      std::vector<int> needed((size_t)L.ngroups);
      for (int g = 0; g < L.ngroups; ++g) {
        const int sgi = layer_first_seg[d] + g;
        L.gbeg[g] = seg_aug[sgi];
        L.gend[g] = seg_aug[sgi + 1];
        needed[g] = (int)((L.gend[g] - L.gbeg[g] + ksd::kTrsvWaves - 1) / ksd::kTrsvWaves);
      }
      L.begin = L.gbeg[0];
      L.end = L.gend[L.ngroups - 1];
      KS_HIP(hipMalloc(&L.needed_d, (size_t)L.ngroups * sizeof(int)));
      KS_HIP(hipMemcpy(L.needed_d, needed.data(), (size_t)L.ngroups * sizeof(int), hipMemcpyHostToDevice));
      f.layers.push_back(L);
    }
    if (env_int("KS_LU_PREPASS", 1)) {
      // every launch but the first: the entries that refer to rows of EARLIER launches (complete by then) are summed by a
      // pre-pass; rowmid[r] = the first entry of row r that refers to its own launch
      std::vector<int64_t> mid(arp.begin(), arp.end());
      auto mark = [&](int64_t b, int64_t e) {
        for (int64_t r = b; r < e; ++r) mid[r] = std::lower_bound(aci.begin() + arp[r], aci.begin() + arp[r + 1], (int32_t)b) - aci.begin();
      };
      for (auto& L : f.layers) mark(L.begin, L.end);
      mark(f.top_begin, f.top_end);
      KS_HIP(hipMalloc(&f.rowmid, ((size_t)N + 1) * 8));
      KS_HIP(hipMemcpy(f.rowmid, mid.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice));
      KS_HIP(hipMalloc(&f.pre, (size_t)N * sizeof(D)));
      KS_HIP(hipMemset(f.pre, 0, (size_t)N * sizeof(D)));
    }
  }
  f.nnz = (int64_t)nci.size();
  const bool any_pivot = with_diag || N > n;
  KS_HIP(hipMalloc(&f.rowptr, ((size_t)N + 1) * 8));
  KS_HIP(hipMalloc(&f.colind, std::max<size_t>(aci.size(), 1) * 4));
  KS_HIP(hipMalloc(&f.val, std::max<size_t>(av.size(), 1) * sizeof(D)));
  KS_HIP(hipMemcpy(f.rowptr, arp.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice));
  if (!aci.empty()) {
    KS_HIP(hipMemcpy(f.colind, aci.data(), aci.size() * 4, hipMemcpyHostToDevice));
    KS_HIP(hipMemcpy(f.val, av.data(), av.size() * sizeof(D), hipMemcpyHostToDevice));
  }
  f.stored = (int64_t)aci.size();
  if (any_pivot) {
    KS_HIP(hipMalloc(&f.diag, (size_t)N * sizeof(D)));
    KS_HIP(hipMemcpy(f.diag, adg.data(), (size_t)N * sizeof(D), hipMemcpyHostToDevice));
  }
  const size_t words = (size_t)N * ksd::LLWords<D>::W;
  KS_HIP(hipMalloc(&f.sol, std::max<size_t>(words, 1) * 8));
  KS_HIP(hipMemset(f.sol, 0, std::max<size_t>(words, 1) * 8));  // sequence number 0 is never used by a solve
}

template <class D> struct LuOp : ks_operator {
  TriFactor<D> L, U;
  int32_t* src_l = nullptr;  // row i of L's numbering takes x[src_l[i]]
  double* scale_l = nullptr; //   ... times scale_l[i]
  int32_t* src_u = nullptr;  // row i of U's numbering takes entry src_u[i] of L's solution
  int32_t* dst_u = nullptr;  //   ... and its result goes to y[dst_u[i]]
  static constexpr int kWords = 4 + ksd::kLuMaxLayers * 32;  // control words per factor: 4 of the top launch + 4 per group, up to 4 layers of 8
  int* tickets = nullptr;
  int xcc_group[16];         // XCC id -> group (probed once; 127: no such XCC)
  int nxcc = 1;
  int local = 3;             // form of the tail launch (KS_LU_XCD): 3 = one XCD, stores to its L2; 4 = one XCD, stores through; 0 = all XCDs
  int* err_d = nullptr;      // the context's pinned error word (checked at every synchronisation point of the context)
  uint32_t seq = 0;
  int grid = 0, grid_groups = 256, backoff = 2, nap_lds = 1;
  int64_t stall_row = -1;
  long long timeout_ticks = 0;
  unsigned long long* stats = nullptr;  // KS_LU_STATS=1
  unsigned long long* timeline = nullptr;  // KS_LU_TIMELINE=<file prefix>: 2 x 4 x tl_rows
  int64_t tl_rows = 0;
  ~LuOp() override {
    L.release(); U.release();
    (void)hipFree(src_l); (void)hipFree(scale_l); (void)hipFree(src_u); (void)hipFree(dst_u); (void)hipFree(tickets);
    (void)hipFree(stats);
    (void)hipFree(timeline);
  }
  // Top part: one launch on one XCD (or, KS_LU_XCD=0, on 64 workgroups anywhere: fewer waiting waves are faster there,
  // each polls through the fabric -- with more than ~64 workgroups the all-XCD form collapses to 15-65 ms per product).
  void launch_rows(ksd::TrsvArgs a, int64_t row0, int64_t row1, int* words) {
    if (row1 <= row0) return;
    a.row0 = row0; a.n = row1; a.ticket = words;
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>((row1 - row0 + ksd::kTrsvWaves - 1) / ksd::kTrsvWaves, grid));
    const bool probe = a.stats || a.timeline;
    switch (local) {
      case 3:
        if (probe) ksd::k_sptrsv<D, 3, true><<<g * 8, ksd::kTrsvWaves * 64, 0, ctx->stream>>>(a);
        else ksd::k_sptrsv<D, 3, false><<<g * 8, ksd::kTrsvWaves * 64, 0, ctx->stream>>>(a);
        break;
      case 4: ksd::k_sptrsv<D, 4, false><<<g * 8, ksd::kTrsvWaves * 64, 0, ctx->stream>>>(a); break;
      default: ksd::k_sptrsv<D, 0, false><<<g, ksd::kTrsvWaves * 64, 0, ctx->stream>>>(a);
    }
    KS_HIP(hipGetLastError());
  }
  // Independent parts: ONE launch per layer, every XCD works on its own group (one 1024-thread workgroup per CU).
  void launch_groups(ksd::TrsvArgs a, const typename TriFactor<D>::Layer& L, int* words) {
    for (int g = 0; g < 8; ++g) { a.gbeg[g] = g < L.ngroups ? L.gbeg[g] : 0; a.gend[g] = g < L.ngroups ? L.gend[g] : 0; }
    for (int k = 0; k < 16; ++k) a.xcc_group[k] = (signed char)(xcc_group[k] < L.ngroups ? xcc_group[k] : -1);
    a.ticket = words;
    if (a.stats || a.timeline) ksd::k_sptrsv<D, 5, true><<<grid_groups, ksd::kTrsvWaves * 64, 0, ctx->stream>>>(a);
    else ksd::k_sptrsv<D, 5, false><<<grid_groups, ksd::kTrsvWaves * 64, 0, ctx->stream>>>(a);
    KS_HIP(hipGetLastError());
    ksd::k_trsv_check<<<1, 64, 0, ctx->stream>>>(words, L.needed_d, L.ngroups, err_d, a.st);
    KS_HIP(hipGetLastError());
  }
  void prepass(const TriFactor<D>& f, int64_t b, int64_t e, const ksd::DevState* st) {
    if (e <= b) return;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((e - b + 3) / 4, (int64_t)ctx->num_cu * 8));
    ksd::k_trsv_pre<D><<<nb, 256, 0, ctx->stream>>>(b, e, f.rowptr, f.rowmid, f.colind, f.val, f.sol, f.pre, st);
    KS_HIP(hipGetLastError());
  }
  // words: 4 control words for the top launch, then 32 per layer
  void solve(ksd::TrsvArgs a, const TriFactor<D>& f, int* words) {
    a.rowbegin = f.rowptr;
    a.pre = nullptr;
    if (f.layers.empty()) { launch_rows(a, 0, f.rows, words); return; }
    KS_REQUIRE((int)f.layers.size() <= ksd::kLuMaxLayers, KS_ERR_INTERNAL, "more layers than the control words of a factor provide for");
    ksd::TrsvArgs a2 = a;  // every launch but the first sums only what its pre-pass left
    if (f.rowmid) { a2.rowbegin = f.rowmid; a2.pre = f.pre; }
    bool first = true;
    auto top = [&] {
      if (!first && f.rowmid) prepass(f, f.top_begin, f.top_end, a.st);
      launch_rows(first ? a : a2, f.top_begin, f.top_end, words);
      first = false;
    };
    if (f.top_first) top();
    for (size_t k = 0; k < f.layers.size(); ++k) {
      if (!first && f.rowmid) prepass(f, f.layers[k].begin, f.layers[k].end, a.st);
      launch_groups(first ? a : a2, f.layers[k], words + 4 + 32 * (int)k);
      first = false;
    }
    if (!f.top_first) top();
  }
  void apply(const void* x, void* y, const DevState* st) override {
    ctx->check_comm();  // (of earlier products: the word is written by the device)
    ProfScope ps(ctx, KSP_SPMV, (double)(L.stored + U.stored) * (sizeof(D) + 4.0 + 8.0 * ksd::LLWords<D>::W) + (double)n_local * (4.0 * sizeof(D) + 2.0 * 8.0 + 3.0 * 4.0 + 3.0 * 8.0 * ksd::LLWords<D>::W));
    if (++seq == 0) {  // 2^32 solves: start the sequence numbers over
      KS_HIP(hipMemsetAsync(L.sol, 0, (size_t)L.rows * ksd::LLWords<D>::W * 8, ctx->stream));
      KS_HIP(hipMemsetAsync(U.sol, 0, (size_t)U.rows * ksd::LLWords<D>::W * 8, ctx->stream));
      seq = 1;
    }
    ksd::TrsvArgs a{};
    a.st = st;
    a.n = n_local;
    a.err = err_d;
    a.timeout_ticks = timeout_ticks;
    a.seq = seq;
    a.backoff = backoff;
    a.nap_lds = nap_lds;
    a.stall_row = stall_row;
    a.stats = stats;
    if (stats) KS_HIP(hipMemsetAsync(stats, 0, 24 * 8, ctx->stream));
    // L z = P_in (s o x)
    a.rowptr = L.rowptr; a.colind = L.colind; a.val = L.val; a.diag = L.diag; a.sol = L.sol;
    a.rhs = x; a.src = src_l; a.scale = scale_l;
    KS_HIP(hipMemsetAsync(tickets, 0, 2 * kWords * sizeof(int), ctx->stream));
    a.timeline = timeline;
    solve(a, L, tickets);
    // U w = z;  y[perm_out] = w
    a.rowptr = U.rowptr; a.colind = U.colind; a.val = U.val; a.diag = U.diag; a.sol = U.sol;
    a.rhs = nullptr; a.rhs_ll = L.sol; a.src = src_u; a.scale = nullptr;
    a.out = y; a.dst = dst_u;
    if (stats) a.stats = stats + 12;
    const size_t nchunks = 4 * (size_t)tl_rows;
    if (timeline) a.timeline = timeline + nchunks;
    solve(a, U, tickets + kWords);
    if (timeline) {
      std::vector<unsigned long long> h(2 * nchunks);
      KS_HIP(hipMemcpyAsync(h.data(), timeline, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
      KS_HIP(hipStreamSynchronize(ctx->stream));
      const std::string path = std::string(std::getenv("KS_LU_TIMELINE")) + ".u64";
      if (FILE* fp = std::fopen(path.c_str(), "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
    }
    if (stats) {
      unsigned long long h[24];
      KS_HIP(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
      KS_HIP(hipStreamSynchronize(ctx->stream));
      for (int f = 0; f < 2; ++f) {
        const unsigned long long* q = h + 12 * f;
        const double rows = (double)std::max<unsigned long long>(q[1], 1);
        std::fprintf(stderr, "[lu %s] per row: ticket %.2f us, row %.2f us (gate %.2f, lds-only %.2f); gate polls %.1f, attempts %.1f; entries per row: cached hit %.1f miss %.1f, coherent tries %.1f; %llu workgroups on XCDs 0x%llx\n",
                     f == 0 ? "L" : "U", 0.01 * q[0] / rows, 0.01 * q[2] / rows, 0.01 * q[3] / rows, 0.01 * q[4] / rows, q[5] / rows, q[6] / rows, q[7] / rows, q[8] / rows, q[9] / rows, q[11], q[10]);
      }
    }
  }
};

template <class D>
ks_operator* make_lu(ks_ctx* ctx, int64_t n, const int64_t* lrp, const int32_t* lci, const void* lv, const int64_t* urp,
                     const int32_t* uci, const void* uv, const int32_t* pin, const int32_t* pout, const double* sc) {
  auto op = std::make_unique<LuOp<D>>();
  op->ctx = ctx;
  op->n_local = n;
  op->dtype = sizeof(D) == 8 ? KS_F64 : KS_C64;
  for (const int32_t* p : {pin, pout}) {
    if (!p) continue;
    std::vector<char> seen((size_t)n, 0);
    for (int64_t i = 0; i < n; ++i) {
      KS_REQUIRE(p[i] >= 0 && p[i] < n && !seen[p[i]], KS_ERR_ARGUMENT, std::string("ks_operator_lu: ") + (p == pin ? "perm_in" : "perm_out") + " is not a permutation of 0..n-1");
      seen[p[i]] = 1;
    }
  }
  {
    // which XCDs does this device have?  (8 on an unpartitioned MI355X; the groups are dealt to the ids that answer)
    unsigned* mask_d = nullptr;
    unsigned mask = 0;
    KS_HIP(hipMalloc(&mask_d, sizeof(unsigned)));
    KS_HIP(hipMemsetAsync(mask_d, 0, sizeof(unsigned), ctx->stream));
    ksd::k_xcc_probe<<<ctx->num_cu * 8, 64, 0, ctx->stream>>>(mask_d);
    KS_HIP(hipGetLastError());
    KS_HIP(hipMemcpyAsync(&mask, mask_d, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(mask_d);
    op->nxcc = 0;
    for (int k = 0; k < 16; ++k) op->xcc_group[k] = (mask >> k & 1u) && op->nxcc < 8 ? op->nxcc++ : 127;
  }
  const int want_groups = std::min(op->nxcc, std::max(1, env_int("KS_LU_GROUPS", 8)));
  upload_factor<D>(op->L, n, lrp, lci, static_cast<const D*>(lv), true, "ks_operator_lu: L", want_groups);
  upload_factor<D>(op->U, n, urp, uci, static_cast<const D*>(uv), false, "ks_operator_lu: U", want_groups);
  op->nnz = op->L.nnz + op->U.nnz + n;
  // index arrays between the caller's vectors and the rows of the two systems the kernel solves
  const int64_t nl = op->L.rows, nu = op->U.rows;
  std::vector<int32_t> sl((size_t)nl), su((size_t)nu), du((size_t)nu);
  std::vector<double> scl;
  if (sc) scl.assign((size_t)nl, 1.0);
  for (int64_t i = 0; i < nl; ++i) {
    const int32_t rl = op->L.order[i];  // row of the triangular system
    if (op->L.kind[i] == 2) { sl[i] = -1; continue; }
    sl[i] = pin ? pin[rl] : rl;
    if (sc) scl[i] = sc[sl[i]];
  }
  for (int64_t i = 0; i < nu; ++i) {
    const int32_t ru = op->U.order[i];
    su[i] = op->U.kind[i] == 2 ? -1 : op->L.pos[ru];
    du[i] = op->U.kind[i] == 1 ? -1 : (pout ? pout[ru] : ru);
  }
  auto up = [&](const void* h, size_t bytes, void** d) {
    KS_HIP(hipMalloc(d, std::max<size_t>(bytes, 8)));
    KS_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
  };
  up(sl.data(), (size_t)nl * 4, (void**)&op->src_l);
  up(su.data(), (size_t)nu * 4, (void**)&op->src_u);
  up(du.data(), (size_t)nu * 4, (void**)&op->dst_u);
  if (sc) up(scl.data(), (size_t)nl * 8, (void**)&op->scale_l);
  std::vector<int32_t>().swap(op->L.order); std::vector<int32_t>().swap(op->L.pos);
  std::vector<int32_t>().swap(op->U.order); std::vector<int32_t>().swap(op->U.pos);
  std::vector<char>().swap(op->L.kind); std::vector<char>().swap(op->U.kind);
  KS_HIP(hipMalloc(&op->tickets, 2 * LuOp<D>::kWords * sizeof(int)));
  KS_HIP(hipMemset(op->tickets, 0, 2 * LuOp<D>::kWords * sizeof(int)));
  op->err_d = ctx->operr_dev();
  // two 1024-thread workgroups per CU of one XCD (launched 8x over, KS_LU_XCD=3/4) or 64 workgroups anywhere (=0: fewer
  // waiting waves are faster there, each polls through the fabric)
  op->local = env_int("KS_LU_XCD", 3);
  op->grid = std::max(1, env_int("KS_LU_GRID", op->local ? std::max(1, ctx->num_cu / 8) * 2 : 64));
  op->grid_groups = std::max(8, env_int("KS_LU_GRID_GROUPS", ctx->num_cu));  // one workgroup per CU
  op->backoff = std::max(0, env_int("KS_LU_BACKOFF", 2));  // nap between two polls of a missing entry, x 128 clocks
  op->nap_lds = std::max(0, env_int("KS_LU_NAP_LDS", 1));
  op->stall_row = env_int("KS_LU_INJECT_STALL", -1);  // tests: the bounded waits must end in KS_ERR_OPERATOR, not in a hung device
  op->timeout_ticks = (long long)env_int("KS_LU_TIMEOUT_S", 20) * 100000000LL;
  if (std::getenv("KS_LU_TIMELINE")) {
    op->tl_rows = std::max(op->L.rows, op->U.rows);
    const size_t nchunks = 4 * (size_t)op->tl_rows;
    KS_HIP(hipMalloc(&op->timeline, 2 * nchunks * 8));
    KS_HIP(hipMemset(op->timeline, 0, 2 * nchunks * 8));
  }
  if (env_int("KS_LU_STATS", 0)) {
    KS_HIP(hipMalloc(&op->stats, 24 * 8));
    KS_HIP(hipMemset(op->stats, 0, 24 * 8));
  }
  return op.release();
}

}  // namespace
