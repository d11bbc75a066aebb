// Peer-to-peer transport for the row-partitioned solver (one process per GPU).
//
// The collectives of the Arnoldi hot path are latency-bound: two sums of <= maxdim+2 doubles per step
// (src/expansion.jl:37,46 are the reference's single-process gemv's that become reductions once V is
// split by rows) and one exchange of ghost entries of x before the SpMV.  Instead of a library
// collective per exchange, every rank owns one UNCACHED, IPC-shared region that all peers map over xGMI:
//
//   region = [ LL words : 2 slots x nranks senders x cap elements x 2 words ]
//            [ halo flags: 2 slots x nranks senders                          ]
//            [ ghost arena: operators carve their (double-buffered) ghost vectors out of it ]
//
// Reduction ("LL" protocol): a double is split into two 32-bit halves, each stored together with a 32-bit
// sequence number in ONE 8-byte store -- 8-byte stores are single-copy atomic over xGMI, so data and flag
// arrive together and no fence or separate flag is needed.  Rank r pushes its value to every peer
// (including itself) and then polls its own region until all nranks contributions of that sequence
// number have landed; the contributions are added in rank order 0..nranks-1, so every rank obtains the
// bit-identical sum (the DGKS branches of src/expansion.jl:91,99 must agree across ranks).
// Sequence numbers are kept per element ON THE DEVICE and advance only when an exchange really runs
// (kernels of a batch exit early after a Krylov breakdown), which is what makes two slots sufficient:
// a rank can only start exchange k+2 of an element after finishing k+1, which needed the peer's k+1
// push, which the peer issues in a kernel that runs after its kernel of exchange k completed.
//
// Halo: the sender writes its boundary entries straight into the neighbour's ghost slot (remote stores),
// fences at system scope, and the last workgroup raises the neighbour's flag; it then waits for the
// flags of the ranks it receives from.  The following SpMV kernel reads the ghosts from uncached memory.
//
// Every spin is bounded (wall clock) and reports through a pinned host flag: a lost peer produces
// KS_ERR_COMM at the next synchronisation instead of a hung GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ksd {

constexpr int kP2pMaxRanks = 16;
constexpr int kP2pMaxNeigh = 16;

struct P2pDev {
  uint64_t* region[kP2pMaxRanks];  // region[q]: rank q's region as mapped HERE (region[rank] is the local one)
  uint32_t* seqc;                  // [cap] per-element sequence counters (plain local memory)
  int* err;                        // pinned host word: != 0 once any spin timed out
  int rank, nranks, cap;
  long long timeout_ticks;         // wall_clock64() ticks (100 MHz)
};

__device__ __forceinline__ size_t ll_index(const P2pDev& p, int slot, int sender, int elem) {
  return (((size_t)slot * p.nranks + sender) * p.cap + elem) * 2;
}
__host__ __device__ inline size_t p2p_ll_words(int nranks, int cap) { return (size_t)2 * nranks * cap * 2; }
__host__ __device__ inline size_t p2p_flag_words(int nranks) { return (size_t)2 * nranks; }

__device__ __forceinline__ void ll_store(uint64_t* p, uint32_t data, uint32_t seq) {
  __hip_atomic_store(p, ((uint64_t)seq << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t ll_load(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ bool p2p_expired(const P2pDev& p, long long t0, long spins) {
  if ((spins & 1023) != 1023) return false;
  if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return true;  // someone already gave up
  if (wall_clock64() - t0 > p.timeout_ticks) {
    __hip_atomic_store(p.err, 1 + p.rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
  return false;
}

// Sum over ranks of NV consecutive elements elem0..elem0+NV-1.  Must be called by all 64 lanes of ONE wave
// (lane = vi * 16 + q handles value vi and peer q); every lane returns all NV sums.
template <int NV>
__device__ __forceinline__ void p2p_sum_wave(const P2pDev& p, int elem0, const double (&x)[NV], double (&out)[NV]) {
  static_assert(NV >= 1 && NV <= 64 / kP2pMaxRanks, "at most 4 values per wave");
  const int lane = threadIdx.x & 63;
  const int vi = lane / kP2pMaxRanks, q = lane % kP2pMaxRanks;
  const bool active = vi < NV && q < p.nranks;
  double mine = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) mine = (vi == i) ? x[i] : mine;
  const int elem = elem0 + (vi < NV ? vi : 0);
  uint32_t seq = 0;
  if (vi < NV && q == 0) {
    seq = p.seqc[elem] + 1u;
    if (seq == 0u) seq = 1u;
    p.seqc[elem] = seq;
  }
  seq = __shfl(seq, vi * kP2pMaxRanks);
  const int slot = seq & 1u;
  double got = 0.0;
  if (active) {
    const uint64_t bits = (uint64_t)__double_as_longlong(mine);
    uint64_t* dst = p.region[q] + ll_index(p, slot, p.rank, elem);
    ll_store(dst, (uint32_t)bits, seq);
    ll_store(dst + 1, (uint32_t)(bits >> 32), seq);
    const uint64_t* src = p.region[p.rank] + ll_index(p, slot, q, elem);
    const long long t0 = wall_clock64();
    long spins = 0;
    uint64_t lo, hi;
    for (;;) {
      lo = ll_load(src);
      hi = ll_load(src + 1);
      if ((uint32_t)(lo >> 32) == seq && (uint32_t)(hi >> 32) == seq) break;
      if (p2p_expired(p, t0, spins++)) { lo = 0; hi = 0x7ff80000u; break; }  // NaN marks the failure
      // (polling uncached memory goes out to the memory side every time: a little backoff keeps a thousand polling
      // lanes from saturating the one or two channels the LL window lives in)
      __builtin_amdgcn_s_sleep(3);
    }
    got = __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = 0.0;
    for (int r = 0; r < p.nranks; ++r) s += __shfl(got, i * kP2pMaxRanks + r);  // fixed order: identical on all ranks
    out[i] = s;
  }
}

// Stand-alone in-place sum over ranks of `count` doubles (the generic verbs; the lazy expansion path
// folds the exchange into its reduction kernels instead).  One wave per 4 elements.
static __global__ void __launch_bounds__(256) k_p2p_allreduce(double* __restrict__ v, int count, P2pDev p) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int e0 = wave * 4;
  if (e0 >= count) return;
  double x[4], out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (e0 + i < count) ? v[e0 + i] : 0.0;
  p2p_sum_wave<4>(p, e0, x, out);
  const int lane = threadIdx.x & 63;
  if (lane < 4 && e0 + lane < count) {
    double r = out[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = (lane == i) ? out[i] : r;
    v[e0 + lane] = r;
  }
}

// ---- halo ------------------------------------------------------------------------------------------
struct HaloArgs {
  int nneigh, nrecv;
  long long send_ptr[kP2pMaxNeigh + 1];  // entries for neighbour p: [send_ptr[p], send_ptr[p+1])
  void* dst[kP2pMaxNeigh];               // neighbour's ghost slot 0, already offset to where MY entries go
  long long dst_stride[kP2pMaxNeigh];    // elements between the neighbour's two ghost slots
  uint64_t* flag_dst[kP2pMaxNeigh];      // neighbour's halo flag for sender == me, slot 0 (slot 1 is + nranks words)
  int recv_from[kP2pMaxNeigh];           // ranks whose flags I wait for
};

// SEQUENCE NUMBERS of the halo exchange are kept by the HOST (one counter per context, advanced at every enqueue of an
// exchange) and travel in the kernel arguments; the slot of the double-buffered ghost vectors is seq & 1.  A kernel of a
// batch that exits early after a Krylov breakdown skips its exchange on EVERY rank alike (the breakdown decision is
// replicated), so numbers may be skipped -- flags are compared for equality, which does not care -- and the slot argument
// survives a skip too: the breakdown was detected by a reduction kernel that every rank entered after its last executed
// SpMV and left only with every peer's contribution, so no rank can push exchange e2 while a peer still reads the ghosts
// of the last executed exchange e1 < e2, whatever their parities.  (Without skips it is the usual double-buffer argument:
// push e+2 follows this rank's SpMV e+1, which waited for the peer's push e+1, which the peer issued after its SpMV e.)
//
// The work of one exchange, as device functions so that an SpMV kernel can do it ITSELF (k_spmv_stencil, k_spmv_csr with
// `HaloFused::enabled`): workgroups 0 .. npush-1 of the launch -- the first to be dispatched -- store this rank's
// boundary entries into the neighbours' ghost slots before they turn to their own tile, the last of them to finish raises
// the neighbours' flags; a workgroup whose rows may read ghost entries waits for the flags of the ranks it receives from
// before its loads, every other workgroup never waits.  One launch and ~9 us less per step than a push kernel in front
// of the SpMV (its own launch, and the whole grid behind the slowest neighbour), and the wait overlaps the interior.
// hstate[1] = finished-pusher counter (reset by the last pusher; the next exchange starts in a later launch).
struct HaloFused {
  int enabled;                     // 0: this launch has no exchange folded in
  uint32_t seq;
  int npush;                       // workgroups that push (<= gridDim.x)
  const int32_t* send_idx;
  uint32_t* counter;               // hstate + 1
  long long ghost_lo_end, ghost_hi_begin;  // only rows < ghost_lo_end or >= ghost_hi_begin may read ghost entries
  int tile_shift;                  // tiles are rotated by this many so that the leading boundary rows are dispatched LAST
};

// returns true in the workgroup that finished last (it raised the flags)
template <class D>
__device__ __forceinline__ bool halo_push_share(const D* __restrict__ x, const int32_t* __restrict__ send_idx, const HaloArgs& a,
                                                const P2pDev& p, uint32_t seq, int wg, int nwg, uint32_t* __restrict__ counter) {
  const int slot = seq & 1u;
  const long long total = a.send_ptr[a.nneigh];
  // four entries per thread and trip: the push is a chain of two dependent loads and a remote store per entry, so the
  // trips of a thread cost a memory round trip each -- few pushers with independent entries in flight, not many pushers
  const long long stride = (long long)nwg * blockDim.x;
  for (long long i0 = (long long)wg * blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
    int32_t si[4];
    D v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) si[u] = (i0 + u * stride < total) ? send_idx[i0 + u * stride] : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x[si[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * stride;
      if (i < total) {
        int pn = 0;
        while (i >= a.send_ptr[pn + 1]) ++pn;
        D* d = static_cast<D*>(a.dst[pn]) + (long long)slot * a.dst_stride[pn] + (i - a.send_ptr[pn]);
        *d = v[u];
      }
    }
  }
  // A RELEASE fence at system scope: this workgroup's remote stores are performed at the neighbours before it reports in.
  // Release only -- __threadfence_system() is release + ACQUIRE, and the acquire half invalidates this XCD's whole L2:
  // inside an SpMV launch that threw away the x the other tiles re-read through L2 seven times (35 us per launch).
  // (Waiting for the memory counter alone is NOT enough: the stores are acknowledged before they are performed at
  // the destination, and a neighbour polling from another XCD read stale ghosts -- caught by tools/dist_overhead.py.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  __shared__ int last_pusher;
  if (threadIdx.x == 0) last_pusher = (atomicAdd(counter, 1u) == (unsigned)nwg - 1u) ? 1 : 0;
  __syncthreads();
  if (!last_pusher) return false;
  const int t = threadIdx.x;
  if (t < a.nneigh && a.send_ptr[t + 1] > a.send_ptr[t])
    __hip_atomic_store(a.flag_dst[t] + (size_t)slot * p.nranks, (uint64_t)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (t == 0) *counter = 0u;
  return true;
}

// all threads of the workgroup: returns once the ghost entries of exchange `seq` have arrived from every sender
__device__ __forceinline__ void halo_wait(const HaloArgs& a, const P2pDev& p, uint32_t seq) {
  const int slot = seq & 1u;
  const int t = threadIdx.x;
  if (t < a.nrecv) {
    const uint64_t* f = p.region[p.rank] + p2p_ll_words(p.nranks, p.cap) + (size_t)slot * p.nranks + a.recv_from[t];
    const long long t0 = wall_clock64();
    long spins = 0;
    // relaxed polls: flag and ghost entries live in uncached memory, there is nothing to invalidate -- an ACQUIRE load at
    // system scope invalidates this XCD's whole L2 at every poll, under the feet of the tiles that are streaming x
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (uint64_t)seq) {
      if (p2p_expired(p, t0, spins++)) break;
      __builtin_amdgcn_s_sleep(2);
    }
  }
  // No acquire fence here (an acquire invalidates caches that every other tile is streaming x through).  What makes the
  // plain ghost loads that follow safe: the arena is uncached fine-grained memory (nothing of it lives in an L2), the
  // vector L1 is invalidated at every kernel launch, and inside a launch only workgroups that have passed this wait ever
  // load from the ghost vector (the kernels fold every other ghost-range address back onto local rows).
  asm volatile("" ::: "memory");
  __syncthreads();
}

// the prologue of an SpMV kernel with the exchange folded in; [row_lo, row_hi) = rows of this workgroup
template <class D>
__device__ __forceinline__ void halo_fused_prologue(const D* __restrict__ x, const HaloFused& h, const HaloArgs& a, const P2pDev& p,
                                                    long long row_lo, long long row_hi) {
  if ((int)blockIdx.x < h.npush) halo_push_share<D>(x, h.send_idx, a, p, h.seq, (int)blockIdx.x, h.npush, h.counter);
  if (row_lo < h.ghost_lo_end || row_hi > h.ghost_hi_begin) halo_wait(a, p, h.seq);
}

// stand-alone exchange in front of an SpMV kernel that does not fold it in: push, raise, wait
template <class D>
__global__ void __launch_bounds__(256)
    k_halo_push(const D* __restrict__ x, const int32_t* __restrict__ send_idx, HaloArgs a, P2pDev p,
                uint32_t* __restrict__ hstate, uint32_t seq, const int* __restrict__ breakdown) {
  if (breakdown && *breakdown >= 0) return;
  // the last workgroup to finish raised the flags; it also waits for this rank's own ghosts, so that the SpMV launched
  // behind this kernel finds them in place
  if (halo_push_share<D>(x, send_idx, a, p, seq, (int)blockIdx.x, (int)gridDim.x, hstate + 1)) halo_wait(a, p, seq);
}

}  // namespace ksd
