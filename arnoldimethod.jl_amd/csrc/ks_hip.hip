// libkschur_hip.so -- host side of the MI355X Krylov-Schur hot path: context / operator / workspace
// objects, the HIP backend of the driver (async expansion, in-place MFMA rotation), RCCL plumbing and
// the extern "C" entry points declared in include/kschur.h.
//
// There is NO CPU fallback in this library: without a gfx950 device every compute entry point
// returns KS_ERR_NO_DEVICE / KS_ERR_HIP.  Only the ks_host_* small-dense exports run without a GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/kschur.h"
#include "ks_driver.hpp"
#include "ks_kernels.hpp"

using ks::cplx;
using ksd::cd;
using ksd::DevState;
using ksd::kBlock;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_last_error;

struct KsError {
  int code;
  std::string msg;
};

#define KS_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess)                                                                        \
      throw KsError{KS_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(e__) + " (" + \
                                    __FILE__ + ":" + std::to_string(__LINE__) + ")"};             \
  } while (0)

#define KS_NCCL(expr)                                                                               \
  do {                                                                                              \
    ncclResult_t r__ = (expr);                                                                      \
    if (r__ != ncclSuccess)                                                                         \
      throw KsError{KS_ERR_RCCL, std::string(#expr) + " failed: " + ncclGetErrorString(r__) + " (" + \
                                     __FILE__ + ":" + std::to_string(__LINE__) + ")"};              \
  } while (0)

#define KS_REQUIRE(cond, code, text)           \
  do {                                         \
    if (!(cond)) throw KsError{(code), (text)}; \
  } while (0)

template <class F> int guarded(F&& f) {
  try {
    f();
    return KS_OK;
  } catch (const KsError& e) {
    g_last_error = e.msg;
    return e.code;
  } catch (const ks::QRNotConverged& e) {
    g_last_error = e.what();
    return KS_ERR_QR;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return KS_ERR_INTERNAL;
  } catch (...) {
    g_last_error = "unknown error";
    return KS_ERR_INTERNAL;
  }
}

template <class T> struct DevT;
template <> struct DevT<double> { using type = double; };
template <> struct DevT<cplx> { using type = cd; };
template <class D> struct HostT;
template <> struct HostT<double> { using type = double; };
template <> struct HostT<cd> { using type = cplx; };

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int env_int(const char* name, int dflt) {
  const char* s = std::getenv(name);
  return s ? std::atoi(s) : dflt;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
// Optional per-kernel-class timing with HIP events recorded on the context's stream (bench.py):
// class ids index ks_profile_* below.
enum { KSP_SPMV = 0, KSP_DOTS = 1, KSP_AXPY = 2, KSP_SCALE = 3, KSP_ROTATE = 4, KSP_FIN = 5, KSP_FUSED = 6, KSP_NCLASS = 7 };

struct ProfRecord {
  hipEvent_t a, b;
  int cls;
  double bytes;
};

struct ks_ctx {
  bool profiling = false;
  std::vector<ProfRecord> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[KSP_NCLASS] = {};
  double prof_bytes[KSP_NCLASS] = {};
  int64_t prof_count[KSP_NCLASS] = {};
  int device = 0;
  hipStream_t stream = nullptr;
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;
  // peer-to-peer transport (ks_p2p.hpp): one uncached, IPC-shared region per rank
  struct P2p {
    bool allocated = false, attached = false;
    void* region = nullptr;
    size_t region_bytes = 0, arena_off = 0, arena_bytes = 0, arena_used = 0;
    void* peer[ksd::kP2pMaxRanks] = {};
    uint32_t* seqc = nullptr;
    uint32_t* hstate = nullptr;
    uint32_t hseq = 0;          // sequence number of the last halo exchange enqueued on this context (ks_p2p.hpp)
    int* err_h = nullptr;
    int cap = 0;
    ksd::P2pDev dev{};
  } p2p;
  // host-staged transport (ks_ctx_create_hostcomm): the SAME launch structure as the RCCL transport (reduce-only
  // kernels -> all-reduce -> post kernels; pack kernel -> neighbour exchange -> SpMV on the ghost buffer), but each
  // exchange is staged through pinned host memory and executed by two caller-supplied functions (MPI, gloo, ...).
  // Exists so that the sequence around every ncclAllReduce / ncclSend / ncclRecv call site can run with several real
  // ranks on a ONE-GPU box (RCCL refuses two ranks per device) and as a transport of last resort on fabrics RCCL
  // does not cover.  Communication only: no arithmetic ever happens on the host.
  struct HostComm {
    ks_host_allreduce_fn allreduce = nullptr;
    ks_host_exchange_fn exchange = nullptr;
    void* user = nullptr;
    double* stage = nullptr;  // pinned
    size_t stage_doubles = 0;
  } hc;
  int num_cu = 256;
  int bpc = 6;  // streaming workgroups per CU (KS_BPC; 6 measured best on MI355X, tools/streambench.hip)
  int nblocks() const { return num_cu * bpc; }
  void use() const { KS_HIP(hipSetDevice(device)); }
  // A context created with ks_ctx_create_dist / ks_ctx_create_p2p always takes the collective code path
  // (even with nranks == 1, which is how that path is exercised on a single-GPU box).
  bool distributed() const { return comm != nullptr || p2p.attached || hc.allreduce != nullptr; }
  // in-place sum over ranks of `count` doubles living in device memory
  void allreduce(double* dev, int count) {
    if (p2p.attached) {
      KS_REQUIRE(count <= p2p.cap, KS_ERR_ARGUMENT, "reduction longer than the peer-to-peer window (KS_P2P_CAP)");
      const int waves = (count + 3) / 4;
      ksd::k_p2p_allreduce<<<(waves + 3) / 4, 256, 0, stream>>>(dev, count, p2p.dev);
    } else if (comm) {
      KS_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, comm, stream));
    } else if (hc.allreduce) {
      if ((size_t)count > hc.stage_doubles) {
        if (hc.stage) { KS_HIP(hipStreamSynchronize(stream)); (void)hipHostFree(hc.stage); hc.stage = nullptr; }
        hc.stage_doubles = (size_t)std::max(count, 256);
        KS_HIP(hipHostMalloc(&hc.stage, hc.stage_doubles * 8));
      }
      KS_HIP(hipMemcpyAsync(hc.stage, dev, (size_t)count * 8, hipMemcpyDeviceToHost, stream));
      KS_HIP(hipStreamSynchronize(stream));
      const int rc = hc.allreduce(hc.user, hc.stage, count);
      KS_REQUIRE(rc == 0, KS_ERR_COMM, "host all-reduce callback returned " + std::to_string(rc));
      KS_HIP(hipMemcpyAsync(dev, hc.stage, (size_t)count * 8, hipMemcpyHostToDevice, stream));
    }
  }
  // a bounded spin of the peer-to-peer kernels gave up: report instead of computing on garbage
  void check_comm() const {
    if (p2p.err_h && *p2p.err_h != 0)
      throw KsError{KS_ERR_COMM, "peer-to-peer exchange timed out waiting for a peer (first reported by rank " +
                                     std::to_string(*p2p.err_h - 1) + ")"};
  }
};

namespace {
// RAII scope: records an event pair around the enclosed launches when profiling is on.
struct ProfScope {
  ks_ctx* c;
  ProfRecord r;
  bool on;
  ProfScope(ks_ctx* ctx, int cls, double bytes) : c(ctx), on(ctx->profiling) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
      else KS_HIP(hipEventCreate(&e));
      return e;
    };
    r.a = get(); r.b = get(); r.cls = cls; r.bytes = bytes;
    KS_HIP(hipEventRecord(r.a, c->stream));
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, c->stream);
    c->prof_pending.push_back(r);
  }
};
// fold finished event pairs into the per-class totals (call after a stream synchronize)
void prof_collect(ks_ctx* c) {
  for (auto& r : c->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->prof_ms[r.cls] += ms;
      c->prof_bytes[r.cls] += r.bytes;
      c->prof_count[r.cls] += 1;
    }
    c->prof_pool.push_back(r.a);
    c->prof_pool.push_back(r.b);
  }
  c->prof_pending.clear();
}
}  // namespace

static void ctx_init_device(ks_ctx* c, int device) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    throw KsError{KS_ERR_NO_DEVICE, "no HIP device visible: libkschur_hip has no CPU fallback"};
  KS_REQUIRE(device >= 0 && device < ndev, KS_ERR_ARGUMENT, "device index out of range");
  c->device = device;
  KS_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  KS_HIP(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount;
  c->bpc = env_int("KS_BPC", 6);
  KS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
}

// ---- peer-to-peer region ---------------------------------------------------------------------------
static void p2p_alloc(ks_ctx* c) {
  auto& P = c->p2p;
  KS_REQUIRE(c->nranks <= ksd::kP2pMaxRanks, KS_ERR_ARGUMENT, "peer-to-peer transport supports at most 16 ranks");
  P.cap = env_int("KS_P2P_CAP", 2048);
  P.arena_bytes = (size_t)env_int("KS_P2P_ARENA_MB", 64) << 20;
  P.arena_off = (size_t)round_up((int64_t)(ksd::p2p_ll_words(c->nranks, P.cap) + ksd::p2p_flag_words(c->nranks)) * 8, 4096);
  P.region_bytes = P.arena_off + P.arena_bytes;
  // uncached (fine-grained) device memory: remote stores of the peers must not be shadowed by stale L2 lines
  KS_HIP(hipExtMallocWithFlags(&P.region, P.region_bytes, hipDeviceMallocUncached));
  KS_HIP(hipMemset(P.region, 0, P.region_bytes));
  KS_HIP(hipMalloc(&P.seqc, (size_t)P.cap * 4));
  KS_HIP(hipMemset(P.seqc, 0, (size_t)P.cap * 4));
  KS_HIP(hipMalloc(&P.hstate, 16));
  KS_HIP(hipMemset(P.hstate, 0, 16));
  KS_HIP(hipHostMalloc(&P.err_h, sizeof(int), hipHostMallocMapped));
  *P.err_h = 0;
  KS_HIP(hipDeviceSynchronize());
  P.allocated = true;
}

// `handles`: nranks x 64 bytes in rank order (this rank's own entry is ignored)
static void p2p_attach(ks_ctx* c, const void* handles) {
  auto& P = c->p2p;
  KS_REQUIRE(P.allocated && !P.attached, KS_ERR_ARGUMENT, "context has no (or an already attached) peer-to-peer region");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  for (int q = 0; q < c->nranks; ++q) {
    if (q == c->rank) { P.peer[q] = P.region; continue; }
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + (size_t)q * 64, 64);
    KS_HIP(hipIpcOpenMemHandle(&P.peer[q], h, hipIpcMemLazyEnablePeerAccess));
  }
  ksd::P2pDev d{};
  for (int q = 0; q < c->nranks; ++q) d.region[q] = static_cast<uint64_t*>(P.peer[q]);
  d.seqc = P.seqc;
  int* err_d = nullptr;
  KS_HIP(hipHostGetDevicePointer((void**)&err_d, P.err_h, 0));
  d.err = err_d;
  d.rank = c->rank;
  d.nranks = c->nranks;
  d.cap = P.cap;
  d.timeout_ticks = (long long)env_int("KS_P2P_TIMEOUT_S", 30) * 100000000LL;  // wall_clock64 runs at 100 MHz
  P.dev = d;
  P.attached = true;
}

static void p2p_release(ks_ctx* c) {
  auto& P = c->p2p;
  if (!P.allocated) return;
  for (int q = 0; q < c->nranks; ++q)
    if (q != c->rank && P.peer[q]) (void)hipIpcCloseMemHandle(P.peer[q]);
  (void)hipFree(P.region);
  (void)hipFree(P.seqc);
  (void)hipFree(P.hstate);
  (void)hipHostFree(P.err_h);
  P = ks_ctx::P2p{};
}

// gather `k` small non-negative integers (< 2^53) from every rank: rank r contributes row r of a
// nranks x k table of doubles, the others add zeros -- the sum IS the all-gather (exact in Float64)
static std::vector<int64_t> p2p_allgather_i64(ks_ctx* c, const std::vector<int64_t>& mine) {
  const int k = (int)mine.size(), total = k * c->nranks;
  std::vector<double> h((size_t)total, 0.0);
  for (int i = 0; i < k; ++i) h[(size_t)c->rank * k + i] = (double)mine[i];
  double* d = nullptr;
  KS_HIP(hipMalloc(&d, (size_t)total * 8));
  KS_HIP(hipMemcpyAsync(d, h.data(), (size_t)total * 8, hipMemcpyHostToDevice, c->stream));
  c->allreduce(d, total);
  KS_HIP(hipMemcpyAsync(h.data(), d, (size_t)total * 8, hipMemcpyDeviceToHost, c->stream));
  KS_HIP(hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  c->check_comm();
  std::vector<int64_t> out((size_t)total);
  for (int i = 0; i < total; ++i) out[i] = (int64_t)h[i];
  return out;
}

// ------------------------------------------------------------------------------------------------
// operators
// ------------------------------------------------------------------------------------------------
struct ks_operator {
  ks_ctx* ctx = nullptr;
  int64_t n_local = 0, nnz = 0;
  int dtype = KS_F64;
  bool async_capable = true;  // may be enqueued ahead without host involvement
  double bytes_per_nnz = 0.0;  // what the SpMV streams per stored non-zero (0: not a stored-matrix operator)
  double aux_bytes = 0.0;      // index structures next to the non-zeros (row pointers, slice offsets, permutation)
  int layout = -1;             // KS_LAYOUT_* of include/kschur.h (-1: not a stored sparse matrix)
  // Inside an expansion the newest basis column is stored unnormalised (x = beta * v up to a correction in span(V)); the
  // library's own operators are linear and do not care, a HOST callback is handed the vector scaled to unit norm and its
  // result is scaled back (a user's inner solver may use absolute tolerances).  Set by the expansion before apply().
  double in_scale = 1.0;
  virtual ~ks_operator() = default;
  // y = A x on device pointers, enqueued on ctx->stream; `st` lets the kernels of a batch skip work
  // after a breakdown.
  virtual void apply(const void* x, void* y, const DevState* st) = 0;
};

namespace {

template <class D> struct CsrOp : ks_operator {
  void* rowptr = nullptr;   // int32[n+1], or int64[n+1] when ptr64 (nnz >= 2^31: int64-nnz CSR)
  bool ptr64 = false;
  int32_t* colidx = nullptr;
  D* val = nullptr;
  // row blocks of k_spmv_csr (CSR-adaptive tiling): rows [blkrow[b], blkrow[b+1]), non-zeros [blkptr[b], blkptr[b+1])
  void* blkptr = nullptr;   // same integer type as rowptr
  int32_t* blkrow = nullptr;
  int nblk = 0;
  // column-blocked layout: the matrix split into column blocks, each a CSR-row-block sub-operator of its own; apply() runs
  // them in order, each continuing the row sums of the previous one (k_spmv_csr's yacc)
  std::vector<std::unique_ptr<CsrOp<D>>> cblocks;
  int cb_rpt = 0, cb_ni = 0;  // single-launch form of the column-blocked layout (k_spmv_csr_cb): 256-row sub-tiles per workgroup, LDS depth; 0: one launch per block
  int ni = 7;               // non-zeros per thread and block: a block holds at most ni * 256 entries in LDS
  bool row_gather = false;  // k_spmv_csr: one thread per row gathers x itself (banded matrices) instead of the non-zero-parallel gathers
  int nlong = 0;            // rows longer than that: cut into chunk blocks, partial sums added by k_spmv_longfix
  int32_t* blkpart = nullptr;  // per block: -1, or the index of the chunk's partial sum
  D* lpart = nullptr;
  int32_t* lrow = nullptr;
  int32_t* lfirst = nullptr;
  // stencil-mask layout (k_spmv_stencil): one bit per dictionary slot and row, the dictionary in the kernel arguments
  int nstencil = 0;          // slots (0: layout not in use)
  int stencil_mask_bytes = 1;
  void* smask = nullptr;
  void* smask2 = nullptr;    // == smask (the array is padded to an even number of rows): masks of rows 2t, 2t+1 in one word
  ksd::StencilDict<D> sdict{};
  // sliced-ELLPACK layout (k_spmv_sell): slices of 64 rows, column-major, padded to the slice's longest row
  void* sliceptr = nullptr;  // entry offsets of the slices, same integer type as rowptr
  int32_t* sperm = nullptr;  // slice position -> row (sigma > 1 only)
  int nslices = 0;
  int sell_un = 8;
  int64_t sell_entries = 0;  // stored entries including padding
  int ndict = 0;      // > 0: value-indexed layout (k_spmv_csr<.., VI>): colidx = (dict index << 24) | column, val = dictionary
  // delta-value-indexed layout (k_spmv_dvi): one byte per non-zero into a dictionary of (column - row, value)
  int ndvi = 0;
  uint8_t* codes = nullptr;
  int32_t* ddelta = nullptr;
  int dvi_unroll = 8;
  int dvi_rpt = 4;          // rows per thread of k_spmv_dvi
  // halo plan (distributed)
  int64_t nghost = 0;
  D* ghost = nullptr;
  D* sendbuf = nullptr;
  int32_t* send_idx = nullptr;
  std::vector<int> neigh;
  std::vector<int64_t> send_ptr, recv_ptr;
  std::vector<int64_t> send_first;  // >= 0: neighbour p's rows are the contiguous run starting here (no packing)
  std::vector<int64_t> pack_ptr;    // offset of neighbour p's packed values in sendbuf (scattered lists only)
  int64_t nscatter = 0;             // number of packed entries (send_idx holds only these)
  // peer-to-peer halo (ks_p2p.hpp): ghost lives (double-buffered) in this rank's shared arena, neighbours
  // store into it directly
  // host-staged halo (ks_ctx_create_hostcomm): pinned send / receive images of the plan
  D* hsend = nullptr;
  D* hrecv = nullptr;
  bool p2p_halo = false;
  int64_t ghost_stride = 0;         // elements between the two ghost slots
  int64_t ghost_lo_end = 0, ghost_hi_begin = 0;  // only rows < ghost_lo_end or >= ghost_hi_begin reference ghost columns (fused exchange)
  size_t arena_lo = 0, arena_hi = 0;
  int32_t* send_idx_all = nullptr;  // every send entry (contiguous runs included), neighbour by neighbour
  ksd::HaloArgs hargs{};

  ~CsrOp() override {
    (void)hipFree(rowptr); (void)hipFree(colidx); (void)hipFree(val); (void)hipFree(blkptr); (void)hipFree(blkrow); (void)hipFree(smask); (void)hipFree(sliceptr); (void)hipFree(sperm); (void)hipFree(blkpart); (void)hipFree(lpart); (void)hipFree(lrow); (void)hipFree(lfirst);
    if (p2p_halo) {
      if (ctx->p2p.arena_used == arena_hi) ctx->p2p.arena_used = arena_lo;  // stack discipline; otherwise kept until the context dies
    } else {
      (void)hipFree(ghost);
    }
    (void)hipHostFree(hsend); (void)hipHostFree(hrecv);
    (void)hipFree(sendbuf); (void)hipFree(send_idx); (void)hipFree(send_idx_all);
    (void)hipFree(codes); (void)hipFree(ddelta);
  }
  // the CSR-row-block kernel of THIS operator's arrays on x -> y, continuing the row sums in `yacc` (column-blocked
  // layout: plain CSR, single GPU, no long rows -- make_csr only builds column blocks under those conditions)
  void launch_csr_blocks(const D* x, D* y, const DevState* st, const D* yacc, int plain_store) {
    hipStream_t s = ctx->stream;
    auto go = [&](auto ip_tag) {
      using IP = decltype(ip_tag);
      auto launch = [&](auto ni_tag) {
        constexpr int NI = decltype(ni_tag)::value;
        if constexpr ((size_t)NI * kBlock * sizeof(D) <= (size_t)ksd::kSpmvCapBytes)
          ksd::k_spmv_csr<D, IP, false, NI><<<nblk, kBlock, 0, s>>>(static_cast<const IP*>(blkptr), blkrow, static_cast<const IP*>(rowptr), colidx, val, x,
                                                                    nullptr, y, n_local, nblk, st, nullptr, 0, 0, nullptr, nullptr, yacc, plain_store,
                                                                    ksd::HaloFused{}, ksd::HaloArgs{}, ksd::P2pDev{}, env_int("KS_SPMV_CSR_NT", 1) != 0, row_gather);
        else
          throw KsError{KS_ERR_INTERNAL, "CSR row blocks of " + std::to_string(NI) + " x 256 entries exceed the LDS budget of this element type"};
      };
      switch (ni) {
        case 4: launch(std::integral_constant<int, 4>{}); break;
        case 7: launch(std::integral_constant<int, 7>{}); break;
        case 8: launch(std::integral_constant<int, 8>{}); break;
        case 12: launch(std::integral_constant<int, 12>{}); break;
        default: launch(std::integral_constant<int, 16>{}); break;
      }
    };
    if (ptr64) go(int64_t{});
    else go(int32_t{});
  }
  void apply(const void* xv, void* yv, const DevState* st) override {
    const D* x = static_cast<const D*>(xv);
    D* y = static_cast<D*>(yv);
    hipStream_t s = ctx->stream;
    // peer-to-peer mode: sequence number and ghost slot of this exchange (host counter, ks_p2p.hpp); the stencil and the
    // CSR-row-block kernels do the exchange themselves, every other layout gets the push kernel in front
    ksd::HaloFused hf{};
    const D* xg = ghost;
    if (p2p_halo && !neigh.empty()) {
      uint32_t seq = ctx->p2p.hseq + 1u;
      if (seq == 0u) seq = 1u;
      ctx->p2p.hseq = seq;
      xg = ghost + (int64_t)(seq & 1u) * ghost_stride;
      static const int fuse_env = env_int("KS_HALO_FUSED", 1);
      const bool fusable = fuse_env && n_local > 0 && cblocks.empty() && ndvi == 0 && nslices == 0;
      const int64_t total = send_ptr.back();
      if (fusable) {
        hf.enabled = 1;
        hf.seq = seq;
        // pushers: ~8 entries per thread (two trips of four independent entries), at most 64 workgroups -- each pusher ends
        // with a system-scope release fence (an L2 write-back), which is what an SpMV launch can afford only a few of
        static const int npush_env = env_int("KS_HALO_NPUSH", 0);
        hf.npush = npush_env > 0 ? npush_env : (int)std::max<int64_t>(1, std::min<int64_t>((total + 2047) / 2048, 64));
        hf.send_idx = send_idx_all;
        hf.counter = ctx->p2p.hstate + 1;
        hf.ghost_lo_end = ghost_lo_end;
        hf.ghost_hi_begin = ghost_hi_begin;
      } else {
        const int gb = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 512));
        ksd::k_halo_push<D><<<gb, 256, 0, s>>>(x, send_idx_all, hargs, ctx->p2p.dev, ctx->p2p.hstate, seq,
                                                 st ? &st->breakdown : nullptr);
      }
    } else if (!p2p_halo && !neigh.empty()) {
      // neighbours whose send list is one contiguous run of rows (grid planes of a slab partition) are sent
      // straight out of x; only genuinely scattered lists go through the pack kernel
      if (nscatter > 0) {
        const int gb = (int)std::min<int64_t>((nscatter + kBlock - 1) / kBlock, 4096);
        ksd::k_gather<D><<<gb, kBlock, 0, s>>>(x, send_idx, sendbuf, nscatter, st);
      }
      constexpr int dpe = sizeof(D) / 8;  // doubles per element
      if (ctx->hc.exchange) {
        // host-staged: the same plan (in-place runs, packed lists, consecutive ghost slots) through pinned memory
        const size_t np_ = neigh.size();
        std::vector<const void*> sp_(np_);
        std::vector<void*> rp_(np_);
        std::vector<int64_t> sb_(np_), rb_(np_);
        for (size_t p = 0; p < np_; ++p) {
          const int64_t sc = send_ptr[p + 1] - send_ptr[p], rc = recv_ptr[p + 1] - recv_ptr[p];
          if (sc > 0) {
            const D* src = send_first[p] >= 0 ? x + send_first[p] : sendbuf + pack_ptr[p];
            KS_HIP(hipMemcpyAsync(hsend + send_ptr[p], src, (size_t)sc * sizeof(D), hipMemcpyDeviceToHost, s));
          }
          sp_[p] = hsend + send_ptr[p]; sb_[p] = sc * (int64_t)sizeof(D);
          rp_[p] = hrecv + recv_ptr[p]; rb_[p] = rc * (int64_t)sizeof(D);
        }
        KS_HIP(hipStreamSynchronize(s));
        const int rc = ctx->hc.exchange(ctx->hc.user, (int)np_, neigh.data(), sp_.data(), sb_.data(), rp_.data(), rb_.data());
        KS_REQUIRE(rc == 0, KS_ERR_COMM, "host exchange callback returned " + std::to_string(rc));
        if (nghost > 0) KS_HIP(hipMemcpyAsync(ghost, hrecv, (size_t)nghost * sizeof(D), hipMemcpyHostToDevice, s));
      } else {
      KS_NCCL(ncclGroupStart());
      for (size_t p = 0; p < neigh.size(); ++p) {
        const int64_t sc = send_ptr[p + 1] - send_ptr[p], rc = recv_ptr[p + 1] - recv_ptr[p];
        if (sc > 0) {
          const D* src = send_first[p] >= 0 ? x + send_first[p] : sendbuf + pack_ptr[p];
          KS_NCCL(ncclSend(src, (size_t)sc * dpe, ncclDouble, neigh[p], ctx->comm, s));
        }
        if (rc > 0) KS_NCCL(ncclRecv(ghost + recv_ptr[p], (size_t)rc * dpe, ncclDouble, neigh[p], ctx->comm, s));
      }
      KS_NCCL(ncclGroupEnd());
      }
    }
    if (n_local > 0 && !cblocks.empty()) {
      // column-blocked CSR: one launch per column block; block b > 0 reads the partial row sums block b-1 left in y
      ProfScope ps(ctx, KSP_SPMV, (double)nnz * bytes_per_nnz + aux_bytes + 2.0 * sizeof(D) * n_local);
      if (cb_rpt) {
        // ONE launch: a workgroup keeps the sums of its rows in registers while it walks the column blocks (k_spmv_csr_cb)
        ksd::CbArgs<D> a{};
        a.nb = (int)cblocks.size();
        for (int b = 0; b < a.nb; ++b) {
          a.rowptr[b] = static_cast<const int32_t*>(cblocks[b]->rowptr);
          a.colidx[b] = cblocks[b]->colidx;
          a.val[b] = cblocks[b]->val;
        }
        const int nt = (int)((n_local + (int64_t)kBlock * cb_rpt - 1) / ((int64_t)kBlock * cb_rpt));
        auto go = [&](auto ni_tag, auto rpt_tag) {
          constexpr int NI = decltype(ni_tag)::value, RPT = decltype(rpt_tag)::value;
          if constexpr ((size_t)NI * kBlock * sizeof(D) <= (size_t)ksd::kSpmvCapBytes)
            ksd::k_spmv_csr_cb<D, NI, RPT><<<nt, kBlock, 0, s>>>(a, x, y, n_local, nt, st);
          else
            throw KsError{KS_ERR_INTERNAL, "column-blocked CSR: LDS depth exceeds the budget of this element type"};
        };
        auto by_rpt = [&](auto ni_tag) {
          switch (cb_rpt) {
            case 1: go(ni_tag, std::integral_constant<int, 1>{}); break;
            case 2: go(ni_tag, std::integral_constant<int, 2>{}); break;
            case 4: go(ni_tag, std::integral_constant<int, 4>{}); break;
            case 8: go(ni_tag, std::integral_constant<int, 8>{}); break;
            default: go(ni_tag, std::integral_constant<int, 16>{}); break;
          }
        };
        if (cb_ni == 8) by_rpt(std::integral_constant<int, 8>{});
        else by_rpt(std::integral_constant<int, 16>{});
        KS_HIP(hipGetLastError());
        return;
      }
      for (size_t b = 0; b < cblocks.size(); ++b) cblocks[b]->launch_csr_blocks(x, y, st, b > 0 ? y : nullptr, b + 1 < cblocks.size() ? 1 : 0);
      KS_HIP(hipGetLastError());
      return;
    }
    if (n_local > 0) {
      // algorithmic bytes: 12 nnz + 4 (n+1) + 16 n   (8 -> 16 for complex); 4 nnz in the
      // value-indexed layout, 1 nnz in the delta-value-indexed one
      ProfScope ps(ctx, KSP_SPMV, (double)nnz * bytes_per_nnz + aux_bytes + 2.0 * sizeof(D) * n_local);
      const uint32_t* hseq = nullptr;  // (the host picked the ghost slot: xg)
      auto with_ip = [&](auto f) {
        if (ptr64) f(int64_t{});
        else f(int32_t{});
      };
      if (nstencil > 0 && nghost == 0 && smask2 && n_local >= 2 && env_int("KS_STENCIL_PAIRS", 1)) {
        // two rows per lane, 16-byte gathers (no ghost columns: single GPU)
        const int nt = (int)(((n_local + 1) / 2 + kBlock - 1) / kBlock);
        if (stencil_mask_bytes == 1)
          ksd::k_spmv_stencil2<D, uint16_t><<<nt, kBlock, 0, s>>>(static_cast<const uint16_t*>(smask2), sdict, nstencil, x, y, n_local, nt, st);
        else
          ksd::k_spmv_stencil2<D, uint64_t><<<nt, kBlock, 0, s>>>(static_cast<const uint64_t*>(smask2), sdict, nstencil, x, y, n_local, nt, st);
        KS_HIP(hipGetLastError());
        return;
      }
      if (nstencil > 0) {
        static const int rpt_env = env_int("KS_STENCIL_RPT", 1);
        auto go = [&](auto mt_tag, auto rpt_tag) {
          using MT = decltype(mt_tag);
          constexpr int RPT = decltype(rpt_tag)::value;
          const int nt = (int)((n_local + kBlock * RPT - 1) / (kBlock * RPT));
          ksd::HaloFused h = hf;
          h.npush = std::min(h.npush, nt);
          h.tile_shift = (h.enabled && ghost_lo_end < n_local) ? (int)((ghost_lo_end + kBlock * RPT - 1) / (kBlock * RPT)) % std::max(nt, 1) : 0;
          ksd::k_spmv_stencil<D, MT, RPT><<<nt, kBlock, 0, s>>>(static_cast<const MT*>(smask), sdict, nstencil, x, xg, y, n_local,
                                                                std::max<int64_t>(nghost, 0), nt, st, h, hargs, ctx->p2p.dev);
        };
        auto by_rpt = [&](auto mt_tag) {
          if (rpt_env <= 1) go(mt_tag, std::integral_constant<int, 1>{});
          else if (rpt_env == 2) go(mt_tag, std::integral_constant<int, 2>{});
          else go(mt_tag, std::integral_constant<int, 4>{});
        };
        if (stencil_mask_bytes == 1) by_rpt(uint8_t{});
        else by_rpt(uint32_t{});
        KS_HIP(hipGetLastError());
        return;
      }
      if (ndvi > 0) {
        with_ip([&](auto ip_tag) {
          using IP = decltype(ip_tag);
          const IP* rp = static_cast<const IP*>(rowptr);
          auto go = [&](auto un_tag, auto rpt_tag) {
            constexpr int UN = decltype(un_tag)::value, RPT = decltype(rpt_tag)::value;
            const int nt = (int)((n_local + kBlock * RPT - 1) / (kBlock * RPT));
            ksd::k_spmv_dvi<D, IP, UN, RPT><<<nt, kBlock, 0, s>>>(rp, codes, ddelta, val, x, xg, y, n_local, nt, ndvi, st, hseq, ghost_stride);
          };
          using I = std::integral_constant<int, 0>;
          (void)sizeof(I);
          if (dvi_unroll == 4) {
            if (dvi_rpt == 1) go(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
            else if (dvi_rpt == 2) go(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
            else go(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
          } else {
            if (dvi_rpt == 1) go(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
            else if (dvi_rpt == 2) go(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});
            else go(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
          }
        });
        KS_HIP(hipGetLastError());
        return;
      }
      if (nslices > 0) {
        with_ip([&](auto ip_tag) {
          using IP = decltype(ip_tag);
          const int ng = (nslices + 3) / 4;
          auto go = [&](auto vi_tag, auto un_tag) {
            constexpr bool VI = decltype(vi_tag)::value;
            constexpr int UN = decltype(un_tag)::value;
            static const int plain_loads = env_int("KS_SELL_PLAIN_LOADS", 0);  // experiment: default-policy loads of the matrix streams
            if (plain_loads)
              ksd::k_spmv_sell<D, IP, VI, UN, false><<<ng, kBlock, 0, s>>>(static_cast<const IP*>(sliceptr), colidx, val, sperm, x, xg, y,
                                                                           n_local, nslices, ng, st, hseq, ghost_stride, ndict);
            else
              ksd::k_spmv_sell<D, IP, VI, UN, true><<<ng, kBlock, 0, s>>>(static_cast<const IP*>(sliceptr), colidx, val, sperm, x, xg, y,
                                                                          n_local, nslices, ng, st, hseq, ghost_stride, ndict);
          };
          auto by_un = [&](auto vi_tag) {
            if (sell_un <= 4) go(vi_tag, std::integral_constant<int, 4>{});
            else go(vi_tag, std::integral_constant<int, 8>{});
          };
          if (ndict > 0) by_un(std::true_type{});
          else by_un(std::false_type{});
        });
        KS_HIP(hipGetLastError());
        return;
      }
      with_ip([&](auto ip_tag) {
        using IP = decltype(ip_tag);
        auto launch = [&](auto vi_tag, auto ni_tag) {
          constexpr bool VI = decltype(vi_tag)::value;
          constexpr int NI = decltype(ni_tag)::value;
          if constexpr ((size_t)NI * kBlock * sizeof(D) <= (size_t)ksd::kSpmvCapBytes)
          {
            static const int csr_nt = env_int("KS_SPMV_CSR_NT", 1);
            ksd::HaloFused h = hf;
            h.npush = std::min(h.npush, nblk);
            ksd::k_spmv_csr<D, IP, VI, NI><<<nblk, kBlock, 0, s>>>(static_cast<const IP*>(blkptr), blkrow, static_cast<const IP*>(rowptr), colidx,
                                                                   val, x, xg, y, n_local, nblk, st, hseq, ghost_stride, ndict, blkpart, lpart,
                                                                   nullptr, 0, h, hargs, ctx->p2p.dev, csr_nt != 0, row_gather);
          }
          else
            throw KsError{KS_ERR_INTERNAL, "CSR row blocks of " + std::to_string(NI) + " x 256 entries exceed the LDS budget of this element type"};
        };
        auto by_ni = [&](auto vi_tag) {
          switch (ni) {
            case 4: launch(vi_tag, std::integral_constant<int, 4>{}); break;
            case 7: launch(vi_tag, std::integral_constant<int, 7>{}); break;
            case 8: launch(vi_tag, std::integral_constant<int, 8>{}); break;
            case 12: launch(vi_tag, std::integral_constant<int, 12>{}); break;
            default: launch(vi_tag, std::integral_constant<int, 16>{}); break;
          }
        };
        if (ndict > 0) by_ni(std::true_type{});
        else by_ni(std::false_type{});
      });
      if (nlong > 0) ksd::k_spmv_longfix<D><<<(nlong + kBlock - 1) / kBlock, kBlock, 0, s>>>(lrow, lfirst, lpart, y, nlong, st);
    }
    KS_HIP(hipGetLastError());
  }
};

// dense matrix resident in HBM, row-major with padded rows
template <class D> struct DenseOp : ks_operator {
  D* A = nullptr;
  int64_t lda = 0;
  ~DenseOp() override { (void)hipFree(A); }
  void apply(const void* xv, void* yv, const DevState* st) override {
    ProfScope ps(ctx, KSP_SPMV, (double)n_local * n_local * sizeof(D) + 2.0 * sizeof(D) * n_local);
    const int64_t want = (n_local + 3) / 4;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cu * 8));
    ksd::k_gemv_rows<D><<<nb, kBlock, 0, ctx->stream>>>(A, lda, n_local, n_local, static_cast<const D*>(xv), static_cast<D*>(yv), st);
    KS_HIP(hipGetLastError());
  }
};

struct HostCallbackOp : ks_operator {
  ks_host_apply_fn fn = nullptr;
  void* user = nullptr;
  void* xh = nullptr;
  void* yh = nullptr;
  ~HostCallbackOp() override { (void)hipHostFree(xh); (void)hipHostFree(yh); }
  void apply(const void* x, void* y, const DevState*) override {
    const size_t bytes = (size_t)n_local * (dtype == KS_F64 ? 8 : 16);
    KS_HIP(hipMemcpyAsync(xh, x, bytes, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t nd = n_local * (dtype == KS_F64 ? 1 : 2);
    if (in_scale != 1.0) {
      double* xd = static_cast<double*>(xh);
      for (int64_t i = 0; i < nd; ++i) xd[i] *= in_scale;
    }
    const int rc = fn(user, xh, yh);
    KS_REQUIRE(rc == 0, KS_ERR_OPERATOR, "host operator callback returned " + std::to_string(rc));
    if (in_scale != 1.0) {
      const double back = 1.0 / in_scale;
      double* yd = static_cast<double*>(yh);
      for (int64_t i = 0; i < nd; ++i) yd[i] *= back;
    }
    KS_HIP(hipMemcpyAsync(y, yh, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
};

struct DeviceCallbackOp : ks_operator {
  ks_device_apply_fn fn = nullptr;
  void* user = nullptr;
  void apply(const void* x, void* y, const DevState*) override {
    const int rc = fn(user, x, y, (void*)ctx->stream);
    KS_REQUIRE(rc == 0, KS_ERR_OPERATOR, "device operator callback returned " + std::to_string(rc));
  }
};

// Host conversion of whatever the caller has into int32 0-based CSR.
template <class I> inline int64_t idx_at(const void* p, int64_t i) { return (int64_t) static_cast<const I*>(p)[i]; }

template <class D>
void build_csr_host(int64_t nrows, int64_t ncols, int64_t nnz, const void* ptr, const void* idx, const void* val,
                    int layout, int base, int itype, std::vector<int64_t>& rp, std::vector<int32_t>& ci,
                    std::vector<D>& vv) {
  auto P = [&](int64_t i) { return (itype == KS_I32 ? idx_at<int32_t>(ptr, i) : idx_at<int64_t>(ptr, i)) - base; };
  auto J = [&](int64_t i) { return (itype == KS_I32 ? idx_at<int32_t>(idx, i) : idx_at<int64_t>(idx, i)) - base; };
  const D* v = static_cast<const D*>(val);
  // column indices are 32-bit on the device; the non-zero offsets (rowptr) switch to 64 bits when nnz >= 2^31
  KS_REQUIRE(nrows < (int64_t)2147483647 && ncols < (int64_t)2147483647, KS_ERR_ARGUMENT, "matrix order must fit int32");
  {
    // the pointer array must be monotone and stay inside [0, nnz]: a malformed one would index host (CSC
    // conversion) or device (SpMV) arrays out of bounds
    const int64_t np = (layout == KS_CSR ? nrows : ncols);
    int64_t prev = P(0);
    KS_REQUIRE(prev == 0, KS_ERR_ARGUMENT, layout == KS_CSR ? "row pointer does not match nnz" : "column pointer does not match nnz");
    for (int64_t i = 1; i <= np; ++i) {
      const int64_t cur = P(i);
      KS_REQUIRE(cur >= prev && cur <= nnz, KS_ERR_ARGUMENT, "pointer array is not monotone within [0, nnz]");
      prev = cur;
    }
    KS_REQUIRE(prev == nnz, KS_ERR_ARGUMENT, layout == KS_CSR ? "row pointer does not match nnz" : "column pointer does not match nnz");
  }
  rp.assign(nrows + 1, 0);
  ci.resize(nnz);
  vv.resize(nnz);
  if (layout == KS_CSR) {
    for (int64_t i = 0; i <= nrows; ++i) rp[i] = P(i);
    for (int64_t p = 0; p < nnz; ++p) {
      const int64_t c = J(p);
      KS_REQUIRE(c >= 0 && c < ncols, KS_ERR_ARGUMENT, "column index out of range");
      ci[p] = (int32_t)c;
      vv[p] = v[p];
    }
  } else {  // CSC (Julia SparseMatrixCSC: colptr, rowval, nzval) -> CSR by counting sort
    for (int64_t p = 0; p < nnz; ++p) {
      const int64_t r = J(p);
      KS_REQUIRE(r >= 0 && r < nrows, KS_ERR_ARGUMENT, "row index out of range");
      rp[r + 1]++;
    }
    for (int64_t i = 0; i < nrows; ++i) rp[i + 1] += rp[i];
    std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
    for (int64_t c = 0; c < ncols; ++c)
      for (int64_t p = P(c); p < P(c + 1); ++p) {
        const int64_t r = J(p);
        const int64_t q = fill[r]++;
        ci[q] = (int32_t)c;
        vv[q] = v[p];
      }
  }
}

// device copy of the non-zero offsets in the width the kernels will use
inline void* upload_ptr(const std::vector<int64_t>& v, bool ptr64) {
  void* d = nullptr;
  const size_t cnt = v.size();
  if (ptr64) {
    KS_HIP(hipMalloc(&d, std::max<size_t>(cnt * 8, 16)));
    KS_HIP(hipMemcpy(d, v.data(), cnt * 8, hipMemcpyHostToDevice));
  } else {
    std::vector<int32_t> t(v.begin(), v.end());
    KS_HIP(hipMalloc(&d, std::max<size_t>(cnt * 4, 16)));
    KS_HIP(hipMemcpy(d, t.data(), cnt * 4, hipMemcpyHostToDevice));
  }
  return d;
}

template <class D>
CsrOp<D>* make_csr(ks_ctx* ctx, int64_t nrows, int64_t nnz, const std::vector<int64_t>& rp,
                   const std::vector<int32_t>& ci, const std::vector<D>& vv, int cb_mode = 0) {
  // cb_mode: 0 = column blocks not allowed (distributed operators: ghost columns), 1 = allowed (decided below),
  //          2 = this IS a column block (plain CSR row blocks, nothing else is tried)
  auto op = std::make_unique<CsrOp<D>>();
  op->ctx = ctx;
  op->n_local = nrows;
  op->nnz = nnz;
  op->dtype = sizeof(D) == 8 ? KS_F64 : KS_C64;
  // int64-nnz CSR: offsets need 64 bits from 2^31 stored entries on (KS_SPMV_PTR64=1 forces it, for tests)
  op->ptr64 = nnz >= (int64_t)2147483647 || env_int("KS_SPMV_PTR64", 0) != 0;
  // Delta-value-indexed layout (k_spmv_dvi): at most 256 distinct (column - row, value) pairs -> one byte per
  // non-zero.  KS_SPMV_FORMAT = csr | vi | dvi restricts the choice (default: the most compact that applies).
  {
    const char* fmt = cb_mode == 2 ? "csr" : std::getenv("KS_SPMV_FORMAT");
    const bool try_dvi = nnz > 0 && (!fmt || std::string(fmt) == "dvi" || std::string(fmt) == "stencil");
    if (try_dvi) {
      struct Key {
        uint64_t a, b;
        int64_t d;
        bool operator==(const Key& o) const { return a == o.a && b == o.b && d == o.d; }
      };
      struct KeyHash {
        size_t operator()(const Key& k) const {
          return std::hash<uint64_t>()((k.a * 0x9E3779B97F4A7C15ull ^ k.b) + (uint64_t)k.d * 0xC2B2AE3D27D4EB4Full);
        }
      };
      std::unordered_map<Key, int, KeyHash> index;
      std::vector<uint8_t> codes((size_t)nnz);
      std::vector<int32_t> dd;
      std::vector<D> dv;
      bool ok = true;
      int64_t max_row = 0;
      Key ckey[8];
      int cid[8], ncache = 0, cnext = 0;
      for (int64_t r = 0; r < nrows && ok; ++r) {
        max_row = std::max(max_row, rp[r + 1] - rp[r]);
        for (int64_t p = rp[r]; p < rp[r + 1]; ++p) {
          Key k{0, 0, (int64_t)ci[p] - r};
          std::memcpy(&k, &vv[p], sizeof(D));
          // stencils cycle through a handful of keys: a tiny recent-key cache in front of the hash map
          // (n = 1e8 rows / 7e8 non-zeros convert in seconds instead of half a minute)
          bool hit = false;
          for (int q = 0; q < ncache; ++q)
            if (ckey[q] == k) { codes[p] = (uint8_t)cid[q]; hit = true; break; }
          if (hit) continue;
          auto it = index.find(k);
          int id;
          if (it == index.end()) {
            if (dd.size() == 256) { ok = false; break; }
            id = (int)dd.size();
            index.emplace(k, id);
            dd.push_back((int32_t)k.d);
            dv.push_back(vv[p]);
          } else {
            id = it->second;
          }
          codes[p] = (uint8_t)id;
          ckey[cnext] = k;
          cid[cnext] = id;
          cnext = (cnext + 1) & 7;
          if (ncache < 8) ++ncache;
        }
      }
      // Stencil-mask layout: <= 32 dictionary entries and every row a sub-sequence of ONE ordering of them (a
      // topological order of "entry a precedes entry b in some row"): one bit per slot and row.  KS_SPMV_FORMAT=dvi
      // keeps the byte-per-entry layout, =stencil insists on this one.
      if (ok && dd.size() <= (size_t)ksd::kStencilSlots && !(fmt && std::string(fmt) == "dvi")) {
        const int ns = (int)dd.size();
        std::vector<uint32_t> succ((size_t)ns, 0u);  // succ[a] bit b: a directly precedes b in some row
        for (int64_t r = 0; r < nrows; ++r)
          for (int64_t p = rp[r] + 1; p < rp[r + 1]; ++p) succ[codes[p - 1]] |= 1u << codes[p];
        // Kahn's algorithm on <= 32 nodes; ties broken by dictionary id (first appearance) -> deterministic
        std::vector<int> indeg((size_t)ns, 0), order;
        for (int a = 0; a < ns; ++a)
          for (int b = 0; b < ns; ++b)
            if (succ[a] >> b & 1u) indeg[b]++;
        std::vector<char> done((size_t)ns, 0);
        for (int it = 0; it < ns; ++it) {
          int pick = -1;
          for (int a = 0; a < ns; ++a)
            if (!done[a] && indeg[a] == 0) { pick = a; break; }
          if (pick < 0) break;  // a cycle: no common order
          done[pick] = 1;
          order.push_back(pick);
          for (int b = 0; b < ns; ++b)
            if (succ[pick] >> b & 1u) indeg[b]--;
        }
        bool sten = (int)order.size() == ns;
        std::vector<int> slot((size_t)ns, 0);
        for (int k = 0; k < (int)order.size(); ++k) slot[order[k]] = k;
        const int mbytes = ns <= 8 ? 1 : 4;
        std::vector<uint8_t> m8;
        std::vector<uint32_t> m32;
        if (sten) {
          if (mbytes == 1) m8.assign((size_t)nrows, 0); else m32.assign((size_t)nrows, 0u);
          for (int64_t r = 0; r < nrows && sten; ++r) {
            uint32_t m = 0;
            int last = -1;
            for (int64_t p = rp[r]; p < rp[r + 1]; ++p) {
              const int k = slot[codes[p]];
              if (k <= last) { sten = false; break; }  // (a repeated entry in one row: not a sub-sequence)
              last = k;
              m |= 1u << k;
            }
            if (mbytes == 1) m8[r] = (uint8_t)m; else m32[r] = m;
          }
        }
        if (sten) {
          op->nstencil = ns;
          op->stencil_mask_bytes = mbytes;
          for (int k = 0; k < ns; ++k) {
            op->sdict.delta[k] = dd[order[k]];
            op->sdict.val[k] = dv[order[k]];
          }
          for (int k = ns; k < ksd::kStencilSlots; ++k) { op->sdict.delta[k] = 0; op->sdict.val[k] = D{}; }
          op->ndvi = 0;
          op->layout = KS_LAYOUT_STENCIL;
          op->bytes_per_nnz = (double)mbytes * (double)nrows / (double)nnz;
          op->aux_bytes = 0.0;
          const size_t mbytes_al = (size_t)round_up((int64_t)nrows + 2, 8) * mbytes;
          KS_HIP(hipMalloc(&op->smask, mbytes_al));
          KS_HIP(hipMemset(op->smask, 0, mbytes_al));
          KS_HIP(hipMemcpy(op->smask, mbytes == 1 ? (const void*)m8.data() : (const void*)m32.data(), (size_t)nrows * mbytes, hipMemcpyHostToDevice));
          op->smask2 = op->smask;
          return op.release();
        }
        KS_REQUIRE(!(fmt && std::string(fmt) == "stencil"), KS_ERR_ARGUMENT, "KS_SPMV_FORMAT=stencil: the rows are not sub-sequences of one entry order");
      }
      if (ok) {
        op->ndvi = (int)dd.size();
        op->dvi_unroll = max_row <= 4 ? 4 : 8;
        // rows per thread: 4 once there are enough 1024-row tiles to fill the device twice over, else fewer
        // (KS_DVI_RPT overrides: 1, 2 or 4)
        // rows per thread (KS_DVI_RPT = 1, 2 or 4).  Measured on the 216^3 Laplacian: 77.8 / 78.7 / 117 us for
        // 1 / 2 / 4 -- the kernel is bound by instruction issue (byte decode, two dictionary reads and one gather per
        // entry), not by memory latency, so more rows per thread only cost occupancy.
        op->dvi_rpt = env_int("KS_DVI_RPT", 1);
        op->bytes_per_nnz = 1.0;
        op->layout = KS_LAYOUT_DVI;
        op->aux_bytes = (op->ptr64 ? 8.0 : 4.0) * (double)(nrows + 1);
        op->rowptr = upload_ptr(rp, op->ptr64);
        KS_HIP(hipMalloc(&op->codes, (size_t)nnz + 64));
        KS_HIP(hipMemset(op->codes, 0, (size_t)nnz + 64));
        KS_HIP(hipMalloc(&op->ddelta, 256 * 4));
        KS_HIP(hipMalloc(&op->val, 256 * sizeof(D)));
        KS_HIP(hipMemcpy(op->codes, codes.data(), (size_t)nnz, hipMemcpyHostToDevice));
        KS_HIP(hipMemcpy(op->ddelta, dd.data(), dd.size() * 4, hipMemcpyHostToDevice));
        KS_HIP(hipMemcpy(op->val, dv.data(), dv.size() * sizeof(D), hipMemcpyHostToDevice));
        return op.release();
      }
    }
  }
  // Value-indexed layout (k_spmv_csr<.., VI>): at most 256 distinct stored values (compared bit for bit, so
  // -0.0 and NaN payloads survive) and every column index below 2^24.  KS_SPMV_FORMAT=csr keeps plain CSR.
  std::vector<D> dict;
  std::vector<int32_t> packed;
  {
    const char* fmt = cb_mode == 2 ? "csr" : std::getenv("KS_SPMV_FORMAT");
    bool try_vi = nnz > 0 && !(fmt && (std::string(fmt) == "csr" || std::string(fmt) == "dvi" || std::string(fmt) == "sell"));
    if (try_vi) {
      struct Key {
        uint64_t a, b;
        bool operator==(const Key& o) const { return a == o.a && b == o.b; }
      };
      struct KeyHash {
        size_t operator()(const Key& k) const { return std::hash<uint64_t>()(k.a * 0x9E3779B97F4A7C15ull ^ k.b); }
      };
      std::unordered_map<Key, int, KeyHash> index;
      Key last_key{0, 0};
      int last_id = 0;
      packed.resize((size_t)nnz);
      for (int64_t p = 0; p < nnz && try_vi; ++p) {
        Key k{0, 0};
        std::memcpy(&k, &vv[p], sizeof(D));
        int id;
        if (p > 0 && k == last_key) {  // runs of equal values are the common case
          if (ci[p] >= (1 << 24)) { try_vi = false; break; }
          packed[p] = (int32_t)(((uint32_t)last_id << 24) | (uint32_t)ci[p]);
          continue;
        }
        auto it = index.find(k);
        if (it == index.end()) {
          if (dict.size() == 256) { try_vi = false; break; }
          id = (int)dict.size();
          index.emplace(k, id);
          dict.push_back(vv[p]);
        } else {
          id = it->second;
        }
        if (ci[p] >= (1 << 24)) { try_vi = false; break; }
        packed[p] = (int32_t)(((uint32_t)id << 24) | (uint32_t)ci[p]);
        last_key = k;
        last_id = id;
      }
    }
    if (!try_vi) { dict.clear(); packed.clear(); }
  }
  op->ndict = (int)dict.size();
  // Storage order.  Sliced ELLPACK (k_spmv_sell, lane = row: coalesced index / value loads and, for banded matrices,
  // coalesced gathers) when slicing the rows 64 at a time pads the matrix by at most 15 % -- uniform row lengths:
  // stencils with variable coefficients, structured finite-element meshes, banded matrices; otherwise (ragged rows,
  // where a lane per row would idle and the gathers are scattered anyway) the non-zero-parallel CSR blocks of k_spmv_csr.
  // KS_SPMV_FORMAT=sell / sellvi force it (KS_SELL_SIGMA = window for sorting rows by length, multiple of 64, default:
  // 1 = no permutation); csr / vi force the CSR blocks.
  {
    const char* fmt = cb_mode == 2 ? "csr" : std::getenv("KS_SPMV_FORMAT");
    const std::string f = fmt ? fmt : "";
    const bool force_sell = f == "sell" || f == "sellvi";
    const bool allow_sell = force_sell || f.empty();
    int sigma = std::max(1, env_int("KS_SELL_SIGMA", 1));
    if (sigma > 1) sigma = (int)round_up(sigma, 64);
    if (f == "sell") { dict.clear(); packed.clear(); op->ndict = 0; }
    if (allow_sell && nrows > 0 && nnz > 0) {
      // slice position -> row (identity unless sigma > 1: stable sort by descending length inside each window)
      std::vector<int32_t> perm;
      if (sigma > 1) {
        perm.resize((size_t)nrows);
        for (int64_t i = 0; i < nrows; ++i) perm[i] = (int32_t)i;
        for (int64_t w0 = 0; w0 < nrows; w0 += sigma) {
          const int64_t w1 = std::min<int64_t>(nrows, w0 + sigma);
          std::stable_sort(perm.begin() + w0, perm.begin() + w1,
                           [&](int32_t x_, int32_t y_) { return rp[x_ + 1] - rp[x_] > rp[y_ + 1] - rp[y_]; });
        }
      }
      auto row_at = [&](int64_t pos) { return sigma > 1 ? (int64_t)perm[pos] : pos; };
      const int64_t nsl = (nrows + 63) / 64;
      std::vector<int64_t> sp((size_t)nsl + 1, 0);
      int64_t wmax = 0;
      for (int64_t sl = 0; sl < nsl; ++sl) {
        int64_t w = 0;
        for (int64_t pos = sl * 64; pos < std::min<int64_t>(nrows, sl * 64 + 64); ++pos) {
          const int64_t r = row_at(pos);
          w = std::max(w, rp[r + 1] - rp[r]);
        }
        wmax = std::max(wmax, w);
        sp[sl + 1] = sp[sl] + 64 * w;
      }
      const int64_t padded = sp[nsl];
      if (force_sell || (double)padded <= 1.15 * (double)nnz + 64.0) {
        KS_REQUIRE(padded < ((int64_t)1 << 40), KS_ERR_ARGUMENT, "sliced-ELLPACK padding explodes: use KS_SPMV_FORMAT=csr");
        if (padded >= (int64_t)2147483647) op->ptr64 = true;
        const bool vi = op->ndict > 0;
        std::vector<int32_t> sc((size_t)padded, -1);
        std::vector<D> sv(vi ? 0 : (size_t)padded);
        for (int64_t sl = 0; sl < nsl; ++sl)
          for (int64_t pos = sl * 64; pos < std::min<int64_t>(nrows, sl * 64 + 64); ++pos) {
            const int64_t r = row_at(pos);
            const int64_t lane = pos - sl * 64;
            for (int64_t p = rp[r], k = 0; p < rp[r + 1]; ++p, ++k) {
              const int64_t q = sp[sl] + k * 64 + lane;
              sc[q] = vi ? packed[p] : ci[p];
              if (!vi) sv[q] = vv[p];
            }
          }
        op->nslices = (int)nsl;
        op->sell_un = wmax <= 4 ? 4 : 8;
        op->sell_entries = padded;
        op->layout = vi ? KS_LAYOUT_SELL_VI : KS_LAYOUT_SELL;
        op->bytes_per_nnz = (vi ? 4.0 : 4.0 + sizeof(D)) * (double)padded / (double)nnz;
        op->aux_bytes = (op->ptr64 ? 8.0 : 4.0) * (double)(nsl + 1) + (sigma > 1 ? 4.0 * (double)nrows : 0.0);
        op->sliceptr = upload_ptr(sp, op->ptr64);
        KS_HIP(hipMalloc(&op->colidx, (size_t)padded * 4 + 16));
        KS_HIP(hipMemcpy(op->colidx, sc.data(), (size_t)padded * 4, hipMemcpyHostToDevice));
        if (vi) {
          KS_HIP(hipMalloc(&op->val, 256 * sizeof(D)));
          KS_HIP(hipMemcpy(op->val, dict.data(), dict.size() * sizeof(D), hipMemcpyHostToDevice));
        } else {
          KS_HIP(hipMalloc(&op->val, (size_t)padded * sizeof(D) + 16));
          KS_HIP(hipMemcpy(op->val, sv.data(), (size_t)padded * sizeof(D), hipMemcpyHostToDevice));
        }
        if (sigma > 1) {
          KS_HIP(hipMalloc(&op->sperm, (size_t)nrows * 4));
          KS_HIP(hipMemcpy(op->sperm, perm.data(), (size_t)nrows * 4, hipMemcpyHostToDevice));
        }
        return op.release();
      }
    }
  }
  // COLUMN BLOCKS (KS_LAYOUT_CSR_CB).  A matrix with scattered columns whose x is larger than one XCD's L2 (4 MiB) runs at
  // the device's random-gather rate (config 3: 59 us at n = 1e6, 5.1x its algorithmic traffic through the fabric).  Split
  // into column blocks -- block b holds the entries with column in [b n/NB, (b+1) n/NB) -- each launch gathers from an
  // x block that stays L2 resident, and because the entries of a row are sorted by column the row sums are simply
  // continued from launch to launch (k_spmv_csr's yacc): same additions in the same order, bit-identical y.  Measured
  // (tools/colblock_probe.py, n = 1e6): 59.5 us whole, 2 blocks 23 + 23 us, 4 blocks 4 x 13 us (launch floor), 8: 8 x 9.
  // Auto: plain CSR row blocks would be used, single GPU, x between 6 and 160 MiB, rows sorted by column and short, and
  // at least half of the entries further than n/16 from the diagonal -> blocks of ~4 MiB of x, at most 8.
  // KS_SPMV_COLBLOCKS = 0 off / k >= 2 force.
  if (cb_mode == 1 && op->ndict == 0 && nnz > 0) {
    const int cb_env = env_int("KS_SPMV_COLBLOCKS", -1);  // (read per upload: tests switch it inside one process)
    int nbk = 0;
    if (cb_env != 0) {
      bool sorted = true;
      int64_t far = 0, maxrow = 0;
      const int64_t fardist = std::max<int64_t>(1, nrows / 16);
      for (int64_t r = 0; r < nrows && sorted; ++r) {
        maxrow = std::max(maxrow, rp[r + 1] - rp[r]);
        for (int64_t q = rp[r]; q < rp[r + 1]; ++q) {
          if (q > rp[r] && ci[q] < ci[q - 1]) { sorted = false; break; }
          far += std::llabs((int64_t)ci[q] - r) > fardist;
        }
      }
      const double xmb = (double)nrows * sizeof(D) / (1 << 20);
      if (sorted && maxrow <= 4 * kBlock) {
        if (cb_env >= 2) nbk = cb_env;
        // block width ~ 4 MiB of x (measured optimum at n = 1e6: 2 blocks, 2e6: 4 blocks); beyond 8 blocks the y that is
        // written and read back between the launches (16 n bytes each) eats the gain (n = 1e7: 8 blocks -11 %, 16: +35 %)
        else if (xmb >= 6.0 && xmb <= 160.0 && 2 * far >= nnz) nbk = std::min(8, std::max(2, (int)std::lround(xmb / 4.0)));
      }
    }
    if (nbk >= 2) {
      nbk = std::min(nbk, ksd::kCbMaxBlocks);
      // single-launch form (k_spmv_csr_cb): largest segment (entries of a tile of 256 * RPT rows inside one column block)
      // for every candidate RPT
      constexpr int kRptCand[5] = {1, 2, 4, 8, 16};
      int64_t maxseg[5] = {0, 0, 0, 0, 0};
      bool small_ptrs = true;
      for (int b = 0; b < nbk; ++b) {
        const int64_t lo = (int64_t)b * nrows / nbk, hi = (b + 1 == nbk) ? (int64_t)1 << 40 : (int64_t)(b + 1) * nrows / nbk;
        std::vector<int64_t> rpb((size_t)nrows + 1, 0);
        std::vector<int32_t> cib;
        std::vector<D> vvb;
        for (int64_t r = 0; r < nrows; ++r) {
          for (int64_t q = rp[r]; q < rp[r + 1]; ++q)
            if (ci[q] >= lo && ci[q] < hi) { cib.push_back(ci[q]); vvb.push_back(vv[q]); }
          rpb[r + 1] = (int64_t)cib.size();
        }
        for (int k = 0; k < 5; ++k) {
          const int64_t tr = (int64_t)kBlock * kRptCand[k];
          for (int64_t r0 = 0; r0 < nrows; r0 += tr) maxseg[k] = std::max(maxseg[k], rpb[std::min(nrows, r0 + tr)] - rpb[r0]);
        }
        op->cblocks.emplace_back(make_csr<D>(ctx, nrows, (int64_t)cib.size(), rpb, cib, vvb, 2));
        small_ptrs = small_ptrs && !op->cblocks.back()->ptr64;
      }
      // Measured (tools/cb_single_ab.py, profiles/r03_column_blocks.txt): the single launch wins where the y round trips of
      // many blocks hurt (n = 1e7, 8 blocks: 858 -> 823 us) and loses a little where two to four launches were already close
      // to what bounds this product -- the rate at which an XCD's L2 hands out randomly addressed lines, 5e6 of them for
      // 1e6 rows: 46 us either way at n = 1e6, 100 vs 107 us at 2e6.  So: single launch from 5 blocks on
      // (KS_SPMV_CB_SINGLE=0 never, KS_SPMV_CB_RPT=k forces it with k sub-tiles per workgroup).
      const int rpt_force = env_int("KS_SPMV_CB_RPT", 0);
      if (small_ptrs && env_int("KS_SPMV_CB_SINGLE", 1) && (nbk > 4 || rpt_force > 0)) {
        // all tiles resident at once (one round of workgroups keeps them in step on the same column block): the smallest RPT
        // whose tile count fits, among those whose segments fit the LDS depth (8 x 256 products, 16 x 256 for Float64)
        const int nimax = (int)(ksd::kSpmvCapBytes / (kBlock * sizeof(D)));  // 16 (Float64) / 8 (ComplexF64)
        const int rpt_env = env_int("KS_SPMV_CB_RPT", 0);
        int best = -1;
        for (int k = 0; k < 5; ++k) {
          const int ni = maxseg[k] <= 8 * kBlock ? 8 : (maxseg[k] <= 16 * kBlock && nimax >= 16 ? 16 : 0);
          if (!ni) break;  // (segments only grow with RPT)
          best = k;
          const int64_t ntiles = (nrows + (int64_t)kBlock * kRptCand[k] - 1) / ((int64_t)kBlock * kRptCand[k]);
          if (rpt_env ? kRptCand[k] >= rpt_env : ntiles <= (int64_t)ctx->num_cu * (ni == 8 ? 8 : 4)) break;
        }
        if (best >= 0) {
          op->cb_rpt = kRptCand[best];
          op->cb_ni = maxseg[best] <= 8 * kBlock ? 8 : 16;
        }
      }
      op->layout = KS_LAYOUT_CSR_CB;
      op->bytes_per_nnz = 4.0 + sizeof(D);
      op->aux_bytes = 0.0;
      for (auto& cbk : op->cblocks) op->aux_bytes += cbk->aux_bytes;
      if (!op->cb_rpt) op->aux_bytes += (double)(nbk - 1) * 2.0 * sizeof(D) * (double)nrows;  // y written and read back between the blocks
      return op.release();
    }
  }
  // Row blocks of k_spmv_csr.  A block holds at most ni * 256 products in LDS (<= 32 KiB; KS_SPMV_NI overrides), so
  // regular matrices get full 256-row blocks and the LDS footprint (occupancy) follows the matrix.  Greedy pass over the
  // rows: close the block at 256 rows or when the next row would overflow it; a row longer than the capacity becomes a
  // block of its own (handled by all 256 threads).
  {
    const int nimax = (int)(ksd::kSpmvCapBytes / (kBlock * sizeof(D)));  // 16 (Float64) / 8 (ComplexF64)
    // depth from the 90th percentile of the non-zeros of fixed 256-row tiles: a regular matrix gets exactly what its
    // tiles need (7-point stencil: 1792 -> 7; 12 measured 14 % slower than 7 or 8 there: LDS footprint), the heavy tail
    // of a skewed one gets shorter blocks instead of inflating everybody's LDS
    std::vector<int64_t> tile_nnz;
    for (int64_t r0 = 0; r0 < nrows; r0 += ksd::kSpmvRows) tile_nnz.push_back(rp[std::min<int64_t>(nrows, r0 + ksd::kSpmvRows)] - rp[r0]);
    int64_t t90 = 0;
    if (!tile_nnz.empty()) {
      const size_t k = (tile_nnz.size() - 1) * 9 / 10;
      std::nth_element(tile_nnz.begin(), tile_nnz.begin() + k, tile_nnz.end());
      t90 = tile_nnz[k];
    }
    const int need = (int)((t90 + kBlock - 1) / kBlock);
    int ni = need <= 4 ? 4 : need <= 7 ? 7 : need <= 8 ? 8 : need <= 12 ? 12 : 16;
    ni = env_int("KS_SPMV_NI", ni);
    if (ni != 4 && ni != 7 && ni != 8 && ni != 12 && ni != 16) ni = 16;
    ni = std::min(ni, nimax);
    op->ni = ni;
    {
      // ROW-GATHER or NON-ZERO-PARALLEL gathers (k_spmv_csr): with lane = row the gathers of one instruction are coalesced
      // when neighbouring rows reference neighbouring columns (banded / stencil / FEM matrices: 212 -> 204 us on the 216^3
      // Laplacian, 0.62 -> 0.64 of the HBM spec), and a chain of dependent LDS reads and scattered loads when they do not
      // (hashed columns: 46.5 -> 49.5 us, heavy-tailed rows 109 -> 125 us).  Decided once from the matrix: the share of
      // consecutive row pairs whose first stored columns are at most 16 apart.  KS_SPMV_CSR_ROWGATHER=0/1 forces.
      int64_t pairs = 0, close = 0;
      const int64_t stride = std::max<int64_t>(1, nrows / 65536);
      for (int64_t r = 0; r + 1 < nrows; r += stride) {
        if (rp[r + 1] == rp[r] || rp[r + 2] == rp[r + 1]) continue;
        ++pairs;
        const int64_t d = (int64_t)ci[rp[r + 1]] - (int64_t)ci[rp[r]];
        if (d >= -16 && d <= 16) ++close;
      }
      const int rg_env = env_int("KS_SPMV_CSR_ROWGATHER", -1);
      op->row_gather = rg_env >= 0 ? rg_env != 0 : (pairs > 0 && 2 * close >= pairs);
    }
    const int64_t cap = (int64_t)ni * kBlock;
    std::vector<int64_t> bp{0};
    std::vector<int32_t> br{0}, part, lrow, lfirst{0};
    int64_t r = 0;
    while (r < nrows) {
      const int64_t first = rp[r + 1] - rp[r];
      if (first > cap) {  // long row: chunk blocks of <= cap entries, all with row range [r, r+1)
        for (int64_t q = rp[r]; q < rp[r + 1]; q += cap) {
          part.push_back((int32_t)lfirst.back() + (int32_t)((q - rp[r]) / cap));
          br.push_back((int32_t)(r + 1));
          bp.push_back(std::min(q + cap, rp[r + 1]));
          if (q + cap < rp[r + 1]) br.back() = (int32_t)r;  // the next chunk starts at the same row
        }
        lrow.push_back((int32_t)r);
        lfirst.push_back(lfirst.back() + (int32_t)((first + cap - 1) / cap));
        op->nlong++;
        r += 1;
        continue;
      }
      int64_t e = r + 1;
      while (e < nrows && e - r < ksd::kSpmvRows && rp[e + 1] - rp[r] <= cap && rp[e + 1] - rp[e] <= cap) ++e;
      part.push_back(-1);
      br.push_back((int32_t)e);
      bp.push_back(rp[e]);
      r = e;
    }
    op->nblk = (int)br.size() - 1;
    KS_REQUIRE((int64_t)br.size() - 1 < (int64_t)2147483647, KS_ERR_ARGUMENT, "too many row blocks");
    op->blkptr = upload_ptr(bp, op->ptr64);
    KS_HIP(hipMalloc(&op->blkrow, std::max<size_t>(br.size() * 4, 16)));
    KS_HIP(hipMemcpy(op->blkrow, br.data(), br.size() * 4, hipMemcpyHostToDevice));
    if (op->nlong > 0) {
      KS_HIP(hipMalloc(&op->blkpart, part.size() * 4));
      KS_HIP(hipMemcpy(op->blkpart, part.data(), part.size() * 4, hipMemcpyHostToDevice));
      KS_HIP(hipMalloc(&op->lpart, (size_t)lfirst.back() * sizeof(D)));
      KS_HIP(hipMalloc(&op->lrow, lrow.size() * 4));
      KS_HIP(hipMemcpy(op->lrow, lrow.data(), lrow.size() * 4, hipMemcpyHostToDevice));
      KS_HIP(hipMalloc(&op->lfirst, lfirst.size() * 4));
      KS_HIP(hipMemcpy(op->lfirst, lfirst.data(), lfirst.size() * 4, hipMemcpyHostToDevice));
    }
  }
  op->layout = op->ndict > 0 ? KS_LAYOUT_CSR_VI : KS_LAYOUT_CSR;
  op->bytes_per_nnz = op->ndict > 0 ? 4.0 : 4.0 + sizeof(D);
  op->aux_bytes = (op->ptr64 ? 8.0 : 4.0) * (double)(nrows + 1 + 2 * ((int64_t)op->nblk + 1));
  op->rowptr = upload_ptr(rp, op->ptr64);
  KS_HIP(hipMalloc(&op->colidx, (size_t)(nnz + 2) * 4 + 16));
  if (op->ndict > 0) {
    KS_HIP(hipMalloc(&op->val, 256 * sizeof(D)));
    KS_HIP(hipMemcpy(op->colidx, packed.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
    KS_HIP(hipMemcpy(op->val, dict.data(), dict.size() * sizeof(D), hipMemcpyHostToDevice));
  } else {
    KS_HIP(hipMalloc(&op->val, (size_t)(nnz + 2) * sizeof(D) + 16));
    if (nnz) {
      KS_HIP(hipMemcpy(op->colidx, ci.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
      KS_HIP(hipMemcpy(op->val, vv.data(), (size_t)nnz * sizeof(D), hipMemcpyHostToDevice));
    }
  }
  return op.release();
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
constexpr size_t kCtlStateSlot = 128;  // bytes reserved for the DevState inside the control block

struct ks_workspace {
  ks_ctx* ctx = nullptr;
  int dtype = KS_F64;
  int64_t n = 0, n_global = 0, row_begin = 0, ld = 0;
  int maxdim = 0;
  size_t esz = 8;
  void* V = nullptr;        // device, ld x (maxdim+1)
  void* Vbase = nullptr;    // what hipFree gets: == V, or V - guard when KS_GUARD=1 put canary zones around the basis
  size_t guard = 0, vbytes = 0;
  int place_failed = 0;         // candidate allocations the search was refused
  int place_candidates = 0;     // placement search of ks_workspace_create: candidates timed, fastest / slowest calibration time
  double place_best_ms = 0.0, place_worst_ms = 0.0;
  void* H = nullptr;        // pinned host, (maxdim+1) x maxdim
  void* Q = nullptr;        // pinned host, maxdim x maxdim
  void* Hd = nullptr;       // device mirror the expansion kernels write into
  void* Hstage = nullptr;   // pinned host staging for Hd
  void* Hscratch = nullptr; // device, maxdim+1 elements (verbs that must not touch H)
  void* partial = nullptr;  // device, nblocks x pstride elements
  void* partial_s = nullptr; // device, same size: k_dots' partial sums in the two-pass expansion (both sets live at once)
  double* partial2 = nullptr;  // device, nblocks doubles
  void* coef = nullptr;     // device, pstride + 136 elements
  void* red = nullptr;      // device, 2 pstride + 8 elements (all-reduce buffer)
  double* scal = nullptr;   // device, 8 doubles
  double* scal_h = nullptr; // pinned host, 8 doubles
  void* coef_h = nullptr;   // pinned host, pstride elements
  // CONTROL BLOCK: one device allocation [ Hd | DevState | colscale ] mirrored by one pinned host allocation
  // [ Hstage | st_h | cs_h ], so that what an expansion batch needs from / hands back to the host travels in ONE copy
  // each way (each hipMemcpyAsync is a blit kernel plus a launch gap; round 1 issued four per restart cycle with two
  // host synchronisations in between: profiles/r02_restart_bubble.txt)
  size_t hd_bytes = 0;      // bytes of Hd up to the DevState (64-byte aligned)
  DevState* st = nullptr;   // device, inside the Hd allocation
  DevState* st_h = nullptr; // pinned host, inside the Hstage allocation
  double* cs_h = nullptr;   // pinned host image of colscale, inside the Hstage allocation
  bool colscale_dirty = false;  // hostscale changed on the host side: upload with the next batch's state
  // early hand-over of H at the end of the expansion a restart follows (HipBackend::iterate_arnoldi_early): a second
  // pinned image of [ Hd | DevState ], published by the device right after the last step's k_fin_mid_def
  void* Hstage_early = nullptr;
  std::vector<char> Hbackup;    // host H before the early part of the restart step touched it
  // MAILBOX: the device publishes [ H columns | DevState ] into the pinned images itself (k_publish) and releases a
  // sequence number; the host spins on it.  mbox[0]: early hand-over, mbox[8]: end of the batch (64 bytes apart).
  uint64_t* mbox = nullptr;     // pinned host
  uint64_t* mbox_dev = nullptr; // the same memory through its device pointer
  void* Hstage_dev = nullptr;   // device pointers of the pinned images
  void* Hstage_early_dev = nullptr;
  uint64_t mbox_seq = 0;
  bool use_mbox = true;         // KS_MAILBOX (read at creation); 0: hipMemcpyAsync + hipStreamSynchronize
  // IMPLICIT SECOND PASS (ks_kernels.hpp, k_fin_dots_t / k_fin_mid_t): V_true = S * T.  T and the vector g live in the
  // control block behind the column factors; columns < ntrue are ordinary, columns ntrue..t_hi are "T-lazy".
  int passes = 2;               // 2 = implicit second pass (default), 3 = second pass applied to the vector (KS_PASSES at
                                // creation, ks_workspace_set_passes afterwards)
  double max_ratio = 1e-3;      // largest ||c|| / beta an implicit second pass carries (DevState::max_ratio); a step beyond
                                // it is redone in the explicit form.  KS_IMPLICIT_MAX_RATIO at creation, ks_workspace_set_passes
  // PROVENANCE of the factorisation.  The implicit second pass reads EARLIER columns of H (g = H c) and
  // relies on the Arnoldi relation A V[:, 0:k) = V[:, 0:k+1) H[0:k+1, 0:k) for them; the reference's iterate_arnoldi!
  // never does.  prov_k >= 0: the library itself produced (or the caller asserted, ks_workspace_assert_arnoldi) steps
  // 1..prov_k and nobody wrote to V since; Hshadow = the host H as the library last left it.  A batch starting at
  // step `from` takes the implicit form only if prov_k >= from - 1 AND the caller's H[:, 0:from-1) still equals the
  // shadow bit for bit; otherwise it runs the explicit three-pass form, which needs neither.  -1: unknown.
  int prov_k = -1;
  std::vector<char> Hshadow;
  size_t off_T = 0, off_g = 0, ctl_bytes = 0;
  int ldt = 0;
  void* Td = nullptr;           // device, ldt x ldt, inside the Hd allocation
  void* gd = nullptr;           // device, ldt elements, inside the Hd allocation
  void* Th = nullptr;           // pinned host image of T, inside the Hstage allocation (valid after a batch)
  unsigned* ctr = nullptr;      // device: arrival counter of the reduction kernels
  bool t_lazy = false;
  int ntrue = 0, t_hi = -1;
  void* Qd = nullptr;       // device, maxdim x maxdim
  void* Qstage = nullptr;   // pinned host
  void* oop = nullptr;      // device, 2 x ld elements (zero pads): scratch vectors of the out-of-place updates
  bool oop_full = false;    // the previous batch took the second DGKS pass in >= 90 % of its steps
  int oop_mode = 2;         // KS_OOP at creation: 0 in place, 2 scratch product (default), 1 both projections out of place
  void* tmp = nullptr;      // device scratch, lazily sized
  size_t tmp_bytes = 0;
  void* tmp2 = nullptr;
  size_t tmp2_bytes = 0;
  int pstride = 0;
  // lazy normalisation (fused Float64 path): columns lazy_lo..lazy_hi are stored unnormalised in HBM with
  // factor hostscale[c] (device mirror colscale[c]); everything else has factor 1
  double* colscale = nullptr;   // device, maxdim+2 doubles
  std::vector<double> hostscale;
  std::vector<double> ones;
  int lazy_lo = 1 << 30, lazy_hi = -1;
  bool has_lazy() const { return lazy_hi >= lazy_lo || t_lazy; }
  int nb = 0;               // streaming workgroups (capped for small problems)
  int pnb = 0;              // column stride of `partial` (>= every producer's grid)
  uint64_t seed = 20240917ull;
  uint64_t rng_count = 0;

  void* col(int j) const { return static_cast<char*>(V) + (size_t)j * ld * esz; }
  void* ensure_tmp(size_t bytes) {
    if (bytes > tmp_bytes) {
      (void)hipFree(tmp);
      tmp = nullptr;
      KS_HIP(hipMalloc(&tmp, bytes));
      tmp_bytes = bytes;
    }
    return tmp;
  }
  void* ensure_tmp2(size_t bytes) {
    if (bytes > tmp2_bytes) {
      (void)hipFree(tmp2);
      tmp2 = nullptr;
      KS_HIP(hipMalloc(&tmp2, bytes));
      tmp2_bytes = bytes;
    }
    return tmp2;
  }
  ~ks_workspace() {
    (void)hipFree(Vbase ? Vbase : V); (void)hipHostFree(H); (void)hipHostFree(Q); (void)hipFree(Hd); (void)hipHostFree(Hstage);
    (void)hipFree(Hscratch); (void)hipFree(partial); (void)hipFree(partial_s); (void)hipFree(partial2); (void)hipFree(coef); (void)hipFree(red);
    (void)hipFree(scal); (void)hipHostFree(scal_h); (void)hipHostFree(coef_h);
    (void)hipFree(Qd); (void)hipHostFree(Qstage); (void)hipFree(tmp); (void)hipFree(tmp2); (void)hipFree(oop);
    (void)hipHostFree(Hstage_early); (void)hipHostFree(mbox); (void)hipFree(ctr);
  }
};

namespace {

inline int cap_blocks(const ks_workspace* ws, int nb, int packs_per_iter) {
  const int64_t npacks = ws->ld * (int64_t)ws->esz / 16;
  const int64_t want = std::max<int64_t>(1, npacks / (2 * (int64_t)packs_per_iter));
  return (int)std::min<int64_t>(nb, want);
}

uint64_t next_seed(ks_workspace* ws) {
  const uint64_t s = ws->seed + ws->rng_count * 0x9E3779B97F4A7C15ull;
  ws->rng_count++;
  return s;
}

// ------------------------------------------------------------------------------------------------
// kernel launch helpers (T = host scalar type; D = device scalar type)
// ------------------------------------------------------------------------------------------------
// Streaming kernels partition the rows into one contiguous range per workgroup, so every workgroup
// must be co-resident: the grid is num_cu x min(KS_BPC, occupancy of that kernel).
template <class K> int resident_blocks(ks_ctx* ctx, K kernel, size_t smem, int& cache) {
  if (cache < 0) {
    int occ = 0;
    KS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kBlock, smem));
    cache = std::max(1, std::min(occ, ctx->bpc));
  }
  return ctx->num_cu * cache;
}
// Small problems (n = 1e6, or 1/8 of 1e7 per GPU): do not launch more workgroups than there are
// `packs_per_iter`-sized pieces of work, two iterations each.

// (k_dots with 2 / 4 packs per lane measured 3 % / 70 % slower than 1: the accumulators already fill the
// register file; the update kernels on the other hand gain 8 % from 8 packs per lane.)
template <class D, int NC4> int dots_blocks(ks_workspace* ws) {
  static int cache = -1;
  return resident_blocks(ws->ctx, ksd::k_dots<D, NC4, 1>, 0, cache);
}

template <class D> int dots_blocks_for(ks_workspace* ws, int nc4) {
  switch (nc4) {
    case 1: return dots_blocks<D, 1>(ws);
    case 2: return dots_blocks<D, 2>(ws);
    case 3: return dots_blocks<D, 3>(ws);
    case 4: return dots_blocks<D, 4>(ws);
    case 5: return dots_blocks<D, 5>(ws);
    case 6: return dots_blocks<D, 6>(ws);
    case 7: return dots_blocks<D, 7>(ws);
    case 8: return dots_blocks<D, 8>(ws);
    case 9: return dots_blocks<D, 9>(ws);
    default: return dots_blocks<D, 10>(ws);
  }
}

template <class D, int NC4>
void launch_dots_nc(ks_workspace* ws, int nb, const D* V, int jc, const D* w, D* partial, int norm_slot, int pass,
                    const DevState* st) {
  ksd::k_dots<D, NC4, 1><<<nb, kBlock, 0, ws->ctx->stream>>>(V, ws->ld, jc, w, partial, ws->pnb, norm_slot, pass, st);
}

// partial[b][0..j) = V[:,0:j)^H w (block-local), partial[b][j] = |w|^2 (block-local); returns the
// number of workgroups that wrote partials
template <class D> int launch_dots(ks_workspace* ws, int j, const D* w, int pass, const DevState* st, D* partial_out = nullptr) {
  const D* V = static_cast<const D*>(ws->V);
  D* partial = partial_out ? partial_out : static_cast<D*>(ws->partial);
  const int nb = cap_blocks(ws, dots_blocks_for<D>(ws, (std::min(j, 40) + 3) / 4), kBlock);  // first chunk is the widest
  for (int c0 = 0; c0 < j; c0 += 40) {
    const int jc = std::min(40, j - c0);
    const int norm_slot = (c0 + 40 >= j) ? (j - c0) : -1;
    const D* Vc = V + (size_t)c0 * ws->ld;
    D* pc = partial + (size_t)c0 * ws->pnb;  // partial is [column][workgroup], column stride ws->pnb
    switch ((jc + 3) / 4) {
      case 1: launch_dots_nc<D, 1>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 2: launch_dots_nc<D, 2>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 3: launch_dots_nc<D, 3>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 4: launch_dots_nc<D, 4>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 5: launch_dots_nc<D, 5>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 6: launch_dots_nc<D, 6>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 7: launch_dots_nc<D, 7>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 8: launch_dots_nc<D, 8>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      case 9: launch_dots_nc<D, 9>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
      default: launch_dots_nc<D, 10>(ws, nb, Vc, jc, w, pc, norm_slot, pass, st); break;
    }
  }
  return nb;
}

// reduce the per-workgroup partials (+ all-reduce over ranks) and post-process on the device
template <class D> void launch_fin_dots(ks_workspace* ws, int nbd, int j, D* Hcol, int pass, DevState* st) {
  ks_ctx* c = ws->ctx;
  D* partial = static_cast<D*>(ws->partial);
  D* red = static_cast<D*>(ws->red);
  D* coef = static_cast<D*>(ws->coef);
  // one workgroup per column (0..j-1 = inner products, j = |w|^2; pass 2 ignores column j)
  const int ncol = pass == 1 ? j + 1 : j;
  if (!c->distributed()) {
    ksd::k_fin_dots<D><<<ncol, kBlock, 0, c->stream>>>(partial, nbd, ws->pnb, j, red, Hcol, coef, pass, 0, st);
  } else {
    ksd::k_fin_dots<D><<<ncol, kBlock, 0, c->stream>>>(partial, nbd, ws->pnb, j, red, Hcol, coef, pass, 1, st);
    c->allreduce(reinterpret_cast<double*>(red), ncol * (int)(sizeof(D) / 8));
    ksd::k_fin_dots<D><<<ncol, 64, 0, c->stream>>>(partial, nbd, ws->pnb, j, red, Hcol, coef, pass, 2, st);
  }
}

template <class D> void launch_fin_norm(ks_workspace* ws, int nbp, int j, D* Hsub, int pass, DevState* st) {
  ks_ctx* c = ws->ctx;
  double* red = reinterpret_cast<double*>(ws->red);
  if (!c->distributed()) {
    ksd::k_fin_norm<D><<<1, kBlock, 0, c->stream>>>(ws->partial2, nbp, red, Hsub, j, pass, 0, st);
  } else {
    ksd::k_fin_norm<D><<<1, kBlock, 0, c->stream>>>(ws->partial2, nbp, red, Hsub, j, pass, 1, st);
    c->allreduce(red, 1);
    ksd::k_fin_norm<D><<<1, kBlock, 0, c->stream>>>(ws->partial2, nbp, red, Hsub, j, pass, 2, st);
  }
}

// fused first projection + second-pass inner products (k_axpy_dots_cs, j <= 64); returns workgroups used.
// NCW = ceil(j/4) columns per wave; U packs per lane and iteration: 4 up to NCW = 10, 2 above (register budget).
constexpr int kFusedMaxJ = 64;
template <class D, int NCW, int U, int WB> int launch_axpy_dots_nc(ks_workspace* ws, int j, D* w, int defer, D* wdst = nullptr) {
  static int cache = -1;
  static const int plain = env_int("KS_FUSED_PLAIN_STORE", 0);
  const int nb = cap_blocks(ws, resident_blocks(ws->ctx, ksd::k_axpy_dots_cs<D, NCW, U, WB>, 0, cache), 64 * U);
  ksd::k_axpy_dots_cs<D, NCW, U, WB><<<nb, kBlock, 0, ws->ctx->stream>>>(static_cast<const D*>(ws->V), ws->ld, j, w,
                                                                       static_cast<const D*>(ws->coef),
                                                                       static_cast<D*>(ws->partial), ws->pnb, ws->partial2,
                                                                       ws->st, defer, wdst, plain && ws->passes == 2);
  return nb;
}
// Write-back staging depth WB of the projection kernel.  The kernel writes ONE column next to the j+1 it reads, and that
// write stream is what keeps it below k_dots: with its stores removed it runs at 7.0 TB/s, with 32 KiB bursts (WB = 8
// at U = 4) at 5.9, with 64 KiB bursts at 6.6, with 96 KiB bursts at 6.7 (tools/fused_probe.hip,
// profiles/r02_write_bursts.txt) -- every burst makes the memory channels turn around, so fewer and larger ones win even
// at one workgroup per CU (the LDS of a gfx950 CU is 160 KiB: tbuf 32 KiB + 96 KiB of staged rows).  KS_FUSED_WB=8 / 24
// forces one setting.
template <class D> int launch_axpy_dots(ks_workspace* ws, int j, D* w, int defer, D* wdst = nullptr) {
  KS_REQUIRE(j >= 1 && j <= kFusedMaxJ, KS_ERR_INTERNAL, "fused projection kernel covers 1 <= j <= 64");
  // (one workgroup per CU is too little parallelism while the basis is cache resident: 8 MiB columns lose 3 % with the deep
  // staging, 80 MiB columns gain 11 % -- deep staging from KS_FUSED_WB_MIN_MB (24) MiB per column on)
  static const int wb_env = env_int("KS_FUSED_WB", 0);
  static const int wb_min_mb = env_int("KS_FUSED_WB_MIN_MB", 24);
  const bool wb_small = wb_env ? wb_env <= 8 : (ws->ld * (int64_t)sizeof(D) < ((int64_t)wb_min_mb << 20));
  if (wb_small) {
    switch ((j + 3) / 4) {
      case 1: return launch_axpy_dots_nc<D, 1, 4, 8>(ws, j, w, defer, wdst);
      case 2: return launch_axpy_dots_nc<D, 2, 4, 8>(ws, j, w, defer, wdst);
      case 3: return launch_axpy_dots_nc<D, 3, 4, 8>(ws, j, w, defer, wdst);
      case 4: return launch_axpy_dots_nc<D, 4, 4, 8>(ws, j, w, defer, wdst);
      case 5: return launch_axpy_dots_nc<D, 5, 4, 8>(ws, j, w, defer, wdst);
      case 6: return launch_axpy_dots_nc<D, 6, 4, 8>(ws, j, w, defer, wdst);
      case 7: return launch_axpy_dots_nc<D, 7, 4, 8>(ws, j, w, defer, wdst);
      case 8: return launch_axpy_dots_nc<D, 8, 4, 8>(ws, j, w, defer, wdst);
      case 9: return launch_axpy_dots_nc<D, 9, 4, 8>(ws, j, w, defer, wdst);
      case 10: return launch_axpy_dots_nc<D, 10, 4, 8>(ws, j, w, defer, wdst);
      case 11: return launch_axpy_dots_nc<D, 11, 2, 8>(ws, j, w, defer, wdst);
      case 12: return launch_axpy_dots_nc<D, 12, 2, 8>(ws, j, w, defer, wdst);
      case 13: return launch_axpy_dots_nc<D, 13, 2, 8>(ws, j, w, defer, wdst);
      case 14: return launch_axpy_dots_nc<D, 14, 2, 8>(ws, j, w, defer, wdst);
      case 15: return launch_axpy_dots_nc<D, 15, 2, 8>(ws, j, w, defer, wdst);
      default: return launch_axpy_dots_nc<D, 16, 2, 8>(ws, j, w, defer, wdst);
    }
  }
  switch ((j + 3) / 4) {
    case 1: return launch_axpy_dots_nc<D, 1, 4, 24>(ws, j, w, defer, wdst);
    case 2: return launch_axpy_dots_nc<D, 2, 4, 24>(ws, j, w, defer, wdst);
    case 3: return launch_axpy_dots_nc<D, 3, 4, 24>(ws, j, w, defer, wdst);
    case 4: return launch_axpy_dots_nc<D, 4, 4, 24>(ws, j, w, defer, wdst);
    case 5: return launch_axpy_dots_nc<D, 5, 4, 24>(ws, j, w, defer, wdst);
    case 6: return launch_axpy_dots_nc<D, 6, 4, 24>(ws, j, w, defer, wdst);
    case 7: return launch_axpy_dots_nc<D, 7, 4, 24>(ws, j, w, defer, wdst);
    case 8: return launch_axpy_dots_nc<D, 8, 4, 24>(ws, j, w, defer, wdst);
    case 9: return launch_axpy_dots_nc<D, 9, 4, 24>(ws, j, w, defer, wdst);
    case 10: return launch_axpy_dots_nc<D, 10, 4, 24>(ws, j, w, defer, wdst);
    case 11: return launch_axpy_dots_nc<D, 11, 2, 48>(ws, j, w, defer, wdst);
    case 12: return launch_axpy_dots_nc<D, 12, 2, 48>(ws, j, w, defer, wdst);
    case 13: return launch_axpy_dots_nc<D, 13, 2, 48>(ws, j, w, defer, wdst);
    case 14: return launch_axpy_dots_nc<D, 14, 2, 48>(ws, j, w, defer, wdst);
    case 15: return launch_axpy_dots_nc<D, 15, 2, 48>(ws, j, w, defer, wdst);
    default: return launch_axpy_dots_nc<D, 16, 2, 48>(ws, j, w, defer, wdst);
  }
}

// Enqueue orthogonalize!(arnoldi, j) (src/expansion.jl:69-109) entirely on the device, EAGER form (maxdim > 64, or
// KS_NO_DEFER=1 for debugging): two un-fused DGKS passes (the second one skips itself unless the first requested
// it), H column into Hd, v ./= wnorm.  Four passes over V plus the scaling pass; everything up to maxdim = 64
// takes the fused, lazily normalised path below instead.
template <class D> void enqueue_orthogonalize(ks_workspace* ws, int j) {
  hipStream_t s = ws->ctx->stream;
  D* w = static_cast<D*>(ws->col(j));
  D* Hd = static_cast<D*>(ws->Hd);
  const int ldh = ws->maxdim + 1;
  D* Hcol = Hd + (size_t)(j - 1) * ldh;
  const D* V = static_cast<const D*>(ws->V);
  const double nb8 = (double)ws->n * sizeof(D);  // bytes of one column
  for (int pass = 1; pass <= 2; ++pass) {
    int nbd;
    {
      ProfScope ps(ws->ctx, KSP_DOTS, nb8 * (j + 1));       // read V[:,0:j) and w
      nbd = launch_dots<D>(ws, j, w, pass, ws->st);
    }
    {
      ProfScope ps(ws->ctx, KSP_FIN, 0.0);
      launch_fin_dots<D>(ws, nbd, j, Hcol, pass, ws->st);
    }
    {
      ProfScope ps(ws->ctx, KSP_AXPY, nb8 * (j + 2));       // read V[:,0:j), read + write w
      ksd::k_axpy<D><<<ws->nb, kBlock, 0, s>>>(V, ws->ld, j, w, static_cast<const D*>(ws->coef), ws->partial2, pass, ws->st);
    }
    {
      ProfScope ps(ws->ctx, KSP_FIN, 0.0);
      launch_fin_norm<D>(ws, ws->nb, j, Hcol + j, pass, ws->st);
    }
  }
  {
    ProfScope ps(ws->ctx, KSP_SCALE, nb8 * 2);
    ksd::k_scale<D><<<ws->nb, kBlock, 0, s>>>(w, ws->ld, 0.0, ws->st);
  }
  KS_HIP(hipGetLastError());
}

// Lazy columns -> ordinary columns: one scaling pass per lazy column (only needed when something other than
// the expansion / restart-rotation pair is about to read V).
inline void reset_lazy(ks_workspace* ws) {
  ws->t_lazy = false;
  ws->t_hi = -1;
  if (!(ws->lazy_hi >= ws->lazy_lo)) return;
  for (int c = ws->lazy_lo; c <= ws->lazy_hi; ++c) ws->hostscale[c] = 1.0;
  ws->colscale_dirty = true;  // the device copy is only read inside expansion batches: uploaded with the next one's state
  ws->lazy_lo = 1 << 30;
  ws->lazy_hi = -1;
}
template <class D> void materialize_t(ks_workspace* ws);
inline void materialize(ks_workspace* ws) {
  if (ws->t_lazy) {  // implicit second pass: V_true = S T, one in-place triangular product over the T-lazy columns
    if (ws->dtype == KS_F64) materialize_t<double>(ws);
    else materialize_t<cd>(ws);
  }
  if (!(ws->lazy_hi >= ws->lazy_lo)) return;
  for (int c = ws->lazy_lo; c <= ws->lazy_hi; ++c)
    if (ws->hostscale[c] != 1.0) {
      if (ws->dtype == KS_F64) ksd::k_scale<double><<<ws->nb, kBlock, 0, ws->ctx->stream>>>(static_cast<double*>(ws->col(c)), ws->ld, ws->hostscale[c], nullptr);
      else ksd::k_scale<cd><<<ws->nb, kBlock, 0, ws->ctx->stream>>>(static_cast<cd*>(ws->col(c)), ws->ld, ws->hostscale[c], nullptr);
    }
  KS_HIP(hipGetLastError());
  reset_lazy(ws);
}

// MAILBOX (ks_workspace::mbox).  publish_control: the H columns of steps from.. and the DevState go to the pinned image
// `image_dev` by a one-workgroup kernel that then releases `seq` in flag slot `slot`; mbox_wait: the host spins on it.
// Replaces hipMemcpyAsync + hipStreamSynchronize around every expansion batch: the copy cost ~100 us of host enqueue
// time when issued in mid-stream and stalled the submission of what followed; the synchronisation woke the host
// 15-20 us after the fact (profiles/r02_restart_bubble.txt).
// end of the range a batch hands back: H columns + DevState, and with the implicit second pass also T
inline size_t control_end(const ks_workspace* ws, bool with_T) {
  return with_T ? ws->off_g : ws->hd_bytes + sizeof(DevState);
}
inline void publish_control(ks_workspace* ws, int from, void* image_dev, int slot, uint64_t seq, bool with_T = false) {
  const size_t off = from >= 1 ? (size_t)(from - 1) * (ws->maxdim + 1) * ws->esz : ws->hd_bytes;
  const int nwords = (int)((control_end(ws, with_T) - off) / 8);
  ksd::k_publish<<<1, kBlock, 0, ws->ctx->stream>>>(reinterpret_cast<const uint64_t*>(static_cast<const char*>(ws->Hd) + off),
                                                     reinterpret_cast<uint64_t*>(static_cast<char*>(image_dev) + off), nwords,
                                                     ws->mbox_dev + 8 * slot, seq);
  KS_HIP(hipGetLastError());
}
inline void mbox_wait(ks_workspace* ws, int slot, uint64_t seq) {
  const uint64_t* f = ws->mbox + 8 * slot;
  unsigned spins = 0;
  while (__atomic_load_n(f, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
    if ((++spins & 0xFFFu) == 0) {  // every ~50 us: is the stream still alive?
      const hipError_t q = hipStreamQuery(ws->ctx->stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == seq) break;
        throw KsError{KS_ERR_HIP, "the stream drained without publishing the control block"};
      }
      if (q != hipErrorNotReady) KS_HIP(q);
    }
  }
}

// Fused expansion steps from..to with LAZY NORMALISATION (see ks_kernels.hpp; Float64 and ComplexF64, to <= 64): per step
//   SpMV -> DOTS -> FIN_DOTS_DEF -> AXPY+DOTS -> FIN_MID_DEF -> AXPY
// (6 launches, 2 reductions, 3 passes over V, no v ./= wnorm pass) and one FIN_PEND at the end of the batch.  `op` may be null
// (ks_orthogonalize: the column is already there).
template <class D> void enqueue_steps_deferred(ks_workspace* ws, ks_operator* op, int from, int to, uint64_t early_seq = 0) {
  ks_ctx* cx = ws->ctx;
  hipStream_t s = cx->stream;
  const int ldh = ws->maxdim + 1;
  D* Hd = static_cast<D*>(ws->Hd);
  const D* V = static_cast<const D*>(ws->V);
  D* red = static_cast<D*>(ws->red);
  D* coef = static_cast<D*>(ws->coef);
  const D* part = static_cast<const D*>(ws->partial);
  double* redd = reinterpret_cast<double*>(ws->red);
  constexpr int dpe = (int)(sizeof(D) / 8);  // doubles per element (all-reduce counts)
  const double nb8 = (double)ws->n * sizeof(D);
  const bool dist = cx->distributed();
  // peer-to-peer: exchange folded into the reduction kernels (mode 3).  KS_P2P_NO_FOLD=1 keeps the three-launch
  // structure of the collective transports (reduce -> all-reduce -> post) on the peer-to-peer all-reduce kernel.
  static const int no_fold = env_int("KS_P2P_NO_FOLD", 0);
  const bool p2p = cx->p2p.attached && !no_fold;
  const ksd::P2pDev pd = cx->p2p.dev;
  // OUT-OF-PLACE first projection (KS_OOP, read at workspace creation).  The two kernels that update the new vector used
  // to read and write the SAME addresses (w' = w - V h in place).  The same binary lands in a "slow" or a "fast" mode from
  // process to process (k_axpy_dots_cs 5.35 vs 5.77 TB/s, 678 vs 702 iterations/s: the physical placement of the basis,
  // round 1's "placement lottery"), and the slow mode is slow only for in-place read-modify-write streams.
  //   KS_OOP=2 (default): the product y = A v goes into a scratch vector S0 instead of column j; the inner products and
  //              the first projection read S0, the projection WRITES column j; the second-pass update stays in place.
  //              One extra n-vector, no data-dependent behaviour.  Slow mode 678 -> 684-685, fast mode within 1 %.
  //   KS_OOP=1:  additionally, while the second DGKS pass is the rule (>= 90 % of the steps of the previous batch), the
  //              projection writes a second scratch vector S1 and the second-pass update reads S1 and writes column j (a
  //              step that then does NOT take the second pass moves S1 home).  Same speed as 2 on the headline; the
  //              SpMV loses the warm x the in-place update leaves in the memory-side cache (42 -> 53 us).
  //   KS_OOP=0:  everything in place (round 1).
  // Pure data movement: H, V and every decision are bit-identical in all forms (tested).  `op == nullptr`
  // (ks_orthogonalize: the vector already sits in column j) always runs in place.  profiles/r02_out_of_place_ab.txt.
  D* S0 = (op && ws->oop) ? static_cast<D*>(ws->oop) : nullptr;
  D* S1 = (S0 && ws->oop_full && ws->oop_mode == 1) ? S0 + ws->ld : nullptr;
  for (int j = from; j <= to; ++j) {
    D* w = static_cast<D*>(ws->col(j));
    D* y = S0 ? S0 : w;           // where the product lands and what the inner products / first projection read
    D* w1 = S1 ? S1 : w;          // where the first projection writes (and the second-pass update reads)
    D* Hcol = Hd + (size_t)(j - 1) * ldh;
    D* Hsub_prev = (j >= 2) ? Hd + (size_t)(j - 2) * ldh + (j - 1) : Hcol;  // only touched when a norm is pending
    if (op) {
      op->in_scale = op->async_capable ? 1.0 : ws->hostscale[j - 1];
      op->apply(ws->col(j - 1), y, ws->st);
    }
    int nbd;
    {
      ProfScope ps(cx, KSP_DOTS, nb8 * (j + 1));
      nbd = launch_dots<D>(ws, j, y, 1, ws->st);
    }
    {
      ProfScope ps(cx, KSP_FIN, 0.0);
      if (!dist || p2p) {
        ksd::k_fin_dots_def<D><<<j + 1, kBlock, 0, s>>>(part, nbd, ws->pnb, ws->partial2, ws->nb, j, red, Hcol, Hsub_prev, coef, ws->colscale, p2p ? 3 : 0, ws->st, pd);
      } else {
        ksd::k_fin_dots_def<D><<<j + 2, kBlock, 0, s>>>(part, nbd, ws->pnb, ws->partial2, ws->nb, j, red, Hcol, Hsub_prev, coef, ws->colscale, 1, ws->st, pd);
        cx->allreduce(redd, (j + 2) * dpe);
        ksd::k_fin_dots_def<D><<<j + 1, 64, 0, s>>>(part, nbd, ws->pnb, ws->partial2, ws->nb, j, red, Hcol, Hsub_prev, coef, ws->colscale, 2, ws->st, pd);
      }
    }
    int nbf;
    {
      ProfScope ps(cx, KSP_FUSED, nb8 * (j + 2));  // reads V[:,0:j) and y, writes w'
      nbf = launch_axpy_dots<D>(ws, j, y, 1, w1 == y ? nullptr : w1);
    }
    {
      ProfScope ps(cx, KSP_FIN, 0.0);
      if (!dist || p2p) {
        ksd::k_fin_mid_def<D><<<j + 1, kBlock, 0, s>>>(part, ws->partial2, nbf, ws->pnb, j, red, Hcol, coef, ws->colscale, p2p ? 3 : 0, ws->st, pd);
      } else {
        ksd::k_fin_mid_def<D><<<j + 1, kBlock, 0, s>>>(part, ws->partial2, nbf, ws->pnb, j, red, Hcol, coef, ws->colscale, 1, ws->st, pd);
        cx->allreduce(redd, (j + 1) * dpe);
        ksd::k_fin_mid_def<D><<<j + 1, 64, 0, s>>>(part, ws->partial2, nbf, ws->pnb, j, red, Hcol, coef, ws->colscale, 2, ws->st, pd);
      }
    }
    if (early_seq && j == to) {
      // H[0:to, from-1:to) is final here (the second-pass correction is in); what is still to come -- the second-pass
      // update of the vector and the reduction of H[to, to-1] -- does not touch it: hand it to the host now
      publish_control(ws, from, ws->Hstage_early_dev, 0, early_seq);
    }
    {
      ProfScope ps(cx, KSP_AXPY, nb8 * (j + 2));
      // packs per lane per iteration: at n = 1e7 going 2 -> 4 -> 8 gained 3 % + 8 % (16 lost 18 %); small
      // problems (<= 3072 packs per workgroup) are ~1 % better off with 4
      const int64_t ppb = (ws->ld * (int64_t)sizeof(D) / 16) / std::max(1, ws->nb);
      const D* src = w1 == w ? nullptr : w1;
      static const int plain_st = env_int("KS_OOP_PLAIN_STORE", 0);
      if (src && plain_st && ppb >= 3072) ksd::k_axpy<D, 8, true><<<ws->nb, kBlock, 0, s>>>(V, ws->ld, j, w, coef, ws->partial2, 2, ws->st, src);
      else if (ppb >= 3072) ksd::k_axpy<D, 8><<<ws->nb, kBlock, 0, s>>>(V, ws->ld, j, w, coef, ws->partial2, 2, ws->st, src);
      else ksd::k_axpy<D, 4><<<ws->nb, kBlock, 0, s>>>(V, ws->ld, j, w, coef, ws->partial2, 2, ws->st, src);
    }
    if (j == to) {  // settle the norm of the last column (it stays unnormalised in HBM: colscale)
      {
        ProfScope ps(cx, KSP_FIN, 0.0);
        if (!dist || p2p) {
          ksd::k_fin_pend<D><<<1, kBlock, 0, s>>>(ws->partial2, ws->nb, redd, Hcol + j, j, ws->colscale, p2p ? 3 : 0, ws->st, pd);
        } else {
          ksd::k_fin_pend<D><<<1, kBlock, 0, s>>>(ws->partial2, ws->nb, redd, Hcol + j, j, ws->colscale, 1, ws->st, pd);
          cx->allreduce(redd, 1);
          ksd::k_fin_pend<D><<<1, 64, 0, s>>>(ws->partial2, ws->nb, redd, Hcol + j, j, ws->colscale, 2, ws->st, pd);
        }
      }
    }
  }
  KS_HIP(hipGetLastError());
}

// Fused expansion steps from..to with the IMPLICIT SECOND PASS (ks_kernels.hpp): per step
//   SpMV -> DOTS -> FIN_STEP_T -> AXPY+DOTS            and one more FIN_STEP_T after the last step
// (4 launches, ONE reduction / exchange, TWO passes over the basis whether or not the DGKS test asks for the second
// projection).  FIN_STEP_T of step j settles the second reduction of step j-1 together with the first one of step j.
template <class D> void enqueue_steps_t(ks_workspace* ws, ks_operator* op, int from, int to) {
  ks_ctx* cx = ws->ctx;
  hipStream_t s = cx->stream;
  const int ldh = ws->maxdim + 1;
  D* Hd = static_cast<D*>(ws->Hd);
  D* Tm = static_cast<D*>(ws->Td);
  D* gv = static_cast<D*>(ws->gd);
  D* red = static_cast<D*>(ws->red);
  D* coef = static_cast<D*>(ws->coef);
  const D* part_c = static_cast<const D*>(ws->partial);    // written by the projection kernel (c_raw)
  D* part_s = static_cast<D*>(ws->partial_s);              // written by k_dots (s): must survive next to part_c
  double* redd = reinterpret_cast<double*>(ws->red);
  constexpr int dpe = (int)(sizeof(D) / 8);
  const double nb8 = (double)ws->n * sizeof(D);
  const bool dist = cx->distributed();
  static const int no_fold = env_int("KS_P2P_NO_FOLD", 0);
  const bool p2p = cx->p2p.attached && !no_fold;
  const ksd::P2pDev pd = cx->p2p.dev;
  const int nt = ws->ntrue;
  D* S0 = ws->oop ? static_cast<D*>(ws->oop) : nullptr;
  int nbf = 0;  // grid of the previous step's projection kernel (= number of its partial sums per column)
  auto fin = [&](int jm, int jd, int nbd) {
    ProfScope ps(cx, KSP_FIN, 0.0);
    const int nwg = (jm ? jm + 1 : 0) + (jd ? jd + 1 : 0);
    if (!dist || p2p) {
      ksd::k_fin_step_t<D><<<nwg, kBlock, 0, s>>>(part_s, nbd, part_c, ws->partial2, nbf, ws->pnb, jm, jd, red, Hd, ldh, Tm, ws->ldt, nt, gv, coef,
                                                  p2p ? 3 : 0, ws->st, pd, ws->ctr);
    } else {
      ksd::k_fin_step_t<D><<<nwg, kBlock, 0, s>>>(part_s, nbd, part_c, ws->partial2, nbf, ws->pnb, jm, jd, red, Hd, ldh, Tm, ws->ldt, nt, gv, coef, 1,
                                                  ws->st, pd, ws->ctr);
      cx->allreduce(redd, nwg * dpe);
      ksd::k_fin_step_t<D><<<1, kBlock, 0, s>>>(part_s, nbd, part_c, ws->partial2, nbf, ws->pnb, jm, jd, red, Hd, ldh, Tm, ws->ldt, nt, gv, coef, 2,
                                                ws->st, pd, ws->ctr);
    }
  };
  for (int j = from; j <= to; ++j) {
    D* w = static_cast<D*>(ws->col(j));
    D* y = S0 ? S0 : w;  // where the product lands; the projection reads it and writes column j
    // (host callbacks run one step per batch: the factor of the input column is on the host by now)
    op->in_scale = (!op->async_capable && ws->t_lazy && j - 1 >= ws->ntrue && j - 1 <= ws->t_hi)
                       ? reinterpret_cast<const double*>(static_cast<const char*>(ws->Th) + ((size_t)(j - 1) + (size_t)(j - 1) * ws->ldt) * ws->esz)[0]
                       : 1.0;
    op->apply(ws->col(j - 1), y, ws->st);
    int nbd;
    {
      ProfScope ps(cx, KSP_DOTS, nb8 * (j + 1));
      nbd = launch_dots<D>(ws, j, y, 1, ws->st, part_s);
    }
    fin(j > from ? j - 1 : 0, j, nbd);
    {
      ProfScope ps(cx, KSP_FUSED, nb8 * (j + 2));  // reads S[:,0:j) and y', writes w'
      nbf = launch_axpy_dots<D>(ws, j, y, 1, y == w ? nullptr : w);
    }
  }
  fin(to, 0, 0);  // settle the last step
  KS_HIP(hipGetLastError());
}

inline bool use_deferred(const ks_workspace* ws, int to) {
  static const int no_fuse = env_int("KS_NO_FUSE", 0), no_defer = env_int("KS_NO_DEFER", 0);
  return to <= kFusedMaxJ && !no_fuse && !no_defer;
}

// Provenance (ks_workspace::prov_k): see the field's comment.
inline size_t h_bytes(const ks_workspace* ws) { return (size_t)(ws->maxdim + 1) * ws->maxdim * ws->esz; }
inline void prov_set(ks_workspace* ws, int k) {
  ws->prov_k = k;
  if (k < 0) return;
  ws->Hshadow.resize(h_bytes(ws));
  std::memcpy(ws->Hshadow.data(), ws->H, h_bytes(ws));
}
inline void prov_drop(ks_workspace* ws) { ws->prov_k = -1; }
// may a batch that starts at step `from` lean on H[:, 0:from-1) and the relation of those steps?
inline bool prov_ok(const ks_workspace* ws, int from) {
  if (ws->prov_k < from - 1) return false;
  const size_t bytes = (size_t)(ws->maxdim + 1) * (size_t)(from - 1) * ws->esz;
  if (bytes == 0) return true;
  return ws->Hshadow.size() >= bytes && std::memcmp(ws->H, ws->Hshadow.data(), bytes) == 0;
}

// Start of a batch: fresh DevState and, when the host changed column factors since the last batch, the factors --
// one asynchronous copy from the pinned control block, no synchronisation.
inline void reset_state(ks_workspace* ws, bool upload_H = false, double sigma0 = 1.0) {
  // (no synchronisation: the state image is always the same bytes, and the factor image is only rewritten after a
  // host-side change, which follows the synchronising fetch of the previous batch)
  std::memset(ws->st_h, 0, sizeof(DevState));
  ws->st_h->breakdown = -1;
  ws->st_h->bail = -1;
  ws->st_h->max_ratio = ws->max_ratio;
  ws->st_h->sigma = sigma0;
  if (upload_H) {
    // implicit second pass: the device needs the CURRENT H (the restart rewrote its leading block on the host) for
    // g = H c -- the whole array travels with the state, still one copy
    std::memcpy(ws->Hstage, ws->H, (size_t)(ws->maxdim + 1) * ws->maxdim * ws->esz);
    KS_HIP(hipMemcpyAsync(ws->Hd, ws->Hstage, ws->hd_bytes + sizeof(DevState), hipMemcpyHostToDevice, ws->ctx->stream));
    return;  // (the column factors are not used by this path)
  }
  size_t bytes = sizeof(DevState);
  if (ws->colscale_dirty) {
    std::memcpy(ws->cs_h, ws->hostscale.data(), (size_t)(ws->maxdim + 2) * 8);
    bytes = kCtlStateSlot + (size_t)(ws->maxdim + 2) * 8;
    ws->colscale_dirty = false;
  }
  KS_HIP(hipMemcpyAsync(ws->st, ws->st_h, bytes, hipMemcpyHostToDevice, ws->ctx->stream));
}
// End of a batch: the H columns of steps from.. (to the end of Hd) and the DevState in ONE copy, one synchronisation.
inline void fetch_state_enqueue(ks_workspace* ws, int from = 0, bool with_T = false) {
  const size_t off = from >= 1 ? (size_t)(from - 1) * (ws->maxdim + 1) * ws->esz : ws->hd_bytes;
  KS_HIP(hipMemcpyAsync(static_cast<char*>(ws->Hstage) + off, static_cast<char*>(ws->Hd) + off, control_end(ws, with_T) - off,
                        hipMemcpyDeviceToHost, ws->ctx->stream));
}
inline void fetch_state_wait(ks_workspace* ws) {
  KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  if (ws->ctx->profiling) prof_collect(ws->ctx);
  ws->ctx->check_comm();
}
inline void fetch_state(ks_workspace* ws, int from = 0) {
  fetch_state_enqueue(ws, from);
  fetch_state_wait(ws);
}

// global 2-norm of column j (synchronous)
template <class D> double col_norm(ks_workspace* ws, int j) {
  ks_ctx* c = ws->ctx;
  ksd::k_norm2<D><<<ws->nb, kBlock, 0, c->stream>>>(static_cast<const D*>(ws->col(j)), ws->ld, ws->partial2);
  ksd::k_sum<<<1, kBlock, 0, c->stream>>>(ws->partial2, ws->nb, ws->scal);
  c->allreduce(ws->scal, 1);
  KS_HIP(hipMemcpyAsync(ws->scal_h, ws->scal, 8, hipMemcpyDeviceToHost, c->stream));
  KS_HIP(hipStreamSynchronize(c->stream));
  c->check_comm();
  return std::sqrt(ws->scal_h[0]);
}

template <class D> void col_scale(ks_workspace* ws, int j, double factor) {
  ksd::k_scale<D><<<ws->nb, kBlock, 0, ws->ctx->stream>>>(static_cast<D*>(ws->col(j)), ws->ld, factor, nullptr);
  KS_HIP(hipGetLastError());
}

// h = V[:,0:j)^H V[:,jv]  -> coef (device) and, if h_host, the host copy (synchronous)
template <class D> void gemv_t(ks_workspace* ws, int j, int jv, void* h_host) {
  ks_ctx* c = ws->ctx;
  const int nbd = launch_dots<D>(ws, j, static_cast<const D*>(ws->col(jv)), 1, nullptr);
  // post into the scratch column so H is untouched; st must be valid for k_fin_dots -> use ws->st with
  // breakdown cleared (verbs run outside batches)
  launch_fin_dots<D>(ws, nbd, j, static_cast<D*>(ws->Hscratch), 1, ws->st);
  if (h_host) {
    KS_HIP(hipMemcpyAsync(ws->coef_h, ws->coef, (size_t)j * sizeof(D), hipMemcpyDeviceToHost, c->stream));
    KS_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(h_host, ws->coef_h, (size_t)j * sizeof(D));
  }
}

// V[:,jv] -= V[:,0:j) h ; returns nothing (partial2 holds block-local |v|^2)
template <class D> void gemv_n_sub(ks_workspace* ws, int j, int jv, const void* h_host) {
  ks_ctx* c = ws->ctx;
  if (h_host) {
    std::memcpy(ws->coef_h, h_host, (size_t)j * sizeof(D));
    KS_HIP(hipMemcpyAsync(ws->coef, ws->coef_h, (size_t)j * sizeof(D), hipMemcpyHostToDevice, c->stream));
  }
  ksd::k_axpy<D><<<ws->nb, kBlock, 0, c->stream>>>(static_cast<const D*>(ws->V), ws->ld, j, static_cast<D*>(ws->col(jv)),
                                                    static_cast<const D*>(ws->coef), ws->partial2, 1, nullptr);
  KS_HIP(hipGetLastError());
  if (h_host) KS_HIP(hipStreamSynchronize(c->stream));  // coef_h is reused by the next verb
}

template <class D> double norm_from_partial2(ks_workspace* ws) {
  ks_ctx* c = ws->ctx;
  ksd::k_sum<<<1, kBlock, 0, c->stream>>>(ws->partial2, ws->nb, ws->scal);
  c->allreduce(ws->scal, 1);
  KS_HIP(hipMemcpyAsync(ws->scal_h, ws->scal, 8, hipMemcpyDeviceToHost, c->stream));
  KS_HIP(hipStreamSynchronize(c->stream));
  c->check_comm();
  return std::sqrt(ws->scal_h[0]);
}

// copyto!(view(V,:,j), host) incl. zeroing the pad rows
template <class D> void col_upload(ks_workspace* ws, int j, const void* host) {
  ks_ctx* c = ws->ctx;
  KS_HIP(hipMemcpyAsync(ws->col(j), host, (size_t)ws->n * sizeof(D), hipMemcpyHostToDevice, c->stream));
  if (ws->ld > ws->n)
    KS_HIP(hipMemsetAsync(static_cast<char*>(ws->col(j)) + (size_t)ws->n * sizeof(D), 0,
                          (size_t)(ws->ld - ws->n) * sizeof(D), c->stream));
  KS_HIP(hipStreamSynchronize(c->stream));
}

// reinitialize!(arnoldi, j, populate!)  src/expansion.jl:12-59 (synchronous; rare)
template <class D> bool reinit_column(ks_workspace* ws, int j, const void* v1_host) {
  ks_ctx* c = ws->ctx;
  materialize(ws);
  D* v = static_cast<D*>(ws->col(j));
  if (v1_host) {
    col_upload<D>(ws, j, v1_host);
  } else {
    const int gb = (int)std::min<int64_t>((ws->ld + kBlock - 1) / kBlock, 8192);
    ksd::k_fill_uniform<D><<<gb, kBlock, 0, c->stream>>>(v, ws->n, ws->ld, next_seed(ws), (uint64_t)ws->row_begin);
  }
  double rnorm = col_norm<D>(ws, j);                      // :24
  if (j == 0) {                                           // :27-30
    col_scale<D>(ws, j, 1.0 / rnorm);
    return true;
  }
  // st->breakdown must read -1 for the fin kernels
  reset_state(ws);
  gemv_t<D>(ws, j, j, nullptr);                           // :37
  gemv_n_sub<D>(ws, j, j, nullptr);                       // :38
  double wnorm = norm_from_partial2<D>(ws);               // :41
  if (wnorm < ksd::kEta * rnorm) {                        // :44
    rnorm = wnorm;
    gemv_t<D>(ws, j, j, nullptr);
    gemv_n_sub<D>(ws, j, j, nullptr);
    wnorm = norm_from_partial2<D>(ws);
  }
  if (wnorm <= ksd::kEta * rnorm) return false;           // :51
  col_scale<D>(ws, j, 1.0 / wnorm);                       // :56
  return true;
}

// columns built by a lazily-normalised batch are stored as beta * v with beta = H[j, j-1] (read from the staged image:
// the host H may already have been transformed by the early part of the restart step)
template <class T> void lazy_factors_from_stage(ks_workspace* ws, int from, int to, const void* stage) {
  const int ldh = ws->maxdim + 1;
  const T* hs = static_cast<const T*>(stage);
  for (int j = from; j <= to; ++j) {
    const double beta = ks::real_(hs[(size_t)(j - 1) * ldh + j]);
    if (beta != 0.0) {
      ws->hostscale[j] = 1.0 / beta;
      ws->lazy_lo = std::min(ws->lazy_lo, j);
      ws->lazy_hi = std::max(ws->lazy_hi, j);
    }
  }
}

// copy the H columns produced on the device for steps from..to (already staged by fetch_state(ws, from), or by the early
// copy into `stage`) into the host H
template <class T> void fetch_H_columns(ks_workspace* ws, int from, int to, const ks::Mat<T>& H, bool lazy = false, const void* stage = nullptr) {
  if (to < from) return;
  const int ldh = ws->maxdim + 1;
  if (!stage) stage = ws->Hstage;  // filled by fetch_state(ws, from) together with the DevState
  const T* hs = static_cast<const T*>(stage);
  for (int j = from; j <= to; ++j)
    for (int i = 0; i <= j; ++i) H(i, j - 1) = hs[(size_t)(j - 1) * ldh + i];
  if (lazy) lazy_factors_from_stage<T>(ws, from, to, stage);
}

// out[:, 0:r) = V[:, 0:c) * Y[0:c, 0:r)  (device Y, column-major ldy) for any c, r: the coefficient block
// a launch keeps in LDS is limited to 48 KiB, wider products are split over output-column chunks.
template <class TV, class TY>
void gemm_tall_chunked(ks_workspace* ws, const TV* V, int c, int r, const TY* Yd, int ldy, TY* out, int64_t ldo) {
  ks_ctx* ctx = ws->ctx;
  const int rc_max = std::max<int>(1, (int)((48 * 1024) / ((size_t)c * sizeof(TY))));
  KS_REQUIRE((size_t)c * sizeof(TY) <= 48 * 1024, KS_ERR_ARGUMENT, "too many columns for the generic tall-skinny kernel");
  for (int r0 = 0; r0 < r; r0 += rc_max) {
    const int rc = std::min(rc_max, r - r0);
    const size_t smem = (size_t)c * rc * sizeof(TY);
    ksd::k_gemm_tall<TV, TY><<<ctx->num_cu * 4, kBlock, smem, ctx->stream>>>(V, ws->ld, ws->n, c, rc, Yd + (size_t)r0 * ldy, ldy,
                                                                          out + (size_t)r0 * ldo, ldo);
  }
  KS_HIP(hipGetLastError());
}

// V[:, c0+out0 : c0+out0+r) <- V[:, c0:c0+c) Q  with Q already on the device (column-major, ld = c); in place.  out0 = 0 is
// the plain rotation; extra_out >= 0 sends the LAST output to column c0 + extra_out instead (T-folded restart: the
// residual direction lands next to the truncated basis).
template <class D> void rotate_device(ks_workspace* ws, int c0, int c, int r, int out0 = 0, int extra_out = -1) {
  ks_ctx* ctx = ws->ctx;
  hipStream_t s = ctx->stream;
  D* Vc = static_cast<D*>(ws->col(c0));
  const D* Qd = static_cast<const D*>(ws->Qd);
  ProfScope ps(ctx, KSP_ROTATE, (double)ws->n * sizeof(D) * (c + r));  // in place: read c, write r columns
  const bool force_valu = env_int("KS_ROTATE_VALU", 0) != 0;
  if constexpr (sizeof(D) == 8) {
    // KS_ROTATE = fma (default: vector-ALU kernel -- the FP64 matrix cores of gfx950 run at half the vector rate) |
    // mfma (v_mfma_f64_16x16x4_f64 tiles)
    const char* rot_env = std::getenv("KS_ROTATE");
    const std::string rot = rot_env ? rot_env : "fma";
    if (!force_valu && rot == "fma" && c <= 64) {
      static const int bpc_env = env_int("KS_ROTATE_FMA_BPC", 2);
      auto go = [&](auto ct_tag) {
        constexpr int CT = decltype(ct_tag)::value;
        const size_t smem = (size_t)r * CT * 8;
        static int occ = -1;
        if (occ < 0) {
          KS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ksd::k_rotate_fma<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
          int o = 0;
          KS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, ksd::k_rotate_fma<CT>, kBlock, (size_t)48 * CT * 8));
          occ = std::max(1, std::min(o, bpc_env));
        }
        const int nbr = cap_blocks(ws, ctx->num_cu * occ, kBlock);
        ksd::k_rotate_fma<CT><<<nbr, kBlock, smem, s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out);
      };
      KS_REQUIRE((size_t)r * 64 * 8 <= (size_t)64 * 1024, KS_ERR_INTERNAL, "rotation wider than the coefficient tile");
      if (c <= 24) go(std::integral_constant<int, 24>{});
      else if (c <= 32) go(std::integral_constant<int, 32>{});
      else if (c <= 44) go(std::integral_constant<int, 44>{});
      else go(std::integral_constant<int, 64>{});
      KS_HIP(hipGetLastError());
      return;
    }
    if (!force_valu && c <= 64) {
      const int ntile = (r + 15) / 16;
      auto smem = [&](int KC) { return (size_t)ntile * 16 * (4 * KC + 1) * 8; };
      static const int rt = env_int("KS_ROTATE_RT", 2);
      static const int nbm = env_int("KS_ROTATE_BPC", 4);
      const int nbr = ctx->num_cu * nbm;
      if (c <= 24) { if (rt == 2) ksd::k_rotate_mfma<6, 2><<<nbr, kBlock, smem(6), s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out); else ksd::k_rotate_mfma<6, 1><<<nbr, kBlock, smem(6), s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out); }
      else if (c <= 40) { if (rt == 2) ksd::k_rotate_mfma<10, 2><<<nbr, kBlock, smem(10), s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out); else ksd::k_rotate_mfma<10, 1><<<nbr, kBlock, smem(10), s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out); }
      else if (c <= 44) { if (rt == 2) ksd::k_rotate_mfma<11, 2><<<nbr, kBlock, smem(11), s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out); else ksd::k_rotate_mfma<11, 1><<<nbr, kBlock, smem(11), s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out); }
      else ksd::k_rotate_mfma<16, 1><<<nbr, kBlock, smem(16), s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out);
      KS_HIP(hipGetLastError());
      return;
    }
  }
  const size_t smem = (size_t)c * r * sizeof(D);
  const int nb = ctx->num_cu * 2;
  D* Vo = Vc + (size_t)out0 * ws->ld;
  const int xo = extra_out >= 0 ? extra_out - out0 : -1;  // relative to the output base
  if (c <= 8) ksd::k_rotate_valu<D, 8><<<nb, kBlock, smem, s>>>(Vc, ws->ld, c, r, Qd, c, Vo, ws->ld, xo);
  else if (c <= 16) ksd::k_rotate_valu<D, 16><<<nb, kBlock, smem, s>>>(Vc, ws->ld, c, r, Qd, c, Vo, ws->ld, xo);
  else if (c <= 24) ksd::k_rotate_valu<D, 24><<<nb, kBlock, smem, s>>>(Vc, ws->ld, c, r, Qd, c, Vo, ws->ld, xo);
  else if (c <= 40) ksd::k_rotate_valu<D, 40><<<nb, kBlock, smem, s>>>(Vc, ws->ld, c, r, Qd, c, Vo, ws->ld, xo);
  else {
    // out of place through scratch, then copy back
    D* tmp = static_cast<D*>(ws->ensure_tmp((size_t)ws->ld * r * sizeof(D)));
    KS_HIP(hipMemsetAsync(tmp, 0, (size_t)ws->ld * r * sizeof(D), s));  // keeps the pad rows zero
    gemm_tall_chunked<D, D>(ws, Vc, c, r, Qd, c, tmp, ws->ld);
    const int rmain = extra_out >= 0 ? r - 1 : r;
    if (rmain > 0) KS_HIP(hipMemcpyAsync(Vo, tmp, (size_t)ws->ld * rmain * sizeof(D), hipMemcpyDeviceToDevice, s));
    if (extra_out >= 0)
      KS_HIP(hipMemcpyAsync(Vc + (size_t)extra_out * ws->ld, tmp + (size_t)(r - 1) * ws->ld, (size_t)ws->ld * sizeof(D), hipMemcpyDeviceToDevice, s));
  }
  KS_HIP(hipGetLastError());
}

// T as the host sees it after a batch (pinned image); columns outside ntrue..t_hi are unit vectors
template <class T> inline T t_entry(const ks_workspace* ws, int k, int i) {
  if (!ws->t_lazy || i < ws->ntrue || i > ws->t_hi) return k == i ? T(1) : T(0);
  if (k > i) return T(0);
  return static_cast<const T*>(ws->Th)[k + (size_t)i * ws->ldt];
}

// Implicit second pass: V_true[:, a] = S[:, 0:a+1) T[0:a+1, a].  General T-folded product, in place:
//   V[:, out0 : out0+r) <- V_true[:, c0 : c0+c) Q[0:c, 0:r)      (Q host, column-major ldq; may be null with r == 0)
//   V[:, dst]           <- V_true[:, src]                         (src < 0: none)
// computed as S[:, 0:cin) (T Q) with cin = max(c0 + c, src + 1).  Afterwards NO column is T-lazy: every T-lazy column that
// is not among the outputs is dead (the caller guarantees it: restart, or materialisation of all of them).
template <class T> void rotate_tfold(ks_workspace* ws, int c0, int c, int r, const T* Qh, int ldq, int out0, int src, int dst) {
  using D = typename DevT<T>::type;
  ws->ctx->use();
  KS_HIP(hipStreamSynchronize(ws->ctx->stream));  // Qstage may still be in flight from a previous rotation
  const int rr = r + (src >= 0 ? 1 : 0);
  const int cin = std::max(c0 + c, src + 1);
  KS_REQUIRE(cin <= ws->maxdim + 1 && rr <= ws->maxdim + 1, KS_ERR_INTERNAL, "T-folded rotation out of range");
  T* qs = static_cast<T*>(ws->Qstage);  // cin x rr, ld = cin
  // T Q, column by column of T (upper triangular; unit vectors outside the T-lazy range): qs[0:i+1, jj] += T[0:i+1, i] q_i --
  // contiguous in k, so the inner loop vectorises (a per-entry accessor cost 30-50 us per restart, a third of the Schur
  // step it follows)
  const bool tl = ws->t_lazy;
  const int tlo = ws->ntrue, thi = ws->t_hi;
  const T* Th = static_cast<const T*>(ws->Th);
  const size_t ldt = (size_t)ws->ldt;
  std::fill(qs, qs + (size_t)cin * rr, T(0));
  for (int jj = 0; jj < r; ++jj) {
    T* __restrict__ out = qs + (size_t)jj * cin;
    for (int i = c0; i < c0 + c; ++i) {
      const T q = Qh[(i - c0) + (size_t)jj * ldq];
      if (tl && i >= tlo && i <= thi) {
        const T* __restrict__ tc = Th + (size_t)i * ldt;
        for (int k = 0; k <= i; ++k) out[k] += tc[k] * q;
      } else {
        out[i] += q;
      }
    }
  }
  if (src >= 0) {
    T* __restrict__ out = qs + (size_t)r * cin;
    if (tl && src >= tlo && src <= thi) {
      const T* __restrict__ tc = Th + (size_t)src * ldt;
      for (int k = 0; k <= src; ++k) out[k] = tc[k];
    } else {
      out[src] = T(1);
    }
  }
  KS_HIP(hipMemcpyAsync(ws->Qd, qs, (size_t)cin * rr * sizeof(T), hipMemcpyHostToDevice, ws->ctx->stream));
  const bool extra_elsewhere = src >= 0 && dst != out0 + r;
  rotate_device<D>(ws, 0, cin, rr, out0, extra_elsewhere ? dst : -1);
  ws->t_lazy = false;
  ws->t_hi = -1;
}

// all T-lazy columns -> ordinary columns, in place (verbs outside the expansion / restart pair are about to read V)
template <class D> void materialize_t(ks_workspace* ws) {
  using T = typename HostT<D>::type;
  if (!ws->t_lazy) return;
  const int lo = ws->ntrue, hi = ws->t_hi;
  if (hi < lo) { ws->t_lazy = false; return; }
  const int c = hi - lo + 1;
  std::vector<T> I((size_t)c * c, T(0));
  for (int i = 0; i < c; ++i) I[i + (size_t)i * c] = T(1);
  rotate_tfold<T>(ws, lo, c, c, I.data(), c, lo, -1, -1);
}

// V[:, c0:c0+r) <- V[:, c0:c0+c) * Q with Q on the HOST (column-major, leading dimension ldq), aware of lazily
// normalised columns: a lazy column is stored as beta * v, so  V Q = (stored) diag(1/beta) Q  -- the factors are
// folded into the rows of Q instead of touching n-sized data, and the rotated columns come out ordinary.  Lazy columns
// BELOW the rotated range are not absorbed by this rotation and are made ordinary first.
template <class T> void rotate_lazy(ks_workspace* ws, int c0, int c, int r, const T* Qh, int ldq, bool update_device_factors = true) {
  using D = typename DevT<T>::type;
  if (c <= 0 || r <= 0) return;
  ws->ctx->use();
  if (ws->t_lazy) materialize(ws);  // (T-lazy columns outside the rotated range would lose the columns they refer to)
  KS_HIP(hipStreamSynchronize(ws->ctx->stream));  // Qstage may still be in flight from a previous rotation
  if (ws->has_lazy() && ws->lazy_lo < c0) materialize(ws);
  T* qs = static_cast<T*>(ws->Qstage);
  for (int jj = 0; jj < r; ++jj)
    for (int ii = 0; ii < c; ++ii) qs[ii + (size_t)jj * c] = Qh[ii + (size_t)jj * ldq] * ws->hostscale[c0 + ii];
  KS_HIP(hipMemcpyAsync(ws->Qd, qs, (size_t)c * r * sizeof(T), hipMemcpyHostToDevice, ws->ctx->stream));
  rotate_device<D>(ws, c0, c, r);
  // the rotated columns are ordinary again; the factors of columns c0+r .. c0+c-1 (inputs only) stay as they are
  bool any = false;
  for (int ii = 0; ii < r; ++ii) {
    any = any || ws->hostscale[c0 + ii] != 1.0;
    ws->hostscale[c0 + ii] = 1.0;
  }
  if (any) ws->colscale_dirty = true;
  (void)update_device_factors;
}

// V[:, dst] <- V[:, src] (src/run.jl:365), lazy-aware: the factor of a lazy source is applied on the way, the
// destination comes out ordinary, every other column keeps its state.
template <class D> void col_copy_lazy(ks_workspace* ws, int dst, int src) {
  ws->ctx->use();
  if (ws->t_lazy) materialize(ws);
  const double f = ws->hostscale[src];
  if (dst == src) {
    if (f != 1.0) ksd::k_scale<D><<<ws->nb, kBlock, 0, ws->ctx->stream>>>(static_cast<D*>(ws->col(src)), ws->ld, f, nullptr);
  } else {
    ksd::k_copy<D><<<ws->nb, kBlock, 0, ws->ctx->stream>>>(static_cast<const D*>(ws->col(src)), static_cast<D*>(ws->col(dst)), ws->ld, f);
  }
  KS_HIP(hipGetLastError());
  if (ws->hostscale[dst] != 1.0 || (dst == src && f != 1.0)) {
    ws->hostscale[dst] = 1.0;
    ws->colscale_dirty = true;
  }
}

// ------------------------------------------------------------------------------------------------
// the HIP backend of the driver
// ------------------------------------------------------------------------------------------------
template <class T> struct HipBackend : ks::Backend<T> {
  using D = typename DevT<T>::type;
  ks_operator* op;
  ks_workspace* ws;
  HipBackend(ks_operator* o, ks_workspace* w) : op(o), ws(w) {}

  int64_t n_global() const override { return ws->n_global; }

  void iterate_arnoldi(int from, int to, const ks::Mat<T>& H, ks::ExpandStats& stats) override {
    guarded_expand([&] { iterate_arnoldi_impl(from, to, H, stats, nullptr); });
  }

  // The expansion a restart follows: H[0:to, :] is handed to `early` (restart_host_early: Schur form, Ritz values,
  // unit residuals, ordering) as soon as the last step's k_fin_mid_def ran, while the device still runs that step's
  // second-pass update and the reduction of H[to, to-1] -- at n = 1e7 the update alone (0.4 ms) outlasts the whole early
  // part (0.13 ms), at n = 1e6 about 30 us of it are hidden (profiles/r02_restart_bubble.txt).
  // KS_EARLY_RESTART=0 keeps the strictly sequential order.  Same operations on the same numbers either way.
  // Explicit-second-pass path (KS_PASSES=3) only: with the implicit second pass (default) H is final only when the batch
  // ends -- there is no tail to hide behind -- and this returns false (the caller then runs the whole host step).
  bool iterate_arnoldi_early(int from, int to, const ks::Mat<T>& H, ks::ExpandStats& stats, const std::function<void()>& early) override {
    const char* e = std::getenv("KS_EARLY_RESTART");
    const bool on = !(e && e[0] == '0');
    bool done = false;
    guarded_expand([&] { done = iterate_arnoldi_impl(from, to, H, stats, on ? &early : nullptr); });
    return done;
  }

  template <class F> void guarded_expand(F&& f) {
    try {
      f();
    } catch (...) {
      // an operator callback (or a HIP / transport error) aborted a batch midway: steps were enqueued whose H columns
      // and lazy-normalisation factors were never fetched.  Drain the stream and return the bookkeeping to "every
      // column is ordinary"; the factorisation itself is undefined from here on -- the caller must re-initialise
      // (ks_reinitialize(ws, 0, ...) or ks_partialschur with initialize = 1) before using the workspace again.
      (void)hipStreamSynchronize(ws->ctx->stream);
      try { reset_lazy(ws); } catch (...) {}
      prov_drop(ws);
      throw;
    }
  }

  // returns true iff *early ran and its effects on H stand
  bool iterate_arnoldi_impl(int from, int to, const ks::Mat<T>& H, ks::ExpandStats& stats, const std::function<void()>* early) {
    ws->ctx->use();
    bool early_stands = false;
    int j0 = from;
    int explicit_step = -1;  // a step the implicit form handed back (DevState::bail): redone with the explicit second pass
    // Provenance: the implicit second pass reads earlier columns of the caller's H and assumes the Arnoldi relation for
    // them.  Only a factorisation the library produced itself (or the caller vouched for) qualifies; anything else runs
    // the explicit form, which -- like the reference's iterate_arnoldi! -- reads neither.
    const bool trusted = prov_ok(ws, from);
    while (j0 <= to) {
      const double tb0 = ks::now_s();
      int jend = to;
      if (!op->async_capable) jend = j0;  // host operators: one step per batch
      if (explicit_step >= 0) jend = j0;
      const bool lazy = use_deferred(ws, jend);
      const bool tpath = lazy && ws->passes == 2 && trusted && explicit_step < 0;  // implicit second pass: two reads of the basis per step
      if (tpath && !(ws->t_lazy && j0 == ws->t_hi + 1)) {
        materialize(ws);   // whatever is lazy (either kind) becomes ordinary: this batch starts a new T
        ws->ntrue = j0;
      }
      double sigma0 = 1.0;
      if (tpath && ws->t_lazy && j0 - 1 >= ws->ntrue && j0 - 1 <= ws->t_hi) {
        // the batch continues on a factored column (host callbacks: every batch): k_dots' scale factor from its 1 / beta
        const double binv = reinterpret_cast<const double*>(static_cast<const char*>(ws->Th) + ((size_t)(j0 - 1) + (size_t)(j0 - 1) * ws->ldt) * ws->esz)[0];
        if (binv > 0.0 && std::isfinite(binv)) sigma0 = std::ldexp(1.0, std::ilogb(binv));
      }
      reset_state(ws, tpath, sigma0);
      const bool mb = lazy && ws->use_mbox;          // the device publishes the results itself, the host spins
      const bool do_early = early && mb && !tpath && jend == to;  // (with two passes H is final only at the very end)
      const uint64_t seq = ++ws->mbox_seq;
      if (tpath) {
        enqueue_steps_t<D>(ws, op, j0, jend);
      } else if (lazy) {
        if (ws->t_lazy) materialize(ws);
        enqueue_steps_deferred<D>(ws, op, j0, jend, do_early ? seq : 0);
      } else {
        materialize(ws);  // the eager kernels expect ordinary columns
        for (int j = j0; j <= jend; ++j) {
          op->in_scale = 1.0;
          op->apply(ws->col(j - 1), ws->col(j), ws->st);
          enqueue_orthogonalize<D>(ws, j);
        }
      }
      bool early_ran = false;
      // if anything throws after the early part of the restart step ran (transport time-out, operator error), the host H
      // is put back as it was: a caller that catches the error must not find a half-restarted matrix
      struct EarlyGuard {
        ks_workspace* w; void* Hp; bool armed;
        ~EarlyGuard() { if (armed && !w->Hbackup.empty()) std::memcpy(Hp, w->Hbackup.data(), w->Hbackup.size()); }
      } early_guard{ws, H.p, false};
      static const int dbg = env_int("KS_EARLY_DEBUG", 0);
      double tq0 = dbg ? ks::now_s() : 0.0, tq1 = 0, tq2 = 0;
      if (mb) publish_control(ws, j0, ws->Hstage_dev, 1, seq, tpath);
      else fetch_state_enqueue(ws, j0, tpath);
      if (do_early) {
        mbox_wait(ws, 0, seq);
        if (dbg) tq1 = ks::now_s();
        const DevState* se = reinterpret_cast<const DevState*>(static_cast<const char*>(ws->Hstage_early) + ws->hd_bytes);
        if (se->breakdown < 0) {  // (a breakdown of the LAST step is only known after the final reduction: see below)
          const size_t hb = (size_t)H.ld * H.n * sizeof(T);
          ws->Hbackup.resize(hb);
          std::memcpy(ws->Hbackup.data(), H.p, hb);
          early_guard.armed = true;
          fetch_H_columns<T>(ws, j0, jend, H, false, ws->Hstage_early);  // H[jend, jend-1] is not final yet, nobody reads it
          (*early)();
          early_ran = true;
        }
        if (dbg) tq2 = ks::now_s();
      }
      if (mb) {
        mbox_wait(ws, 1, seq);
        if (ws->ctx->profiling) {
          KS_HIP(hipStreamSynchronize(ws->ctx->stream));
          prof_collect(ws->ctx);
        }
        ws->ctx->check_comm();
      } else {
        fetch_state_wait(ws);
      }
      if (dbg && do_early) {
        const double tq3 = ks::now_s();
        std::fprintf(stderr, "[early] enqueue %.1f us | wait H %.1f us | early host part %.1f us | final wait %.1f us | batch %.1f us\n", 1e6 * (tq0 - tb0), 1e6 * (tq1 - tq0), 1e6 * (tq2 - tq1), 1e6 * (tq3 - tq2), 1e6 * (tq3 - tb0));
      } else if (dbg) {
        const double tq3 = ks::now_s();
        std::fprintf(stderr, "[early off] enqueue %.1f us | wait %.1f us | batch %.1f us\n", 1e6 * (tq0 - tb0), 1e6 * (tq3 - tq0), 1e6 * (tq3 - tb0));
      }
      // bail: the implicit form refuses step `bail` (its second-pass correction is not small) -- steps before it stand,
      // the step itself is redone below in the explicit form; not a breakdown
      const int bail = tpath ? ws->st_h->bail : -1;
      const int bd = bail >= 0 ? -1 : ws->st_h->breakdown;
      const int last_done = bail >= 0 ? bail - 1 : (bd >= 0 ? bd : jend);
      if (early_ran && bd >= 0) {  // the last step broke down: withdraw (rare; the caller redoes the early part)
        std::memcpy(H.p, ws->Hbackup.data(), ws->Hbackup.size());
        early_ran = false;
      }
      early_guard.armed = false;
      if (early_ran) {
        const T* hs = static_cast<const T*>(ws->Hstage);
        H(jend, jend - 1) = hs[(size_t)(jend - 1) * (ws->maxdim + 1) + jend];
        lazy_factors_from_stage<T>(ws, j0, jend, ws->Hstage);
        early_stands = true;
      } else {
        fetch_H_columns<T>(ws, j0, last_done, H, lazy && !tpath);
      }
      if (tpath) {
        // columns j0 .. (last completed step) are T-lazy now; a column that broke down is garbage until reinit_column
        const int good = bail >= 0 ? bail - 1 : (bd >= 0 ? bd - 1 : jend);
        if (good >= ws->ntrue) {
          ws->t_lazy = true;
          ws->t_hi = good;
        }
      }
      stats.steps += last_done - j0 + 1;
      stats.reorth += ws->st_h->n_reorth;
      if (explicit_step >= 0) explicit_step = -1;  // the handed-back step is done
      if (bail >= 0) {
        explicit_step = bail;
        stats.explicit_steps++;
      }
      if (lazy && jend > j0) ws->oop_full = 10 * ws->st_h->n_reorth >= 9 * (last_done - j0 + 1);  // (batches of one step keep the setting)
      if (bd >= 0) {
        // orthogonalize! returned false at step bd: H[bd, bd-1] = 0 is already in place
        // (src/expansion.jl:99-102); draw a fresh vector unless bd == n (src/expansion.jl:127-129)
        if ((int64_t)bd != ws->n_global) {
          reinit_column<D>(ws, bd, nullptr);
          stats.breakdowns++;
        }
      }
      j0 = last_done + 1;
    }
    // the factorisation up to `to` is the library's own again (or stays unknown)
    if (trusted) prov_set(ws, to);
    else prov_drop(ws);
    return early_stands;
  }

  bool reinitialize(int j, const T* v1_host) override {
    ws->ctx->use();
    return reinit_column<D>(ws, j, v1_host);
  }

  void rotate(int c0, int c, int r, const ks::Mat<T>& Q) override {
    if (c <= 0 || r <= 0) return;
    rotate_lazy<T>(ws, c0, c, r, &Q(c0, c0), Q.ld, /*update_device_factors=*/false);  // col_copy resets them right after
  }

  // V[:, dst] <- V[:, src]: the last act of a restart (src = maxdim holds the residual direction).  Every
  // column that is still lazy afterwards is dead (it lies beyond the truncated basis) -> reset the factors.
  void col_copy(int dst, int src) override {
    if (ws->t_lazy) materialize(ws);
    const double copy_factor = ws->hostscale[src];
    if (dst != src || copy_factor != 1.0) {
      if (dst == src)
        ksd::k_scale<D><<<ws->nb, kBlock, 0, ws->ctx->stream>>>(static_cast<D*>(ws->col(src)), ws->ld, copy_factor, nullptr);
      else
        ksd::k_copy<D><<<ws->nb, kBlock, 0, ws->ctx->stream>>>(static_cast<const D*>(ws->col(src)), static_cast<D*>(ws->col(dst)), ws->ld, copy_factor);
      KS_HIP(hipGetLastError());
    }
    reset_lazy(ws);
  }

  // src/run.jl:363-365 as ONE operation.  With the implicit second pass the basis is S T: the rotation and the move of
  // the residual direction are one T-folded product S[:, 0:src+1) [ T Q | T[:, src] ] (the columns it does not write are
  // the ones the restart discards).
  void rotate_and_move(int c0, int c, int r, const ks::Mat<T>& Q, int dst, int src) override {
    const bool own = ws->prov_k == src;  // the library's own factorisation of `src` steps is being truncated to `dst`
    if (ws->t_lazy) {
      rotate_tfold<T>(ws, c0, c, r, r > 0 ? &Q(c0, c0) : nullptr, Q.ld, c0, src, dst);
      reset_lazy(ws);
    } else {
      rotate(c0, c, r, Q);
      col_copy(dst, src);
    }
    if (own) prov_set(ws, dst);  // (the host step rewrote H: new shadow)
    else prov_drop(ws);
  }
};

template <class F> void dispatch_dtype(int dtype, F&& f) {
  if (dtype == KS_F64) f(double{});
  else if (dtype == KS_C64) f(cplx{});
  else throw KsError{KS_ERR_ARGUMENT, "unknown dtype"};
}

void check_col(const ks_workspace* ws, int j) {
  KS_REQUIRE(ws != nullptr, KS_ERR_ARGUMENT, "null workspace");
  KS_REQUIRE(j >= 0 && j <= ws->maxdim, KS_ERR_ARGUMENT, "column index out of range");
}

}  // namespace

namespace {

// Frobenius norm of (A X - Y C) where X = V[:, 0:nx), Y = V[:, 0:ny), C (ny x nx) host; and of (Y^H Y - I).
template <class T>
void relation_norms(ks_operator* A, ks_workspace* ws, int nx, int ny, const T* C, int ldc, double* resid, double* orth) {
  using D = typename DevT<T>::type;
  ks_ctx* c = ws->ctx;
  hipStream_t s = c->stream;
  D* y = static_cast<D*>(ws->ensure_tmp((size_t)ws->ld * sizeof(D)));
  KS_HIP(hipMemsetAsync(y, 0, (size_t)ws->ld * sizeof(D), s));
  const int nbk = c->num_cu * 4;
  double r2 = 0.0;
  for (int i = 0; i < nx; ++i) {
    A->in_scale = 1.0;
    A->apply(ws->col(i), y, nullptr);
    std::memcpy(ws->coef_h, C + (size_t)i * ldc, (size_t)ny * sizeof(T));
    KS_HIP(hipMemcpyAsync(ws->coef, ws->coef_h, (size_t)ny * sizeof(T), hipMemcpyHostToDevice, s));
    ksd::k_sub_lincomb<D><<<nbk, kBlock, 0, s>>>(y, static_cast<const D*>(ws->V), ws->ld, ny, static_cast<const D*>(ws->coef), ws->n);
    ksd::k_norm2<D><<<ws->nb, kBlock, 0, s>>>(y, ws->ld, ws->partial2);
    ksd::k_sum<<<1, kBlock, 0, s>>>(ws->partial2, ws->nb, ws->scal);
    c->allreduce(ws->scal, 1);
    KS_HIP(hipMemcpyAsync(ws->scal_h, ws->scal, 8, hipMemcpyDeviceToHost, s));
    KS_HIP(hipStreamSynchronize(s));
    r2 += ws->scal_h[0];
  }
  *resid = std::sqrt(r2);
  // Gram matrix in 8x8 tiles
  const int gnb = std::min(ws->nb, c->num_cu * 2);
  D* gp = static_cast<D*>(ws->ensure_tmp2((size_t)gnb * 64 * sizeof(D) + 64 * sizeof(D)));
  D* gout = gp + (size_t)gnb * 64;
  std::vector<T> tile(64);
  double o2 = 0.0;
  for (int i0 = 0; i0 < ny; i0 += 8)
    for (int j0 = 0; j0 < ny; j0 += 8) {
      const int na = std::min(8, ny - i0), nbc = std::min(8, ny - j0);
      ksd::k_gram_tile<D><<<gnb, kBlock, 0, s>>>(static_cast<const D*>(ws->col(i0)), ws->ld, na, static_cast<const D*>(ws->col(j0)),
                                                  ws->ld, nbc, ws->n, gp);
      ksd::k_reduce_cols<D><<<1, kBlock, 0, s>>>(gp, gnb, 64, 64, gout);
      c->allreduce(reinterpret_cast<double*>(gout), 64 * (int)(sizeof(D) / 8));
      KS_HIP(hipMemcpyAsync(tile.data(), gout, 64 * sizeof(D), hipMemcpyDeviceToHost, s));
      KS_HIP(hipStreamSynchronize(s));
      for (int jj = 0; jj < nbc; ++jj)
        for (int ii = 0; ii < na; ++ii) {
          T g = tile[ii + 8 * jj];
          if (i0 + ii == j0 + jj) g -= T(1);
          o2 += ks::abs2_(g);
        }
    }
  *orth = std::sqrt(o2);
}

}  // namespace

namespace {
// Placement tuning.  How fast the streaming kernels run on a basis of several GB depends on WHICH physical
// pages back it: K simultaneous allocations of the same size in one process differ reproducibly by 6-8 %
// (tools/placement_probe2.hip; the launches of three real steps take 5.31..5.67 ms on eight candidates at
// n = 1e7) while offsets inside one allocation and the leading dimension make no difference
// (tools/placement_probe.hip).  This is what made identical runs land on two plateaus 4 % apart.  So a large
// workspace allocates a few candidates for V, times the launches of real steps at three basis sizes on each
// (zeros in, zeros out) and keeps the fastest (search policy: tune_placement below); only for a basis of at
// least KS_PLACE_MIN_MB (1024) MB; KS_PLACE_TRIALS=1 disables.
template <class D> double placement_trio_ms(ks_workspace* w, hipEvent_t a, hipEvent_t b) {
  ks_ctx* c = w->ctx;
  const int jmax = std::min(w->maxdim, 40);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {  // rep 0 warms up
    KS_HIP(hipEventRecord(a, c->stream));
    for (int j = jmax; j >= 1 && j > jmax - 20; j -= 6) {  // the launches of real steps at a few basis sizes
      D* col = static_cast<D*>(w->col(j));
      launch_dots<D>(w, j, col, 1, nullptr);
      launch_axpy_dots<D>(w, std::min(j, kFusedMaxJ), col, 0);
      const int64_t ppb = (w->ld / 2) / std::max(1, w->nb);
      if (sizeof(D) == 8 && ppb >= 3072)
        ksd::k_axpy<D, 8><<<w->nb, kBlock, 0, c->stream>>>(static_cast<const D*>(w->V), w->ld, j, col, static_cast<const D*>(w->coef), w->partial2, 1, nullptr);
      else
        ksd::k_axpy<D, 4><<<w->nb, kBlock, 0, c->stream>>>(static_cast<const D*>(w->V), w->ld, j, col, static_cast<const D*>(w->coef), w->partial2, 1, nullptr);
    }
    KS_HIP(hipEventRecord(b, c->stream));
    KS_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    KS_HIP(hipEventElapsedTime(&ms, a, b));
    if (rep > 0) best = std::min(best, ms);
  }
  return best;
}

// Search policy (round 2: opt-in, bounded and exception-safe).  KS_PLACE_TRIALS=N (N >= 2) times N candidate
// allocations of V and keeps the fastest; it HOLDS its candidates while it runs (a freed block would simply be handed
// out again), at most KS_PLACE_MAX_X (default 2) times the basis size and only while half of the free memory stays
// untouched, within KS_PLACE_BUDGET_MS.  Default KS_PLACE_TRIALS=1: no search.  Candidates live in an RAII holder:
// whatever happens, every loser is freed and w->V / w->Vbase name the kept allocation.
struct PlacementCandidates {
  ks_workspace* w;
  std::vector<void*> cand;
  size_t keep = 0;
  int failed = 0;  // allocations that were refused (reported through ks_workspace_placement)
  explicit PlacementCandidates(ks_workspace* ws) : w(ws), cand{ws->V} {}
  ~PlacementCandidates() {
    for (size_t k = 0; k < cand.size(); ++k)
      if (k != keep) (void)hipFree(cand[k]);
    w->V = cand[keep];
    w->Vbase = w->V;
  }
};

template <class D> void tune_placement(ks_workspace* w, size_t vbytes) {
  // OFF by default (round 2).  The gain of round 1 (+3 % on the streaming kernels) came from ONE kind of candidate, a
  // physically contiguous allocation (hipDeviceMallocContiguous, now behind KS_PLACE_CONTIGUOUS=1) -- and in such memory
  // the SpMV, which re-reads every x element seven times and lives on L2 hits, runs 2.3x SLOWER (42 -> 96 us): the
  // solver as a whole loses (669 vs 677 iterations/s, profiles/r02_placement_ab.txt).  Plain candidates are
  // indistinguishable from each other on the boxes measured.  KS_PLACE_TRIALS >= 2 opts in.
  static const int trials = env_int("KS_PLACE_TRIALS", 1);
  // measured: +3 % at 3.3 GB, +1.5 % at 1.6 GB, nothing at 0.8 GB, -2 % at 0.4 GB (there the calibration, which
  // revisits the same columns, sees the memory-side cache more than the placement)
  static const int min_mb = env_int("KS_PLACE_MIN_MB", 1024);
  static const int budget_ms = env_int("KS_PLACE_BUDGET_MS", 1500);
  static const int max_x = std::max(2, env_int("KS_PLACE_MAX_X", 2));  // total footprint of held candidates / basis size
  static const int debug = env_int("KS_PLACE_DEBUG", 0);
  if (trials <= 1 || vbytes < ((size_t)min_mb << 20) || w->guard) return;
  ks_ctx* c = w->ctx;
  struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  KS_HIP(hipEventCreate(&ev.a));
  KS_HIP(hipEventCreate(&ev.b));
  PlacementCandidates pc(w);
  double best_ms = 1e30, worst_ms = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t k = 0;; ++k) {
    w->V = pc.cand[k];
    const double ms = placement_trio_ms<D>(w, ev.a, ev.b);
    if (debug) std::fprintf(stderr, "[ks] placement candidate %zu @%p: %.3f ms\n", k, pc.cand[k], ms);
    if (ms < best_ms) { best_ms = ms; pc.keep = k; }
    worst_ms = std::max(worst_ms, ms);
    const double spent = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if ((int)pc.cand.size() >= trials || spent > budget_ms) break;
    if ((int)pc.cand.size() + 1 > max_x) break;                      // footprint cap: held candidates <= max_x * V
    if (pc.cand.size() >= 4 && best_ms <= 0.955 * worst_ms) break;   // a candidate from the fast cluster was found
    size_t free_b = 0, total_b = 0;
    KS_HIP(hipMemGetInfo(&free_b, &total_b));
    if (free_b / 2 < vbytes) { pc.failed++; break; }                 // never take more than half of what is left
    void* p = nullptr;
    static const int try_contig = env_int("KS_PLACE_CONTIGUOUS", 0);
    if (try_contig && pc.cand.size() == 1 && hipExtMallocWithFlags(&p, vbytes, hipDeviceMallocContiguous) != hipSuccess) {
      (void)hipGetLastError();
      pc.failed++;
      p = nullptr;
    }
    if (!p && hipMalloc(&p, vbytes) != hipSuccess) { (void)hipGetLastError(); pc.failed++; break; }
    pc.cand.push_back(p);  // owned by the holder from here on
    KS_HIP(hipMemsetAsync(p, 0, vbytes, c->stream));
  }
  w->place_candidates = (int)pc.cand.size();
  w->place_failed = pc.failed;
  w->place_best_ms = best_ms;
  w->place_worst_ms = worst_ms;
  if (debug) std::fprintf(stderr, "[ks] placement: kept candidate %zu of %zu (%.3f ms, slowest %.3f ms, %d refused)\n", pc.keep, pc.cand.size(), best_ms, worst_ms, pc.failed);
  // ~PlacementCandidates frees the losers and points w->V at the kept one; the calibration wrote (zeros) into the
  // scratch of the reductions only, V is still all zero
}
}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* ks_last_error_string(void) { return g_last_error.c_str(); }

int ks_version(int* major, int* minor) {
  if (major) *major = KS_VERSION_MAJOR;
  if (minor) *minor = KS_VERSION_MINOR;
  return KS_OK;
}

int ks_ctx_create(int device, ks_ctx** out) {
  return guarded([&] {
    KS_REQUIRE(out, KS_ERR_ARGUMENT, "null out");
    auto c = std::make_unique<ks_ctx>();
    ctx_init_device(c.get(), device);
    *out = c.release();
  });
}

int ks_comm_unique_id(void* out128) {
  return guarded([&] {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    KS_NCCL(ncclGetUniqueId(&id));
    std::memcpy(out128, &id, 128);
  });
}

int ks_ctx_create_dist(int device, int rank, int nranks, const void* unique_id128, ks_ctx** out) {
  return guarded([&] {
    KS_REQUIRE(out && unique_id128, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, KS_ERR_ARGUMENT, "bad rank/nranks");
    auto c = std::make_unique<ks_ctx>();
    ctx_init_device(c.get(), device);
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, 128);
    KS_NCCL(ncclCommInitRank(&c->comm, nranks, id, rank));
    // KS_TRANSPORT=p2p: keep RCCL for bootstrap only (IPC handles travel through one all-gather) and run
    // the solver's reductions and halo over the peer-to-peer region
    const char* tr = std::getenv("KS_TRANSPORT");
    if (tr && std::string(tr) == "p2p") {
      p2p_alloc(c.get());
      hipIpcMemHandle_t mine;
      KS_HIP(hipIpcGetMemHandle(&mine, c->p2p.region));
      char* dbuf = nullptr;
      KS_HIP(hipMalloc(&dbuf, (size_t)64 * nranks));
      KS_HIP(hipMemcpy(dbuf + (size_t)64 * rank, &mine, 64, hipMemcpyHostToDevice));
      KS_NCCL(ncclAllGather(dbuf + (size_t)64 * rank, dbuf, 64, ncclChar, c->comm, c->stream));
      std::vector<char> all((size_t)64 * nranks);
      KS_HIP(hipMemcpyAsync(all.data(), dbuf, all.size(), hipMemcpyDeviceToHost, c->stream));
      KS_HIP(hipStreamSynchronize(c->stream));
      (void)hipFree(dbuf);
      p2p_attach(c.get(), all.data());
    }
    *out = c.release();
  });
}

int ks_ctx_create_hostcomm(int device, int rank, int nranks, ks_host_allreduce_fn allreduce, ks_host_exchange_fn exchange,
                           void* user, ks_ctx** out) {
  return guarded([&] {
    KS_REQUIRE(out && allreduce && exchange, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, KS_ERR_ARGUMENT, "bad rank/nranks");
    auto c = std::make_unique<ks_ctx>();
    ctx_init_device(c.get(), device);
    c->rank = rank;
    c->nranks = nranks;
    c->hc.allreduce = allreduce;
    c->hc.exchange = exchange;
    c->hc.user = user;
    *out = c.release();
  });
}

int ks_ctx_create_p2p(int device, int rank, int nranks, ks_ctx** out) {
  return guarded([&] {
    KS_REQUIRE(out, KS_ERR_ARGUMENT, "null out");
    KS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, KS_ERR_ARGUMENT, "bad rank/nranks");
    auto c = std::make_unique<ks_ctx>();
    ctx_init_device(c.get(), device);
    c->rank = rank;
    c->nranks = nranks;
    p2p_alloc(c.get());
    if (nranks == 1) p2p_attach(c.get(), nullptr);
    *out = c.release();
  });
}

int ks_ctx_p2p_handle(ks_ctx* ctx, void* out64) {
  return guarded([&] {
    KS_REQUIRE(ctx && out64, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(ctx->p2p.allocated, KS_ERR_ARGUMENT, "context was not created with ks_ctx_create_p2p");
    ctx->use();
    hipIpcMemHandle_t h;
    KS_HIP(hipIpcGetMemHandle(&h, ctx->p2p.region));
    std::memcpy(out64, &h, 64);
  });
}

int ks_ctx_p2p_attach(ks_ctx* ctx, const void* handles) {
  return guarded([&] {
    KS_REQUIRE(ctx && handles, KS_ERR_ARGUMENT, "null argument");
    ctx->use();
    p2p_attach(ctx, handles);
  });
}

int ks_ctx_destroy(ks_ctx* ctx) {
  return guarded([&] {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    if (ctx->hc.stage) (void)hipHostFree(ctx->hc.stage);
    p2p_release(ctx);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
  });
}

int ks_ctx_synchronize(ks_ctx* ctx) {
  return guarded([&] {
    KS_REQUIRE(ctx, KS_ERR_ARGUMENT, "null ctx");
    ctx->use();
    KS_HIP(hipStreamSynchronize(ctx->stream));
    ctx->check_comm();
  });
}

int ks_ctx_rank(const ks_ctx* ctx, int* rank, int* nranks) {
  return guarded([&] {
    KS_REQUIRE(ctx, KS_ERR_ARGUMENT, "null ctx");
    if (rank) *rank = ctx->rank;
    if (nranks) *nranks = ctx->nranks;
  });
}

int ks_ctx_stream(ks_ctx* ctx, void** hip_stream) {
  return guarded([&] {
    KS_REQUIRE(ctx && hip_stream, KS_ERR_ARGUMENT, "null argument");
    *hip_stream = (void*)ctx->stream;
  });
}

// ---- operators -----------------------------------------------------------------------------------
int ks_operator_csr(ks_ctx* ctx, int64_t nrows_local, int64_t ncols, int64_t nnz, const void* ptr, const void* idx,
                    const void* val, int layout, int index_base, int index_type, int dtype, ks_operator** out) {
  return guarded([&] {
    KS_REQUIRE(ctx && out, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(nrows_local >= 0 && ncols >= 0 && nnz >= 0, KS_ERR_ARGUMENT, "negative size");
    KS_REQUIRE(ctx->nranks > 1 || nrows_local == ncols, KS_ERR_DIMENSION,
               "matrix is not square: dimensions are (" + std::to_string(nrows_local) + ", " + std::to_string(ncols) + ")");
    KS_REQUIRE(ctx->nranks == 1, KS_ERR_ARGUMENT, "use ks_operator_csr_dist on a multi-rank context");
    KS_REQUIRE(layout == KS_CSR || layout == KS_CSC, KS_ERR_ARGUMENT, "bad layout");
    KS_REQUIRE(index_type == KS_I32 || index_type == KS_I64, KS_ERR_ARGUMENT, "bad index type");
    ctx->use();
    std::vector<int64_t> rp;
    std::vector<int32_t> ci;
    dispatch_dtype(dtype, [&](auto tag) {
      using T = decltype(tag);
      using D = typename DevT<T>::type;
      std::vector<D> vv;
      build_csr_host<D>(nrows_local, ncols, nnz, ptr, idx, val, layout, index_base, index_type, rp, ci, vv);
      *out = make_csr<D>(ctx, nrows_local, nnz, rp, ci, vv, /*cb_mode=*/(ctx->nranks == 1 && nrows_local == ncols) ? 1 : 0);
    });
  });
}

int ks_operator_csr_dist(ks_ctx* ctx, int64_t nrows_local, int64_t nghost, int64_t nnz, const int64_t* rowptr,
                         const int32_t* colidx, const void* val, int dtype, int nneigh, const int32_t* neigh,
                         const int64_t* send_ptr, const int32_t* send_idx, const int64_t* recv_cnt, ks_operator** out) {
  return guarded([&] {
    KS_REQUIRE(ctx && out && rowptr, KS_ERR_ARGUMENT, "null argument");
    ctx->use();
    KS_REQUIRE(nrows_local < (int64_t)2147483647 && nghost < (int64_t)2147483647 - nrows_local, KS_ERR_ARGUMENT,
               "local-extended column range must fit int32");
    std::vector<int64_t> rp(nrows_local + 1);
    KS_REQUIRE(rowptr[0] == 0 && rowptr[nrows_local] == nnz, KS_ERR_ARGUMENT, "row pointer does not match nnz");
    for (int64_t i = 0; i <= nrows_local; ++i) {
      KS_REQUIRE(i == 0 || (rowptr[i] >= rowptr[i - 1] && rowptr[i] <= nnz), KS_ERR_ARGUMENT,
                 "pointer array is not monotone within [0, nnz]");
      rp[i] = rowptr[i];
    }
    std::vector<int32_t> ci(colidx, colidx + nnz);
    for (int64_t p = 0; p < nnz; ++p)
      KS_REQUIRE(ci[p] >= 0 && ci[p] < nrows_local + nghost, KS_ERR_ARGUMENT, "local-extended column index out of range");
    dispatch_dtype(dtype, [&](auto tag) {
      using T = decltype(tag);
      using D = typename DevT<T>::type;
      std::vector<D> vv(static_cast<const D*>(val), static_cast<const D*>(val) + nnz);
      CsrOp<D>* op = make_csr<D>(ctx, nrows_local, nnz, rp, ci, vv);
      std::unique_ptr<CsrOp<D>> guard(op);
      op->nghost = nghost;
      op->p2p_halo = ctx->p2p.attached;
      {
        // rows that reference ghost columns: all of them lie outside ONE interval [ghost_lo_end, ghost_hi_begin) -- the
        // largest gap between such rows (a slab of a structured grid: everything between its first and its last plane).
        // Workgroups whose rows fall inside never wait for the exchange (fused halo, ks_p2p.hpp).
        int64_t best_a = 0, best_b = nrows_local, prev = -1;
        bool any = false;
        int64_t gap_best = -1;
        for (int64_t r = 0; r < nrows_local; ++r) {
          bool touches = false;
          for (int64_t q = rp[r]; q < rp[r + 1] && !touches; ++q) touches = ci[q] >= nrows_local;
          if (!touches) continue;
          any = true;
          if (r - prev - 1 > gap_best) { gap_best = r - prev - 1; best_a = prev + 1; best_b = r; }
          prev = r;
        }
        if (any && nrows_local - prev - 1 > gap_best) { best_a = prev + 1; best_b = nrows_local; }
        op->ghost_lo_end = any ? best_a : 0;
        op->ghost_hi_begin = any ? best_b : nrows_local;
      }
      if (!op->p2p_halo) {
        KS_HIP(hipMalloc(&op->ghost, std::max<size_t>((size_t)nghost * sizeof(D), 16)));
        KS_HIP(hipMemset(op->ghost, 0, std::max<size_t>((size_t)nghost * sizeof(D), 16)));
      }
      op->neigh.assign(neigh, neigh + nneigh);
      op->send_ptr.assign(send_ptr, send_ptr + nneigh + 1);
      op->recv_ptr.assign(nneigh + 1, 0);
      for (int p = 0; p < nneigh; ++p) op->recv_ptr[p + 1] = op->recv_ptr[p] + recv_cnt[p];
      KS_REQUIRE(op->recv_ptr[nneigh] == nghost, KS_ERR_ARGUMENT, "recv counts do not add up to nghost");
      // split the send lists into contiguous runs (sent in place) and scattered ones (packed)
      std::vector<int32_t> packed;
      op->send_first.assign(nneigh, -1);
      op->pack_ptr.assign(nneigh + 1, 0);
      for (int p = 0; p < nneigh; ++p) {
        const int64_t a = send_ptr[p], b = send_ptr[p + 1];
        bool contiguous = b > a;
        for (int64_t q = a; q < b; ++q) {
          KS_REQUIRE(send_idx[q] >= 0 && send_idx[q] < nrows_local, KS_ERR_ARGUMENT, "send index out of range");
          if (q > a && send_idx[q] != send_idx[q - 1] + 1) contiguous = false;
        }
        if (contiguous) op->send_first[p] = send_idx[a];
        else packed.insert(packed.end(), send_idx + a, send_idx + b);
        op->pack_ptr[p + 1] = (int64_t)packed.size();
      }
      op->nscatter = (int64_t)packed.size();
      KS_HIP(hipMalloc(&op->sendbuf, std::max<size_t>(packed.size() * sizeof(D), 16)));
      KS_HIP(hipMalloc(&op->send_idx, std::max<size_t>(packed.size() * 4, 16)));
      if (!packed.empty()) KS_HIP(hipMemcpy(op->send_idx, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
      KS_REQUIRE(nneigh == 0 || ctx->distributed(), KS_ERR_ARGUMENT, "halo plan needs a distributed context");
      if (ctx->hc.exchange) {
        KS_HIP(hipHostMalloc(&op->hsend, std::max<size_t>((size_t)send_ptr[nneigh] * sizeof(D), 16)));
        KS_HIP(hipHostMalloc(&op->hrecv, std::max<size_t>((size_t)nghost * sizeof(D), 16)));
      }
      if (op->p2p_halo) {
        // COLLECTIVE in peer-to-peer mode: every rank publishes where its ghost vector lives in its shared
        // arena and, per sender, at which offset that sender's entries go (and how many it expects)
        auto& P = ctx->p2p;
        const int R = ctx->nranks;
        KS_REQUIRE(nneigh <= ksd::kP2pMaxNeigh, KS_ERR_ARGUMENT, "peer-to-peer halo supports at most 16 neighbours per rank");
        op->ghost_stride = round_up(std::max<int64_t>(nghost, 1), 32);
        const size_t bytes = (size_t)round_up(2 * op->ghost_stride * (int64_t)sizeof(D), 256);
        KS_REQUIRE(P.arena_used + bytes <= P.arena_bytes, KS_ERR_ARGUMENT,
                   "ghost vectors do not fit the shared arena (raise KS_P2P_ARENA_MB)");
        op->arena_lo = P.arena_used;
        op->arena_hi = P.arena_used + bytes;
        P.arena_used = op->arena_hi;
        op->ghost = reinterpret_cast<D*>(static_cast<char*>(P.region) + P.arena_off + op->arena_lo);
        KS_HIP(hipMemset(op->ghost, 0, bytes));
        KS_HIP(hipDeviceSynchronize());
        std::vector<int64_t> row((size_t)2 + 2 * R, 0);
        row[0] = (int64_t)(P.arena_off + op->arena_lo);
        row[1] = op->ghost_stride;
        for (int p = 0; p < nneigh; ++p) {
          KS_REQUIRE(neigh[p] >= 0 && neigh[p] < R, KS_ERR_ARGUMENT, "bad neighbour rank");
          row[2 + 2 * neigh[p]] = op->recv_ptr[p];
          row[3 + 2 * neigh[p]] = recv_cnt[p];
        }
        const std::vector<int64_t> tab = p2p_allgather_i64(ctx, row);
        const size_t K = row.size();
        ksd::HaloArgs a{};
        a.nneigh = nneigh;
        a.nrecv = 0;
        std::vector<int32_t> all;
        for (int p = 0; p < nneigh; ++p) {
          const int q = neigh[p];
          const int64_t sc = send_ptr[p + 1] - send_ptr[p];
          a.send_ptr[p] = send_ptr[p];
          a.send_ptr[p + 1] = send_ptr[p + 1];
          const int64_t* tq = tab.data() + (size_t)q * K;
          KS_REQUIRE(tq[3 + 2 * ctx->rank] == sc, KS_ERR_ARGUMENT,
                     "halo plans disagree: rank " + std::to_string(q) + " expects " + std::to_string(tq[3 + 2 * ctx->rank]) +
                         " entries from rank " + std::to_string(ctx->rank) + ", which sends " + std::to_string(sc));
          a.dst[p] = static_cast<char*>(P.peer[q]) + tq[0] + tq[2 + 2 * ctx->rank] * (int64_t)sizeof(D);
          a.dst_stride[p] = tq[1];
          a.flag_dst[p] = static_cast<uint64_t*>(P.peer[q]) + ksd::p2p_ll_words(R, P.cap) + ctx->rank;
          if (recv_cnt[p] > 0) a.recv_from[a.nrecv++] = q;
          all.insert(all.end(), send_idx + send_ptr[p], send_idx + send_ptr[p + 1]);
        }
        op->hargs = a;
        KS_HIP(hipMalloc(&op->send_idx_all, std::max<size_t>(all.size() * 4, 16)));
        if (!all.empty()) KS_HIP(hipMemcpy(op->send_idx_all, all.data(), all.size() * 4, hipMemcpyHostToDevice));
      }
      *out = guard.release();
    });
  });
}

int ks_operator_dense(ks_ctx* ctx, int64_t n, const void* a, int64_t ld, int layout, int dtype, ks_operator** out) {
  return guarded([&] {
    KS_REQUIRE(ctx && out && (a || n == 0), KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(n >= 0 && ld >= n, KS_ERR_ARGUMENT, "leading dimension smaller than the matrix order");
    KS_REQUIRE(layout == KS_ROW_MAJOR || layout == KS_COL_MAJOR, KS_ERR_ARGUMENT, "bad layout");
    KS_REQUIRE(ctx->nranks == 1, KS_ERR_ARGUMENT, "the dense operator is single-GPU (shard a dense matrix through a device callback)");
    ctx->use();
    dispatch_dtype(dtype, [&](auto tag) {
      using T = decltype(tag);
      using D = typename DevT<T>::type;
      auto op = std::make_unique<DenseOp<D>>();
      op->ctx = ctx; op->n_local = n; op->nnz = n * n; op->dtype = dtype;
      op->bytes_per_nnz = sizeof(D);
      op->lda = round_up(std::max<int64_t>(n, 1), 2);
      const size_t bytes = (size_t)op->lda * std::max<int64_t>(n, 1) * sizeof(D);
      KS_HIP(hipMalloc(&op->A, bytes));
      KS_HIP(hipMemset(op->A, 0, bytes));
      const D* src = static_cast<const D*>(a);
      if (n > 0 && layout == KS_ROW_MAJOR) {
        KS_HIP(hipMemcpy2D(op->A, (size_t)op->lda * sizeof(D), src, (size_t)ld * sizeof(D), (size_t)n * sizeof(D), (size_t)n, hipMemcpyHostToDevice));
      } else if (n > 0) {  // column-major (Julia): transpose on the host in row panels, upload panel by panel
        const int64_t panel = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(64 << 20) / (int64_t)(op->lda * sizeof(D))));
        std::vector<D> buf((size_t)panel * op->lda);
        for (int64_t r0 = 0; r0 < n; r0 += panel) {
          const int64_t rows = std::min(panel, n - r0);
          std::memset(buf.data(), 0, (size_t)rows * op->lda * sizeof(D));
          for (int64_t c = 0; c < n; ++c)
            for (int64_t r = 0; r < rows; ++r) buf[(size_t)r * op->lda + c] = src[(size_t)c * ld + r0 + r];
          KS_HIP(hipMemcpy(op->A + (size_t)r0 * op->lda, buf.data(), (size_t)rows * op->lda * sizeof(D), hipMemcpyHostToDevice));
        }
      }
      *out = op.release();
    });
  });
}

int ks_operator_host_callback(ks_ctx* ctx, int64_t n_local, int dtype, ks_host_apply_fn apply, void* user,
                              ks_operator** out) {
  return guarded([&] {
    KS_REQUIRE(ctx && out && apply, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(dtype == KS_F64 || dtype == KS_C64, KS_ERR_ARGUMENT, "unknown dtype");
    ctx->use();
    auto op = std::make_unique<HostCallbackOp>();
    op->ctx = ctx; op->n_local = n_local; op->dtype = dtype; op->fn = apply; op->user = user;
    op->async_capable = false;
    const size_t bytes = std::max<size_t>((size_t)n_local * (dtype == KS_F64 ? 8 : 16), 16);
    KS_HIP(hipHostMalloc(&op->xh, bytes));
    KS_HIP(hipHostMalloc(&op->yh, bytes));
    *out = op.release();
  });
}

int ks_operator_device_callback(ks_ctx* ctx, int64_t n_local, int dtype, ks_device_apply_fn apply, void* user,
                                ks_operator** out) {
  return guarded([&] {
    KS_REQUIRE(ctx && out && apply, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(dtype == KS_F64 || dtype == KS_C64, KS_ERR_ARGUMENT, "unknown dtype");
    auto op = std::make_unique<DeviceCallbackOp>();
    op->ctx = ctx; op->n_local = n_local; op->dtype = dtype; op->fn = apply; op->user = user;
    *out = op.release();
  });
}

int ks_operator_destroy(ks_operator* op) {
  return guarded([&] {
    if (!op) return;
    (void)hipSetDevice(op->ctx->device);
    (void)hipStreamSynchronize(op->ctx->stream);
    delete op;
  });
}

int ks_operator_size(const ks_operator* op, int64_t* n_local, int64_t* nnz, int* dtype) {
  return guarded([&] {
    KS_REQUIRE(op, KS_ERR_ARGUMENT, "null operator");
    if (n_local) *n_local = op->n_local;
    if (nnz) *nnz = op->nnz;
    if (dtype) *dtype = op->dtype;
  });
}

int ks_operator_format(const ks_operator* op, double* bytes_per_nnz, int* ndict, int* layout) {
  return guarded([&] {
    KS_REQUIRE(op, KS_ERR_ARGUMENT, "null operator");
    if (layout) *layout = op->layout;
    if (bytes_per_nnz) *bytes_per_nnz = op->bytes_per_nnz;
    if (ndict) {
      *ndict = 0;
      if (op->dtype == KS_F64) { if (auto* c = dynamic_cast<const CsrOp<double>*>(op)) *ndict = c->nstencil > 0 ? c->nstencil : c->ndvi > 0 ? c->ndvi : c->ndict; }
      else if (auto* c = dynamic_cast<const CsrOp<cd>*>(op)) *ndict = c->nstencil > 0 ? c->nstencil : c->ndvi > 0 ? c->ndvi : c->ndict;
    }
  });
}

int ks_operator_apply_raw(ks_operator* op, const void* x_dev, void* y_dev) {
  return guarded([&] {
    KS_REQUIRE(op && x_dev && y_dev, KS_ERR_ARGUMENT, "null argument");
    op->ctx->use();
    op->in_scale = 1.0;
    op->apply(x_dev, y_dev, nullptr);
  });
}

// ---- workspace -----------------------------------------------------------------------------------
int ks_workspace_create(ks_ctx* ctx, int64_t n_local, int64_t n_global, int64_t row_begin, int maxdim, int dtype,
                        ks_workspace** out) {
  return guarded([&] {
    KS_REQUIRE(ctx && out, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(dtype == KS_F64 || dtype == KS_C64, KS_ERR_ARGUMENT, "unknown dtype");
    KS_REQUIRE(maxdim >= 1, KS_ERR_ARGUMENT, "maxdim must be positive");
    // ArnoldiWorkspace(T, n, k): "Krylov dimension should be less than matrix order." src/ArnoldiMethod.jl:62-63
    KS_REQUIRE((int64_t)maxdim <= n_global, KS_ERR_ARGUMENT, "Krylov dimension should be less than matrix order.");
    KS_REQUIRE(n_local >= 0 && n_local <= n_global, KS_ERR_ARGUMENT, "bad n_local");
    ctx->use();
    auto w = std::make_unique<ks_workspace>();
    w->ctx = ctx;
    w->dtype = dtype;
    w->esz = dtype == KS_F64 ? 8 : 16;
    w->n = n_local;
    w->n_global = n_global;
    w->row_begin = row_begin;
    w->maxdim = maxdim;
    w->esz = dtype == KS_F64 ? 8 : 16;
    w->ld = std::max<int64_t>(round_up(n_local, 64), 64);
    {
      // COLUMN STRIDE.  A step streams 20-40 columns of V at the same row offset; how those streams fall onto the memory
      // channels depends on the stride between columns modulo the interleave.  Measured (tools/stride_scan.sh,
      // profiles/r02_column_stride.txt): with the stride at 0xEA00..0xFE00 modulo 128 KiB k_dots runs at 6.6-6.8 TB/s and
      // k_axpy_dots_cs at 5.8-5.9, against 6.1-6.2 / 5.4-5.6 at the strides n happens to give (216^3, 215^3, 200^3,
      // 232^3, 160^3: +3.5 ... +8 % iterations/s, every size).  A fixed rule, not a timing search: results stay
      // reproducible.  Costs at most 128 KiB per column, applied from 4 MiB columns on.  KS_LD_PAD=<512-byte units>
      // overrides (0 = none), KS_STRIDE_RULE=0 disables.
      const int pad_env = env_int("KS_LD_PAD", -1);
      const int64_t colb = w->ld * (int64_t)w->esz;
      if (pad_env >= 0) {
        w->ld += 64 * (int64_t)pad_env * (w->esz == 8 ? 1 : 1);
      } else if (env_int("KS_STRIDE_RULE", 1) && colb >= ((int64_t)4 << 20)) {
        const int64_t target = 0xF800, window = 0x20000;
        const int64_t pad_bytes = ((target - colb % window) % window + window) % window;
        w->ld += pad_bytes / (int64_t)w->esz;
      }
    }
    w->pstride = (int)round_up(maxdim + 2, 8);
    w->esz = dtype == KS_F64 ? 8 : 16;
    w->pnb = ctx->nblocks();
    w->nb = cap_blocks(w.get(), w->pnb, 2 * kBlock);  // generic streaming grid
    const size_t esz = w->esz;
    const size_t vbytes = (size_t)w->ld * (maxdim + 1) * esz;
    w->vbytes = vbytes;
    if (env_int("KS_GUARD", 0)) {  // debugging: 1 MiB of 0xA5 on both sides of V, verified by ks_workspace_check_guard
      w->guard = (size_t)1 << 20;
      KS_HIP(hipMalloc(&w->Vbase, vbytes + 2 * w->guard));
      KS_HIP(hipMemsetAsync(w->Vbase, 0xA5, vbytes + 2 * w->guard, ctx->stream));
      w->V = static_cast<char*>(w->Vbase) + w->guard;
    } else {
      KS_HIP(hipMalloc(&w->V, vbytes));
      w->Vbase = w->V;
    }
    KS_HIP(hipMemsetAsync(w->V, 0, vbytes, ctx->stream));
    const size_t hbytes = (size_t)(maxdim + 1) * maxdim * esz, qbytes = (size_t)maxdim * maxdim * esz;
    const size_t qxbytes = (size_t)(maxdim + 1) * (maxdim + 1) * esz;  // T-folded rotations: one more row and column
    KS_HIP(hipHostMalloc(&w->H, hbytes));
    KS_HIP(hipHostMalloc(&w->Q, qbytes));
    // control block: [ Hd | DevState (128-byte slot) | colscale ] on the device, the same layout pinned on the host
    w->hd_bytes = (size_t)round_up((int64_t)hbytes, 64);
    w->ldt = maxdim + 1;
    w->off_T = (size_t)round_up((int64_t)(w->hd_bytes + kCtlStateSlot + (size_t)(maxdim + 2) * 8), 64);
    w->off_g = w->off_T + (size_t)w->ldt * w->ldt * esz;
    w->ctl_bytes = (size_t)round_up((int64_t)(w->off_g + (size_t)w->ldt * esz), 64);
    const size_t ctl_bytes = w->ctl_bytes;
    static_assert(sizeof(DevState) <= kCtlStateSlot, "DevState must fit its slot of the control block");
    // the images the device publishes into (k_publish) and the flag words the host spins on must be COHERENT pinned
    // memory whatever the process default is (HIP_HOST_COHERENT=0 makes plain hipHostMalloc memory non-coherent: device
    // stores would only become visible at a synchronisation and the spin would never see them)
    const unsigned coh = hipHostMallocCoherent | hipHostMallocMapped;
    KS_HIP(hipHostMalloc(&w->Hstage, ctl_bytes, coh));
    KS_HIP(hipHostMalloc(&w->Qstage, qxbytes));
    KS_HIP(hipHostMalloc(&w->Hstage_early, ctl_bytes, coh));
    std::memset(w->Hstage_early, 0, ctl_bytes);
    KS_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->mbox), 128, coh));
    std::memset(w->mbox, 0, 128);
    w->use_mbox = env_int("KS_MAILBOX", 1) != 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&w->mbox_dev), w->mbox, 0) != hipSuccess ||
        hipHostGetDevicePointer(&w->Hstage_dev, w->Hstage, 0) != hipSuccess ||
        hipHostGetDevicePointer(&w->Hstage_early_dev, w->Hstage_early, 0) != hipSuccess) {
      (void)hipGetLastError();
      w->use_mbox = false;  // no device view of the pinned images: copies + stream synchronisation instead
    }
    std::memset(w->H, 0, hbytes);   // zeros(T, k+1, k), src/ArnoldiMethod.jl:66
    std::memset(w->Q, 0, qbytes);
    std::memset(w->Hstage, 0, ctl_bytes);
    KS_HIP(hipMalloc(&w->Hd, ctl_bytes));
    KS_HIP(hipMemsetAsync(w->Hd, 0, ctl_bytes, ctx->stream));
    w->st = reinterpret_cast<DevState*>(static_cast<char*>(w->Hd) + w->hd_bytes);
    w->st_h = reinterpret_cast<DevState*>(static_cast<char*>(w->Hstage) + w->hd_bytes);
    w->colscale = reinterpret_cast<double*>(static_cast<char*>(w->Hd) + w->hd_bytes + kCtlStateSlot);
    w->cs_h = reinterpret_cast<double*>(static_cast<char*>(w->Hstage) + w->hd_bytes + kCtlStateSlot);
    w->Td = static_cast<char*>(w->Hd) + w->off_T;
    w->gd = static_cast<char*>(w->Hd) + w->off_g;
    w->Th = static_cast<char*>(w->Hstage) + w->off_T;
    KS_HIP(hipMalloc(reinterpret_cast<void**>(&w->ctr), 64));
    KS_HIP(hipMemsetAsync(w->ctr, 0, 64, ctx->stream));
    w->passes = env_int("KS_PASSES", 2) == 3 ? 3 : 2;
    if (const char* mr = std::getenv("KS_IMPLICIT_MAX_RATIO")) w->max_ratio = std::atof(mr);
    KS_HIP(hipMalloc(&w->Hscratch, (size_t)(maxdim + 2) * esz));
    KS_HIP(hipMalloc(&w->partial, (size_t)w->pnb * w->pstride * esz));
    KS_HIP(hipMalloc(&w->partial_s, (size_t)w->pnb * w->pstride * esz));
    KS_HIP(hipMalloc(&w->partial2, (size_t)std::max(w->pnb, ctx->num_cu * 8) * 8));
    KS_HIP(hipMalloc(&w->coef, (size_t)(w->pstride + 136) * esz));
    KS_HIP(hipMemsetAsync(w->coef, 0, (size_t)(w->pstride + 136) * esz, ctx->stream));
    KS_HIP(hipMalloc(&w->red, (size_t)(2 * w->pstride + 8) * esz));  // two reductions of <= maxdim+2 elements share a launch
    KS_HIP(hipMalloc(&w->scal, 64));
    KS_HIP(hipHostMalloc(&w->scal_h, 64));
    KS_HIP(hipHostMalloc(&w->coef_h, (size_t)w->pstride * esz));
    KS_HIP(hipMalloc(&w->Qd, std::max<size_t>(qxbytes, 16)));
    w->oop_mode = env_int("KS_OOP", 2);
    if (w->oop_mode == 1 || w->oop_mode == 2) {
      const size_t ob = (size_t)(w->oop_mode == 1 ? 2 : 1) * w->ld * esz;
      KS_HIP(hipMalloc(&w->oop, ob));
      KS_HIP(hipMemsetAsync(w->oop, 0, ob, ctx->stream));
    }
    w->hostscale.assign(maxdim + 2, 1.0);
    w->ones.assign(maxdim + 2, 1.0);
    w->colscale_dirty = true;  // first reset_state uploads the (all-one) factors
    reset_state(w.get());
    KS_HIP(hipStreamSynchronize(ctx->stream));
    if (dtype == KS_F64) tune_placement<double>(w.get(), vbytes);
    else tune_placement<cd>(w.get(), vbytes);
    reset_state(w.get());
    KS_HIP(hipStreamSynchronize(ctx->stream));
    *out = w.release();
  });
}

int ks_workspace_passes(const ks_workspace* ws, int* passes) {
  return guarded([&] {
    KS_REQUIRE(ws && passes, KS_ERR_ARGUMENT, "null argument");
    *passes = ws->passes;
  });
}

int ks_workspace_set_passes(ks_workspace* ws, int passes, double max_ratio) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    KS_REQUIRE(passes == 2 || passes == 3, KS_ERR_ARGUMENT, "passes must be 2 (implicit second pass) or 3 (explicit)");
    ws->ctx->use();
    materialize(ws);  // columns in the factored form of the other setting become ordinary first
    ws->passes = passes;
    if (!std::isnan(max_ratio)) ws->max_ratio = max_ratio;
  });
}

int ks_workspace_assert_arnoldi(ks_workspace* ws, int k) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    KS_REQUIRE(k >= 0 && k <= ws->maxdim, KS_ERR_ARGUMENT, "k out of range");
    prov_set(ws, k);
  });
}

int ks_workspace_provenance(const ks_workspace* ws, int* k) {
  return guarded([&] {
    KS_REQUIRE(ws && k, KS_ERR_ARGUMENT, "null argument");
    *k = ws->prov_k;
  });
}

int ks_workspace_placement(const ks_workspace* ws, int* candidates, double* best_ms, double* worst_ms, int* refused) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    if (refused) *refused = ws->place_failed;
    if (candidates) *candidates = ws->place_candidates;
    if (best_ms) *best_ms = ws->place_best_ms;
    if (worst_ms) *worst_ms = ws->place_worst_ms;
  });
}

int ks_workspace_check_guard(ks_workspace* ws, int* intact) {
  return guarded([&] {
    KS_REQUIRE(ws && intact, KS_ERR_ARGUMENT, "null argument");
    *intact = 1;
    if (!ws->guard) return;
    ws->ctx->use();
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
    std::vector<unsigned char> h(ws->guard);
    for (int side = 0; side < 2; ++side) {
      const char* src = static_cast<const char*>(ws->Vbase) + (side ? ws->guard + ws->vbytes : 0);
      KS_HIP(hipMemcpy(h.data(), src, ws->guard, hipMemcpyDeviceToHost));
      for (unsigned char b : h)
        if (b != 0xA5) { *intact = 0; return; }
    }
  });
}

int ks_workspace_destroy(ks_workspace* ws) {
  return guarded([&] {
    if (!ws) return;
    (void)hipSetDevice(ws->ctx->device);
    (void)hipStreamSynchronize(ws->ctx->stream);
    delete ws;
  });
}

int ks_workspace_dims(const ks_workspace* ws, int64_t* n_local, int* maxdim, int* dtype, int64_t* ldv) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    if (n_local) *n_local = ws->n;
    if (maxdim) *maxdim = ws->maxdim;
    if (dtype) *dtype = ws->dtype;
    if (ldv) *ldv = ws->ld;
  });
}

int ks_workspace_H(ks_workspace* ws, void** H, int* ldh) {
  return guarded([&] {
    KS_REQUIRE(ws && H, KS_ERR_ARGUMENT, "null argument");
    *H = ws->H;
    if (ldh) *ldh = ws->maxdim + 1;
  });
}

int ks_workspace_Q(ks_workspace* ws, void** Q, int* ldq) {
  return guarded([&] {
    KS_REQUIRE(ws && Q, KS_ERR_ARGUMENT, "null argument");
    *Q = ws->Q;
    if (ldq) *ldq = ws->maxdim;
  });
}

int ks_workspace_col_ptr(ks_workspace* ws, int j, void** dev_ptr) {
  return guarded([&] {
    check_col(ws, j);
    KS_REQUIRE(dev_ptr, KS_ERR_ARGUMENT, "null argument");
    ws->ctx->use();
    materialize(ws);
    prov_drop(ws);  // the caller may write through the pointer
    *dev_ptr = ws->col(j);
  });
}

int ks_workspace_set_seed(ks_workspace* ws, uint64_t seed) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    ws->seed = seed;
    ws->rng_count = 0;
  });
}

// ---- verbs ---------------------------------------------------------------------------------------
int ks_col_upload(ks_workspace* ws, int j, const void* host) {
  return guarded([&] {
    check_col(ws, j);
    KS_REQUIRE(host, KS_ERR_ARGUMENT, "null host pointer");
    ws->ctx->use();
    prov_drop(ws);  // the caller writes to V: the factorisation is no longer the library's own
    materialize(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) { col_upload<typename DevT<decltype(tag)>::type>(ws, j, host); });
  });
}

int ks_col_download(ks_workspace* ws, int j, void* host) {
  return guarded([&] {
    check_col(ws, j);
    KS_REQUIRE(host, KS_ERR_ARGUMENT, "null host pointer");
    ws->ctx->use();
    materialize(ws);
    KS_HIP(hipMemcpyAsync(host, ws->col(j), (size_t)ws->n * ws->esz, hipMemcpyDeviceToHost, ws->ctx->stream));
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_cols_download(ks_workspace* ws, int j0, int ncols, void* host, int64_t ldhost) {
  return guarded([&] {
    KS_REQUIRE(ws && host, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(j0 >= 0 && ncols >= 0 && j0 + ncols <= ws->maxdim + 1, KS_ERR_ARGUMENT, "column range out of bounds");
    KS_REQUIRE(ldhost >= ws->n, KS_ERR_ARGUMENT, "ldhost too small");
    if (ncols == 0 || ws->n == 0) return;
    ws->ctx->use();
    materialize(ws);
    KS_HIP(hipMemcpy2DAsync(host, (size_t)ldhost * ws->esz, ws->col(j0), (size_t)ws->ld * ws->esz,
                            (size_t)ws->n * ws->esz, (size_t)ncols, hipMemcpyDeviceToHost, ws->ctx->stream));
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_cols_upload(ks_workspace* ws, int j0, int ncols, const void* host, int64_t ldhost) {
  return guarded([&] {
    KS_REQUIRE(ws && host, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(j0 >= 0 && ncols >= 0 && j0 + ncols <= ws->maxdim + 1, KS_ERR_ARGUMENT, "column range out of bounds");
    KS_REQUIRE(ldhost >= ws->n, KS_ERR_ARGUMENT, "ldhost too small");
    if (ncols == 0 || ws->n == 0) return;
    ws->ctx->use();
    prov_drop(ws);  // the caller writes to V: the factorisation is no longer the library's own
    materialize(ws);
    KS_HIP(hipMemcpy2DAsync(ws->col(j0), (size_t)ws->ld * ws->esz, host, (size_t)ldhost * ws->esz,
                            (size_t)ws->n * ws->esz, (size_t)ncols, hipMemcpyHostToDevice, ws->ctx->stream));
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_col_fill_uniform(ks_workspace* ws, int j, uint64_t seed) {
  return guarded([&] {
    check_col(ws, j);
    ws->ctx->use();
    prov_drop(ws);  // the caller writes to V: the factorisation is no longer the library's own
    materialize(ws);
    const int gb = (int)std::min<int64_t>((ws->ld + kBlock - 1) / kBlock, 8192);
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using D = typename DevT<decltype(tag)>::type;
      ksd::k_fill_uniform<D><<<gb, kBlock, 0, ws->ctx->stream>>>(static_cast<D*>(ws->col(j)), ws->n, ws->ld, seed,
                                                                (uint64_t)ws->row_begin);
    });
    KS_HIP(hipGetLastError());
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_col_norm(ks_workspace* ws, int j, double* out) {
  return guarded([&] {
    check_col(ws, j);
    KS_REQUIRE(out, KS_ERR_ARGUMENT, "null out");
    ws->ctx->use();
    materialize(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) { *out = col_norm<typename DevT<decltype(tag)>::type>(ws, j); });
  });
}

int ks_col_div(ks_workspace* ws, int j, double s) {
  return guarded([&] {
    check_col(ws, j);
    ws->ctx->use();
    prov_drop(ws);  // the caller writes to V: the factorisation is no longer the library's own
    materialize(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) { col_scale<typename DevT<decltype(tag)>::type>(ws, j, 1.0 / s); });
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_col_copy(ks_workspace* ws, int dst, int src) {
  return guarded([&] {
    check_col(ws, dst);
    check_col(ws, src);
    prov_drop(ws);
    // lazy-aware: a lazily normalised source is scaled on the way, nothing else is touched
    dispatch_dtype(ws->dtype, [&](auto tag) { col_copy_lazy<typename DevT<decltype(tag)>::type>(ws, dst, src); });
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_apply(ks_operator* A, ks_workspace* ws, int jsrc, int jdst) {
  return guarded([&] {
    KS_REQUIRE(A, KS_ERR_ARGUMENT, "null operator");
    check_col(ws, jsrc);
    check_col(ws, jdst);
    KS_REQUIRE(jsrc != jdst, KS_ERR_ARGUMENT, "source and destination columns must differ");
    KS_REQUIRE(A->n_local == ws->n && A->dtype == ws->dtype, KS_ERR_DIMENSION, "operator / workspace mismatch");
    ws->ctx->use();
    prov_drop(ws);  // the caller writes to V: the factorisation is no longer the library's own
    materialize(ws);
    A->in_scale = 1.0;
    A->apply(ws->col(jsrc), ws->col(jdst), nullptr);
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_gemv_t(ks_workspace* ws, int j, int jv, void* h_host) {
  return guarded([&] {
    check_col(ws, jv);
    KS_REQUIRE(j >= 1 && j <= ws->maxdim + 1 && h_host, KS_ERR_ARGUMENT, "bad arguments");
    ws->ctx->use();
    materialize(ws);
    reset_state(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) { gemv_t<typename DevT<decltype(tag)>::type>(ws, j, jv, h_host); });
  });
}

int ks_gemv_n_sub(ks_workspace* ws, int j, int jv, const void* h_host) {
  return guarded([&] {
    check_col(ws, jv);
    KS_REQUIRE(j >= 1 && j <= ws->maxdim + 1 && h_host, KS_ERR_ARGUMENT, "bad arguments");
    KS_REQUIRE(jv >= j, KS_ERR_ARGUMENT, "the updated column must not be one of the projected-out columns");
    ws->ctx->use();
    prov_drop(ws);  // the caller writes to V: the factorisation is no longer the library's own
    materialize(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) { gemv_n_sub<typename DevT<decltype(tag)>::type>(ws, j, jv, h_host); });
  });
}

int ks_rotate(ks_workspace* ws, int c0, int c, int r, const void* Q_host, int ldq) {
  return guarded([&] {
    KS_REQUIRE(ws && Q_host, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(c0 >= 0 && c >= 1 && r >= 1 && r <= c && c0 + c <= ws->maxdim + 1 && ldq >= c, KS_ERR_ARGUMENT,
               "bad rotation shape");
    KS_REQUIRE((size_t)c * r <= (size_t)ws->maxdim * ws->maxdim, KS_ERR_ARGUMENT, "rotation larger than the workspace Q");
    prov_drop(ws);
    // lazily normalised columns inside the rotated range are absorbed into Q (no scaling passes), see rotate_lazy
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      rotate_lazy<T>(ws, c0, c, r, static_cast<const T*>(Q_host), ldq);
    });
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  });
}

int ks_basis_times(ks_workspace* ws, int c, int r, const void* Y_host, int ldy, int ydtype, void* out_host,
                   int64_t ldout) {
  return guarded([&] {
    KS_REQUIRE(ws && Y_host && out_host, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(c >= 1 && r >= 1 && c <= ws->maxdim + 1 && ldy >= c && ldout >= ws->n, KS_ERR_ARGUMENT, "bad shape");
    KS_REQUIRE(ydtype == KS_C64 || ydtype == ws->dtype, KS_ERR_ARGUMENT, "coefficient dtype must be complex or match the basis");
    ws->ctx->use();
    materialize(ws);
    hipStream_t s = ws->ctx->stream;
    const size_t yes = ydtype == KS_F64 ? 8 : 16;
    const size_t smem = (size_t)c * r * yes;
    // coefficients -> device (contiguous, ld = c)
    std::vector<char> yc(smem);
    for (int jj = 0; jj < r; ++jj)
      std::memcpy(yc.data() + (size_t)jj * c * yes, static_cast<const char*>(Y_host) + (size_t)jj * ldy * yes, (size_t)c * yes);
    void* yd = ws->ensure_tmp2(smem);
    KS_HIP(hipMemcpyAsync(yd, yc.data(), smem, hipMemcpyHostToDevice, s));
    void* out = ws->ensure_tmp((size_t)ws->ld * r * yes);
    if (ws->dtype == KS_F64 && ydtype == KS_F64)
      gemm_tall_chunked<double, double>(ws, (const double*)ws->V, c, r, (const double*)yd, c, (double*)out, ws->ld);
    else if (ws->dtype == KS_F64)
      gemm_tall_chunked<double, cd>(ws, (const double*)ws->V, c, r, (const cd*)yd, c, (cd*)out, ws->ld);
    else
      gemm_tall_chunked<cd, cd>(ws, (const cd*)ws->V, c, r, (const cd*)yd, c, (cd*)out, ws->ld);
    if (ws->n > 0)
      KS_HIP(hipMemcpy2DAsync(out_host, (size_t)ldout * yes, out, (size_t)ws->ld * yes, (size_t)ws->n * yes, (size_t)r,
                              hipMemcpyDeviceToHost, s));
    KS_HIP(hipStreamSynchronize(s));
  });
}

// ---- fused hot path --------------------------------------------------------------------------------
int ks_orthogonalize(ks_workspace* ws, int j, int* ok) {
  return guarded([&] {
    check_col(ws, j);
    KS_REQUIRE(j >= 1, KS_ERR_ARGUMENT, "orthogonalize needs j >= 1");
    ws->ctx->use();
    prov_drop(ws);  // (a column the caller put there)
    reset_state(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      using D = typename DevT<T>::type;
      materialize(ws);
      const bool lazy = use_deferred(ws, j);
      if (lazy) enqueue_steps_deferred<D>(ws, nullptr, j, j);
      else enqueue_orthogonalize<D>(ws, j);
      fetch_state(ws, j);
      ks::Mat<T> H(static_cast<T*>(ws->H), ws->maxdim + 1, ws->maxdim, ws->maxdim + 1);
      fetch_H_columns<T>(ws, j, j, H, lazy && ws->st_h->breakdown < 0);
      materialize(ws);
    });
    if (ok) *ok = ws->st_h->breakdown < 0;
  });
}

int ks_reinitialize(ks_workspace* ws, int j, const void* v1_host, int* ok) {
  return guarded([&] {
    check_col(ws, j);
    ws->ctx->use();
    materialize(ws);
    bool good = true;
    dispatch_dtype(ws->dtype, [&](auto tag) { good = reinit_column<typename DevT<decltype(tag)>::type>(ws, j, v1_host); });
    // column j is new: a factorisation of zero steps (j = 0) is the library's own; otherwise steps < j keep their status
    if (j == 0 && good) prov_set(ws, 0);
    else if (ws->prov_k > j - 1) ws->prov_k = j - 1;
    if (ok) *ok = good ? 1 : 0;
  });
}

int ks_iterate_arnoldi(ks_operator* A, ks_workspace* ws, int from, int to, ks_expand_stats* stats) {
  return guarded([&] {
    KS_REQUIRE(A && ws, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(A->n_local == ws->n && A->dtype == ws->dtype, KS_ERR_DIMENSION, "operator / workspace mismatch");
    KS_REQUIRE(from >= 1 && to <= ws->maxdim, KS_ERR_ARGUMENT, "step range out of bounds");
    ws->ctx->use();
    ks::ExpandStats st;
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      HipBackend<T> be(A, ws);
      ks::Mat<T> H(static_cast<T*>(ws->H), ws->maxdim + 1, ws->maxdim, ws->maxdim + 1);
      be.iterate_arnoldi(from, to, H, st);
    });
    if (stats) {
      stats->steps = st.steps;
      stats->reorth = st.reorth;
      stats->breakdowns = st.breakdowns;
      stats->explicit_steps = st.explicit_steps;
    }
  });
}

// ---- driver ------------------------------------------------------------------------------------------
int ks_params_default(int64_t n, ks_params* p) {
  return guarded([&] {
    KS_REQUIRE(p, KS_ERR_ARGUMENT, "null params");
    p->nev = (int32_t)std::min<int64_t>(6, n);                              // src/run.jl:103
    p->which = KS_LM;                                                        // :104
    p->tol = std::sqrt(ks::kEps);                                            // :105
    p->mindim = (int32_t)std::min<int64_t>(std::max(10, p->nev), n);         // :106
    p->maxdim = (int32_t)std::min<int64_t>(std::max(20, 2 * p->nev), n);     // :107
    p->restarts = 200;                                                       // :108
    p->start_from = 1;
    p->initialize = 1;
    p->reserved = 0;
  });
}

int ks_partialschur(ks_operator* A, ks_workspace* ws, const ks_params* p, const void* v1_host,
                    double* eigenvalues_c64, ks_history* history) {
  return guarded([&] {
    KS_REQUIRE(A && ws && p, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(A->n_local == ws->n && A->dtype == ws->dtype, KS_ERR_DIMENSION, "operator / workspace mismatch");
    ks::Params prm{p->nev, p->which, p->tol, p->mindim, p->maxdim, p->restarts, p->start_from, p->initialize};
    std::string msg;
    if (ks::check_params(ws->n_global, ws->maxdim + 1, prm, msg)) throw KsError{KS_ERR_ARGUMENT, msg};
    ws->ctx->use();
    materialize(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      const int ldh = ws->maxdim + 1;
      T* Hh = static_cast<T*>(ws->H);
      // fill!(view(H, :, start_from:end), 0)   src/run.jl:176
      for (int j = prm.start_from - 1; j < ws->maxdim; ++j)
        for (int i = 0; i < ldh; ++i) Hh[i + (size_t)j * ldh] = T(0);
      ks::Mat<T> H(Hh, prm.maxdim + 1, prm.maxdim, ldh);
      ks::Mat<T> Q(static_cast<T*>(ws->Q), prm.maxdim, prm.maxdim, ws->maxdim);
      HipBackend<T> be(A, ws);
      if (prm.initialize) be.reinitialize(prm.start_from - 1, prm.start_from == 1 ? static_cast<const T*>(v1_host) : nullptr);
      // partialschur! trusts the workspace it is handed (src/run.jl:152-179: V[:, 1:start_from-1] and H hold a partial
      // Schur decomposition, the start column is in place): so does the provenance from here on
      prov_set(ws, prm.start_from - 1);
      std::vector<cplx> lams(prm.maxdim);
      ks::History h = ks::partialschur_driver<T>(be, H, Q, prm, prm.start_from - 1, lams.data());
      materialize(ws);
      KS_HIP(hipStreamSynchronize(ws->ctx->stream));
      prov_drop(ws);  // (a partial Schur decomposition now, not an Arnoldi factorisation of maxdim steps)
      if (eigenvalues_c64)
        for (int i = 0; i < h.nconverged; ++i) {
          eigenvalues_c64[2 * i] = lams[i].real();
          eigenvalues_c64[2 * i + 1] = lams[i].imag();
        }
      if (history) {
        history->mvproducts = h.mvproducts;
        history->nconverged = h.nconverged;
        history->converged = h.converged;
        history->nev = h.nev;
        history->restarts = h.restarts;
        history->reorth = h.reorth;
        history->breakdowns = h.breakdowns;
        history->explicit_steps = h.explicit_steps;
        history->seconds_expand = h.seconds_expand;
        history->seconds_host = h.seconds_host;
        history->seconds_rotate = h.seconds_rotate;
      }
    });
  });
}

int ks_restart(ks_workspace* ws, const ks_params* p, int active, int* k_out, int* nlock_out, int* purge_out,
               double* lams_c64, double* rs, int32_t* groups) {
  return guarded([&] {
    KS_REQUIRE(ws && p, KS_ERR_ARGUMENT, "null argument");
    ks::Params prm{p->nev, p->which, p->tol, p->mindim, p->maxdim, p->restarts, 1, 0};
    std::string msg;
    if (ks::check_params(ws->n_global, ws->maxdim + 1, prm, msg)) throw KsError{KS_ERR_ARGUMENT, msg};
    KS_REQUIRE(active >= 0 && active < prm.maxdim, KS_ERR_ARGUMENT, "active out of range");
    ws->ctx->use();
    // the restart keeps the library's provenance only if it works on the factorisation the library left (all maxdim steps,
    // H untouched since)
    if (!(ws->prov_k == prm.maxdim && prov_ok(ws, prm.maxdim + 1))) prov_drop(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      ks::Mat<T> H(static_cast<T*>(ws->H), prm.maxdim + 1, prm.maxdim, ws->maxdim + 1);
      ks::Mat<T> Q(static_cast<T*>(ws->Q), prm.maxdim, prm.maxdim, ws->maxdim);
      ks::RestartScratch<T> sc(prm.maxdim);
      const ks::RestartResult r =
          ks::restart_host_step(H, Q, prm.maxdim, prm.mindim, prm.nev, ks::Ordering{prm.which}, prm.tol, active, sc);
      HipBackend<T> be(nullptr, ws);
      be.rotate_and_move(r.purge, prm.maxdim - r.purge, r.k - r.purge, Q, r.k, prm.maxdim);  // src/run.jl:363-365
      if (k_out) *k_out = r.k;
      if (nlock_out) *nlock_out = r.nlock;
      if (purge_out) *purge_out = r.purge;
      for (int i = 0; i < prm.maxdim; ++i) {
        if (lams_c64) { lams_c64[2 * i] = sc.lams[i].real(); lams_c64[2 * i + 1] = sc.lams[i].imag(); }
        if (rs) rs[i] = sc.rs[i];
        if (groups) groups[i] = sc.groups[i];
      }
    });
  });
}

int ks_expand_restart(ks_operator* A, ks_workspace* ws, const ks_params* p, int active, int k_in, int* k_out, int* nlock_out,
                      int* purge_out, double* lams_c64, double* rs, int32_t* groups, ks_expand_stats* stats, double* seconds) {
  return guarded([&] {
    KS_REQUIRE(A && ws && p, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(A->n_local == ws->n && A->dtype == ws->dtype, KS_ERR_DIMENSION, "operator / workspace mismatch");
    ks::Params prm{p->nev, p->which, p->tol, p->mindim, p->maxdim, p->restarts, 1, 0};
    std::string msg;
    if (ks::check_params(ws->n_global, ws->maxdim + 1, prm, msg)) throw KsError{KS_ERR_ARGUMENT, msg};
    KS_REQUIRE(active >= 0 && active < prm.maxdim, KS_ERR_ARGUMENT, "active out of range");
    KS_REQUIRE(k_in >= 1 && k_in < prm.maxdim, KS_ERR_ARGUMENT, "k_in out of range");
    ws->ctx->use();
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      ks::Mat<T> H(static_cast<T*>(ws->H), prm.maxdim + 1, prm.maxdim, ws->maxdim + 1);
      ks::Mat<T> Q(static_cast<T*>(ws->Q), prm.maxdim, prm.maxdim, ws->maxdim);
      ks::RestartScratch<T> sc(prm.maxdim);
      const ks::Ordering ordering{prm.which};
      HipBackend<T> be(A, ws);
      ks::ExpandStats st;
      double t0 = ks::now_s();
      const bool early_done = be.iterate_arnoldi_early(k_in + 1, prm.maxdim, H, st,
                                                       [&] { ks::restart_host_early(H, Q, prm.maxdim, ordering, active, sc); });
      double t1 = ks::now_s();
      if (!early_done) ks::restart_host_early(H, Q, prm.maxdim, ordering, active, sc);
      const ks::RestartResult r = ks::restart_host_late(H, Q, prm.maxdim, prm.mindim, prm.nev, prm.tol, active, sc);
      double t2 = ks::now_s();
      be.rotate_and_move(r.purge, prm.maxdim - r.purge, r.k - r.purge, Q, r.k, prm.maxdim);  // src/run.jl:363-365
      double t3 = ks::now_s();
      if (k_out) *k_out = r.k;
      if (nlock_out) *nlock_out = r.nlock;
      if (purge_out) *purge_out = r.purge;
      for (int i = 0; i < prm.maxdim; ++i) {
        if (lams_c64) { lams_c64[2 * i] = sc.lams[i].real(); lams_c64[2 * i + 1] = sc.lams[i].imag(); }
        if (rs) rs[i] = sc.rs[i];
        if (groups) groups[i] = sc.groups[i];
      }
      if (stats) { stats->steps = st.steps; stats->reorth = st.reorth; stats->breakdowns = st.breakdowns; stats->explicit_steps = st.explicit_steps; }
      if (seconds) { seconds[0] = t1 - t0; seconds[1] = t2 - t1; seconds[2] = t3 - t2; }
    });
  });
}

int ks_profile_enable(ks_ctx* ctx, int on) {
  return guarded([&] {
    KS_REQUIRE(ctx, KS_ERR_ARGUMENT, "null ctx");
    ctx->use();
    KS_HIP(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    ctx->profiling = on != 0;
  });
}

int ks_profile_reset(ks_ctx* ctx) {
  return guarded([&] {
    KS_REQUIRE(ctx, KS_ERR_ARGUMENT, "null ctx");
    ctx->use();
    KS_HIP(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    for (int i = 0; i < KSP_NCLASS; ++i) { ctx->prof_ms[i] = 0; ctx->prof_bytes[i] = 0; ctx->prof_count[i] = 0; }
  });
}

int ks_profile_get(ks_ctx* ctx, int nclass, double* ms, double* bytes, int64_t* counts) {
  return guarded([&] {
    KS_REQUIRE(ctx && ms && bytes && counts, KS_ERR_ARGUMENT, "null argument");
    ctx->use();
    KS_HIP(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    for (int i = 0; i < nclass && i < KSP_NCLASS; ++i) { ms[i] = ctx->prof_ms[i]; bytes[i] = ctx->prof_bytes[i]; counts[i] = ctx->prof_count[i]; }
  });
}

// ---- on-device residual checks -------------------------------------------------------------------------
int ks_residual_norms(ks_operator* A, ks_workspace* ws, int ncols, double* resid, double* orth) {
  return guarded([&] {
    KS_REQUIRE(A && ws && resid && orth, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(ncols >= 0 && ncols <= ws->maxdim, KS_ERR_ARGUMENT, "bad ncols");
    ws->ctx->use();
    materialize(ws);
    *resid = 0.0;
    *orth = 0.0;
    if (ncols == 0) return;
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      relation_norms<T>(A, ws, ncols, ncols, static_cast<const T*>(ws->H), ws->maxdim + 1, resid, orth);
    });
  });
}

int ks_arnoldi_relation(ks_operator* A, ks_workspace* ws, int k, double* resid, double* orth) {
  return guarded([&] {
    KS_REQUIRE(A && ws && resid && orth, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(k >= 1 && k <= ws->maxdim, KS_ERR_ARGUMENT, "bad k");
    ws->ctx->use();
    materialize(ws);
    dispatch_dtype(ws->dtype, [&](auto tag) {
      using T = decltype(tag);
      relation_norms<T>(A, ws, k, k + 1, static_cast<const T*>(ws->H), ws->maxdim + 1, resid, orth);
    });
  });
}

// ---- host small dense exports (no device needed) ----------------------------------------------------------
int ks_host_schurfact(int dtype, void* H, int m, int n, int ldh, int start, int to, void* Q, int nq, int ldq) {
  return guarded([&] {
    KS_REQUIRE(H, KS_ERR_ARGUMENT, "null H");
    dispatch_dtype(dtype, [&](auto tag) {
      using T = decltype(tag);
      ks::Mat<T> Hm(static_cast<T*>(H), m, n, ldh), Qm(static_cast<T*>(Q), nq, nq, ldq);
      const bool ok = ks::local_schurfact(Hm, start, to, Qm);
      if (!ok) throw KsError{KS_ERR_QR, "QR algorithm did not converge"};
    });
  });
}

int ks_host_restart_step(int dtype, void* H, int ldh, void* Q, int ldq, int maxdim, int mindim, int nev, int which,
                         double tol, int active, int* k, int* nlock, int* purge, double* lams_c64, double* rs,
                         int32_t* groups) {
  return guarded([&] {
    KS_REQUIRE(H && Q, KS_ERR_ARGUMENT, "null argument");
    dispatch_dtype(dtype, [&](auto tag) {
      using T = decltype(tag);
      ks::Mat<T> Hm(static_cast<T*>(H), maxdim + 1, maxdim, ldh), Qm(static_cast<T*>(Q), maxdim, maxdim, ldq);
      ks::RestartScratch<T> sc(maxdim);
      const ks::RestartResult r = ks::restart_host_step(Hm, Qm, maxdim, mindim, nev, ks::Ordering{which}, tol, active, sc);
      if (k) *k = r.k;
      if (nlock) *nlock = r.nlock;
      if (purge) *purge = r.purge;
      for (int i = 0; i < maxdim; ++i) {
        if (lams_c64) { lams_c64[2 * i] = sc.lams[i].real(); lams_c64[2 * i + 1] = sc.lams[i].imag(); }
        if (rs) rs[i] = sc.rs[i];
        if (groups) groups[i] = sc.groups[i];
      }
    });
  });
}

int ks_host_sortschur(int dtype, void* H, int m, int n, int ldh, void* Q, int nq, int ldq, int nconv, int which) {
  return guarded([&] {
    KS_REQUIRE(H, KS_ERR_ARGUMENT, "null H");
    dispatch_dtype(dtype, [&](auto tag) {
      using T = decltype(tag);
      ks::Mat<T> Hm(static_cast<T*>(H), m, n, ldh), Qm(static_cast<T*>(Q), nq, nq, ldq);
      ks::sortschur(Hm, Qm, nconv, ks::Ordering{which});
    });
  });
}

int ks_host_givens(int dtype, const double* f, const double* g, double* c, double* s, double* r) {
  return guarded([&] {
    if (dtype == KS_F64) {
      ks::givens(f[0], g[0], *c, s[0], r[0]);
      s[1] = 0.0; r[1] = 0.0;
    } else {
      cplx sn, rr;
      ks::givens(cplx(f[0], f[1]), cplx(g[0], g[1]), *c, sn, rr);
      s[0] = sn.real(); s[1] = sn.imag(); r[0] = rr.real(); r[1] = rr.imag();
    }
  });
}

}  // extern "C"
