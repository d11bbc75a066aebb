// errors, the context object (device, stream, RCCL communicator / peer-to-peer region / host-staged transport), per-kernel-class profiling
// Part of the ONE translation unit of libkschur_hip.so: included by ks_hip.hip, in this order --
//     ks_context.hpp -> ks_operators.hpp -> ks_workspace.hpp -> ks_backend.hpp -> (C ABI in ks_hip.hip)
// -- and not meant to be included on its own (needs the includes and using-declarations at the top of ks_hip.hip).
#pragma once

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
  // device-reported failure of an operator kernel (ks_sptrsv.hpp: a bounded spin gave up); pinned + mapped, made on demand
  int* operr_h = nullptr;
  int* operr_dev() {
    if (!operr_h) {
      KS_HIP(hipHostMalloc(&operr_h, sizeof(int), hipHostMallocMapped));
      *operr_h = 0;
    }
    int* d = nullptr;
    KS_HIP(hipHostGetDevicePointer((void**)&d, operr_h, 0));
    return d;
  }
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
    if (operr_h && *operr_h != 0) {
      *operr_h = 0;  // reported once: the operator's next product starts clean (its vectors of THIS product are garbage)
      throw KsError{KS_ERR_OPERATOR, "sparse triangular solve gave up waiting for a solution entry (KS_LU_TIMEOUT_S): malformed factor or a stalled device"};
    }
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

