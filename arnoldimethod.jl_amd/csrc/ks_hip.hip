// libkschur_hip.so -- host side of the MI355X Krylov-Schur hot path: context / operator / workspace
// objects, the HIP backend of the driver (async expansion, in-place MFMA rotation), RCCL plumbing and
// the extern "C" entry points declared in include/kschur.h.
//
// There is NO CPU fallback in this library: without a gfx950 device every compute entry point
// returns KS_ERR_NO_DEVICE / KS_ERR_HIP.  Only the ks_host_* small-dense exports run without a GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/kschur.h"
#include "ks_driver.hpp"
#include "ks_kernels.hpp"
#include "ks_spmv_march.hpp"   // persistent form of the paired stencil SpMV
#include "ks_block_kernels.hpp"  // kernels of the s-step (block) expansion (the streaming ones are instantiated in ks_block_inst.hip)
#include "ks_block_launch.hpp"

using ks::cplx;
using ksd::cd;
using ksd::DevState;
using ksd::kBlock;

#include "ks_context.hpp"    // errors, ks_ctx, transports
#include "ks_operators.hpp"  // ks_operator and its layouts
#include "ks_sptrsv.hpp"     // shift-invert operator from triangular factors (sparse triangular solves)
#include "ks_workspace.hpp"  // ks_workspace, launch helpers, expansion, rotations
#include "ks_block.hpp"      // s-step (block) expansion: launchers, shifts, block sizes
#include "ks_backend.hpp"    // HipBackend, residual checks, placement search

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
    if (ctx->operr_h) (void)hipHostFree(ctx->operr_h);
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
      // ghost slots owned by lower ranks (they precede the local columns in the global order: column blocks keep the
      // single-GPU summation order); -1: the neighbour list is not ascending by rank with consecutive slots -> no column blocks
      int64_t nlow = 0;
      {
        bool above = false, ok = true;
        for (int p = 0; p < nneigh; ++p) {
          KS_REQUIRE(recv_cnt[p] >= 0, KS_ERR_ARGUMENT, "negative receive count");
          if (neigh[p] < ctx->rank) { if (above) ok = false; nlow += recv_cnt[p]; }
          else above = true;
        }
        if (!ok) nlow = -1;
      }
      CsrOp<D>* op = make_csr<D>(ctx, nrows_local, nnz, rp, ci, vv, nlow >= 0 ? 3 : 0, nghost, std::max<int64_t>(nlow, 0));
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

int ks_operator_lu(ks_ctx* ctx, int64_t n, int dtype, const int64_t* l_rowptr, const int32_t* l_colind, const void* l_val,
                   const int64_t* u_rowptr, const int32_t* u_colind, const void* u_val, const int32_t* perm_in,
                   const int32_t* perm_out, const double* scale, ks_operator** out) {
  return guarded([&] {
    KS_REQUIRE(ctx && out && l_rowptr && u_rowptr, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(dtype == KS_F64 || dtype == KS_C64, KS_ERR_ARGUMENT, "unknown dtype");
    KS_REQUIRE(n >= 1 && n < (int64_t)2147483647, KS_ERR_ARGUMENT, "ks_operator_lu: n must be in [1, 2^31)");
    KS_REQUIRE(!ctx->distributed(), KS_ERR_ARGUMENT, "ks_operator_lu: single-GPU contexts only (a triangular solve does not shard by rows)");
    KS_REQUIRE((l_rowptr[n] == 0 || (l_colind && l_val)) && u_colind && u_val, KS_ERR_ARGUMENT, "null factor arrays");
    ctx->use();
    *out = dtype == KS_F64 ? make_lu<double>(ctx, n, l_rowptr, l_colind, l_val, u_rowptr, u_colind, u_val, perm_in, perm_out, scale)
                           : make_lu<cd>(ctx, n, l_rowptr, l_colind, l_val, u_rowptr, u_colind, u_val, perm_in, perm_out, scale);
  });
}

int ks_operator_lu_info(const ks_operator* op, int64_t* nnz_l, int64_t* nnz_u, int64_t* levels_l, int64_t* levels_u) {
  return guarded([&] {
    KS_REQUIRE(op, KS_ERR_ARGUMENT, "null operator");
    auto fill = [&](auto* lu) {
      if (nnz_l) *nnz_l = lu->L.nnz;
      if (nnz_u) *nnz_u = lu->U.nnz;
      if (levels_l) *levels_l = lu->L.levels;
      if (levels_u) *levels_u = lu->U.levels;
    };
    if (auto* a = dynamic_cast<const LuOp<double>*>(op)) fill(a);
    else if (auto* b = dynamic_cast<const LuOp<cd>*>(op)) fill(b);
    else throw KsError{KS_ERR_ARGUMENT, "ks_operator_lu_info: not an operator made by ks_operator_lu"};
  });
}

int ks_operator_lu_layout(const ks_operator* op, int upper, int64_t* rows, int64_t* run_rows, int64_t* top_rows, int* ngroups) {
  return guarded([&] {
    KS_REQUIRE(op, KS_ERR_ARGUMENT, "null operator");
    auto fill = [&](auto* lu) {
      const auto& f = upper ? lu->U : lu->L;
      if (rows) *rows = f.rows;
      if (run_rows) *run_rows = f.run_rows;
      if (top_rows) *top_rows = f.top_end - f.top_begin;
      if (ngroups) *ngroups = f.ngroups;
    };
    if (auto* a = dynamic_cast<const LuOp<double>*>(op)) fill(a);
    else if (auto* b = dynamic_cast<const LuOp<cd>*>(op)) fill(b);
    else throw KsError{KS_ERR_ARGUMENT, "ks_operator_lu_layout: not an operator made by ks_operator_lu"};
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
    {
      // basis small enough for the 256 MiB memory-side cache: cacheable loads keep it there between kernels (ld_v,
      // ks_kernels.hpp: +2...4 % on BASELINE configs 2-4); KS_V_NT=0/1 forces, KS_V_NT_MB moves the threshold
      const int nt_env = env_int("KS_V_NT", -1);
      w->v_nt = nt_env >= 0 ? nt_env != 0 : vbytes > ((size_t)env_int("KS_V_NT_MB", 352) << 20);
    }
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
    if (env_int("KS_ZS_EARLY", 0)) {
      // experiment (round 6, the out-of-place lottery of k_bupdate_mfma: 0.86 against 0.98 ms for the same launch, fixed per process):
      // the scratch columns of the fused rotation allocated right behind the basis instead of lazily after everything else
      const size_t zbytes = (size_t)w->ld * (dtype == KS_F64 ? ksd::kBlkSMax : 10) * esz;
      if (hipMalloc(&w->zscratch, zbytes) == hipSuccess) KS_HIP(hipMemsetAsync(w->zscratch, 0, zbytes, ctx->stream));
      else { (void)hipGetLastError(); w->zscratch = nullptr; }
    }
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
    KS_HIP(hipHostMalloc(&w->Qstage, qxbytes, coh));   // (coherent + mapped: the rotation gate reads it from the device)
    KS_HIP(hipHostMalloc(&w->Hstage_early, ctl_bytes, coh));
    std::memset(w->Hstage_early, 0, ctl_bytes);
    KS_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->mbox), 128, coh));
    std::memset(w->mbox, 0, 128);
    // reverse mailbox (k_rot_gate): pinned parameter block the gate polls, its device copy, device view of the Q stage
    if (hipHostMalloc(reinterpret_cast<void**>(&w->gate_h), sizeof(ksd::RotGate), coh) == hipSuccess) {
      std::memset(w->gate_h, 0, sizeof(ksd::RotGate));
      if (hipHostGetDevicePointer(reinterpret_cast<void**>(&w->gate_hd), w->gate_h, 0) != hipSuccess ||
          hipHostGetDevicePointer(&w->Qstage_dev, w->Qstage, 0) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&w->gate_d), sizeof(ksd::RotGate)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(w->gate_h);
        w->gate_h = nullptr;
      } else {
        KS_HIP(hipMemset(w->gate_d, 0, sizeof(ksd::RotGate)));
      }
    } else {
      (void)hipGetLastError();
      w->gate_h = nullptr;
    }
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
    w->sstep = std::max(0, std::min(env_int("KS_SSTEP", 20), ksd::kBlkSMax));  // s-step expansion: ON by default (KS_SSTEP=0: step by step)
    w->sstep_eff = w->sstep;
    w->rot_defer_on = env_int("KS_ROT_DEFER", 1) != 0;   // restart rotation left pending for the next expansion's fused first pass
    w->spec_on = env_int("KS_SPEC_CHAIN", 1) != 0;        // first products of the next expansion behind the previous one
    w->defl_on = env_int("KS_CHAIN_DEFLATE", 1) != 0;    // Newton chains deflated against locked columns of dominant eigenvalues
    w->true_start_on = env_int("KS_TRUE_START", 0) != 0;  // chains behind a block with a Gram deviation above rounding level start from S T[:, maxdim]
    if (const char* e = std::getenv("KS_SSTEP_GDEV_MAX")) w->blk_gdevmax = std::atof(e);
    if (const char* e = std::getenv("KS_SSTEP_PIVOT_MIN")) w->blk_pivmin = std::atof(e);
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

int ks_workspace_set_sstep(ks_workspace* ws, int s, double pivot_min, double gram_dev_max) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    KS_REQUIRE(s >= 0 && s <= ksd::kBlkSMax, KS_ERR_ARGUMENT, "block size must be 0 (off) .. 20");
    ws->ctx->use();
    materialize(ws);
    ws->sstep = s;
    ws->sstep_eff = s;
    ws->blk_clean = 0;
    if (!std::isnan(pivot_min)) ws->blk_pivmin = pivot_min;
    if (!std::isnan(gram_dev_max)) ws->blk_gdevmax = gram_dev_max;
  });
}

int ks_workspace_sstep_info(const ks_workspace* ws, int* s, int* blocks, int* abandoned, double* diag3) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    if (s) *s = ws->sstep_eff;
    if (blocks) *blocks = ws->blk_count;
    if (abandoned) *abandoned = ws->blk_bails;
    if (diag3) { diag3[0] = ws->blk_diag[0]; diag3[1] = ws->blk_diag[1]; diag3[2] = ws->blk_diag[2]; }
  });
}

int ks_workspace_relation_info(const ks_workspace* ws, int* breaks, double* worst_leak) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    if (breaks) *breaks = ws->relation_breaks;
    if (worst_leak) *worst_leak = ws->relation_leak;
  });
}

int ks_workspace_relation_probes(const ks_workspace* ws, int* probes) {
  return guarded([&] {
    KS_REQUIRE(ws && probes, KS_ERR_ARGUMENT, "null argument");
    *probes = ws->relation_probes;
  });
}

int ks_workspace_fused_rotations(const ks_workspace* ws, int* count, int* spec_adopted, int* spec_dropped) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    if (count) *count = ws->rot_fused_count;
    if (spec_adopted) *spec_adopted = ws->spec_used;
    if (spec_dropped) *spec_dropped = ws->spec_wasted;
  });
}

int ks_workspace_split_rotations(const ks_workspace* ws, int* count) {
  return guarded([&] {
    KS_REQUIRE(ws && count, KS_ERR_ARGUMENT, "null argument");
    *count = ws->rot_split_count;
  });
}

int ks_workspace_deflated_blocks(const ks_workspace* ws, int* blocks, int* columns) {
  return guarded([&] {
    KS_REQUIRE(ws != nullptr, KS_ERR_ARGUMENT, "null argument");
    if (blocks) *blocks = ws->defl_blocks;
    if (columns) *columns = ws->defl_last;
  });
}

int ks_sstep_partition(int dtype, int k0, int count, int smax, int* out, int cap, int* nblocks) {
  return guarded([&] {
    KS_REQUIRE(nblocks && (out || cap == 0), KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(dtype == KS_F64 || dtype == KS_C64, KS_ERR_ARGUMENT, "dtype");
    *nblocks = 0;
    if (k0 < 1 || count < 1 || smax < 1) return;
    const auto sizes = blk_partition(dtype, k0, count, std::min(smax, dtype == KS_F64 ? ksd::kBlkSMax : ksd::blk_smax<cd>()));
    *nblocks = (int)sizes.size();
    for (int i = 0; i < (int)sizes.size() && i < cap; ++i) out[i] = sizes[i];
  });
}

// diagnostics: average duration of `reps` launches of one block kernel on the workspace's basis (contents irrelevant: the
// kernels have no data-dependent control flow); which = 0 k_bdots, 1 k_bupdate.  Leaves columns k..k+s-1 overwritten.
int ks_debug_blk_time(ks_workspace* ws, int k, int s, int which, int reps, int dbg, double* ms_per_launch, int* grid) {
  return guarded([&] {
    KS_REQUIRE(ws && ms_per_launch, KS_ERR_ARGUMENT, "null argument");
    KS_REQUIRE(blk_shape_ok(ws->dtype, k, s) && k + s <= ws->maxdim + 1, KS_ERR_ARGUMENT, "no block kernel for this shape");
    KS_REQUIRE(reps >= 1 && (which == 0 || which == 1), KS_ERR_ARGUMENT, "bad reps / which");
    ws->ctx->use();
    materialize(ws);
    prov_drop(ws);
    blk_ensure_buffers(ws);
    reset_state(ws);
    // (probe flags of this entry point itself: 256 = the second pass reads the block from the SCRATCH columns, as behind a fused
    // rotation; 512 = ... after giving the scratch columns a fresh allocation -- the new one is made before the old one is freed, so
    // that the allocator cannot hand the same pages back: tools/bupdate_lottery.py)
    const bool zs = (dbg & 256) != 0, reroll = (dbg & 512) != 0;
    dbg &= 255;
    if (zs) {
      void* old = reroll ? ws->zscratch : nullptr;
      if (reroll) ws->zscratch = nullptr;
      bool ok = false;
      if (ws->dtype == KS_F64) ok = ensure_zscratch<double>(ws);
      else ok = ensure_zscratch<cd>(ws);
      if (old) { KS_HIP(hipStreamSynchronize(ws->ctx->stream)); (void)hipFree(old); }
      KS_REQUIRE(ok, KS_ERR_INTERNAL, "no room for the scratch columns");
    }
    const int saved = blk_dbg();
    blk_dbg() = dbg;
    hipEvent_t a, b;
    KS_HIP(hipEventCreate(&a));
    KS_HIP(hipEventCreate(&b));
    int nb = 0;
    auto run = [&](int n) {
      for (int i = 0; i < n; ++i) {
        if (ws->dtype == KS_F64) nb = launch_blk<double>(ws, which, k, s, zs);
        else nb = launch_blk<cd>(ws, which, k, s, zs);
      }
    };
    run(2);  // warm-up
    KS_HIP(hipEventRecord(a, ws->ctx->stream));
    run(reps);
    KS_HIP(hipEventRecord(b, ws->ctx->stream));
    KS_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    KS_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    blk_dbg() = saved;
    *ms_per_launch = (double)ms / reps;
    if (grid) *grid = nb;
  });
}

int ks_workspace_assert_arnoldi(ks_workspace* ws, int k) {
  return guarded([&] {
    KS_REQUIRE(ws, KS_ERR_ARGUMENT, "null workspace");
    KS_REQUIRE(k >= -1 && k <= ws->maxdim, KS_ERR_ARGUMENT, "k out of range");
    if (k < 0) prov_drop(ws);  // the caller withdraws: the next expansion runs the explicit form
    else { prov_set(ws, k); ws->prov_vouched = true; }
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
    gate_cancel(ws);   // (a synchronisation behind an armed gate would wait for its time-out)
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
      GateScope gate_scope(ws);  // every expansion of the driver's loop is followed by the restart: its rotation may be pre-enqueued
      ws->mindim_hint = prm.mindim;
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
      be.note_ritz(sc.lams.data(), prm.maxdim, r.leak, r.fro, prm.tol);
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
      GateScope gate_scope(ws);
      ws->mindim_hint = prm.mindim;
      ks::ExpandStats st;
      double t0 = ks::now_s();
      const bool early_done = be.iterate_arnoldi_early(k_in + 1, prm.maxdim, H, st,
                                                       [&] { ks::restart_host_early(H, Q, prm.maxdim, ordering, active, sc); });
      double t1 = ks::now_s();
      if (!early_done) ks::restart_host_early(H, Q, prm.maxdim, ordering, active, sc);
      const ks::RestartResult r = ks::restart_host_late(H, Q, prm.maxdim, prm.mindim, prm.nev, prm.tol, active, sc);
      be.note_ritz(sc.lams.data(), prm.maxdim, r.leak, r.fro, prm.tol);
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

// ---- last words (include/kschur.h, diagnostics) -----------------------------------------------------------------------
namespace {
char* g_last_words = nullptr;           // owned; replaced, never freed while a handler may run
volatile size_t g_last_words_len = 0;
volatile int g_last_words_exit = 0;
void last_words_handler(int) {
  // async-signal-safe only: write(2) and _exit(2)
  if (g_last_words && g_last_words_len) {
    ssize_t r = ::write(1, "\n", 1);
    r = ::write(1, g_last_words, g_last_words_len);
    r = ::write(1, "\n", 1);
    (void)r;
  }
  ::_exit(g_last_words_exit);
}
}  // namespace

int ks_last_words(const char* line, int exit_code) {
  return guarded([&] {
    static const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL, SIGTERM};
    if (!line) {
      for (int sg : sigs) ::signal(sg, SIG_DFL);
      g_last_words_len = 0;
      return;
    }
    const size_t n = std::strlen(line);
    char* copy = static_cast<char*>(std::malloc(n + 1));
    KS_REQUIRE(copy != nullptr, KS_ERR_INTERNAL, "out of memory");
    std::memcpy(copy, line, n + 1);
    g_last_words_len = 0;       // (a signal between these stores prints nothing rather than a torn line)
    g_last_words = copy;        // the previous copy is leaked on purpose: a handler may be reading it
    g_last_words_exit = exit_code;
    g_last_words_len = n;
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_handler = last_words_handler;
    sigemptyset(&sa.sa_mask);
    for (int sg : sigs) ::sigaction(sg, &sa, nullptr);
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
