// operators: anything with mul!(y, A, x) (src/run.jl:21-22) -- stored sparse matrices in their device layouts, dense matrices, host / device callbacks
// Part of the ONE translation unit of libkschur_hip.so: included by ks_hip.hip, in this order --
//     ks_context.hpp -> ks_operators.hpp -> ks_workspace.hpp -> ks_backend.hpp -> (C ABI in ks_hip.hip)
// -- and not meant to be included on its own (needs ks_context.hpp).
#pragma once
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
  // Newton products of a LONG block (s-step expansion, ks_block.hpp): the product is stored cacheably instead of streaming --
  // the next product of the chain gathers exactly this vector (49 instead of 59 us at n = 1e7), and the dirty lines cost the
  // ONE inner-product pass that follows a block of 20 less (+45 us) than the twenty products gain (headline 4 332 -> 4 501
  // iterations/s; blocks of 10 with the ring form of pass 1: 3 836 -> 3 923).  With the register form of pass 1 and blocks of
  // 5-8 it was the other way round (products 61 -> 51 us, pass 1 415 -> 510 us), so shorter blocks keep streaming stores.
  // KS_SHIFT_PLAIN = 0 / 1 forces.
  bool shift_store_cacheable = false;
  virtual ~ks_operator() = default;
  // y = A x on device pointers, enqueued on ctx->stream; `st` lets the kernels of a batch skip work
  // after a breakdown.
  virtual void apply(const void* x, void* y, const DevState* st) = 0;
  // y = sigma (A x - theta x): one step of the Newton basis of the s-step expansion (ks_block.hpp).  Default: the product
  // followed by a streaming pass (24 n bytes); layouts with a fused form override it.  theta = (re, im).
  virtual void apply_shifted(const void* x, void* y, double theta_re, double theta_im, double sigma, int64_t ld, const DevState* st);
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
  std::vector<char> cb_from_ghost;  // per column block: gathers from the ghost vector (row block of a distributed operator)
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
  int nstencil_local = 0;            // leading slots of the stencil dictionary; the rest name ghost columns only
  mutable bool split_said = false;   // (KS_DIST_SPLIT_DEBUG: one line per operator)
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
  void launch_csr_blocks(const D* x, D* y, const DevState* st, const D* yacc, int plain_store, ksd::ShiftArg<D> shift = ksd::ShiftArg<D>{}) {
    hipStream_t s = ctx->stream;
    auto go = [&](auto ip_tag) {
      using IP = decltype(ip_tag);
      auto launch = [&](auto ni_tag) {
        constexpr int NI = decltype(ni_tag)::value;
        if constexpr ((size_t)NI * kBlock * sizeof(D) <= (size_t)ksd::kSpmvCapBytes)
          ksd::k_spmv_csr<D, IP, false, NI><<<nblk, kBlock, 0, s>>>(static_cast<const IP*>(blkptr), blkrow, static_cast<const IP*>(rowptr), colidx, val, x,
                                                                    nullptr, y, n_local, nblk, st, nullptr, 0, 0, nullptr, nullptr, yacc, plain_store,
                                                                    ksd::HaloFused{}, ksd::HaloArgs{}, ksd::P2pDev{}, env_int("KS_SPMV_CSR_NT", 1) != 0, row_gather,
                                                                    plain_store ? ksd::ShiftArg<D>{} : shift);
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
  // fused Newton-basis step (s-step expansion): the paired stencil kernel applies y = sigma (A x - theta x) itself
  bool shift_on = false;
  D shift_theta{};
  double shift_sigma = 1.0;
  bool shift_plain() const {
    static const int env = env_int("KS_SHIFT_PLAIN", -1);
    return env >= 0 ? env != 0 : shift_store_cacheable;
  }
  // ---- marching form of the paired stencil kernel (ks_spmv_march.hpp; Float64, one GPU, at most 8 slots) ----
  static bool march_on() {
    return env_int("KS_STENCIL_MARCH", 1) != 0;   // (read per launch: the tests switch the forms inside one process)
  }
  // Workgroups per XCD (= tiles of 512 rows an XCD takes per round).  With a far stride P (the xy-plane of a 3-D grid) a round
  // that covers a WHOLE NUMBER OF PLANES keeps the z - 1 / z + 1 taps of a round in step with the rows other rounds own.  How much
  // that buys depends on where x comes from (tools/spmv_slab.hip, profiles/r06_spmv_sweep.txt, r06_spmv_columns.txt, 216^3):
  //   x and y ping-pong between two buffers that stay in the 256-MB Infinity Cache:   34.0 us at 92 (one plane), 40 at 184-192
  //     (two planes), 41-45 in between and beyond (k_spmv_stencil2: 43-44);
  //   the solver's chain (column i -> column i + 1 of a basis that does not fit the cache):   48.7 at 192, 50.5 at 92, 49-55 for
  //     k_spmv_stencil2 -- a plain copy of the same columns takes 29-31 there, the same kernel with ONE slot 36.
  // The solver lives in the second regime: two planes per round (5-6 workgroups per CU), a multiple of one plane in general.
  int march_slots(int ntiles) const {
    const int env = env_int("KS_MARCH_S", 0);
    int S = 192;
    if (env > 0) S = env;
    else {
      int64_t P = 0;
      for (int k = 0; k < nstencil; ++k) P = std::max<int64_t>(P, std::llabs((long long)sdict.delta[k]));
      const int64_t per = P / 512 + 1;                         // tiles of one far stride, rounded up
      if (per >= 16 && per <= 224) S = (int)(per * ((160 + per - 1) / per)) <= 224 ? (int)(per * ((160 + per - 1) / per)) : (int)per;
    }
    S = std::min(S, std::max(1, (ntiles + 7) / 8));
    return S;
  }
  void launch_march(const D* x, D* y, int nt, const DevState* st, hipStream_t s) const {
    if constexpr (std::is_same<D, double>::value) {
      const int sh = shift_on ? (1 | (shift_plain() ? 2 : 0)) : 0;
      const uint16_t* m2 = static_cast<const uint16_t*>(smask2);
      const int G = 8 * march_slots(nt);
      int kown = -1;
      for (int k = 0; k < nstencil; ++k)
        if (sdict.delta[k] == 0) kown = k;
      auto go = [&](auto ns_tag, auto ko_tag) {
        ksd::k_spmv_stencil_march<decltype(ns_tag)::value, decltype(ko_tag)::value><<<G, kBlock, 0, s>>>(m2, sdict, x, y, n_local, nt, st, sh, shift_theta, shift_sigma);
      };
      // the 3-D 7-point shape (far, near, near, own, near, near, far with the near taps within 256 rows): the window form -- the five
      // near taps of a tile from one copy of its rows in LDS (ks_spmv_march.hpp; 47 against 49 us in the solver's chain,
      // profiles/r06_spmv_columns.txt).  KS_MARCH_WINDOW=0: the register form for every shape.
      const int window = env_int("KS_MARCH_WINDOW", 1);
      if (window && nstencil == 7 && kown == 3) {
        bool shape = std::llabs((long long)sdict.delta[0]) > 256 && std::llabs((long long)sdict.delta[6]) > 256;
        unsigned odd = 0;
        for (int k = 1; k <= 5; ++k) {
          shape = shape && std::llabs((long long)sdict.delta[k]) <= 256;
          if (sdict.delta[k] & 1) odd |= 1u << k;
        }
        // ... and where the planes are large enough to give every XCD a dozen in-plane tiles, the Z-MARCHING form: a workgroup
        // keeps its tile and walks up the planes, the far taps come from registers / the next plane's window (41-42 us against
        // 46-47 for the window form and 50-56 for k_spmv_stencil2 in the solver's chain, profiles/r06_spmv_columns.txt).
        // KS_MARCH_Z=0 switches it off, KS_MARCH_ZR sets the number of z-ranges.
        const int zmarch = env_int("KS_MARCH_Z", 1), zr_env = env_int("KS_MARCH_ZR", 0);
        const int64_t Pz = sdict.delta[6];
        if (shape && zmarch && sdict.delta[0] == -Pz && (Pz & 1) == 0 && Pz >= 8 * 8 * 512) {
          const int nzp = (int)((n_local + Pz - 1) / Pz), cmax = (int)(((Pz + 511) / 512 + 7) / 8);
          int nzr = zr_env > 0 ? zr_env : std::max(1, (288 + cmax / 2) / cmax);
          nzr = std::max(1, std::min(nzr, nzp / 4));
          if (nzp >= 8) {
            auto gz = [&](auto odd_tag) {
              ksd::k_spmv_stencil_marchz<7, 0x3eu, decltype(odd_tag)::value, 3, 0, 6><<<8 * cmax * nzr, kBlock, 0, s>>>(m2, sdict, x, y, n_local, nzr, st, sh, shift_theta, shift_sigma);
            };
            if (odd == 0x14u) gz(std::integral_constant<unsigned, 0x14u>{});
            else if (odd == 0x36u) gz(std::integral_constant<unsigned, 0x36u>{});
            else gz(std::integral_constant<unsigned, 0x3eu>{});
            return;
          }
        }
        if (shape) {
          auto gw = [&](auto odd_tag) {
            ksd::k_spmv_stencil_marchw<7, 0x3eu, decltype(odd_tag)::value, 3><<<G, kBlock, 0, s>>>(m2, sdict, x, y, n_local, nt, st, sh, shift_theta, shift_sigma);
          };
          if (odd == 0x14u) gw(std::integral_constant<unsigned, 0x14u>{});        // nx even: only +-1 are odd
          else if (odd == 0x36u) gw(std::integral_constant<unsigned, 0x36u>{});   // nx odd
          else gw(std::integral_constant<unsigned, 0x3eu>{});                     // anything else: every near tap as two 8-byte reads
          return;
        }
      }
      using M1 = std::integral_constant<int, -1>;
      // the slot of the rows' own entry is a compile-time constant for the common stencils (3-, 5-, 7-point); otherwise one more load
      if (nstencil == 7 && kown == 3) go(std::integral_constant<int, 7>{}, std::integral_constant<int, 3>{});
      else if (nstencil == 5 && kown == 2) go(std::integral_constant<int, 5>{}, std::integral_constant<int, 2>{});
      else if (nstencil == 3 && kown == 1) go(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
      else switch (nstencil) {
        case 1: go(std::integral_constant<int, 1>{}, M1{}); break;
        case 2: go(std::integral_constant<int, 2>{}, M1{}); break;
        case 3: go(std::integral_constant<int, 3>{}, M1{}); break;
        case 4: go(std::integral_constant<int, 4>{}, M1{}); break;
        case 5: go(std::integral_constant<int, 5>{}, M1{}); break;
        case 6: go(std::integral_constant<int, 6>{}, M1{}); break;
        case 7: go(std::integral_constant<int, 7>{}, M1{}); break;
        default: go(std::integral_constant<int, 8>{}, M1{}); break;
      }
    }
  }
  ksd::ShiftArg<D> shift_arg() const {
    ksd::ShiftArg<D> a;
    a.on = shift_on ? 1 : 0;
    a.theta = shift_theta;
    a.sigma = shift_sigma;
    return a;
  }
  void apply_shifted(const void* xv, void* yv, double tre, double tim, double sigma, int64_t ld, const DevState* st) override {
    static const int fuse = env_int("KS_SHIFT_FUSED", 1);
    // (every stored-matrix layout stores y = sigma (A x - theta x) itself -- the row's own x entry is one more cached load
    // -- except rows cut into chunks, whose sums are finished by a second kernel)
    const bool fused = fuse && n_local > 0 && nlong == 0;
    if (!fused) { ks_operator::apply_shifted(xv, yv, tre, tim, sigma, ld, st); return; }
    shift_on = true;
    if constexpr (sizeof(D) == 8) shift_theta = tre; else shift_theta = D{tre, tim};
    shift_sigma = sigma;
    try { apply(xv, yv, st); } catch (...) { shift_on = false; throw; }
    shift_on = false;
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
          a.xb[b] = (!cb_from_ghost.empty() && cb_from_ghost[b]) ? xg : nullptr;
        }
        const int nt = (int)((n_local + (int64_t)kBlock * cb_rpt - 1) / ((int64_t)kBlock * cb_rpt));
        auto go = [&](auto ni_tag, auto rpt_tag) {
          constexpr int NI = decltype(ni_tag)::value, RPT = decltype(rpt_tag)::value;
          if constexpr ((size_t)NI * kBlock * sizeof(D) <= (size_t)ksd::kSpmvCapBytes)
            ksd::k_spmv_csr_cb<D, NI, RPT><<<nt, kBlock, 0, s>>>(a, x, y, n_local, nt, st, shift_arg());
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
      for (size_t b = 0; b < cblocks.size(); ++b) cblocks[b]->launch_csr_blocks(x, y, st, b > 0 ? y : nullptr, b + 1 < cblocks.size() ? 1 : 0, shift_arg());
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
        if constexpr (std::is_same<D, double>::value) {
          // Float64, at most 8 slots: the persistent software-pipelined form (ks_spmv_march.hpp)
          if (stencil_mask_bytes == 1 && nstencil <= 8 && march_on()) {
            launch_march(x, y, nt, st, s);
            KS_HIP(hipGetLastError());
            return;
          }
        }
        if (stencil_mask_bytes == 1)
          ksd::k_spmv_stencil2<D, uint16_t><<<nt, kBlock, 0, s>>>(static_cast<const uint16_t*>(smask2), sdict, nstencil, x, y, n_local, nt, st, shift_on ? (1 | (shift_plain() ? 2 : 0)) : 0, shift_theta, shift_sigma);
        else
          ksd::k_spmv_stencil2<D, uint64_t><<<nt, kBlock, 0, s>>>(static_cast<const uint64_t*>(smask2), sdict, nstencil, x, y, n_local, nt, st, shift_on ? (1 | (shift_plain() ? 2 : 0)) : 0, shift_theta, shift_sigma);
        KS_HIP(hipGetLastError());
        return;
      }
      if (nstencil > 0 && nghost > 0 && !p2p_halo && smask2 && (stencil_mask_bytes == 1 || stencil_mask_bytes == 4) && n_local >= 2 * kBlock) {
        // SPLIT PRODUCT of a rank on the collective transports (round 6c).  The ghost-aware kernel below is the one-row-per-lane
        // form (8-byte gathers: 16.2 us at the 8-way share of 216^3, 1.26e6 rows) where the paired kernel of the single-GPU path
        // takes 7.7 us.  Only the rows outside [ghost_lo_end, ghost_hi_begin) -- a slab's first and last plane -- reference ghost
        // columns: the paired kernel runs over ALL local rows (a ghost column is clamped to a local row there: the boundary rows
        // come out wrong), then the ghost-aware kernel rewrites the boundary tiles.  Same products in the same order as either
        // kernel alone (both add the slots in dictionary order).  A slab with two neighbours has NINE dictionary slots (seven of the
        // stencil, one ghost stride per neighbour): with all nine the paired kernel issues two groups of eight pair loads per lane and
        // the split is SLOWER (0.0491 against 0.0471 ms per iteration at the 8-way share of 216^3); with the slots that local
        // columns use only (nstencil_local: the ghost strides come last, only boundary rows carry their bits) it is one group:
        // 0.0444 against 0.0475 (profiles/r06c_ab.txt).  KS_DIST_SPLIT=0: the ghost-aware kernel for every row.
        if constexpr (std::is_same<D, double>::value) {
          static const int split_env = env_int("KS_DIST_SPLIT", 1);
          const int nt1 = (int)((n_local + kBlock - 1) / kBlock);
          const int nlow = (int)((ghost_lo_end + kBlock - 1) / kBlock), first_high = (int)(ghost_hi_begin / kBlock);
          const int nhigh = nt1 - first_high;
          if (split_env && env_int("KS_STENCIL_PAIRS", 1) && nlow <= first_high && 4 * (nlow + nhigh) <= nt1 && nlow + nhigh > 0) {
            const int nt2 = (int)(((n_local + 1) / 2 + kBlock - 1) / kBlock);
            static const int split_dbg = env_int("KS_DIST_SPLIT_DEBUG", 0);
            if (split_dbg && !split_said) {
              split_said = true;
              std::fprintf(stderr, "[split] rows %lld: paired kernel on all of them (%d of %d slots), ghost-aware kernel on %d + %d boundary tiles of %d\n", (long long)n_local, nstencil_local, nstencil, nlow, nhigh, nt1);
            }
            // (a slab in the middle of the partition has NINE slots -- the seven of the stencil and one ghost stride per neighbour --,
            // hence 4-byte masks: the paired kernel takes them as 64-bit words of two rows)
            const int shf = shift_on ? (1 | (shift_plain() ? 2 : 0)) : 0;
            ksd::HaloFused h0{};
            // (the paired kernel only takes the slots that local columns use: the ghost strides come last in the dictionary, only
            // boundary rows have their bits, and those rows are rewritten below -- one group of eight pair loads instead of two)
            const int nsl = nstencil_local >= 1 ? nstencil_local : nstencil;
            if (stencil_mask_bytes == 1) {
              ksd::k_spmv_stencil2<D, uint16_t><<<nt2, kBlock, 0, s>>>(static_cast<const uint16_t*>(smask2), sdict, nsl, x, y, n_local, nt2, st, shf, shift_theta, shift_sigma);
              ksd::k_spmv_stencil<D, uint8_t, 1><<<nlow + nhigh, kBlock, 0, s>>>(static_cast<const uint8_t*>(smask), sdict, nstencil, x, xg, y, n_local,
                                                                                std::max<int64_t>(nghost, 0), nlow + nhigh, st, h0, hargs, ctx->p2p.dev, shift_on ? 1 : 0,
                                                                                shift_theta, shift_sigma, nlow, first_high - nlow);
            } else {
              ksd::k_spmv_stencil2<D, uint64_t><<<nt2, kBlock, 0, s>>>(static_cast<const uint64_t*>(smask2), sdict, nsl, x, y, n_local, nt2, st, shf, shift_theta, shift_sigma);
              ksd::k_spmv_stencil<D, uint32_t, 1><<<nlow + nhigh, kBlock, 0, s>>>(static_cast<const uint32_t*>(smask), sdict, nstencil, x, xg, y, n_local,
                                                                                 std::max<int64_t>(nghost, 0), nlow + nhigh, st, h0, hargs, ctx->p2p.dev, shift_on ? 1 : 0,
                                                                                 shift_theta, shift_sigma, nlow, first_high - nlow);
            }
            KS_HIP(hipGetLastError());
            return;
          }
        }
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
                                                                std::max<int64_t>(nghost, 0), nt, st, h, hargs, ctx->p2p.dev, shift_on ? 1 : 0,
                                                                shift_theta, shift_sigma);
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
            ksd::k_spmv_dvi<D, IP, UN, RPT><<<nt, kBlock, 0, s>>>(rp, codes, ddelta, val, x, xg, y, n_local, nt, ndvi, st, hseq, ghost_stride, shift_arg());
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
                                                                           n_local, nslices, ng, st, hseq, ghost_stride, ndict, shift_arg());
            else
              ksd::k_spmv_sell<D, IP, VI, UN, true><<<ng, kBlock, 0, s>>>(static_cast<const IP*>(sliceptr), colidx, val, sperm, x, xg, y,
                                                                          n_local, nslices, ng, st, hseq, ghost_stride, ndict, shift_arg());
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
                                                                   nullptr, 0, h, hargs, ctx->p2p.dev, csr_nt != 0, row_gather, shift_arg());
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
                   const std::vector<int32_t>& ci, const std::vector<D>& vv, int cb_mode = 0, int64_t nghost = 0, int64_t nlow = 0) {
  // cb_mode: 0 = column blocks not allowed, 1 = allowed (decided below), 2 = this IS a column block (plain CSR row blocks,
  //          nothing else is tried), 3 = allowed, row block of a distributed operator: columns >= nrows are ghost slots
  //          (nghost of them, the first nlow owned by lower ranks -- they precede the local columns in the global order)
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
      std::vector<uint8_t> local_used(256, 0);   // dictionary entry used by at least one LOCAL column (ghost-only entries: the split product)
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
          if (hit) { if (ci[p] < nrows) local_used[codes[p]] = 1; continue; }
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
          if (ci[p] < nrows) local_used[id] = 1;
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
          // (trailing slots that only ever name ghost columns -- one stride per neighbour of a slab: rows using them are boundary rows)
          op->nstencil_local = ns;
          for (int k = ns - 1; k >= 0 && !local_used[order[k]]; --k) op->nstencil_local = k;
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
  if ((cb_mode == 1 || cb_mode == 3) && op->ndict == 0 && nnz > 0) {
    const int cb_env = env_int("KS_SPMV_COLBLOCKS", -1);  // (read per upload: tests switch it inside one process)
    // Distributed operators (cb_mode 3): the referenced columns in GLOBAL order are [ghosts of lower ranks | local columns |
    // ghosts of higher ranks]; key(c) is the position of local-extended column c in that order.  Blocks are ranges of keys
    // that do not straddle a segment, so every block gathers either from x or from the ghost vector, and a row stored in
    // global column order (what a row block of a sorted CSR matrix is) is summed in the same order as on one GPU.
    const int64_t next = nrows + nghost;
    auto key = [&](int64_t c) { return c < nrows ? nlow + c : (c - nrows < nlow ? c - nrows : c); };
    const int64_t seg_lo[3] = {0, nlow, nlow + nrows}, seg_hi[3] = {nlow, nlow + nrows, next};
    int nbk = 0;
    if (cb_env != 0) {
      bool sorted = true;
      int64_t far = 0, maxrow = 0;
      const int64_t fardist = std::max<int64_t>(1, next / 16);
      for (int64_t r = 0; r < nrows && sorted; ++r) {
        maxrow = std::max(maxrow, rp[r + 1] - rp[r]);
        for (int64_t q = rp[r]; q < rp[r + 1]; ++q) {
          if (q > rp[r] && key(ci[q]) < key(ci[q - 1])) { sorted = false; break; }
          far += std::llabs(key(ci[q]) - (nlow + r)) > fardist;
        }
      }
      const double xmb = (double)next * sizeof(D) / (1 << 20);
      if (sorted && maxrow <= 4 * kBlock) {
        if (cb_env >= 2) nbk = cb_env;
        // block width ~ 4 MiB of x (measured optimum at n = 1e6: 2 blocks, 2e6: 4 blocks); beyond 8 blocks the y that is
        // written and read back between the launches (16 n bytes each) eats the gain (n = 1e7: 8 blocks -11 %, 16: +35 %)
        else if (xmb >= 6.0 && xmb <= 160.0 && 2 * far >= nnz) nbk = std::min(8, std::max(2, (int)std::lround(xmb / 4.0)));
      }
    }
    if (nbk >= 2) {
      nbk = std::min(nbk, ksd::kCbMaxBlocks);
      // block boundaries in key space: nbk blocks shared out over the non-empty segments in proportion to their width
      // (one segment -- a single GPU --: b n / nbk, as before)
      std::vector<int64_t> bounds{0};
      std::vector<int> bseg;
      {
        int nseg = 0;
        for (int g = 0; g < 3; ++g) nseg += seg_hi[g] > seg_lo[g];
        nbk = std::max(nbk, nseg);
        int cnt[3] = {0, 0, 0}, used = 0;
        for (int g = 0; g < 3; ++g)
          if (seg_hi[g] > seg_lo[g]) { cnt[g] = std::max(1, (int)((double)nbk * (double)(seg_hi[g] - seg_lo[g]) / (double)next)); used += cnt[g]; }
        while (used > std::min(nbk, ksd::kCbMaxBlocks)) {  // (rounding up the narrow segments): take from the segment with the most blocks
          int g = 0;
          for (int h = 1; h < 3; ++h) if (cnt[h] > cnt[g]) g = h;
          if (cnt[g] <= 1) break;
          --cnt[g]; --used;
        }
        while (used < nbk) {  // give the rest to the segment with the widest blocks
          int g = -1;
          for (int h = 0; h < 3; ++h)
            if (cnt[h] > 0 && (g < 0 || (double)(seg_hi[h] - seg_lo[h]) / cnt[h] > (double)(seg_hi[g] - seg_lo[g]) / cnt[g])) g = h;
          ++cnt[g]; ++used;
        }
        KS_REQUIRE(used <= ksd::kCbMaxBlocks, KS_ERR_INTERNAL, "column blocks: more segments than blocks");
        for (int g = 0; g < 3; ++g)
          for (int b = 0; b < cnt[g]; ++b) {
            const int64_t w = seg_hi[g] - seg_lo[g];
            bounds.push_back(b + 1 == cnt[g] ? seg_hi[g] : seg_lo[g] + (int64_t)(b + 1) * w / cnt[g]);
            bseg.push_back(g);
          }
        nbk = used;
      }
      // single-launch form (k_spmv_csr_cb): largest segment (entries of a tile of 256 * RPT rows inside one column block)
      // for every candidate RPT
      constexpr int kRptCand[5] = {1, 2, 4, 8, 16};
      int64_t maxseg[5] = {0, 0, 0, 0, 0};
      bool small_ptrs = true;
      for (int b = 0; b < nbk; ++b) {
        const int64_t lo = bounds[b], hi = (b + 1 == nbk) ? (int64_t)1 << 40 : bounds[b + 1];
        const bool from_ghost = bseg[b] != 1;
        std::vector<int64_t> rpb((size_t)nrows + 1, 0);
        std::vector<int32_t> cib;
        std::vector<D> vvb;
        for (int64_t r = 0; r < nrows; ++r) {
          for (int64_t q = rp[r]; q < rp[r + 1]; ++q) {
            const int64_t kq = key(ci[q]);
            if (kq >= lo && kq < hi) { cib.push_back(from_ghost ? (int32_t)(ci[q] - nrows) : ci[q]); vvb.push_back(vv[q]); }
          }
          rpb[r + 1] = (int64_t)cib.size();
        }
        for (int k = 0; k < 5; ++k) {
          const int64_t tr = (int64_t)kBlock * kRptCand[k];
          for (int64_t r0 = 0; r0 < nrows; r0 += tr) maxseg[k] = std::max(maxseg[k], rpb[std::min(nrows, r0 + tr)] - rpb[r0]);
        }
        op->cblocks.emplace_back(make_csr<D>(ctx, nrows, (int64_t)cib.size(), rpb, cib, vvb, 2));
        op->cb_from_ghost.push_back(from_ghost ? 1 : 0);
        small_ptrs = small_ptrs && !op->cblocks.back()->ptr64;
      }
      // Measured (tools/cb_single_ab.py, profiles/r03_column_blocks.txt): the single launch wins where the y round trips of
      // many blocks hurt (n = 1e7, 8 blocks: 858 -> 823 us) and loses a little where two to four launches were already close
      // to what bounds this product -- the rate at which an XCD's L2 hands out randomly addressed lines, 5e6 of them for
      // 1e6 rows: 46 us either way at n = 1e6, 100 vs 107 us at 2e6.  So: single launch from 5 blocks on
      // (KS_SPMV_CB_SINGLE=0 never, KS_SPMV_CB_RPT=k forces it with k sub-tiles per workgroup).  A distributed operator
      // always takes the single launch (the per-block launches have one x; the kernel takes a base per block).
      const int rpt_force = env_int("KS_SPMV_CB_RPT", 0);
      if (small_ptrs && (cb_mode == 3 || (env_int("KS_SPMV_CB_SINGLE", 1) && (nbk > 4 || rpt_force > 0)))) {
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
      if (cb_mode == 3 && !op->cb_rpt) {
        // (no single-launch shape fits: row blocks of plain CSR below)
        op->cblocks.clear();
        op->cb_from_ghost.clear();
      } else {
        op->layout = KS_LAYOUT_CSR_CB;
        op->bytes_per_nnz = 4.0 + sizeof(D);
        op->aux_bytes = 0.0;
        for (auto& cbk : op->cblocks) op->aux_bytes += cbk->aux_bytes;
        if (!op->cb_rpt) op->aux_bytes += (double)(nbk - 1) * 2.0 * sizeof(D) * (double)nrows;  // y written and read back between the blocks
        return op.release();
      }
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

// default of ks_operator::apply_shifted: product + one streaming pass
inline void ks_operator::apply_shifted(const void* x, void* y, double theta_re, double theta_im, double sigma, int64_t ld, const DevState* st) {
  apply(x, y, st);
  ProfScope ps(ctx, KSP_SCALE, 3.0 * (double)n_local * (dtype == KS_F64 ? 8.0 : 16.0));
  const int nbk = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->num_cu * 8, ld / (2 * kBlock) + 1));
  if (dtype == KS_F64) {
    ksd::k_shift_scale<double><<<nbk, kBlock, 0, ctx->stream>>>(static_cast<double*>(y), static_cast<const double*>(x), theta_re, sigma, ld, st);
  } else {
    ksd::k_shift_scale<cd><<<nbk, kBlock, 0, ctx->stream>>>(static_cast<cd*>(y), static_cast<const cd*>(x), cd{theta_re, theta_im}, sigma, ld, st);
  }
  KS_HIP(hipGetLastError());
}
