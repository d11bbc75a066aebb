// host side of the S-STEP (block) Arnoldi expansion (kernels: ks_block_kernels.hpp; model: tests/sstep_model.py).
// Part of the ONE translation unit of libkschur_hip.so: included by ks_hip.hip after ks_workspace.hpp.
//
// iterate_arnoldi!(A, arnoldi, from:to) (src/expansion.jl:116-133) in blocks of s steps: per block s operator products
// (Newton basis, shifts = Leja-ordered Ritz values of the previous restart), TWO passes over the basis and two reductions --
// instead of two passes and one reduction per STEP of the default (implicit second pass) expansion, four passes per step of
// the reference.  Same Krylov space, hence the same H up to rounding (implicit-Q); blocks whose Gram matrix is numerically
// rank deficient (breakdown, ill-conditioned basis) are abandoned before anything is committed and redone step by step.
#pragma once

// (kernels: ks_block_kernels.hpp, included at the top of ks_hip.hip; this file sits inside the anonymous namespace that
// ks_workspace.hpp opens and ks_backend.hpp closes)


// ---- block sizes ------------------------------------------------------------------------------------------------------------
// instantiated shapes: ks_blk_shape_ok (ks_block_launch.hpp)
inline bool blk_shape_ok(int dtype, int k, int s) { return ks_blk_shape_ok(dtype == KS_F64 ? 0 : 1, k, s); }
// split `count` steps starting with k0 existing columns into block sizes <= smax; where the kernels stop (ComplexF64 beyond
// 32 / 48 columns) the partition ends early: the steps behind it run one at a time
inline std::vector<int> blk_partition(int dtype, int k0, int count, int smax) {
  std::vector<int> out;
  int k = k0;
  while (count > 0) {
    int s = std::min(count, smax);
    while (s > 1 && !blk_shape_ok(dtype, k, s)) --s;
    if (!blk_shape_ok(dtype, k, s)) break;
    out.push_back(s);
    k += s;
    count -= s;
  }
  return out;
}

// ---- Newton shifts ------------------------------------------------------------------------------------------------------------
// Leja ordering of the Ritz values of the previous restart (tests/sstep_model.py: leja_order / newton_shifts).  Real element
// type: real parts only, a conjugate pair contributes once.
inline std::vector<std::complex<double>> leja_shifts(const std::vector<std::complex<double>>& ritz, int s, bool real) {
  std::vector<std::complex<double>> pts;
  for (auto z : ritz) {
    if (real) z = std::complex<double>(z.real(), 0.0);
    if (!std::isfinite(z.real()) || !std::isfinite(z.imag())) continue;
    bool dup = false;
    for (auto& q : pts) dup = dup || std::abs(q - z) <= 1e-14 * std::max(1.0, std::abs(z));
    if (!dup) pts.push_back(z);
  }
  std::vector<std::complex<double>> out;
  if (pts.empty()) { out.assign(s, 0.0); return out; }
  std::vector<double> logprod(pts.size(), 0.0);
  std::vector<bool> used(pts.size(), false);
  size_t cur = 0;
  for (size_t i = 1; i < pts.size(); ++i) if (std::abs(pts[i]) > std::abs(pts[cur])) cur = i;
  while (out.size() < pts.size() && (int)out.size() < s) {
    used[cur] = true;
    out.push_back(pts[cur]);
    size_t best = pts.size();
    for (size_t i = 0; i < pts.size(); ++i) {
      if (used[i]) continue;
      logprod[i] += std::log(std::max(std::abs(pts[i] - pts[cur]), 1e-300));
      if (best == pts.size() || logprod[i] > logprod[best]) best = i;
    }
    if (best == pts.size()) break;
    cur = best;
  }
  const size_t have = out.size();
  while ((int)out.size() < s) out.push_back(out[out.size() % have]);
  return out;
}

inline void blk_ensure_buffers(ks_workspace* ws) {
  if (ws->bpart) return;
  const size_t entries = (size_t)ksd::kBlkKMax * ksd::kBlkSMax + ksd::kBlkGram;
  KS_HIP(hipMalloc(&ws->bpart, entries * (size_t)ws->pnb * ws->esz));
  KS_HIP(hipMalloc(&ws->bred, entries * ws->esz));
  const size_t sb = ws->dtype == KS_F64 ? sizeof(ksd::BlkScratch<double>) : sizeof(ksd::BlkScratch<cd>);
  KS_HIP(hipMalloc(&ws->bscr, sb));
  KS_HIP(hipMemset(ws->bscr, 0, sb));
  KS_HIP(hipMalloc(&ws->bzero, 256));
  KS_HIP(hipMemset(ws->bzero, 0, 256));
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
// (the ~200 instantiations of the two streaming kernels live in translation units of their own: ks_block_inst.hip)
inline int& blk_dbg() { static int v = env_int("KS_BLK_DBG", 0); return v; }
template <class D> int launch_blk(ks_workspace* ws, int which, int k, int s, bool zscratch = false) {
  auto* bs = static_cast<ksd::BlkScratch<D>*>(ws->bscr);
  BlkLaunchArgs a{};
  a.V = ws->V; a.ld = ws->ld; a.dtype = sizeof(D) == 8 ? 0 : 1; a.k = k; a.s = s;
  if (zscratch) { a.zsrc = ws->zscratch; a.ldz = ws->ld; }
  if (which == 2) { a.cin = ws->rot_cin; a.out0 = ws->rot_out0; a.rotm = ws->Qd; }
  a.partial = ws->bpart; a.pnb = ws->pnb; a.coefp = bs->coefp; a.r1inv = bs->r1inv; a.zeros = ws->bzero; a.st = ws->st;
  a.dbg = blk_dbg(); a.nt = ws->v_nt ? 1 : 0; a.num_cu = ws->ctx->num_cu; a.bpc = ws->ctx->bpc; a.stream = ws->ctx->stream;
  try {
    return ks_blk_launch(which, a);
  } catch (const std::runtime_error& e) {
    throw KsError{KS_ERR_INTERNAL, e.what()};
  }
}

// Steps from..to as blocks of the given sizes.  Every column is an ordinary one on entry (the caller materialised): T = I
// below `from`, ws->ntrue == from.
// The scratch columns of a block whose first pass is fused with the restart rotation (and of the speculative chain):
// blk_smax<D>() columns of ld elements -- 1.6 GB at n = 1e7, Float64, on top of a 3.3-GB basis (documented in include/kschur.h).
// Allocated on first use; false when the device has no room: the caller then takes the path that needs none (ordinary rotation,
// chain written to the basis) -- a workspace that fitted the device before the fused rotation existed must still run.
template <class D> bool ensure_zscratch(ks_workspace* ws) {
  if (ws->zscratch) return true;
  const size_t zbytes = (size_t)ws->ld * ksd::blk_smax<D>() * sizeof(D);
  if (hipMalloc(&ws->zscratch, zbytes) != hipSuccess) {
    (void)hipGetLastError();
    ws->zscratch = nullptr;
    ws->spec_on = false;        // no speculative chains either
    ws->rot_defer_on = false;   // restarts rotate at once from here on
    return false;
  }
  KS_HIP(hipMemsetAsync(ws->zscratch, 0, zbytes, ws->ctx->stream));   // the pad rows (n .. ld) stay zero: operators write rows < n only
  if (env_int("KS_ADDR_DEBUG", 0)) std::fprintf(stderr, "[addr] V %p zscratch %p ld %lld bytes/col %lld\n", ws->V, ws->zscratch, (long long)ws->ld, (long long)(ws->ld * sizeof(D)));
  return true;
}

// The true last column of the basis as it stands, S[:, 0:maxdim] T[0:maxdim, maxdim] with the device-resident T (k_fin_blk's), into
// the scratch column ztrue (ks_workspace::ztrue): the start of a chain behind a block whose Gram deviation is above rounding level.
// In stream order behind the batch that wrote T; before anything rotates the basis or resets T.  false: no kernel for the width
// (maxdim + 1 > 40) or no room -- the caller rotates at once instead.
template <class D> bool true_start_enqueue(ks_workspace* ws) {
  if (ws->ztrue_valid) return true;
  const int cin = ws->maxdim + 1;
  if (cin > 40) return false;
  if (!ws->ztrue) {
    if (hipMalloc(&ws->ztrue, (size_t)ws->ld * sizeof(D)) != hipSuccess) { (void)hipGetLastError(); ws->ztrue = nullptr; return false; }
    KS_HIP(hipMemsetAsync(ws->ztrue, 0, (size_t)ws->ld * sizeof(D), ws->ctx->stream));   // (pad rows stay zero)
  }
  ks_ctx* cx = ws->ctx;
  ProfScope ps(cx, KSP_ROTATE, (double)ws->n * sizeof(D) * (cin + 1));
  const D* Vc = static_cast<const D*>(ws->col(0));
  const D* tcol = static_cast<const D*>(ws->Td) + (size_t)ws->maxdim * ws->ldt;
  D* out = static_cast<D*>(ws->ztrue);
  const size_t smem = (size_t)cin * sizeof(D);
  const int nb = cap_blocks(ws, cx->num_cu * 4, kBlock);
  if (cin <= 8) ksd::k_rotate_valu<D, 8><<<nb, kBlock, smem, cx->stream>>>(Vc, ws->ld, cin, 1, tcol, ws->ldt, out, ws->ld, -1);
  else if (cin <= 16) ksd::k_rotate_valu<D, 16><<<nb, kBlock, smem, cx->stream>>>(Vc, ws->ld, cin, 1, tcol, ws->ldt, out, ws->ld, -1);
  else if (cin <= 24) ksd::k_rotate_valu<D, 24><<<nb, kBlock, smem, cx->stream>>>(Vc, ws->ld, cin, 1, tcol, ws->ldt, out, ws->ld, -1);
  else ksd::k_rotate_valu<D, 40><<<nb, kBlock, smem, cx->stream>>>(Vc, ws->ld, cin, 1, tcol, ws->ldt, out, ws->ld, -1);
  KS_HIP(hipGetLastError());
  ws->ztrue_valid = true;
  ws->true_starts++;
  return true;
}

template <class D>
void enqueue_steps_blk(ks_workspace* ws, ks_operator* op, int from, const std::vector<int>& sizes, const ksd::BlkShifts<D>& sh, int ndefl = 0) {
  ks_ctx* cx = ws->ctx;
  hipStream_t s_ = cx->stream;
  const int ldh = ws->maxdim + 1;
  const double nb8 = (double)ws->n * sizeof(D);
  D* Hd = static_cast<D*>(ws->Hd);
  D* Tm = static_cast<D*>(ws->Td);
  auto* bs = static_cast<ksd::BlkScratch<D>*>(ws->bscr);
  int k = from;   // existing columns == index of the block's first step
  int first = 1;
  for (int s : sizes) {
    // FUSED RESTART ROTATION (ks_workspace::rot_pending -> rot_fuse): the first block after a restart whose rotation is still
    // pending writes its Newton chain to scratch columns -- the chain starts from the stored last column of the OLD basis, the
    // residual direction up to the Gram deviation of the block that wrote it (rotate_tfold admits <= 1e-12) --, then ONE kernel
    // rotates the basis and takes the first pass; the second pass reads the chain from the scratch columns and writes the block
    // to its place.
    const bool fuse = first && ws->rot_fuse;
    const bool split = fuse && ws->rot_split;   // no fused kernel for the shape: the ordinary rotation, then both passes on the scratch columns
    const bool chain_true = fuse && ws->chain_true;   // the chain starts from the true last column (true_start_enqueue), not the stored one
    ws->rot_fuse = false;
    ws->rot_split = false;
    ws->chain_true = false;
    char* zs = nullptr;
    if (fuse) {
      KS_REQUIRE(ws->zscratch != nullptr, KS_ERR_INTERNAL, "fused rotation without scratch columns (the adoption allocates them)");
      zs = static_cast<char*>(ws->zscratch);
    }
    if (split) {
      // (in place: reads the old basis including its last column -- the chain's start, which it does not write: out0 + rr <= maxdim)
      rotate_device<D>(ws, 0, ws->rot_cin, ws->rot_rr, ws->rot_out0, -1);
      ws->rot_split_count++;
    }
    auto zcol = [&](int i) -> void* { return fuse ? static_cast<void*>(zs + (size_t)i * ws->ld * sizeof(D)) : ws->col(k + i); };
    op->shift_store_cacheable = s >= 10;   // (ks_operators.hpp: pays with one inner-product pass per ten or twenty products)
    const int adopted = fuse ? ws->spec_adopt : 0;   // products of this chain that ran speculatively behind the previous expansion
    ws->spec_adopt = 0;
    for (int i = adopted; i < s; ++i) {
      double tre, tim;
      if constexpr (sizeof(D) == 8) { tre = sh.theta[i]; tim = 0.0; }
      else { tre = sh.theta[i].x; tim = sh.theta[i].y; }
      op->in_scale = 1.0;
      const void* src = i == 0 ? (fuse ? (chain_true ? static_cast<const void*>(ws->ztrue) : ws->col(ws->maxdim)) : ws->col(k - 1)) : zcol(i - 1);
      op->apply_shifted(src, zcol(i), tre, tim, sh.sigma[i], ws->ld, ws->st);
      if (ndefl > 0) {
        // in-chain deflation against the locked columns of dominant eigenvalues (ks_block_kernels.hpp: kDeflMax)
        ProfScope ps(cx, KSP_AXPY, nb8 * 2.0 * (ndefl + 1) + nb8);
        if (!ws->defl_part) KS_HIP(hipMalloc(&ws->defl_part, (size_t)ksd::kDeflMax * (1024 + 1) * sizeof(D)));
        const int nbd = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(1024, cx->num_cu * 4), (ws->n + kBlock - 1) / kBlock));
        D* zc = static_cast<D*>(zcol(i));
        const D* U = static_cast<const D*>(ws->col(0));
        D* part = static_cast<D*>(ws->defl_part);
        ksd::k_defl_dots<D><<<nbd, kBlock, 0, s_>>>(U, ws->ld, ndefl, zc, ws->n, part);
        if (!cx->distributed()) {
          ksd::k_defl_apply<D><<<nbd, kBlock, 0, s_>>>(U, ws->ld, ndefl, zc, ws->n, part, nbd, bs->cdefl + (size_t)i * ksd::kDeflMax);
        } else {
          // several ranks (row-partitioned basis): one more collective per product, of ndefl elements -- every rank the same c_i
          D* red = part + (size_t)ksd::kDeflMax * 1024;
          ksd::k_defl_reduce<D><<<1, kBlock, 0, s_>>>(part, nbd, ndefl, red);
          cx->allreduce(reinterpret_cast<double*>(red), ndefl * (int)(sizeof(D) / 8));
          ksd::k_defl_apply<D><<<nbd, kBlock, 0, s_>>>(U, ws->ld, ndefl, zc, ws->n, red, 1, bs->cdefl + (size_t)i * ksd::kDeflMax);
        }
      }
    }
    if (ndefl > 0) ws->defl_blocks++;
    const int ne = k * s + s * (s + 1) / 2;
    int nb1, nb2;
    // reduction + small algebra of one stage.  Several ranks: reduce -> ONE all-reduce of k s + s (s + 1) / 2 elements over
    // the context's transport -> algebra (every rank the same sums, the same decisions); per BLOCK what the per-step paths
    // do per step (src/expansion.jl:84,88,93,96 are the reductions that become collectives)
    auto fin = [&](int stage, int nbp) {
      ProfScope ps(cx, KSP_FIN, 0.0);
      const D* part = static_cast<const D*>(ws->bpart);
      D* red = static_cast<D*>(ws->bred);
      if (!cx->distributed()) {
        ksd::k_fin_blk<D><<<(ne + 3) / 4, kBlock, 0, s_>>>(stage, 0, part, nbp, ws->pnb, k, s, red, Hd, ldh, Tm, ws->ldt, ws->ntrue, bs, sh, first, ws->blk_pivmin,
                                                 ws->blk_gdevmax, ws->st, ws->ctr, ndefl);
      } else {
        ksd::k_fin_blk<D><<<(ne + 3) / 4, kBlock, 0, s_>>>(stage, 1, part, nbp, ws->pnb, k, s, red, Hd, ldh, Tm, ws->ldt, ws->ntrue, bs, sh, first, ws->blk_pivmin,
                                                 ws->blk_gdevmax, ws->st, ws->ctr, ndefl);
        cx->allreduce(reinterpret_cast<double*>(red), ne * (int)(sizeof(D) / 8));
        ksd::k_fin_blk<D><<<1, kBlock, 0, s_>>>(stage, 2, part, nbp, ws->pnb, k, s, red, Hd, ldh, Tm, ws->ldt, ws->ntrue, bs, sh, first, ws->blk_pivmin,
                                                ws->blk_gdevmax, ws->st, ws->ctr, ndefl);
      }
    };
    if (fuse && !split) {
      // reads the old basis and Z, writes the rotated columns
      ProfScope ps(cx, KSP_ROTATE, nb8 * (ws->rot_cin + s + ws->rot_rr));
      nb1 = launch_blk<D>(ws, 2, k, s, true);
      ws->rot_fused_count++;
    } else {
      ProfScope ps(cx, KSP_DOTS, nb8 * (k + s));               // reads S[:, 0:k) and Z
      nb1 = launch_blk<D>(ws, 0, k, s, fuse);
    }
    fin(1, nb1);
    {
      ProfScope ps(cx, KSP_FUSED, nb8 * (k + 2 * s));          // reads S[:, 0:k) and Z, writes the block
      nb2 = launch_blk<D>(ws, 1, k, s, fuse);
    }
    fin(2, nb2);
    k += s;
    first = 0;
  }
  KS_HIP(hipGetLastError());
}

// shifts of a batch from the workspace's Ritz values; false: nothing to take them from (first expansion of a run)
// `excluded`: eigenvalues of the locked columns the chain is deflated against (in-chain deflation): no shift there, and the scale
// follows the rest of the spectrum
template <class D> bool blk_make_shifts(ks_workspace* ws, int smax, ksd::BlkShifts<D>& sh, const std::vector<std::complex<double>>* excluded = nullptr) {
  if (!ws->ritz_valid || ws->ritz.empty()) return false;
  const bool real = sizeof(D) == 8;
  std::vector<std::complex<double>> pool;
  if (excluded && !excluded->empty()) {
    for (auto z : ws->ritz) {
      bool ex = false;
      for (auto q : *excluded) ex = ex || std::abs(z - q) <= 1e-6 * std::max(1.0, std::abs(q));
      if (!ex) pool.push_back(z);
    }
  }
  const std::vector<std::complex<double>>& src = pool.empty() ? ws->ritz : pool;
  const auto th = leja_shifts(src, smax, real);
  double rho = 0.0;
  for (auto z : src)
    if (std::isfinite(std::abs(z))) rho = std::max(rho, std::abs(z));
  if (!(rho > 0.0) || !std::isfinite(rho)) return false;
  const double sigma = std::ldexp(1.0, -(int)std::lround(std::log2(rho)));   // power of two ~ 1 / ||A||: range only, exact
  for (int i = 0; i < ksd::kBlkSMax; ++i) {
    const auto z = th[i < (int)th.size() ? i : 0];
    if constexpr (sizeof(D) == 8) sh.theta[i] = z.real();
    else sh.theta[i] = cd{z.real(), z.imag()};
    sh.sigma[i] = sigma;
  }
  return true;
}
