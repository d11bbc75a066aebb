// the HIP backend of the Krylov-Schur driver (ks_driver.hpp), the on-device residual checks, the placement search
// Part of the ONE translation unit of libkschur_hip.so: included by ks_hip.hip, in this order --
//     ks_context.hpp -> ks_operators.hpp -> ks_workspace.hpp -> ks_backend.hpp -> (C ABI in ks_hip.hip)
// -- and not meant to be included on its own (needs ks_workspace.hpp).
#pragma once
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
      gate_cancel(ws);
      (void)hipStreamSynchronize(ws->ctx->stream);
      try { reset_lazy(ws); } catch (...) {}
      prov_drop(ws);
      // a pending / adopted rotation, a speculative chain and a drift watch in flight all belong to the factorisation that just
      // became undefined: an exception between the adoption (rot_fuse, spec_adopt) and the first block kernel must not leave them
      // for the batch that follows the re-initialisation (it would rotate a fresh basis with a stale Q and skip products)
      ws->rot_pending = ws->rot_fuse = ws->rot_split = false;
      ws->rot_true_start = ws->chain_true = ws->ztrue_valid = false;
      ws->spec_valid = false;
      ws->spec_adopt = 0;
      ws->rp_inflight = false;
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
    bool no_block = false;  // a block of this call was abandoned: the rest of the range runs step by step
    // (columns still in factored form were produced by the library's previous call, no restart in between: nothing to measure,
    // and materialising them here would change the rounding of a run that stays step by step)
    // (a watch enqueued behind the previous expansion: may switch the blocks off for this one.  BEFORE the relation probe: the two
    // share the pinned words -- the probe's coefficients start where the watch keeps its sequence word)
    drift_probe_collect();
    if (trusted && ws->prov_vouched && !ws->t_lazy && ws->sstep_eff >= 2 && op->async_capable && from >= 2 && to - from + 1 >= 2) relation_probe(from, H);
    blk_shifts_from_saved_H();
    while (j0 <= to) {
      const double tb0 = ks::now_s();
      int jend = to;
      if (!op->async_capable) jend = j0;  // host operators: one step per batch
      if (explicit_step >= 0) jend = j0;
      const bool lazy = use_deferred(ws, jend);
      const bool tpath = lazy && ws->passes == 2 && trusted && explicit_step < 0;  // implicit second pass: two reads of the basis per step
      // S-STEP form of the same path (ks_block.hpp): blocks of up to ws->sstep steps, two reads of the basis per BLOCK.
      // Needs Newton shifts (Ritz values of a previous restart: not before the first one) and a device-resident operator;
      // a range of one step gains nothing.  Several ranks: two all-reduces per block (ks_block.hpp).
      std::vector<int> blk_sizes;
      ksd::BlkShifts<D> blk_sh{};
      // in-chain deflation against locked columns of dominant eigenvalues (defl_plan): a problem that abandoned its blocks before
      // those columns were locked gets its block size back when they are
      std::vector<std::complex<double>> defl_ex;
      int ndefl = 0;
      if (tpath && !no_block && ws->sstep >= 2 && ws->sstep_eff >= 1 && op->async_capable && jend - j0 + 1 >= 2) {
        ndefl = defl_plan(H, j0, defl_ex, std::min(std::min(ws->sstep, ksd::kBlkSMax), jend - j0 + 1));
        if (ndefl > ws->defl_last && ws->sstep_eff < ws->sstep) { ws->sstep_eff = ws->sstep; ws->blk_clean = 0; }
        ws->defl_last = ndefl;
      }
      if (tpath && !no_block && ws->sstep_eff >= 2 && op->async_capable && jend - j0 + 1 >= 2 &&
          blk_make_shifts<D>(ws, std::min(ws->sstep_eff, ksd::kBlkSMax), blk_sh, ndefl > 0 ? &defl_ex : nullptr))
        blk_sizes = blk_partition(ws->dtype, j0, jend - j0 + 1, std::min(ws->sstep_eff, ksd::kBlkSMax));
      const bool bpath = !blk_sizes.empty();
      if (bpath) {   // (a partition that ends early: this batch goes as far as the blocks do, the next one takes the rest step by step)
        int covered = 0;
        for (int sz : blk_sizes) covered += sz;
        jend = std::min(jend, j0 + covered - 1);
      }
      if (ws->rot_pending) {
        // the restart's rotation is still pending (rotate_tfold): this batch's first block does it in the sweep of its first
        // pass when there is a kernel for the shape -- otherwise it runs now, the ordinary way
        // (no fused kernel for the shape: the ordinary rotation kernel runs first and both passes read the chain from
        // the scratch columns -- rot_split; what this buys is the speculative chain)
        const bool fused_ok = bpath && ks_blk_rot_ok(ws->dtype == KS_F64 ? 0 : 1, ws->rot_cin, j0, blk_sizes[0]);
        const bool split_ok = bpath && ws->spec_on && ks_blk_zsrc_ok(ws->dtype == KS_F64 ? 0 : 1, j0, blk_sizes[0]);
        static const int defer_dbg = env_int("KS_DEFER_DEBUG", 0);
        if (defer_dbg)
          std::fprintf(stderr, "[adopt] bpath %d j0 %d out0+rr %d fused_ok %d split_ok %d spec_valid %d blk0 %d spec_ne %d backoff %d\n", (int)bpath, j0, ws->rot_out0 + ws->rot_rr,
                       (int)fused_ok, (int)split_ok, (int)ws->spec_valid, bpath ? blk_sizes[0] : 0, ws->spec_ne, ws->spec_backoff);
        // (a rotation granted on condition of a true start -- ks_workspace::rot_true_start --: a speculative chain qualifies only if
        // it started from the true column; without one that column is formed now, while the old basis and its T still stand)
        const bool need_true = ws->rot_true_start;
        const bool spec_fits = ws->spec_valid && bpath && blk_sizes[0] >= ws->spec_ne && ws->spec_sh.size() == sizeof(blk_sh) && (!need_true || ws->spec_true);
        // (a deflated chain needs the ROTATED locked columns before its first product: the rotation runs now)
        if (bpath && ndefl == 0 && j0 == ws->rot_out0 + ws->rot_rr && (fused_ok || split_ok) && ensure_zscratch<D>(ws) &&
            (!need_true || spec_fits || true_start_enqueue<D>(ws))) {
          ws->rot_pending = false;
          ws->rot_true_start = false;
          ws->rot_fuse = true;
          ws->rot_split = !fused_ok;
          ws->chain_true = spec_fits ? ws->spec_true : need_true;
          ws->t_lazy = false;
          ws->t_hi = -1;
          if (spec_fits) {
            // the first spec_ne products of this block's chain are already in the scratch columns (enqueued behind the
            // previous expansion): the whole batch takes the shift sequence they were made with
            std::memcpy(&blk_sh, ws->spec_sh.data(), sizeof(blk_sh));
            ws->spec_adopt = ws->spec_ne;
            ws->spec_used++;
            ws->spec_backoff_len = 1;
            ws->spec_valid = false;
          }
          spec_drop(ws);
        } else {
          rot_flush(ws);
        }
      }
      spec_drop(ws);   // (whatever was not adopted just now is void: this batch writes to V)
      ws->ztrue_valid = false;   // (and so is the true last column of the basis as it stood)
      if (tpath && (bpath || ws->blk_tail || !(ws->t_lazy && j0 == ws->t_hi + 1))) {
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
      if (bpath) {
        blk_ensure_buffers(ws);
        enqueue_steps_blk<D>(ws, op, j0, blk_sizes, blk_sh, ndefl);
      } else if (tpath) {
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
      // drift watch: behind the publication of H, IN FRONT of the speculative chain (the host is not kept waiting for it; the result
      // is read when the next expansion starts: by then the chain may still be running -- the restart no longer synchronises with
      // the stream, ks_workspace.hpp: qstage_wait --, the watch in front of it has long finished: drift_probe_collect)
      if (bpath && jend == to && to == ws->maxdim && j0 >= 2 && !ws->rp_inflight && ++ws->rp_count >= ws->rp_every) {
        ws->rp_count = 0;
        drift_probe_enqueue(j0, H);
      }
      // (only where the next expansion can be expected to adopt them: this one already had the shape of a first block that reads
      // its chain from scratch columns, and the last speculation was not dropped -- after a drop the next 1, 2, 4, 8 cycles go without)
      if (bpath && ndefl == 0 && jend == to && to == ws->maxdim &&
          (ks_blk_rot_ok(ws->dtype == KS_F64 ? 0 : 1, ws->maxdim + 1, j0, blk_sizes[0]) || ks_blk_zsrc_ok(ws->dtype == KS_F64 ? 0 : 1, j0, blk_sizes[0]))) {
        static const int defer_dbg2 = env_int("KS_DEFER_DEBUG", 0);
        if (defer_dbg2) std::fprintf(stderr, "[spec] enqueue? backoff %d j0 %d blk0 %d\n", ws->spec_backoff, j0, blk_sizes[0]);
        // (and the rotation will stay pending only behind a block whose Gram deviation passes the gate of rotate_tfold: where the last
        // batch's did not -- real shifts on a complex spectrum sit at 1e-11 .. 1e-10 every time -- the chain would be dropped)
        // round 6: there the chain starts from the TRUE last column, formed on the device first (ks_workspace::ztrue), and the
        // rotation stays pending behind any accepted block
        const bool start_stored = ws->blk_count == 0 || ws->blk_diag[2] <= 1e-12;
        if (ws->spec_backoff > 0) --ws->spec_backoff;
        else spec_enqueue(blk_sh, j0, !start_stored);
      }
      // reverse mailbox: the restart that follows this (last) batch will rotate the factored basis -- put that rotation into
      // the stream NOW, behind a gate the host releases when it has Q (ks_workspace.hpp: gate_arm / rotate_tfold)
      // (not behind a block batch with the deferral on: that restart leaves its rotation pending for the next expansion's
      // fused first pass -- ks_workspace::rot_pending -- and a gate would only be cancelled)
      const bool will_defer = ws->rot_defer_on && bpath && ws->sstep_eff >= 8 && to == ws->maxdim;
      if (early && tpath && mb && jend == to && !will_defer) gate_arm(ws);
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
      // blk_bail: a block was abandoned before anything of it was committed (rank-deficient Gram matrix: breakdown, or a
      // Newton basis too ill-conditioned to trust) -- the blocks before it stand, the rest of the range runs step by step
      if (ws->gate_armed && (ws->st_h->breakdown >= 0 || ws->st_h->bail >= 0 || ws->st_h->blk_bail >= 0))
        gate_cancel(ws);  // the batch did not run to its end: more batches (or a re-initialisation) follow, no restart yet
      const int blk_bail = bpath ? ws->st_h->blk_bail : -1;
      const int bail = (tpath && !bpath) ? ws->st_h->bail : -1;
      const int bd = (bail >= 0 || blk_bail >= 0) ? -1 : ws->st_h->breakdown;
      const int last_done = blk_bail >= 0 ? blk_bail - 1 : (bail >= 0 ? bail - 1 : (bd >= 0 ? bd : jend));
      if (bpath) {
        ws->blk_diag[0] = ws->st_h->blk_piv1;
        ws->blk_diag[1] = ws->st_h->blk_piv2;
        ws->blk_diag[2] = ws->st_h->blk_gdev;
        int done = 0, done_blocks = 0;
        for (int sz : blk_sizes) { if (j0 + done + sz - 1 <= last_done) { done += sz; ws->blk_count++; ++done_blocks; } }
        if (blk_bail >= 0) {
          // abandoned: the rest of this range goes step by step; and since what fails is usually the conditioning of the
          // Newton basis (a spectrum the real / few shifts do not cover), later batches use smaller blocks -- 5 -> 2 -> off
          ws->blk_bails++; stats.blk_bails++; no_block = true;
          ws->sstep_eff = ws->sstep_eff >= 4 ? ws->sstep_eff / 2 : (ws->sstep_eff > 2 ? 2 : 1);
          ws->blk_clean = 0;
        } else if (++ws->blk_clean >= 16 && ws->sstep_eff < ws->sstep) {
          // (probe a larger block again: the next instantiated size -- 2 3 4 5 8 10 20 --, not one step more: from 10
          // the sizes 11..19 would all run as blocks of 10)
          int nxt = ws->sstep_eff + 1;
          for (int cand : {2, 3, 4, 5, 8, 10, 20}) if (cand > ws->sstep_eff) { nxt = cand; break; }
          ws->sstep_eff = std::min(ws->sstep, std::max(2, nxt));
          ws->blk_clean = 0;
        }
        stats.blocks += done_blocks;   // (blocks behind an abandoned one were skipped on the device: not counted)
      } else if (tpath && ws->sstep_eff == 1 && ws->sstep >= 2 && blk_bail < 0 && bail < 0 && bd < 0) {
        // blocks were lowered all the way to "off" by abandoned ones (not by a broken relation: that is sstep_eff = 0 and stays):
        // clean per-step batches count too, so that the probe of a larger block after 16 of them is reachable from here
        if (++ws->blk_clean >= 16) { ws->sstep_eff = 2; ws->blk_clean = 0; }
      }
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
        const int good = blk_bail >= 0 ? blk_bail - 1 : (bail >= 0 ? bail - 1 : (bd >= 0 ? bd - 1 : jend));
        if (good >= ws->ntrue) {
          ws->t_lazy = true;
          ws->t_hi = good;
          ws->blk_tail = bpath;
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
    if (ws->sstep_eff >= 2 && to == ws->maxdim && !early_stands) {
      // a caller that runs the restart itself (the reference's own _partialschur on a device basis) never tells the library
      // its Ritz values: keep the full Hessenberg matrix, the next expansion takes its shifts from it (blk_shifts_from_saved_H)
      ws->Hfull.assign(static_cast<const char*>(ws->H), static_cast<const char*>(ws->H) + h_bytes(ws));
      ws->hfull_valid = true;
      ws->ritz_valid = false;   // (note_ritz of a driver that does know them follows and takes precedence)
    }
    return early_stands;
  }

  // IN-CHAIN DEFLATION, the plan of a batch (ks_block_kernels.hpp: kDeflMax; tools/model_deflated_chain.py).  The locked columns are
  // the leading decoupled block of the caller's H (src/run.jl:330 zeroes the sub-diagonal entry behind them; an exact breakdown
  // leaves the same pattern, and deflating against an invariant subspace is just as valid).  Deflated: the LEADING locked columns
  // whose eigenvalue exceeds KS_DEFLATE_RATIO (1.5) times the largest Ritz value of the rest -- a chain scaled by 1 / max|rest|
  // multiplies a component along such a vector by that ratio per step ...
  // Locked columns of NON-dominant eigenvalues (every :SR / :SM problem; the headline) are left alone: their components shrink.
  // `ex`: the eigenvalues of the deflated columns (no shift is placed there).  Several ranks: H and the Ritz values are replicated,
  // every rank takes the same plan; the dot products are all-reduced (one more collective per product: ks_block.hpp).
  // ... and for which that growth matters over the block at hand: ratio^(steps - 1) above 1e3.  (The component does not start at
  // the locking tolerance: for a non-normal A the product A z has an O(1) component along a locked Schur vector although z is
  // orthogonal to it -- the coupling R12 of the Schur form.)  Calibration: a Perron eigenvalue 2.1 x the bulk (config 3, blocks of 9:
  // 2.1^8 = 350) does no harm, and deflating against it costs two launches per product, the fused rotation and the speculative
  // chain (measured: +22 % on config 3's whole solve); eigenvalues 25 x the bulk abandon blocks of 4 (25^3 = 1.6e4).
  int defl_plan(const ks::Mat<T>& H, int j0, std::vector<std::complex<double>>& ex, int steps) {
    ex.clear();
    static const int dbg = env_int("KS_DEFLATE_DEBUG", 0);
    if (dbg) std::fprintf(stderr, "[deflate] j0 %d on %d dist %d ritz_valid %d nritz %d sstep_eff %d steps %d\n", j0, (int)ws->defl_on, (int)ws->ctx->distributed(), (int)ws->ritz_valid, (int)ws->ritz.size(), ws->sstep_eff, steps);
    if (!ws->defl_on || !ws->ritz_valid || ws->ritz.empty()) return 0;
    static const double ratio = [] { const char* e = std::getenv("KS_DEFLATE_RATIO"); const double v = e ? std::atof(e) : 1.5; return v > 1.0 ? v : 1.5; }();
    int nl = 0;
    for (int j = 1; j <= j0 - 2; ++j)
      if (H(j, j - 1) == T(0)) nl = j;
    if (nl == 0) return 0;
    using C = std::complex<double>;
    std::vector<C> lam(nl);
    std::vector<int> width(nl, 1);
    for (int j = 0; j < nl;) {
      if (j + 1 < nl && H(j + 1, j) != T(0)) {   // 2 x 2 block of the real Schur form
        const C a = C(H(j, j)), b = C(H(j, j + 1)), c = C(H(j + 1, j)), d = C(H(j + 1, j + 1));
        const C tr2 = 0.5 * (a + d), disc = std::sqrt(tr2 * tr2 - (a * d - b * c));
        lam[j] = tr2 + disc; lam[j + 1] = tr2 - disc;
        width[j] = 2; width[j + 1] = 0;
        j += 2;
      } else {
        lam[j] = C(H(j, j));
        ++j;
      }
    }
    double rest = 0.0;
    for (auto z : ws->ritz) {
      if (!std::isfinite(std::abs(z))) continue;
      bool locked = false;
      for (int j = 0; j < nl; ++j) locked = locked || std::abs(z - lam[j]) <= 1e-6 * std::max(1.0, std::abs(lam[j]));
      if (!locked) rest = std::max(rest, std::abs(z));
    }
    if (dbg) std::fprintf(stderr, "[deflate] locked columns %d, largest other Ritz value %.3g, |lambda_0| %.3g\n", nl, rest, std::abs(lam[0]));
    if (!(rest > 0.0)) return 0;
    int nd = 0;
    for (int j = 0; j < nl;) {
      const int w = width[j] == 2 ? 2 : 1;
      const double mag = w == 2 ? std::max(std::abs(lam[j]), std::abs(lam[j + 1])) : std::abs(lam[j]);
      const double growth = std::pow(mag / rest, std::max(1, steps - 1));
      if (!(mag > ratio * rest) || !(growth > 1e3) || nd + w > ksd::kDeflMax) break;
      for (int q = 0; q < w; ++q) ex.push_back(lam[j + q]);
      nd += w;
      j += w;
    }
    return nd;
  }

  // The first products of the NEXT expansion's Newton chain, behind the batch that just went into the stream (see
  // ks_workspace::spec_valid).  Only where the restart that follows can leave its rotation pending (the library's own drivers,
  // blocks of >= 8) and the operator's product is enqueued without host participation.
  // (Measured and withdrawn, round 5: the same products on a stream of their own behind the last second pass, next to the block's
  // final reduction + algebra kernel -- no gain on any configuration, the cross-stream waits cost what the overlap buys.)
  void spec_enqueue(const ksd::BlkShifts<D>& sh, int k_now, bool from_true) {
    if (!ws->spec_on || !ws->rot_defer_on || !ws->gate_allowed || ws->sstep_eff < 8 || !op->async_capable || ws->ctx->hc.allreduce != nullptr) return;
    if (from_true && (!ws->true_start_on || ws->maxdim + 1 > 40)) return;
    // as many products as the next first block will certainly have: it starts from about as many columns as this one did (Float64:
    // one more is tolerated -- a 2 x 2 block of the real Schur form kept whole; ComplexF64 restarts keep exactly mindim columns)
    const int ne = std::min(10, ws->maxdim - k_now - (sizeof(D) == 8 ? 1 : 0));
    if (ne < 2) return;
    if (!ensure_zscratch<D>(ws)) return;
    if (from_true && !true_start_enqueue<D>(ws)) return;
    char* zs = static_cast<char*>(ws->zscratch);
    op->shift_store_cacheable = true;
    for (int i = 0; i < ne; ++i) {
      op->in_scale = 1.0;
      const void* src = i == 0 ? (from_true ? static_cast<const void*>(ws->ztrue) : ws->col(ws->maxdim)) : static_cast<const void*>(zs + (size_t)(i - 1) * ws->ld * sizeof(D));
      double tre, tim;
      if constexpr (sizeof(D) == 8) { tre = sh.theta[i]; tim = 0.0; }
      else { tre = sh.theta[i].x; tim = sh.theta[i].y; }
      op->apply_shifted(src, zs + (size_t)i * ws->ld * sizeof(D), tre, tim, sh.sigma[i], ws->ld, ws->st);
    }
    ws->spec_sh.assign(reinterpret_cast<const char*>(&sh), reinterpret_cast<const char*>(&sh) + sizeof(sh));
    ws->spec_ne = ne;
    ws->spec_true = from_true;
    ws->spec_valid = true;
  }

  // The factorisation of `from` columns was vouched for by a caller that ran the restart itself (the seam: the reference's
  // _partialschur on a device basis, src/run.jl:298-365).  The library's own restart measures what its truncation drops
  // (note_ritz below); here that restart was not seen, so the relation is MEASURED before blocks lean on it: a truncation
  // that cut a 2 x 2 block of the real Schur form (imaginary-part targets on real operators, src/run.jl:321-326 does not pair
  // the members) drops the sub-diagonal entry of the LAST kept column -- residual of that column's relation,
  //   r = A v_c - V[:, 0:from) H[0:from, c],  c = from - 2,
  // one operator product into the dead column `from` plus a strided row sample of r (a few MB of traffic): > relation_tol
  // ||H||_F switches the blocks off for the run, exactly as a leaking restart of the library's own driver does.
  void relation_probe(int from, const ks::Mat<T>& H) {
    const int c = from - 2, nc = from;
    hipStream_t s_ = ws->ctx->stream;
    D* scratch = static_cast<D*>(ws->col(from));
    op->in_scale = 1.0;
    op->apply(ws->col(c), ws->col(from), nullptr);
    // buffers of its own (nothing the expansion kernels keep state in is touched): [2 doubles | nc coefficients]
    if (!ws->probe_dev) {
      KS_HIP(hipMalloc(&ws->probe_dev, 16 + (size_t)(ws->maxdim + 2) * 16));
      KS_HIP(hipHostMalloc(&ws->probe_host, 16 + (size_t)(ws->maxdim + 2) * 16, hipHostMallocCoherent | hipHostMallocMapped));   // (polled while the stream runs)
    }
    double* out = static_cast<double*>(ws->probe_dev);
    D* coef_d = reinterpret_cast<D*>(static_cast<char*>(ws->probe_dev) + 16);
    D* coef_h = reinterpret_cast<D*>(static_cast<char*>(ws->probe_host) + 16);
    double fro2 = 0.0;
    for (int j = 0; j < from - 1; ++j)
      for (int i = 0; i < from; ++i) fro2 += std::norm(std::complex<double>(H(i, j)));
    for (int i = 0; i < nc; ++i) std::memcpy(&coef_h[i], &H(i, c), sizeof(D));
    KS_HIP(hipMemcpyAsync(coef_d, coef_h, (size_t)nc * sizeof(D), hipMemcpyHostToDevice, s_));
    KS_HIP(hipMemsetAsync(out, 0, 2 * sizeof(double), s_));
    const int64_t nchunks = (ws->n + 15) / 16;
    const int64_t stride = std::max<int64_t>(1, std::min<int64_t>(64, nchunks / 16384));
    const int64_t sampled = (nchunks + stride - 1) / stride * 16;
    const int grid = (int)((sampled + kBlock - 1) / kBlock);
    ksd::k_relation_probe<D><<<grid, kBlock, 0, s_>>>(static_cast<const D*>(ws->V), ws->ld, nc, scratch, coef_d, ws->n, stride, out);
    KS_HIP(hipGetLastError());
    if (ws->ctx->distributed()) ws->ctx->allreduce(out, 2);
    double* res = static_cast<double*>(ws->probe_host);
    KS_HIP(hipMemcpyAsync(res, out, 2 * sizeof(double), hipMemcpyDeviceToHost, s_));
    KS_HIP(hipStreamSynchronize(s_));
    ws->ctx->check_comm();
    ws->relation_probes++;
    const double fro = std::sqrt(fro2);
    const double leak = res[1] > 0.0 ? std::sqrt(res[0] * ((double)ws->n_global / res[1])) : 0.0;
    // (a MEASURED residual carries the rounding of the column's whole history -- 3e-13 .. 1.1e-12 ||H||_F behind blocks of 13-20,
    // where the library's own restart compares an exactly known dropped entry: ten times its tolerance here; a cut 2 x 2 block
    // shows up at 1e-9 .. 1e-5)
    if (!(leak <= 10.0 * ws->relation_tol * fro)) {   // (NaN counts as a break)
      ws->relation_breaks++;
      ws->relation_leak = std::max(ws->relation_leak, fro > 0.0 ? leak / fro : leak);
      ws->sstep_eff = 0;
    }
  }

  // Drift watch (ks_workspace::probe_col): residual of column c = from - 2 of the relation this batch leaned on,
  //   r = A v_c - V[:, 0:from) H[0:from, c]
  // (columns below `from` are ordinary and untouched by the batch, H[:, c] is the caller's), enqueued behind the batch.
  void drift_probe_enqueue(int from, const ks::Mat<T>& H) {
    const int c = from - 2, nc = from;
    hipStream_t s_ = ws->ctx->stream;
    if (!ws->probe_dev) {
      KS_HIP(hipMalloc(&ws->probe_dev, 16 + (size_t)(ws->maxdim + 2) * 16));
      KS_HIP(hipHostMalloc(&ws->probe_host, 16 + (size_t)(ws->maxdim + 2) * 16, hipHostMallocCoherent | hipHostMallocMapped));   // (polled while the stream runs)
    }
    if (!ws->probe_col) {
      if (hipMalloc(&ws->probe_col, (size_t)ws->ld * sizeof(D)) != hipSuccess) {   // (no room for one more column: no watch, no failure)
        (void)hipGetLastError();
        ws->probe_col = nullptr;
        return;
      }
      KS_HIP(hipMemsetAsync(ws->probe_col, 0, (size_t)ws->ld * sizeof(D), s_));
    }
    D* scratch = static_cast<D*>(ws->probe_col);
    op->in_scale = 1.0;
    op->apply(ws->col(c), scratch, ws->st);
    if (!ws->ctx->distributed()) {
      // one launch, no copies: coefficients by value, the sums written to the pinned words by the last workgroup
      if (!ws->watch_acc) {
        KS_HIP(hipMalloc(&ws->watch_acc, 32));
        KS_HIP(hipMemsetAsync(ws->watch_acc, 0, 32, s_));
        KS_HIP(hipHostGetDevicePointer(&ws->probe_host_dev, ws->probe_host, 0));
      }
      ksd::ProbeCoef<D> pc;
      double fro2w = 0.0;
      for (int j = 0; j < from - 1; ++j)
        for (int i = 0; i < from; ++i) fro2w += std::norm(std::complex<double>(H(i, j)));
      for (int i = 0; i < nc; ++i) std::memcpy(&pc.c[i], &H(i, c), sizeof(D));
      const int64_t nchunks_w = (ws->n + 15) / 16;
      const int64_t stride_w = std::max<int64_t>(1, std::min<int64_t>(256, nchunks_w / 4096));   // (~65 000 sampled rows)
      const int64_t sampled_w = (nchunks_w + stride_w - 1) / stride_w * 16;
      const int grid_w = (int)((sampled_w + kBlock - 1) / kBlock);
      double* hres = static_cast<double*>(ws->probe_host);
      hres[0] = 0.0; hres[1] = 0.0; hres[2] = 0.0;   // (a batch that breaks down leaves them: no measurement)
      ws->rp_seq += 1.0;
      ksd::k_relation_watch<D><<<grid_w, kBlock, 0, s_>>>(static_cast<const D*>(ws->V), ws->ld, nc, scratch, pc, ws->n, stride_w, static_cast<double*>(ws->watch_acc),
                                                         static_cast<double*>(ws->probe_host_dev), ws->rp_seq, ws->st);
      KS_HIP(hipGetLastError());
      ws->rp_fro = std::sqrt(fro2w);
      ws->rp_inflight = true;
      return;
    }
    double* out = static_cast<double*>(ws->probe_dev);
    D* coef_d = reinterpret_cast<D*>(static_cast<char*>(ws->probe_dev) + 16);
    D* coef_h = reinterpret_cast<D*>(static_cast<char*>(ws->probe_host) + 16);
    double fro2 = 0.0;
    for (int j = 0; j < from - 1; ++j)
      for (int i = 0; i < from; ++i) fro2 += std::norm(std::complex<double>(H(i, j)));
    for (int i = 0; i < nc; ++i) std::memcpy(&coef_h[i], &H(i, c), sizeof(D));
    KS_HIP(hipMemcpyAsync(coef_d, coef_h, (size_t)nc * sizeof(D), hipMemcpyHostToDevice, s_));
    KS_HIP(hipMemsetAsync(out, 0, 2 * sizeof(double), s_));
    const int64_t nchunks = (ws->n + 15) / 16;
    const int64_t stride = std::max<int64_t>(1, std::min<int64_t>(64, nchunks / 16384));
    const int64_t sampled = (nchunks + stride - 1) / stride * 16;
    const int grid = (int)((sampled + kBlock - 1) / kBlock);
    ksd::k_relation_probe<D><<<grid, kBlock, 0, s_>>>(static_cast<const D*>(ws->V), ws->ld, nc, scratch, coef_d, ws->n, stride, out);
    KS_HIP(hipGetLastError());
    if (ws->ctx->distributed()) ws->ctx->allreduce(out, 2);
    KS_HIP(hipMemcpyAsync(ws->probe_host, out, 2 * sizeof(double), hipMemcpyDeviceToHost, s_));
    // (several ranks: no sequence word travels with the sums -- an event behind the copy is what the collection waits for; a restart
    // whose rotation went through an armed gate does not synchronise the stream, and zeros read too early would pass for "calm")
    if (!ws->rp_event) KS_HIP(hipEventCreateWithFlags(&ws->rp_event, hipEventDisableTiming));
    KS_HIP(hipEventRecord(ws->rp_event, s_));
    ws->rp_fro = std::sqrt(fro2);
    ws->rp_inflight = true;   // (read at the start of the next expansion)
  }
  // (after the batch's results arrived: the copy above completed before the publication behind it started)
  void drift_probe_collect() {
    if (!ws->rp_inflight) return;
    ws->rp_inflight = false;
    const double* res = static_cast<const double*>(ws->probe_host);
    if (!ws->ctx->distributed()) {
      // (written by the last workgroup of k_relation_watch; a batch that stopped early, or a stream that was not synchronised in
      // between -- a caller that went straight into the next expansion --, leaves the sequence word behind: no measurement)
      // (the watch sits in front of the speculative chain and takes one product + a row sample: by the time the host has run its
      // restart step it has normally finished; where the host was quicker, a bounded wait -- the decision this measurement feeds
      // must not slip a cycle: a drift grows 20-90x per cycle)
      const volatile double* seqw = &res[2];
      const double t_w = ks::now_s();
      while (*seqw != ws->rp_seq && ks::now_s() - t_w < 2e-3) {}
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      if (*seqw != ws->rp_seq) return;
    } else {
      // WAIT for the copy (it was enqueued a whole restart ago: normally long done).  Not a query: whether a rank sees the sums
      // must not depend on its timing -- the ranks take every decision alike (a rank that skipped a measurement would skip the
      // next watch's all-reduce too, and the others would wait for it forever)
      if (!ws->rp_event) return;
      KS_HIP(hipEventSynchronize(ws->rp_event));
      if (ws->st_h->breakdown >= 0 || ws->st_h->blk_bail >= 0) return;
    }
    const double fro = ws->rp_fro;
    const double leak = res[1] > 0.0 ? std::sqrt(res[0] * ((double)ws->n_global / res[1])) : 0.0;
    ws->relation_probes++;
    ws->rp_done++;
    const double rel = fro > 0.0 ? leak / fro : leak;
    static const int watch_dbg = env_int("KS_DEFER_DEBUG", 0);
    if (watch_dbg) std::fprintf(stderr, "[watch] probe %d rel %.3e (leak %.3e fro %.3e) every %d prov_k %d\n", ws->rp_done, rel, leak, fro, ws->rp_every, ws->prov_k);
    // (the restart's locking drops couplings of up to tol |lambda| per converged pair, by design -- src/run.jl:330 --, and the
    // kept columns inherit them: 2.8e-11 ||H||_F at tol = 1e-12 on the miniature of config 2; a drift passes any level within a
    // few cycles)
    const double limit = std::max(1e-10, 30.0 * ws->watch_tol), quiet = std::max(2e-12, 0.1 * ws->watch_tol);
    if (!(rel <= limit)) {   // (NaN counts as a break)
      ws->relation_breaks++;
      ws->relation_leak = std::max(ws->relation_leak, rel);
      ws->sstep_eff = 0;
    }
    const bool calm = rel <= quiet && !(rel > 4.0 * ws->rp_last && rel > 5e-13);
    ws->rp_every = (ws->rp_done >= 4 && calm) ? 2 : 1;
    ws->rp_last = rel;
  }

  // Ritz values of the restart that just happened (Newton shifts of the next expansion's blocks)
  void note_ritz(const cplx* lams, int m, double leak = 0.0, double fro = 0.0, double tol = 0.0) override {
    ws->watch_tol = tol > 0.0 && std::isfinite(tol) ? tol : 0.0;   // (what the restart's locking may legitimately drop: scale of the drift watch)
    if (leak > ws->relation_tol * fro) {
      // the restart cut through a 2 x 2 block (RestartResult::leak): the relation of the kept columns is violated by `leak`.
      // Blocks off for the rest of this run (a new start vector re-arms them); counted for ks_workspace_relation_info.
      ws->relation_breaks++;
      ws->relation_leak = std::max(ws->relation_leak, fro > 0.0 ? leak / fro : leak);
      ws->sstep_eff = 0;
    }
    if (ws->sstep < 2) return;
    ws->ritz.assign(lams, lams + m);
    ws->ritz_valid = true;
    ws->hfull_valid = false;
  }
  // ... or, when nobody told us, the eigenvalues of the Hessenberg matrix the last full expansion left (one Schur
  // factorisation of a copy on the host, ~0.1 ms: only on the path of callers that run their own restart)
  void blk_shifts_from_saved_H() {
    using TT = T;
    if (ws->sstep_eff < 2 || ws->ritz_valid || !ws->hfull_valid) return;
    const int m = ws->maxdim;
    std::vector<TT> Hc((size_t)(m + 1) * m), Qc((size_t)m * m, TT(0));
    std::memcpy(Hc.data(), ws->Hfull.data(), Hc.size() * sizeof(TT));
    for (int i = 0; i < m; ++i) Qc[i + (size_t)i * m] = TT(1);
    ks::Mat<TT> Hm(Hc.data(), m + 1, m, m + 1), Qm(Qc.data(), m, m, m);
    try {
      ks::local_schurfact(Hm.top(m), 0, m - 1, Qm);
      std::vector<cplx> lams(m);
      ks::copy_eigenvalues(lams.data(), Hm, 0, m - 1);
      ws->ritz = lams;
      ws->ritz_valid = true;
    } catch (...) {
      ws->ritz_valid = false;   // (QR did not converge on the copy: the expansion simply runs step by step)
    }
    ws->hfull_valid = false;
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
      ws->rot_defer_ok = own && r > 0;   // (may stay pending for the next expansion's fused first pass: ks_workspace::rot_pending)
      rotate_tfold<T>(ws, c0, c, r, r > 0 ? &Q(c0, c0) : nullptr, Q.ld, c0, src, dst);
      ws->rot_defer_ok = false;
      const bool pend = ws->rot_pending;
      reset_lazy(ws);
      if (pend) { ws->t_lazy = true; ws->ntrue = ws->rot_out0 + ws->rot_rr; ws->t_hi = ws->ntrue - 1; }
    } else {
      rotate(c0, c, r, Q);
      col_copy(dst, src);
    }
    if (own) prov_set(ws, dst);  // (the host step rewrote H: new shadow)
    else prov_drop(ws);
  }
};

// While alive, expansions that announce a restart (iterate_arnoldi_early) may pre-enqueue its rotation behind the gate; on
// exit -- normal or by exception (QR failure in the host step, operator error) -- a gate still waiting is cancelled.
struct GateScope {
  ks_workspace* ws;
  explicit GateScope(ks_workspace* w) : ws(w) { ws->gate_allowed = true; }
  ~GateScope() {
    ws->gate_allowed = false;
    gate_cancel(ws);
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

// The probe of a workspace that will run the BLOCK expansion: the second pass of one large block in place (k existing columns,
// s new ones: 8 n (k + 2 s) bytes, zeros in, zeros out).  This is the launch whose time is BIMODAL with the basis' allocation --
// 850-870 against 950-1000 us at n = 1e7, k = 21, s = 20, the same in place and from the scratch columns, unchanged by a fresh
// allocation of the scratch columns, different between two allocations of the basis in one process
// (tools/bupdate_lottery.py, profiles/r06_bupdate_lottery.txt) -- while the per-step launches above do not tell candidates apart.
template <class D> double placement_block_ms(ks_workspace* w, int k, int s, hipEvent_t a, hipEvent_t b) {
  ks_ctx* c = w->ctx;
  blk_ensure_buffers(w);
  reset_state(w);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {  // rep 0 warms up
    KS_HIP(hipEventRecord(a, c->stream));
    launch_blk<D>(w, 1, k, s);
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
  // Round 6: ON (three candidates) for a workspace that runs the block expansion in Float64 and has a block kernel for the
  // probe's shape -- there the candidates DO differ (placement_block_ms) and the search costs three allocations of the basis held
  // for a few milliseconds; everything else as before (KS_PLACE_TRIALS=1 switches it off, >= 2 forces it).
  const int blk_k = w->maxdim / 2 + 1, blk_s = w->maxdim + 1 - blk_k;
  const bool blk_probe = sizeof(D) == 8 && w->sstep >= 8 && blk_s >= 8 && blk_shape_ok(w->dtype, blk_k, blk_s) && !w->ctx->distributed();
  static const int trials_env = env_int("KS_PLACE_TRIALS", -1);
  // (how often a fresh allocation lands in the fast cluster depends on the box: 13 of 18 processes on one, 1 of 18 candidates on
  // another -- up to ten candidates (held for ~20 ms each), and the search ends with the first one that streams at the fast cluster's rate)
  const int trials = trials_env >= 0 ? trials_env : (blk_probe ? 10 : 1);
  const double blk_fast_ms = 8.0 * (double)w->n * (blk_k + 2 * blk_s) / 5.5e12 * 1e3;   // 5.5 TB/s: 0.86-0.89 ms against 0.95-1.00 at n = 1e7
  // measured: +3 % at 3.3 GB, +1.5 % at 1.6 GB, nothing at 0.8 GB, -2 % at 0.4 GB (there the calibration, which
  // revisits the same columns, sees the memory-side cache more than the placement)
  static const int min_mb = env_int("KS_PLACE_MIN_MB", 1024);
  static const int budget_ms = env_int("KS_PLACE_BUDGET_MS", 1500);
  static const int max_x = std::max(2, env_int("KS_PLACE_MAX_X", 10));  // total footprint of held candidates / basis size
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
    const double ms = blk_probe ? placement_block_ms<D>(w, blk_k, blk_s, ev.a, ev.b) : placement_trio_ms<D>(w, ev.a, ev.b);
    if (debug) std::fprintf(stderr, "[ks] placement candidate %zu @%p: %.3f ms\n", k, pc.cand[k], ms);
    if (ms < best_ms) { best_ms = ms; pc.keep = k; }
    worst_ms = std::max(worst_ms, ms);
    const double spent = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if ((int)pc.cand.size() >= trials || spent > budget_ms) break;
    if ((int)pc.cand.size() + 1 > max_x) break;                      // footprint cap: held candidates <= max_x * V
    // (the step-kernel probe's rule; NOT with the block probe, whose slow cluster alone spreads over 5 %: four slow candidates at
    // 0.951 .. 0.999 ms ended the search of a bench run in the slow cluster -- 6 304 it/s against 6 570 .. 6 720 for the processes
    // behind it on the same box, profiles/r06c_bench*.json)
    if (!blk_probe && pc.cand.size() >= 4 && best_ms <= 0.955 * worst_ms) break;   // a candidate from the fast cluster was found
    if (blk_probe && (best_ms <= blk_fast_ms || (pc.cand.size() >= 2 && best_ms <= 0.93 * worst_ms))) break;   // (block probe: the two clusters are 10-15 % apart)
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

