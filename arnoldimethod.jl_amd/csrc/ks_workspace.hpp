// the workspace (ArnoldiWorkspace, src/ArnoldiMethod.jl:41-93), kernel launch helpers, the asynchronous expansion (src/expansion.jl), verbs, rotations
// Part of the ONE translation unit of libkschur_hip.so: included by ks_hip.hip, in this order --
//     ks_context.hpp -> ks_operators.hpp -> ks_workspace.hpp -> ks_backend.hpp -> (C ABI in ks_hip.hip)
// -- and not meant to be included on its own (needs ks_operators.hpp).
#pragma once
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
  bool v_nt = true;             // streaming (non-temporal) loads of V; false when the basis fits the memory-side cache (ld_v, ks_kernels.hpp)
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
  // the provenance in force was ASSERTED by the caller (ks_workspace_assert_arnoldi after a restart the caller ran): the
  // library has not seen that restart, in particular not what its truncation dropped (relation_breaks below).  Before such
  // a factorisation is expanded in blocks the relation of its last column is measured (HipBackend::relation_probe).
  bool prov_vouched = false;
  int relation_probes = 0;      // probes run so far (ks_workspace_relation_probes)
  // DEFERRED RESTART ROTATION (Float64).  The rotation of a library-run restart (rotate_tfold) is not launched at once: its
  // coefficient matrix T Q goes to the device and the rotation stays PENDING, because the expansion that normally follows can do
  // it in the same sweep as the first pass of its first block (k_brotdots_mfma: the rotated columns never travel back in).
  // While pending the workspace reports itself T-lazy with an empty lazy range, so that every reader's materialize() call
  // flushes it (rot_flush: the ordinary rotation kernel) first.  rot_fuse: the expansion being enqueued took it over.
  bool rot_defer_on = true, spec_on = true;   // KS_ROT_DEFER / KS_SPEC_CHAIN at creation
  bool rot_pending = false, rot_fuse = false;
  bool rot_split = false;       // rot_fuse without a fused kernel for the shape: ordinary rotation, then both passes read the chain from scratch columns
  int rot_split_count = 0;
  int rot_cin = 0, rot_rr = 0, rot_out0 = 0;
  int rot_fused_count = 0;      // rotations done by the fused kernel (ks_workspace_fused_rotations)
  bool rot_defer_ok = false;    // set by the library's restart drivers around their rotate_and_move (never by the verbs)
  void* zscratch = nullptr;     // device: ld x kBlkSMax elements, the Newton chain of a block whose first pass is fused
  // TRUE START of a chain (round 6).  The chain of a fused / speculative block starts from the STORED last column, which is the
  // true residual direction only up to the Gram deviation of the block that wrote it.  Behind a block whose deviation is above
  // rounding level (real shifts on a complex spectrum: 1e-11 .. 1e-10 every time) the true column S T[:, maxdim] is formed once
  // (one product of maxdim + 1 columns with a column of the device-resident T, into the scratch column `ztrue`) and the chain
  // starts from it: the rotation may stay pending behind ANY accepted block.
  // IN-CHAIN DEFLATION (ks_block_kernels.hpp: kDeflMax; HipBackend::defl_plan)
  bool defl_on = true;          // KS_CHAIN_DEFLATE at creation
  void* defl_part = nullptr;    // device: kDeflMax x 1024 partial sums
  int defl_blocks = 0;          // blocks whose chain was deflated (ks_workspace_sstep_info: diag3[...] no -- see ks_workspace_deflated_blocks)
  int defl_last = 0;            // columns the last block batch deflated against
  void* ztrue = nullptr;        // device: ld elements, allocated on first use
  bool true_start_on = false;   // KS_TRUE_START at creation (default 0: measured slower on config 3)
  bool ztrue_valid = false;     // ztrue holds S T[:, maxdim] of the basis as it stands (no batch, rotation or reader since)
  bool rot_true_start = false;  // the pending rotation was granted on condition that the chain starts from ztrue
  bool spec_true = false;       // the speculative chain in the scratch columns started from ztrue
  bool chain_true = false;      // the adopted first block starts its chain from ztrue (consumed by enqueue_steps_blk)
  int true_starts = 0;          // products S T[:, maxdim] formed so far
  // SPECULATIVE CHAIN (the host step off the critical path).  When an expansion that ends at maxdim ran in blocks,
  // the first spec_ne products of the NEXT expansion's Newton chain are enqueued right behind it -- before the host has even
  // received H: they need the stored last column (the chain's start, see rot_pending) and shifts, for which the Ritz values of
  // the restart BEFORE the one about to happen are taken (one restart staler than otherwise; tests/sstep_model.py: the
  // conditioning of the Newton basis does not notice as long as ALL shifts of a block come from the same Leja sequence).  The
  // device works on them while the host runs the Schur / reorder / restore step.  The next expansion adopts them (and the
  // shift sequence they were made with) if its first block is a fused one of at least spec_ne steps; anything else that
  // touches V in between drops them.  spec_sh: the BlkShifts<double> they were made with (raw bytes).
  bool spec_valid = false;
  int spec_ne = 0;
  std::vector<char> spec_sh;
  int spec_backoff = 0;                 // cycles to go without speculation after a dropped one
  int spec_backoff_len = 1;             // ... how many that will be next time (doubles on consecutive drops up to 8)
  int spec_adopt = 0;                   // products the block being enqueued adopts (set by the fuse decision, consumed by enqueue_steps_blk)
  int spec_used = 0, spec_wasted = 0;   // speculations adopted / dropped (diagnostics)
  int mindim_hint = 0;                  // mindim of the restart driver in charge (0: unknown -> no speculation)
  void* probe_dev = nullptr;    // device / pinned host scratch of the probe: [sum, rows | coefficients]
  void* probe_host = nullptr;
  // DRIFT WATCH of block runs (HipBackend::drift_probe_enqueue / drift_probe_collect).  The H recovery of a block expresses
  // A q_j through the relation of the EARLIER columns (A V_k = V_{k+1} H_k) with O(1) coefficients; where the new directions are
  // small against the block's components in the old basis and the restart does not damp -- a non-normal operator with hundreds of
  // eigenvalues within 1e-3 of the wanted end, tests/test_gpu_random_stress.py seed 17 -- an error of that relation GROWS from
  // cycle to cycle (1e-15 -> 1e-8 in 12 cycles with blocks of 10 or more; stable with blocks of <= 8 and step by step), and
  // nothing in a block's own diagnostics shows it.  So the relation is measured: the residual of the last kept column of the
  // factorisation the batch started from, one product + a strided row sample enqueued BEHIND the batch (no synchronisation:
  // the result rides back with H), every cycle for the first four block cycles and whenever the residual is above 2e-12 ||H||_F
  // or growing, every second cycle otherwise; above max(1e-10, 30 tol) ||H||_F the blocks go off for the run
  // (ks_workspace_relation_info).
  void* probe_col = nullptr;    // device: one column (the product of the watch)
  void* watch_acc = nullptr;    // device: [sum, rows, ticket] of k_relation_watch
  void* probe_host_dev = nullptr;   // device address of probe_host
  int rp_every = 1, rp_count = 0, rp_done = 0;
  // recorded behind every upload of the rotation coefficients from the pinned stage (rotate_tfold): what the NEXT rotation has to
  // wait for before it rewrites the stage -- not the whole stream, which by then holds the speculative chain of the next expansion
  hipEvent_t qstage_evt = nullptr;
  bool qstage_marked = false;
  hipEvent_t rp_event = nullptr;   // several ranks: recorded behind the copy of a watch's sums (drift_probe_collect asks it)
  bool rp_inflight = false;
  double rp_fro = 0.0, rp_last = 0.0, rp_seq = 0.0;
  double watch_tol = 0.0;       // convergence tolerance of the driver that ran the last restart (note_ritz), 0: not known
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
  // S-STEP (block) expansion (ks_block.hpp): s steps per block, two passes over the basis per BLOCK.  0 / 1: off.
  int sstep = 0;                // KS_SSTEP at creation, ks_workspace_set_sstep afterwards
  double blk_pivmin = 1e-6;     // smallest Cholesky pivot ratio d_i / G_ii a block may have (below: abandoned, steps redone one by one)
  double blk_gdevmax = 1e-8;    // largest entry of |Gram matrix of the written block - I| a block may have (~ eps cond(R_1)^2)
  int sstep_eff = 0;            // block size in force: lowered when blocks are abandoned (ill-conditioned Newton basis), raised
  int blk_clean = 0;            //   again after blk_clean consecutive clean batches
  std::vector<std::complex<double>> ritz;  // Ritz values of the last restart (Newton shifts of the next expansion)
  bool ritz_valid = false;
  std::vector<char> Hfull;      // host H as the last full expansion left it (shifts for callers that run their own restart)
  bool hfull_valid = false;
  void* bpart = nullptr;        // device: partial sums of the block kernels, [entry][workgroup]
  void* bred = nullptr;         // device: reduced entries
  void* bscr = nullptr;         // device: ksd::BlkScratch
  void* bzero = nullptr;        // device: 256 zero bytes (ring kernels copy them into the packs past a workgroup's rows)
  bool blk_tail = false;        // the T-lazy columns were produced by blocks (a batch continuing on them starts a new T)
  double blk_diag[3] = {1.0, 1.0, 0.0};  // of the last batch: worst pivot ratio of stage 1 / stage 2, largest |G_t - I| entry
  int blk_count = 0, blk_bails = 0;      // blocks completed / abandoned since creation
  // restarts that cut through a 2 x 2 block of the real Schur form (ks::RestartResult::leak > relation_tol ||H||_F): the
  // Arnoldi relation of the kept columns is violated from there on, blocks stay off for the rest of the run
  int relation_breaks = 0;
  double relation_leak = 0.0;            // largest leak / ||H||_F seen
  double relation_tol = 1e-12;
  // REVERSE MAILBOX (k_rot_gate, ks_kernels.hpp): the restart rotation pre-enqueued behind a gate the host releases
  ksd::RotGate* gate_h = nullptr;      // pinned host (coherent)
  ksd::RotGate* gate_hd = nullptr;     // the same memory through its device pointer
  ksd::RotGate* gate_d = nullptr;      // device copy the gated rotation reads
  void* Qstage_dev = nullptr;          // device pointer of the pinned Qstage
  bool gate_allowed = false;           // a driver entry point that always follows the expansion by the restart is running
  bool gate_armed = false;
  uint64_t gate_seq = 0;
  int gate_cin = 0, gate_rmax = 0;     // shape the armed rotation was launched for
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
    if (gate_armed && gate_h) {  // never leave a gate waiting, and never free what the rotation behind it reads: cancel, drain
      __atomic_store_n(&gate_h->flag, (gate_seq << 1) | 1u, __ATOMIC_RELEASE);
      gate_armed = false;
      (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(Vbase ? Vbase : V); (void)hipHostFree(H); (void)hipHostFree(Q); (void)hipFree(Hd); (void)hipHostFree(Hstage);
    (void)hipFree(Hscratch); (void)hipFree(partial); (void)hipFree(partial_s); (void)hipFree(partial2); (void)hipFree(coef); (void)hipFree(red);
    (void)hipFree(scal); (void)hipHostFree(scal_h); (void)hipHostFree(coef_h);
    if (zscratch) (void)hipFree(zscratch);
    if (ztrue) (void)hipFree(ztrue);
    if (defl_part) (void)hipFree(defl_part);
    if (rp_event) (void)hipEventDestroy(rp_event);
    if (qstage_evt) (void)hipEventDestroy(qstage_evt);
    if (probe_col) (void)hipFree(probe_col);
    if (watch_acc) (void)hipFree(watch_acc);
    if (probe_dev) (void)hipFree(probe_dev);
    if (probe_host) (void)hipHostFree(probe_host);
    (void)hipFree(Qd); (void)hipHostFree(Qstage); (void)hipFree(tmp); (void)hipFree(tmp2); (void)hipFree(oop);
    (void)hipHostFree(Hstage_early); (void)hipHostFree(mbox); (void)hipFree(ctr);
    (void)hipFree(bpart); (void)hipFree(bred); (void)hipFree(bscr); (void)hipFree(bzero);
    (void)hipHostFree(gate_h); (void)hipFree(gate_d);
  }
};

namespace {

inline void gate_cancel(ks_workspace* ws);  // (reverse mailbox, below: nothing may synchronise the stream behind an armed gate)

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
  return resident_blocks(ws->ctx, ksd::k_dots<D, NC4, 1, true>, 0, cache);
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
  if (ws->v_nt) ksd::k_dots<D, NC4, 1, true><<<nb, kBlock, 0, ws->ctx->stream>>>(V, ws->ld, jc, w, partial, ws->pnb, norm_slot, pass, st);
  else ksd::k_dots<D, NC4, 1, false><<<nb, kBlock, 0, ws->ctx->stream>>>(V, ws->ld, jc, w, partial, ws->pnb, norm_slot, pass, st);
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
  const int nb = cap_blocks(ws, resident_blocks(ws->ctx, ksd::k_axpy_dots_cs<D, NCW, U, WB, true>, 0, cache), 64 * U);
  // (cacheable loads of V only exist for the shallow-staging family: a basis that fits the memory-side cache has small columns)
  if constexpr (WB <= 8) {
    if (!ws->v_nt) {
      ksd::k_axpy_dots_cs<D, NCW, U, WB, false><<<nb, kBlock, 0, ws->ctx->stream>>>(static_cast<const D*>(ws->V), ws->ld, j, w,
                                                                                  static_cast<const D*>(ws->coef),
                                                                                  static_cast<D*>(ws->partial), ws->pnb, ws->partial2,
                                                                                  ws->st, defer, wdst, plain && ws->passes == 2);
      return nb;
    }
  }
  ksd::k_axpy_dots_cs<D, NCW, U, WB, true><<<nb, kBlock, 0, ws->ctx->stream>>>(static_cast<const D*>(ws->V), ws->ld, j, w,
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
  ws->blk_tail = false;
  if (!(ws->lazy_hi >= ws->lazy_lo)) return;
  for (int c = ws->lazy_lo; c <= ws->lazy_hi; ++c) ws->hostscale[c] = 1.0;
  ws->colscale_dirty = true;  // the device copy is only read inside expansion batches: uploaded with the next one's state
  ws->lazy_lo = 1 << 30;
  ws->lazy_hi = -1;
}
template <class D> void materialize_t(ks_workspace* ws);
inline void rot_flush(ks_workspace* ws);
inline void materialize(ks_workspace* ws) {
  gate_cancel(ws);
  rot_flush(ws);
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
  ws->prov_vouched = false;
  if (k < 0) return;
  ws->Hshadow.resize(h_bytes(ws));
  std::memcpy(ws->Hshadow.data(), ws->H, h_bytes(ws));
}
inline void prov_drop(ks_workspace* ws) { ws->prov_k = -1; ws->prov_vouched = false; }
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
  ws->st_h->blk_bail = -1;
  ws->st_h->blk_piv1 = ws->st_h->blk_piv2 = 1.0;
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
    // a NEW factorisation starts here: Ritz values of whatever this workspace solved before say nothing about it (they
    // would only be used as Newton shifts -- valid, but possibly ill-conditioned), and the block size starts afresh
    ws->ritz_valid = false;
    ws->hfull_valid = false;
    ws->sstep_eff = ws->sstep;
    ws->blk_clean = 0;
    ws->rp_inflight = false; ws->rp_every = 1; ws->rp_count = 0; ws->rp_done = 0; ws->rp_last = 0.0; ws->watch_tol = 0.0;   // (the drift watch starts afresh too)
    ws->defl_last = 0;                                    // (nothing is locked in a new factorisation)
    ws->ztrue_valid = false; ws->rot_true_start = false; ws->chain_true = false; ws->spec_true = false;
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
// cancel a pre-enqueued rotation (its gate is released with the cancel mark: the rotation behind it returns at once)
inline void gate_cancel(ks_workspace* ws) {
  if (!ws->gate_armed) return;
  __atomic_store_n(&ws->gate_h->flag, (ws->gate_seq << 1) | 1u, __ATOMIC_RELEASE);
  ws->gate_armed = false;
}
template <class D> void rotate_device(ks_workspace* ws, int c0, int c, int r, int out0 = 0, int extra_out = -1, bool gated = false) {
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
          KS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ksd::k_rotate_fma<CT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
          KS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ksd::k_rotate_fma<CT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
          int o = 0;
          KS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, ksd::k_rotate_fma<CT, true>, kBlock, (size_t)48 * CT * 8));
          occ = std::max(1, std::min(o, bpc_env));
        }
        const int nbr = cap_blocks(ws, ctx->num_cu * occ, kBlock);
        const ksd::RotGate* g = gated ? ws->gate_d : nullptr;
        if (ws->v_nt) ksd::k_rotate_fma<CT, true><<<nbr, kBlock, smem, s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out, g);
        else ksd::k_rotate_fma<CT, false><<<nbr, kBlock, smem, s>>>(Vc, ws->ld, c, r, Qd, c, out0, extra_out, g);
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
// Pre-enqueue the T-folded restart rotation of a Float64 workspace behind its gate (after the expansion batch and its
// publishing kernel are in the stream, before the host starts the Schur step).  The rotation of a restart always reads
// columns 0..maxdim (cin = maxdim + 1) and writes at most maxdim + 1 of them.
inline bool gate_arm(ks_workspace* ws) {
  static const int on = env_int("KS_ROT_GATE", 1);
  static const long long timeout_ticks = (long long)env_int("KS_ROT_GATE_TIMEOUT_S", 60) * 100000000LL;  // wall_clock64: 100 MHz
  const char* rot_env = std::getenv("KS_ROTATE");
  if (!on || !ws->gate_allowed || ws->gate_armed || !ws->gate_h || ws->dtype != KS_F64 || ws->maxdim + 1 > 64 || ws->ctx->profiling ||
      !ws->use_mbox || (rot_env && std::string(rot_env) != "fma") || env_int("KS_ROTATE_VALU", 0))
    return false;
  const int cin = ws->maxdim + 1;
  ws->gate_seq += 1;
  ksd::k_rot_gate<<<1, kBlock, 0, ws->ctx->stream>>>(ws->gate_hd, ws->gate_seq, ws->gate_d, static_cast<const double*>(ws->Qstage_dev),
                                                     static_cast<double*>(ws->Qd), timeout_ticks);
  ws->gate_cin = cin;
  ws->gate_rmax = cin;
  // (launched with the largest coefficient tile: r <= cin outputs)
  rotate_device<double>(ws, 0, cin, cin, 0, -1, /*gated=*/true);
  KS_HIP(hipGetLastError());
  ws->gate_armed = true;
  return true;
}

// the pinned stage of the rotation coefficients is about to be rewritten: wait for the upload that last read it.  One process, one
// rank: only for that upload (an event behind it) -- a hipStreamSynchronize here waited for the whole speculative chain of the next
// expansion, which is in the stream by the time the restart reaches its rotation, and the host could enqueue nothing behind it
// until the chain had run (round 6: ~50 us of idle device per cycle at the headline, 10-15 % of a config-4 cycle).  The host-staged
// transport keeps the stream synchronisation (no chain is speculated there; its exchanges are host work between drained streams);
// RCCL and the peer-to-peer transport speculate chains and take the event (every rank enqueues the same sequence either way).
inline void qstage_wait(ks_workspace* ws) {
  static const int evt_on = env_int("KS_QSTAGE_EVENT", 1);
  if (evt_on && ws->ctx->hc.allreduce == nullptr && ws->qstage_marked) {
    KS_HIP(hipEventSynchronize(ws->qstage_evt));
    return;
  }
  KS_HIP(hipStreamSynchronize(ws->ctx->stream));
}
inline void qstage_mark(ks_workspace* ws) {
  if (!ws->qstage_evt) KS_HIP(hipEventCreateWithFlags(&ws->qstage_evt, hipEventDisableTiming));
  KS_HIP(hipEventRecord(ws->qstage_evt, ws->ctx->stream));
  ws->qstage_marked = true;
}

template <class T> void rotate_tfold(ks_workspace* ws, int c0, int c, int r, const T* Qh, int ldq, int out0, int src, int dst) {
  using D = typename DevT<T>::type;
  ws->ctx->use();
  rot_flush(ws);   // (a rotation still pending is input of this one)
  const bool gated = ws->gate_armed;
  if (!gated) qstage_wait(ws);  // Qstage may still be in flight from a previous rotation
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
  const bool extra_elsewhere = src >= 0 && dst != out0 + r;
  {
    // a restart's rotation (all maxdim + 1 columns in, the residual direction next to the truncated basis) with blocks in
    // force: leave it to the expansion that follows (see ks_workspace::rot_pending).  That expansion starts its Newton chain
    // from the STORED last column instead of the rotated residual direction (which does not exist yet): the two agree to the
    // Gram deviation of the block that wrote it (stored = true column up to R_2 = I + delta), so only after a batch that ended
    // in a block and whose deviation is at rounding level (<= 1e-12; accepted blocks may carry up to gram_dev_max = 1e-8,
    // those take the ordinary sequence)
    // round 6: above that level the chain starts from the TRUE column instead (ks_workspace::ztrue), formed on the device from
    // the basis and T as they stand -- any accepted block qualifies where that product has a kernel (maxdim + 1 <= 40 columns)
    const bool true_start_on = ws->true_start_on;
    const bool start_stored = ws->blk_diag[2] <= 1e-12;
    const bool start_true = !start_stored && true_start_on && cin <= 40 && ws->blk_diag[2] <= ws->blk_gdevmax;
    const bool start_ok = start_stored || start_true;
    static const int defer_dbg = env_int("KS_DEFER_DEBUG", 0);
    if (defer_dbg)
      std::fprintf(stderr, "[defer] on %d ok %d src %d extra %d cin %d sstep_eff %d blk_tail %d tl %d thi %d gdev %.2e\n", (int)ws->rot_defer_on, (int)ws->rot_defer_ok,
                   src, (int)extra_elsewhere, cin, ws->sstep_eff, (int)ws->blk_tail, (int)tl, thi, ws->blk_diag[2]);
    if (ws->rot_defer_on && ws->rot_defer_ok && src == ws->maxdim && !extra_elsewhere && cin == ws->maxdim + 1 && ws->sstep_eff >= 8 && ws->blk_tail && tl && thi == ws->maxdim &&
        start_ok && ws->defl_last == 0) {   // (a deflated chain needs the rotated locked columns first: no deferral in front of one)
      gate_cancel(ws);   // (a pre-enqueued rotation returns at once)
      KS_HIP(hipMemcpyAsync(ws->Qd, qs, (size_t)cin * rr * sizeof(T), hipMemcpyHostToDevice, ws->ctx->stream));
      qstage_mark(ws);
      ws->rot_pending = true;
      ws->rot_true_start = start_true;
      ws->rot_cin = cin; ws->rot_rr = rr; ws->rot_out0 = out0;
      ws->t_lazy = true;            // "lazy" with an empty range: every reader flushes through materialize()
      ws->ntrue = out0 + rr;
      ws->t_hi = out0 + rr - 1;
      ws->blk_tail = false;
      return;
    }
  }
  if (gated) {
    if constexpr (sizeof(T) == 8) {
      if (cin == ws->gate_cin && rr <= ws->gate_rmax && __atomic_load_n(&ws->gate_h->timed_out, __ATOMIC_ACQUIRE) != ws->gate_seq) {
        // reverse mailbox: T Q is in the pinned stage, the gate copies it and the queued rotation runs
        ksd::RotGate* g = ws->gate_h;
        g->c = cin; g->r = rr; g->out0 = out0; g->extra_out = extra_elsewhere ? dst : -1; g->ldq = cin; g->nq = cin * rr; g->cancel = 0;
        __atomic_store_n(&g->flag, ws->gate_seq << 1, __ATOMIC_RELEASE);
        ws->gate_armed = false;
        // which way did the gate go?  It answers within one poll of the word it spins on (a few microseconds); it may have given
        // up between the look at `timed_out` above and the release -- then the rotation behind it has returned without
        // rotating and the ordinary sequence below does it.  (No answer within 2 ms: the gate kernel has not STARTED polling yet --
        // other work sits in front of it -- and will find the word released when it does: it cannot time out any more.)
        bool took = true;
        const double t_ack = ks::now_s();
        for (;;) {
          const uint64_t a = __atomic_load_n(&g->taken, __ATOMIC_ACQUIRE);
          if ((a >> 1) == ws->gate_seq) { took = (a & 1u) == 0; break; }
          if (ks::now_s() - t_ack > 2e-3) break;
        }
        if (took) {
          ws->t_lazy = false;
          ws->t_hi = -1;
          ws->blk_tail = false;
          return;
        }
      }
    }
    gate_cancel(ws);  // (shape does not fit what was enqueued, or the gate gave up: the ordinary sequence)
    KS_HIP(hipStreamSynchronize(ws->ctx->stream));
  }
  KS_HIP(hipMemcpyAsync(ws->Qd, qs, (size_t)cin * rr * sizeof(T), hipMemcpyHostToDevice, ws->ctx->stream));
  qstage_mark(ws);
  rotate_device<D>(ws, 0, cin, rr, out0, extra_elsewhere ? dst : -1);
  ws->t_lazy = false;
  ws->t_hi = -1;
  ws->blk_tail = false;
}

// the pending rotation as the ordinary kernel (somebody other than a fused first pass is about to read V)
inline void spec_drop(ks_workspace* ws) {
  // (a dropped chain costs the device s products it would otherwise not do: back off 1, 2, 4, 8 cycles on consecutive drops --
  // a reader between two calls is a one-off, a shape that never adopts is not --, an adopted chain resets the length)
  if (ws->spec_valid) { ws->spec_valid = false; ws->spec_wasted++; ws->spec_backoff = ws->spec_backoff_len; ws->spec_backoff_len = std::min(8, 2 * ws->spec_backoff_len); }
}
inline void rot_flush(ks_workspace* ws) {
  if (!ws->rot_pending) return;
  spec_drop(ws);   // (the chain's start column is about to be overwritten)
  ws->rot_pending = false;
  ws->rot_true_start = false;
  ws->ztrue_valid = false;
  ws->t_lazy = false;
  ws->t_hi = -1;
  if (ws->dtype == KS_F64) rotate_device<double>(ws, 0, ws->rot_cin, ws->rot_rr, ws->rot_out0, -1);
  else rotate_device<cd>(ws, 0, ws->rot_cin, ws->rot_rr, ws->rot_out0, -1);
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
  gate_cancel(ws);
  if (ws->t_lazy) materialize(ws);  // (T-lazy columns outside the rotated range would lose the columns they refer to)
  KS_HIP(hipStreamSynchronize(ws->ctx->stream));  // Qstage may still be in flight from a previous rotation
  if (ws->has_lazy() && ws->lazy_lo < c0) materialize(ws);
  T* qs = static_cast<T*>(ws->Qstage);
  for (int jj = 0; jj < r; ++jj)
    for (int ii = 0; ii < c; ++ii) qs[ii + (size_t)jj * c] = Qh[ii + (size_t)jj * ldq] * ws->hostscale[c0 + ii];
  KS_HIP(hipMemcpyAsync(ws->Qd, qs, (size_t)c * r * sizeof(T), hipMemcpyHostToDevice, ws->ctx->stream));
  qstage_mark(ws);
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

