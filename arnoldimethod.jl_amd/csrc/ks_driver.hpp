// Krylov-Schur outer loop (`_partialschur`, src/run.jl:224-392) written against an abstract
// compute backend.  The backend owns the n-sized data (V in HBM) and implements the three verbs
// that scale with n -- expansion, reinitialise, rotation -- the host code in this file only
// touches H ((maxdim+1) x maxdim) and Q (maxdim x maxdim).
//
// ATTRIBUTION.  The algorithms in this file restate, routine by routine, the host-side code of ArnoldiMethod.jl v0.4.0
// (https://github.com/JuliaLinearAlgebra/ArnoldiMethod.jl; MIT License, Copyright (c) 2018 Harmen Stoppels): the Julia
// runtime the reference needs does not exist in the build image, and north_star keeps this O(maxdim^3) part on the
// host.  It is support code for running the hot path end to end, not claimed as hot-path coverage.  The MIT licence
// requires this notice to travel with substantial portions of the original: "Permission is hereby granted, free of
// charge, to any person obtaining a copy of this software and associated documentation files (the "Software"), to deal
// in the Software without restriction ... THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND" (full text:
// the reference's LICENSE file).
#pragma once

#include <chrono>
#include <cstring>
#include <functional>

#include "ks_smalldense.hpp"

namespace ks {

struct ExpandStats {
  int steps = 0, reorth = 0, breakdowns = 0;
  int explicit_steps = 0;  // steps a backend with an implicit second DGKS pass redid in the explicit form
  int blocks = 0, blk_bails = 0;  // s-step expansion: blocks completed / abandoned (their steps redone one by one)
};

// What the driver needs from whoever owns V.  Column/step numbering follows the reference with
// 0-based columns: step j (from..to as passed to iterate_arnoldi!) builds column j from column j-1.
template <class T> struct Backend {
  virtual ~Backend() = default;
  virtual int64_t n_global() const = 0;
  // iterate_arnoldi!(A, arnoldi, from:to)  src/expansion.jl:116-133; fills H[0..j, j-1] for each step.
  virtual void iterate_arnoldi(int from, int to, const Mat<T>& H, ExpandStats& st) = 0;
  // The same, for the expansion a restart follows (to == maxdim).  H[0:to, 0:to) is final BEFORE the step's last pass over
  // V and the reduction of H[to, to-1] have run; a backend that can tell may call `early` at that point -- with
  // H[0:to, 0:to) in place and H[to, to-1] still undefined -- so that the part of the restart's host work that does not
  // need H[to, to-1] (restart_host_early) runs while the device finishes.  Returns true iff `early` was called and its
  // effects on H stand (a backend that has to withdraw -- breakdown inside the batch -- restores H and returns false).
  virtual bool iterate_arnoldi_early(int from, int to, const Mat<T>& H, ExpandStats& st, const std::function<void()>& early) {
    (void)early;
    iterate_arnoldi(from, to, H, st);
    return false;
  }
  // reinitialize!(arnoldi, j, populate!)  src/expansion.jl:12-59.  v1 == nullptr -> rand!.
  virtual bool reinitialize(int j, const T* v1_host) = 0;
  // V[:, c0:c0+r) <- V[:, c0:c0+c) * Q[c0:c0+c, c0:c0+r)   (src/run.jl:363-364, :382-383); Q host.
  virtual void rotate(int c0, int c, int r, const Mat<T>& Q) = 0;
  // V[:, dst] <- V[:, src]   (src/run.jl:365)
  virtual void col_copy(int dst, int src) = 0;
  // src/run.jl:363-365 in one call (a backend that keeps the basis in factored form does both at once)
  virtual void rotate_and_move(int c0, int c, int r, const Mat<T>& Q, int dst, int src) {
    rotate(c0, c, r, Q);
    col_copy(dst, src);
  }
  // the Ritz values (all maxdim eigenvalues of the active Hessenberg matrix) of the restart in progress: a backend with an
  // s-step expansion takes the Newton shifts of its next blocks from them
  virtual void note_ritz(const cplx* lams, int m, double leak = 0.0, double fro = 0.0, double tol = 0.0) { (void)lams; (void)m; (void)leak; (void)fro; (void)tol; }
};

struct Params {
  int nev, which;
  double tol;
  int mindim, maxdim, restarts, start_from, initialize;
};

struct History {
  int mvproducts = 0, nconverged = 0, converged = 0, nev = 0, restarts = 0, reorth = 0, breakdowns = 0, explicit_steps = 0;
  double seconds_expand = 0, seconds_host = 0, seconds_rotate = 0;
};

// include_conjugate_pair, src/run.jl:510-517 (i 0-based position in ord).
template <class T> inline int include_conjugate_pair(const cplx* lams, const int* ord, int nord, int i) {
  if constexpr (!is_real_v<T>) return i;
  if (i >= nord - 1) return i;
  const cplx l1 = lams[ord[i]], l2 = lams[ord[i + 1]];
  return (l1.imag() != 0.0 && std::conj(l1) == l2) ? i + 1 : i;
}

// Scratch reused across restarts (src/run.jl:242-252).
template <class T> struct RestartScratch {
  std::vector<cplx> x, lams;
  std::vector<double> rs;
  std::vector<int> ord, groups;
  std::vector<cplx> rdot;  // restart_host_early -> restart_host_late: last component of each Ritz vector (unit residual)
  double fro2 = 0.0;       // sum |H[i,j]|^2 without the entry H[maxdim, maxdim-1]
  Reflector<T> G;
  explicit RestartScratch(int maxdim)
      : x(maxdim), lams(maxdim), rs(maxdim), ord(maxdim), groups(maxdim, 0), rdot(maxdim), G(maxdim) {}
};

struct RestartResult {
  int k, nlock, purge, effective_nev;
  // largest |entry| of the sorted Schur form BELOW the cut (rows k..maxdim-1 of the kept columns) and ||H||_F.  A Schur form
  // is zero there; the one exception is a 2 x 2 block of a real Schur form that the selection splits (a complex pair whose
  // members do not sit next to each other in the target's order: imaginary-part targets on a real matrix, src/run.jl:298-339):
  // its sub-diagonal entry is then dropped by the truncation, and the Arnoldi relation of the kept columns is off by that
  // much from here on -- in the reference as well.  The per-step expansion does not care; the s-step expansion, which leans
  // on the relation of the earlier columns with O(1) coefficients, must not run on such a state (Backend::note_ritz).
  double leak = 0.0, fro = 0.0;
};

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Optional stage timers of the restart step (tools/host_step_profile.cpp compiles with -DKS_TIME_STAGES; the product
// build does not define it and the macro vanishes).
#ifdef KS_TIME_STAGES
inline double g_stage_s[8] = {};
#define KS_STAGE(i, t0)                         \
  do {                                          \
    const double t1__ = now_s();                \
    g_stage_s[i] += t1__ - (t0);                \
    (t0) = t1__;                                \
  } while (0)
#else
#define KS_STAGE(i, t0) ((void)0)
#endif

// One restart's host work: src/run.jl:278-360.  `active` 0-based.  H is the full (maxdim+1) x maxdim
// array, Q is maxdim x maxdim.  Split in two so that a backend can overlap the first part with the tail of the
// expansion (Backend::iterate_arnoldi_early): restart_host_early touches H[0:maxdim, :] only, restart_host_late is
// the first to read H[maxdim, maxdim-1].  early + late performs the reference's operations in the reference's order
// on every quantity (the residual |x . h_last| is formed from the stored dot product, the Frobenius sum gets its
// last term last), so the split is invisible in the results.
template <class T>
inline void restart_host_early(const Mat<T>& H, const Mat<T>& Q, int maxdim, const Ordering& ordering, int active,
                               RestartScratch<T>& s) {
#ifdef KS_TIME_STAGES
  double t0__ = now_s();
#endif
  // Q <- I  (:278)
  for (int j = 0; j < maxdim; ++j)
    for (int i = 0; i < maxdim; ++i) Q(i, j) = (i == j) ? T(1) : T(0);
  // Schur form of the active block of H[0:maxdim, :]  (:281)
  local_schurfact(H.top(maxdim), active, maxdim - 1, Q);
  KS_STAGE(0, t0__);
  for (int i = 0; i < maxdim; ++i) s.ord[i] = i;                       // :284
  copy_eigenvalues(s.lams.data(), H, 0, maxdim - 1);                   // :285
  residual_dots(s.rdot.data(), H, Q, s.x.data(), active, maxdim - 1);  // :286, up to the factor H[maxdim, maxdim-1]
  KS_STAGE(1, t0__);
  sort_perm(s.ord.data(), maxdim, s.lams.data(), ordering);            // :289
  double fro2 = 0.0;                                                   // :292 norm(H), whole array, column-major order:
  for (int j = 0; j < H.n; ++j)                                        // the entry H[maxdim, maxdim-1] is the LAST term
    for (int i = 0; i < H.m; ++i)
      if (!(i == H.m - 1 && j == H.n - 1)) fro2 += abs2_(H(i, j));
  s.fro2 = fro2;
  KS_STAGE(2, t0__);
}

template <class T>
inline RestartResult restart_host_late(const Mat<T>& H, const Mat<T>& Q, int maxdim, int mindim, int nev, double tol,
                                       int active, RestartScratch<T>& s) {
#ifdef KS_TIME_STAGES
  double t0__ = now_s();
#endif
  finish_residuals(s.rs.data(), s.rdot.data(), H(maxdim, maxdim - 1), H.n, active, maxdim - 1);  // :286
  const double fro = std::sqrt(s.fro2 + abs2_(H(H.m - 1, H.n - 1)));
  auto isconverged = [&](int i) {                                      // :206-208
    return s.rs[i] <= std::max(kEps * fro, tol * std::abs(s.lams[i]));
  };
  const int* ord = s.ord.data();
  int* groups = s.groups.data();
  const int effective_nev = include_conjugate_pair<T>(s.lams.data(), ord, maxdim, nev - 1) + 1;  // :298
  int nlock = 0;
  for (int i = 0; i < effective_nev; ++i) {                            // :301-308
    if (isconverged(ord[i])) { groups[ord[i]] = 1; ++nlock; } else groups[ord[i]] = 2;
  }
  const int ideal_size = std::min(nlock + mindim, (mindim + maxdim) / 2);  // :316
  int k = effective_nev;
  int i = effective_nev;
  while (i < maxdim) {                                                 // :320-339
    const bool is_pair = include_conjugate_pair<T>(s.lams.data(), ord, maxdim, i) == i + 1;
    const int num = is_pair ? 2 : 1;
    int group;
    if (k < ideal_size && !isconverged(ord[i])) { group = 2; k += num; } else group = 3;
    if (is_pair) { groups[ord[i]] = group; groups[ord[i + 1]] = group; i += 2; }
    else { groups[ord[i]] = group; i += 1; }
  }
  int purge = 0;                                                       // :350-353
  while (purge < active && groups[purge] == 1) ++purge;
  KS_STAGE(2, t0__);
  partition_schur_three_way(H, Q, groups, maxdim);                     // :355
  KS_STAGE(3, t0__);
  double leak = 0.0;
  for (int j = 0; j < k; ++j)
    for (int i2 = k; i2 < maxdim; ++i2) leak = std::max(leak, (double)std::abs(H(i2, j)));
  restore_arnoldi(H, nlock, k - 1, Q, s.G);                            // :360
  KS_STAGE(4, t0__);
  RestartResult res{k, nlock, purge, effective_nev};
  res.leak = leak;
  res.fro = fro;
  return res;
}

template <class T>
inline RestartResult restart_host_step(const Mat<T>& H, const Mat<T>& Q, int maxdim, int mindim, int nev,
                                       const Ordering& ordering, double tol, int active, RestartScratch<T>& s) {
  restart_host_early(H, Q, maxdim, ordering, active, s);
  return restart_host_late(H, Q, maxdim, mindim, nev, tol, active, s);
}

// _partialschur, src/run.jl:224-392.  H: (maxdim+1) x maxdim host, Q: maxdim x maxdim host.
// eigenvalues: out, at least maxdim entries.  `active` 0-based (= start_from - 1).
template <class T>
inline History partialschur_driver(Backend<T>& be, const Mat<T>& H, const Mat<T>& Q, const Params& p, int active,
                                   cplx* eigenvalues) {
  const int mindim = p.mindim, maxdim = p.maxdim, nev = p.nev;
  RestartScratch<T> scratch(maxdim);
  const Ordering ordering{p.which};
  History hist;
  hist.nev = nev;
  ExpandStats st;

  int k = mindim;
  int prods = std::max(0, mindim - active);  // length(active:mindim), :264
  double t0 = now_s();
  be.iterate_arnoldi(active + 1, mindim, H, st);  // :267
  hist.seconds_expand += now_s() - t0;

  for (int iter = 0; iter < p.restarts; ++iter) {
    t0 = now_s();
    const bool early_done = be.iterate_arnoldi_early(k + 1, maxdim, H, st,                                   // :272
                                                     [&] { restart_host_early(H, Q, maxdim, ordering, active, scratch); });
    hist.seconds_expand += now_s() - t0;  // (includes whatever of the early host part the device did not hide)
    prods += std::max(0, maxdim - k);          // :275
    hist.restarts++;

    t0 = now_s();
    if (!early_done) restart_host_early(H, Q, maxdim, ordering, active, scratch);
    const RestartResult r = restart_host_late(H, Q, maxdim, mindim, nev, p.tol, active, scratch);
    be.note_ritz(scratch.lams.data(), maxdim, r.leak, r.fro, p.tol);
    hist.seconds_host += now_s() - t0;
    k = r.k;

    // :363-365  V[:, purge:k) <- V[:, purge:maxdim) Q[purge:maxdim, purge:k);  V[:, k] <- V[:, maxdim]
    t0 = now_s();
    be.rotate_and_move(r.purge, maxdim - r.purge, k - r.purge, Q, k, maxdim);
    hist.seconds_rotate += now_s() - t0;

    active = r.nlock;              // :368  (jl: active = nlock + 1)
    if (active + 1 > nev) break;   // :370
  }

  const int nconverged = active;   // :373
  t0 = now_s();
  for (int j = 0; j < maxdim; ++j)
    for (int i = 0; i < maxdim; ++i) Q(i, j) = (i == j) ? T(1) : T(0);
  sortschur(H, Q, nconverged, ordering);                              // :379
  hist.seconds_host += now_s() - t0;
  t0 = now_s();
  if (nconverged > 0) be.rotate(0, nconverged, nconverged, Q);        // :382-383
  hist.seconds_rotate += now_s() - t0;
  copy_eigenvalues(eigenvalues, H, 0, nconverged - 1);                // :386

  hist.mvproducts = prods;
  hist.nconverged = nconverged;
  hist.converged = nconverged >= nev;
  hist.reorth = st.reorth;
  hist.breakdowns = st.breakdowns;
  hist.explicit_steps = st.explicit_steps;
  return hist;
}

// Argument checks of partialschur / partialschur!, src/run.jl:110-116, :162-174.
// Returns 0 (ok), 1 (ArgumentError) and fills msg.
inline int check_params(int64_t n, int ncolsV, const Params& p, std::string& msg) {
  if (p.nev < 1) { msg = "nev cannot be less than 1"; return 1; }
  if (!(p.nev <= p.mindim && p.mindim <= p.maxdim && (int64_t)p.maxdim <= n)) {
    msg = "nev ≤ mindim ≤ maxdim ≤ size(A, 1) does not hold, got " + std::to_string(p.nev) + " ≤ " +
          std::to_string(p.mindim) + " ≤ " + std::to_string(p.maxdim) + " ≤ " + std::to_string(n);
    return 1;
  }
  if (!(p.maxdim < ncolsV)) { msg = "maxdim should be strictly less than size(arnoldi.V, 2)"; return 1; }
  if (!(1 <= p.start_from && p.start_from <= p.maxdim)) { msg = "start_from should be between 1 and maxdim"; return 1; }
  if (p.which < 0 || p.which > 4) { msg = "Unknown target"; return 1; }
  return 0;
}

}  // namespace ks
