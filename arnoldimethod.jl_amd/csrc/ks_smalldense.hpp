// Host-side small dense kernels of the Krylov-Schur restart (layer L2 of SURVEY.md).
//
// Everything here works on the (maxdim+1) x maxdim Hessenberg matrix H and the maxdim x maxdim
// accumulator Q that live on the HOST (north_star keeps them there): O(maxdim^3) work that never
// touches the n-sized basis.  Plain C++17, no HIP, so the same code is unit-tested on a CPU-only
// box through the ks_host_* exports of the C ABI.
//
// Each routine cites the reference file:line whose behaviour it reproduces.  Indices are 0-based
// and ranges inclusive unless stated; matrices are column-major views.
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

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace ks {

using cplx = std::complex<double>;

template <class T> struct is_real : std::is_floating_point<T> {};
template <class T> inline constexpr bool is_real_v = is_real<T>::value;

inline double conj_(double x) { return x; }
inline cplx conj_(cplx x) { return std::conj(x); }
inline double abs_(double x) { return std::fabs(x); }
inline double abs_(cplx x) { return std::abs(x); }
inline double abs2_(double x) { return x * x; }
inline double abs2_(cplx x) { return std::norm(x); }
inline double real_(double x) { return x; }
inline double real_(cplx x) { return x.real(); }
inline double imag_(double) { return 0.0; }
inline double imag_(cplx x) { return x.imag(); }
inline bool iszero_(double x) { return x == 0.0; }
inline bool iszero_(cplx x) { return x.real() == 0.0 && x.imag() == 0.0; }

constexpr double kEps = std::numeric_limits<double>::epsilon();

// Column-major matrix view.
template <class T> struct Mat {
  T* p = nullptr;
  int m = 0, n = 0, ld = 0;
  Mat() = default;
  Mat(T* p_, int m_, int n_, int ld_) : p(p_), m(m_), n(n_), ld(ld_) {}
  T& operator()(int i, int j) const { return p[i + (size_t)j * ld]; }
  bool wanted() const { return p != nullptr; }  // NotWanted, src/schurfact.jl:71-79
  // rows [0, rows) of the same columns
  Mat top(int rows) const { return Mat(p, rows, n, ld); }
};

struct QRNotConverged : std::runtime_error {
  QRNotConverged() : std::runtime_error("QR algorithm did not converge") {}  // src/schurfact.jl:406
};

// ---------------------------------------------------------------------------------------------
// givensAlgorithm: Julia stdlib LinearAlgebra (a port of LAPACK 3.x dlartg / zlartg); called at
// src/schurfact.jl:58,66-67, src/schursort.jl:224-237,260-268,288-289, src/restore_hessenberg.jl:91.
//   [ c  s; -conj(s)  c ] [f; g] = [r; 0]
// ---------------------------------------------------------------------------------------------
namespace detail {
inline double safmn2() {
  static const double v = [] {
    const double safmin = std::numeric_limits<double>::min();
    return std::pow(2.0, std::trunc(std::log(safmin / kEps) / std::log(2.0) / 2.0));
  }();
  return v;
}
inline double safmx2() { return 1.0 / safmn2(); }
}  // namespace detail

inline void givens(double f, double g, double& cs, double& sn, double& r) {
  const double mn2 = detail::safmn2(), mx2 = detail::safmx2();
  if (g == 0.0) { cs = 1.0; sn = 0.0; r = f; return; }
  if (f == 0.0) { cs = 0.0; sn = 1.0; r = g; return; }
  double f1 = f, g1 = g;
  double scale = std::max(std::fabs(f1), std::fabs(g1));
  if (scale >= mx2) {
    int count = 0;
    do { ++count; f1 *= mn2; g1 *= mn2; scale = std::max(std::fabs(f1), std::fabs(g1)); } while (scale >= mx2);
    r = std::sqrt(f1 * f1 + g1 * g1);
    cs = f1 / r; sn = g1 / r;
    for (int i = 0; i < count; ++i) r *= mx2;
  } else if (scale <= mn2) {
    int count = 0;
    do { ++count; f1 *= mx2; g1 *= mx2; scale = std::max(std::fabs(f1), std::fabs(g1)); } while (scale <= mn2);
    r = std::sqrt(f1 * f1 + g1 * g1);
    cs = f1 / r; sn = g1 / r;
    for (int i = 0; i < count; ++i) r *= mn2;
  } else {
    r = std::sqrt(f1 * f1 + g1 * g1);
    cs = f1 / r; sn = g1 / r;
  }
  if (std::fabs(f) > std::fabs(g) && cs < 0.0) { cs = -cs; sn = -sn; r = -r; }
}

inline void givens(cplx f, cplx g, double& cs, cplx& sn, cplx& r) {
  const double mn2 = detail::safmn2(), mx2 = detail::safmx2();
  const double safmin = std::numeric_limits<double>::min();
  auto abs1 = [](cplx z) { return std::max(std::fabs(z.real()), std::fabs(z.imag())); };
  double scale = std::max(abs1(f), abs1(g));
  cplx fs = f, gs = g;
  int count = 0;
  if (scale >= mx2) {
    do { ++count; fs *= mn2; gs *= mn2; scale *= mn2; } while (scale >= mx2);
  } else if (scale <= mn2) {
    if (iszero_(g)) { cs = 1.0; sn = 0.0; r = f; return; }
    do { --count; fs *= mx2; gs *= mx2; scale *= mx2; } while (scale <= mn2);
  }
  const double f2 = std::norm(fs), g2 = std::norm(gs);
  if (f2 <= std::max(g2, 1.0) * safmin) {
    if (iszero_(f)) {
      cs = 0.0;
      r = cplx(std::hypot(g.real(), g.imag()), 0.0);
      const double d = std::hypot(gs.real(), gs.imag());
      sn = cplx(gs.real() / d, -gs.imag() / d);
      return;
    }
    const double f2s = std::hypot(fs.real(), fs.imag());
    const double g2s = std::sqrt(g2);
    cs = f2s / g2s;
    cplx ff;
    if (abs1(f) > 1.0) {
      const double d = std::hypot(f.real(), f.imag());
      ff = cplx(f.real() / d, f.imag() / d);
    } else {
      const double dr = mx2 * f.real(), di = mx2 * f.imag();
      const double d = std::hypot(dr, di);
      ff = cplx(dr / d, di / d);
    }
    sn = ff * cplx(gs.real() / g2s, -gs.imag() / g2s);
    r = cs * f + sn * g;
  } else {
    const double f2s = std::sqrt(1.0 + g2 / f2);
    r = cplx(f2s * fs.real(), f2s * fs.imag());
    cs = 1.0 / f2s;
    const double d = f2 + g2;
    sn = cplx(r.real() / d, r.imag() / d) * std::conj(gs);
    if (count > 0) for (int i = 0; i < count; ++i) r *= mx2;
    if (count < 0) for (int i = 0; i < -count; ++i) r *= mn2;
  }
}

// ---------------------------------------------------------------------------------------------
// Rotations, src/schurfact.jl:19-148.  G = [c s; -conj(s) c];  lmul applies G, rmul applies G'.
// ---------------------------------------------------------------------------------------------
template <class T> struct Rot2 { double c; T s; int i; };
template <class T> struct Rot3 { double c1; T s1; double c2; T s2; int i; };

template <class T> inline void lmul(const Rot2<T>& G, const Mat<T>& A, int from, int to) {
  if (!A.wanted()) return;
  const T sc = conj_(G.s);
  for (int j = from; j <= to; ++j) {
    const T a1 = A(G.i, j), a2 = A(G.i + 1, j);
    A(G.i, j) = G.c * a1 + G.s * a2;
    A(G.i + 1, j) = -sc * a1 + G.c * a2;
  }
}
template <class T> inline void rmul(const Mat<T>& A, const Rot2<T>& G, int from, int to) {
  if (!A.wanted()) return;
  const T sc = conj_(G.s);
  const double c = G.c;
  const T ms = -G.s;
  // two distinct columns: contiguous, non-overlapping -> the loop vectorises (same operations per element, no contraction)
  T* __restrict__ p1 = &A(0, G.i);
  T* __restrict__ p2 = &A(0, G.i + 1);
  for (int j = from; j <= to; ++j) {
    const T a1 = p1[j], a2 = p2[j];
    p1[j] = a1 * c + a2 * sc;
    p2[j] = a1 * ms + a2 * c;
  }
}
template <class T> inline void lmul(const Rot3<T>& G, const Mat<T>& A, int from, int to) {
  if (!A.wanted()) return;
  const T s1c = conj_(G.s1), s2c = conj_(G.s2);
  for (int j = from; j <= to; ++j) {
    const T a1 = A(G.i, j), a2 = A(G.i + 1, j), a3 = A(G.i + 2, j);
    const T a2p = G.c1 * a2 + G.s1 * a3;
    const T a3p = -s1c * a2 + G.c1 * a3;
    const T a1pp = G.c2 * a1 + G.s2 * a2p;
    const T a2pp = -s2c * a1 + G.c2 * a2p;
    A(G.i, j) = a1pp; A(G.i + 1, j) = a2pp; A(G.i + 2, j) = a3p;
  }
}
template <class T> inline void rmul(const Mat<T>& A, const Rot3<T>& G, int from, int to) {
  if (!A.wanted()) return;
  const T s1c = conj_(G.s1), s2c = conj_(G.s2);
  const double c1 = G.c1, c2 = G.c2;
  const T ms1 = -G.s1, ms2 = -G.s2;
  T* __restrict__ p1 = &A(0, G.i);
  T* __restrict__ p2 = &A(0, G.i + 1);
  T* __restrict__ p3 = &A(0, G.i + 2);
  for (int j = from; j <= to; ++j) {
    const T a1 = p1[j], a2 = p2[j], a3 = p3[j];
    const T a2p = a2 * c1 + a3 * s1c;
    const T a3p = a2 * ms1 + a3 * c1;
    const T a1pp = a1 * c2 + a2p * s2c;
    const T a2pp = a1 * ms2 + a2p * c2;
    p1[j] = a1pp; p2[j] = a2pp; p3[j] = a3p;
  }
}
// whole-range forms, src/schurfact.jl:76-77
template <class T, class R> inline void lmul(const R& G, const Mat<T>& A) { if (A.wanted()) lmul(G, A, 0, A.n - 1); }
template <class T, class R> inline void rmul(const Mat<T>& A, const R& G) { if (A.wanted()) rmul(A, G, 0, A.m - 1); }

template <class T> inline Rot2<T> get_rotation(T p1, T p2, int i, T& nrm) {  // src/schurfact.jl:57-60
  Rot2<T> G; G.i = i;
  givens(p1, p2, G.c, G.s, nrm);
  return G;
}
template <class T> inline Rot3<T> get_rotation(T p1, T p2, T p3, int i, T& nrm2) {  // :65-69
  Rot3<T> G; G.i = i;
  T nrm1;
  givens(p2, p3, G.c1, G.s1, nrm1);
  givens(p1, nrm1, G.c2, G.s2, nrm2);
  return G;
}

// ---------------------------------------------------------------------------------------------
// QR iterations, src/schurfact.jl
// ---------------------------------------------------------------------------------------------
template <class T> inline bool is_offdiagonal_small(const Mat<T>& H, int i, double tol = kEps) {  // :7-11
  return abs_(H(i + 1, i)) <= tol * (abs_(H(i, i)) + abs_(H(i + 1, i + 1)));
}

// Francis double shift, src/schurfact.jl:150-249 (real arithmetic only).
inline void double_shift_schur(const Mat<double>& H, int from, int to, double trace, double determinant,
                               const Mat<double>& Q) {
  const int m = H.m, n = H.n;
  const double H11 = H(from, from), H21 = H(from + 1, from);
  const double H12 = H(from, from + 1), H22 = H(from + 1, from + 1), H32 = H(from + 2, from + 1);
  const double p1 = H11 * H11 + H12 * H21 - trace * H11 + determinant;
  const double p2 = H21 * (H11 + H22 - trace);
  const double p3 = H32 * H21;
  double nrm;
  auto G1 = get_rotation(p1, p2, p3, from, nrm);
  lmul(G1, H, from, n - 1);
  rmul(H, G1, 0, std::min(from + 3, m - 1));
  rmul(Q, G1);
  for (int i = from + 1; i <= to - 2; ++i) {
    auto G = get_rotation(H(i, i - 1), H(i + 1, i - 1), H(i + 2, i - 1), i, nrm);
    H(i, i - 1) = nrm; H(i + 1, i - 1) = 0.0; H(i + 2, i - 1) = 0.0;
    lmul(G, H, i, n - 1);
    rmul(H, G, 0, std::min(i + 3, m - 1));
    rmul(Q, G);
  }
  auto Gn = get_rotation(H(to - 1, to - 2), H(to, to - 2), to - 1, nrm);
  H(to - 1, to - 2) = nrm; H(to, to - 2) = 0.0;
  lmul(Gn, H, to - 1, n - 1);
  rmul(H, Gn, 0, to);
  rmul(Q, Gn);
}

// Single shift, src/schurfact.jl:251-320.
template <class T> inline void single_shift_schur(const Mat<T>& H, int from, int to, T mu, const Mat<T>& Q) {
  const int m = H.m, n = H.n;
  T nrm;
  auto G1 = get_rotation(T(H(from, from) - mu), H(from + 1, from), from, nrm);
  lmul(G1, H, from, n - 1);
  rmul(H, G1, 0, std::min(from + 2, m - 1));
  rmul(Q, G1);
  for (int i = from + 1; i <= to - 1; ++i) {
    auto G = get_rotation(H(i, i - 1), H(i + 1, i - 1), i, nrm);
    H(i, i - 1) = nrm; H(i + 1, i - 1) = T(0);
    lmul(G, H, i, n - 1);
    rmul(H, G, 0, std::min(i + 2, m - 1));
    rmul(Q, G);
  }
}

inline double sign_(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }

// src/schurfact.jl:327-357; pinned by test/schurfact.jl:160-168.
inline bool upper_triangular_2x2(double H11, double H12, double H21, double H22, double& c, double& s) {
  c = 1.0; s = 0.0;
  if (H21 == 0.0 || ((H11 - H22) == 0.0 && sign_(H12) != sign_(H21))) return false;
  if (H12 == 0.0) { c = 0.0; s = 1.0; return true; }
  const double p = (H11 - H22) / 2;
  const double bcmax = std::max(std::fabs(H12), std::fabs(H21));
  const double bcmis = std::min(std::fabs(H12), std::fabs(H21)) * sign_(H12) * sign_(H21);
  const double scale = std::max(std::fabs(p), bcmax);
  const double z = (p / scale) * p + (bcmax / scale) * bcmis;
  if (z < 0) return false;
  const double h = p + std::copysign(std::sqrt(scale) * std::sqrt(z), p);
  const double nrm = std::hypot(H21, h);
  c = h / nrm; s = H21 / nrm;
  return true;
}

// src/schurfact.jl:363-388; pinned by test/schurfact.jl:170-173.
inline bool use_single_shift(double H11, double H12, double H21, double H22, double& lam) {
  const double scale = std::fabs(H11) + std::fabs(H12) + std::fabs(H21) + std::fabs(H22);
  H11 /= scale; H12 /= scale; H21 /= scale; H22 /= scale;
  const double t = (H11 + H22) / 2;
  const double d = (H11 - t) * (H22 - t) - H12 * H21;
  lam = 0.0;
  if (d > 0.0) return false;
  const double sq = std::sqrt(std::fabs(d));
  const double l1 = t + sq, l2 = t - sq;
  lam = (std::fabs(H22 - l1) < std::fabs(H22 - l2) ? l1 : l2) * scale;
  return true;
}

// Real quasi-triangularisation, src/schurfact.jl:393-487.  Throws QRNotConverged like the reference.
inline bool local_schurfact(const Mat<double>& H, int start, int to, const Mat<double>& Q, double tol = kEps,
                            int maxiter = -1) {
  if (maxiter < 0) maxiter = 100 * H.m;
  int iter = 0;
  while (to > start) {
    if (++iter > maxiter) throw QRNotConverged();
    int from = to;
    while (from > start) {
      if (is_offdiagonal_small(H, from - 1, tol)) { H(from, from - 1) = 0.0; break; }
      --from;
    }
    if (from == to) { --to; continue; }
    const double C11 = H(to - 1, to - 1), C12 = H(to - 1, to), C21 = H(to, to - 1), C22 = H(to, to);
    if (from + 1 == to) {
      double cs, sn;
      if (upper_triangular_2x2(C11, C12, C21, C22, cs, sn)) {
        Rot2<double> G{cs, sn, from};
        lmul(G, H, from, H.n - 1);
        rmul(H, G, 0, to);
        rmul(Q, G);
        H(to, to - 1) = 0.0;
      }
      to -= 2;
      continue;
    }
    double mu;
    if (use_single_shift(C11, C12, C21, C22, mu)) {
      single_shift_schur(H, from, to, mu, Q);
    } else {
      double_shift_schur(H, from, to, C11 + C22, C11 * C22 - C12 * C21, Q);
    }
  }
  return true;
}

// Generic (complex) triangularisation, src/schurfact.jl:492-538.  Returns false on non-convergence
// (ignored by the driver, src/run.jl:281).
inline bool local_schurfact(const Mat<cplx>& H, int start, int to, const Mat<cplx>& Q, double tol = kEps,
                            int maxiter = -1) {
  if (maxiter < 0) maxiter = 100 * H.m;
  int iter = 0;
  while (true) {
    if (++iter > maxiter) return false;
    int from = to;
    while (from > start && !is_offdiagonal_small(H, from - 1, tol)) --from;
    if (from == to) {
      // The reference writes H[from, from-1] under @inbounds even when from == 1; that single
      // out-of-bounds store is skipped here.
      if (from >= 1) H(from, from - 1) = cplx(0);
      --to;
    } else {
      const cplx H11 = H(to - 1, to - 1), H12 = H(to - 1, to), H21 = H(to, to - 1), H22 = H(to, to);
      const cplx d = H11 * H22 - H21 * H12, t = H11 + H22;
      const cplx sq = std::sqrt(t * t - 4.0 * d);
      const cplx l1 = (t + sq) / 2.0, l2 = (t - sq) / 2.0;
      single_shift_schur(H, from, to, std::abs(H22 - l1) < std::abs(H22 - l2) ? l1 : l2, Q);
    }
    if (to <= start) break;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Eigenvalues of a quasi-triangular matrix, src/eigvals.jl:1-65
// ---------------------------------------------------------------------------------------------
template <class T> inline void copy_eigenvalues(cplx* lams, const Mat<T>& A, int first, int last, double tol = kEps) {
  int i = first;
  while (i < last) {
    if (is_offdiagonal_small(A, i, tol)) {
      lams[i] = cplx(A(i, i));
      ++i;
    } else {
      const T d = A(i, i) * A(i + 1, i + 1) - A(i, i + 1) * A(i + 1, i);
      const T x = (A(i, i) + A(i + 1, i + 1)) / 2.0;
      const cplx y = std::sqrt(cplx(x * x - d));
      lams[i] = cplx(x) + y;      // +imag first, src/eigvals.jl:20-24
      lams[i + 1] = cplx(x) - y;
      i += 2;
    }
  }
  if (i == last) lams[i] = cplx(A(i, i));
}

template <class T> inline cplx eigenvalue(const Mat<T>& R, int i) {  // src/eigvals.jl:42-55
  const int n = std::min(R.m, R.n);
  if (i == n - 1 || iszero_(R(i + 1, i))) return cplx(R(i, i));
  const T d = R(i, i) * R(i + 1, i + 1) - R(i, i + 1) * R(i + 1, i);
  const T x = (R(i, i) + R(i + 1, i + 1)) / 2.0;
  return cplx(x) + std::sqrt(cplx(x * x - d));
}

// ---------------------------------------------------------------------------------------------
// One eigenvector of a (quasi) upper triangular matrix, src/eigenvector_uppertriangular.jl
// ---------------------------------------------------------------------------------------------
// real * complex and complex * complex exactly as Julia forms them (Base: *(x::Real, z::Complex) scales both parts,
// *(z, w) is the four-product formula) -- std::complex's operator* adds the C99 Annex G infinity recovery, and promoting a
// real factor to complex first doubles the multiplications.  These loops are a fifth of the restart's host step.
inline cplx mulx(double r, cplx x) { return cplx(r * x.real(), r * x.imag()); }
inline cplx mulx(cplx a, cplx b) {
  return cplx(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real());
}
template <class T> inline void shifted_backward_sub(cplx* x, const Mat<T>& R, cplx lam, int k) {
  // k = number of unknowns (rows 0..k-1).  :6-42 real quasi-triangular, :44-68 generic.
  while (k > 0) {
    const int kk = k - 1;
    if (is_real_v<T> && k > 1 && !iszero_(R(kk, kk - 1))) {
      const cplx R11 = cplx(R(kk - 1, kk - 1)) - lam, R12 = cplx(R(kk - 1, kk));
      const cplx R21 = cplx(R(kk, kk - 1)), R22 = cplx(R(kk, kk)) - lam;
      const cplx det = R11 * R22 - R21 * R12;
      const cplx a1 = (R22 * x[kk - 1] - R12 * x[kk]) / det;
      const cplx a2 = (-R21 * x[kk - 1] + R11 * x[kk]) / det;
      x[kk - 1] = a1; x[kk] = a2;
      for (int i = 0; i < kk - 1; ++i) x[i] -= mulx(R(i, kk - 1), x[kk - 1]) + mulx(R(i, kk), x[kk]);
      k -= 2;
    } else {
      const cplx sigma = cplx(R(kk, kk)) - lam;
      if (iszero_(sigma)) {
        x[kk] = sigma;
      } else {
        x[kk] /= sigma;
        for (int i = 0; i < kk; ++i) x[i] -= mulx(R(i, kk), x[kk]);
      }
      k -= 1;
    }
  }
}

// collect_eigen!(x, R, j) -> length, src/eigenvector_uppertriangular.jl:76-154
template <class T> inline int collect_eigen(cplx* x, const Mat<T>& R, int j) {
  const int n = R.n;
  if constexpr (is_real_v<T>) {
    if (j < n - 1 && R(j + 1, j) != 0.0) ++j;
    if (j > 0 && R(j, j - 1) != 0.0) {
      const double R11 = R(j - 1, j - 1), R21 = R(j, j - 1), R12 = R(j - 1, j), R22 = R(j, j);
      const double det = R11 * R22 - R21 * R12, tr = R11 + R22;
      const cplx lam = (tr + std::sqrt(cplx(tr * tr - 4 * det))) / 2.0;
      x[j - 1] = -R12 / (R11 - lam);
      x[j] = 1.0;
      for (int i = 0; i < j - 1; ++i) x[i] = -mulx(R(i, j - 1), x[j - 1]) - R(i, j);
      shifted_backward_sub(x, R, lam, j - 1);
    } else {
      const cplx lam = R(j, j);
      x[j] = 1.0;
      for (int i = 0; i < j; ++i) x[i] = -R(i, j);
      shifted_backward_sub(x, R, lam, j);
    }
  } else {
    const cplx lam = R(j, j);
    x[j] = 1.0;
    for (int i = 0; i < j; ++i) x[i] = -R(i, j);
    shifted_backward_sub(x, R, lam, j);
  }
  double nrm = 0.0;
  for (int k = 0; k <= j; ++k) nrm += std::norm(x[k]);
  const double scale = 1.0 / std::sqrt(nrm);
  for (int k = 0; k <= j; ++k) x[k] *= scale;
  return j + 1;
}

// copy_residuals!, src/run.jl:524-545, in two halves: the dot product of the last row of Q with each eigenvector of
// R (everything but the factor h_last = H[maxdim, maxdim-1]), and the residual |dot * h_last| itself.
template <class T>
inline void residual_dots(cplx* dots, const Mat<T>& H, const Mat<T>& Q, cplx* x, int first, int last) {
  const int m = H.n;
  for (int i = first; i <= last; ++i) {
    for (int t = 0; t < m; ++t) x[t] = 0.0;
    const int len = collect_eigen(x, H, i);
    cplx tmp = 0.0;
    for (int j = 0; j < len; ++j) tmp += mulx(Q(m - 1, j), x[j]);
    dots[i] = tmp;
  }
}
template <class T> inline void finish_residuals(double* rs, const cplx* dots, T h_last, int m, int first, int last) {
  for (int i = 0; i < m; ++i) rs[i] = 0.0;
  for (int i = first; i <= last; ++i) rs[i] = std::abs(dots[i] * cplx(h_last));
}
template <class T>
inline void copy_residuals(double* rs, const Mat<T>& H, const Mat<T>& Q, T h_last, cplx* x, int first, int last) {
  std::vector<cplx> dots(H.n);
  residual_dots(dots.data(), H, Q, x, first, last);
  finish_residuals(rs, dots.data(), h_last, H.n, first, last);
}

// ---------------------------------------------------------------------------------------------
// Targets, src/targets.jl
// ---------------------------------------------------------------------------------------------
enum Which { LM = 0, LR = 1, SR = 2, LI = 3, SI = 4 };

inline bool isless_(double a, double b) {  // Julia isless: total order, NaN last, -0.0 < 0.0
  if (std::isnan(a)) return false;
  if (std::isnan(b)) return true;
  if (a == b) return std::signbit(a) && !std::signbit(b);
  return a < b;
}

struct Ordering {  // get_order, src/targets.jl:71-75
  int which;
  bool lt(cplx a, cplx b) const {
    switch (which) {
      case LM: return isless_(std::abs(b), std::abs(a));
      case LR: return isless_(b.real(), a.real());
      case SR: return isless_(a.real(), b.real());
      case LI: return isless_(b.imag(), a.imag());
      default: return isless_(a.imag(), b.imag());
    }
  }
};

// sort!(ord, QuickSort, OrderPerm(lams, ordering)): total order with index tie-break,
// src/targets.jl:61-67, src/run.jl:289.
inline void sort_perm(int* ord, int n, const cplx* lams, const Ordering& o) {
  std::sort(ord, ord + n, [&](int i, int j) {
    if (o.lt(lams[i], lams[j])) return true;
    if (o.lt(lams[j], lams[i])) return false;
    return i < j;
  });
}

// ---------------------------------------------------------------------------------------------
// Reordering the Schur form, src/schursort.jl
// ---------------------------------------------------------------------------------------------
template <class T> inline bool is_start_of_11_block(const Mat<T>& R, int i) { return i == R.n - 1 || iszero_(R(i + 1, i)); }
template <class T> inline bool is_end_of_11_block(const Mat<T>& R, int i) { return i == 0 || iszero_(R(i, i - 1)); }

// Completely pivoted LU of an N x N (N <= 4) system + solve, src/schursort.jl:79-168.
template <class T> struct TinyLU {
  T A[4][4];
  int p[4], q[4], N;
  bool singular;
  void factor() {
    for (int i = 0; i < N; ++i) p[i] = q[i] = N - 1;
    singular = false;
    for (int k = 0; k < N - 1; ++k) {
      int m = 0, n = 0;
      double maxval = 0.0;
      for (int j = k; j < N; ++j)
        for (int i = k; i < N; ++i)
          if (abs_(A[i][j]) > maxval) { m = i; n = j; maxval = abs_(A[i][j]); }
      p[k] = m; q[k] = n;
      for (int j = k; j < N; ++j) std::swap(A[k][j], A[m][j]);
      for (int j = k; j < N; ++j) std::swap(A[j][k], A[j][n]);
      const T Akk = A[k][k];
      if (iszero_(Akk)) { singular = true; break; }
      for (int i = k + 1; i < N; ++i) A[i][k] /= Akk;
      for (int j = k + 1; j < N; ++j) {
        const T Akj = A[k][j];
        for (int i = k + 1; i < N; ++i) A[i][j] -= A[i][k] * Akj;
      }
    }
    if (iszero_(A[N - 1][N - 1])) singular = true;
  }
  void solve(T* x) const {
    for (int i = 0; i < N; ++i) {
      std::swap(x[i], x[p[i]]);
      for (int j = i + 1; j < N; ++j) x[j] -= A[j][i] * x[i];
    }
    for (int i = N - 1; i >= 0; --i) {
      for (int j = N - 1; j > i; --j) x[i] -= A[i][j] * x[j];
      x[i] /= A[i][i];
      std::swap(x[i], x[q[i]]);
    }
  }
};

// sylv: A X - X B = C with A (na x na), B (nb x nb), na, nb in {1,2}.  X, C column-major na x nb.
// src/schursort.jl:170-202.  Returns `singular`.
template <class T> inline bool sylv(const T* A, int na, const T* B, int nb, const T* C, T* X) {
  auto a = [&](int i, int j) { return A[i + j * na]; };
  auto b = [&](int i, int j) { return B[i + j * nb]; };
  TinyLU<T> lu;
  lu.N = na * nb;
  for (auto& row : lu.A) for (auto& v : row) v = T(0);
  if (na == 1 && nb == 2) {
    lu.A[0][0] = a(0, 0) - b(0, 0); lu.A[0][1] = -b(1, 0);
    lu.A[1][0] = -b(0, 1);          lu.A[1][1] = a(0, 0) - b(1, 1);
  } else if (na == 2 && nb == 1) {
    lu.A[0][0] = a(0, 0) - b(0, 0); lu.A[0][1] = a(0, 1);
    lu.A[1][0] = a(1, 0);           lu.A[1][1] = a(1, 1) - b(0, 0);
  } else {
    lu.A[0][0] = a(0, 0) - b(0, 0); lu.A[0][1] = a(0, 1); lu.A[0][2] = -b(1, 0); lu.A[0][3] = T(0);
    lu.A[1][0] = a(1, 0); lu.A[1][1] = a(1, 1) - b(0, 0); lu.A[1][2] = T(0); lu.A[1][3] = -b(1, 0);
    lu.A[2][0] = -b(0, 1); lu.A[2][1] = T(0); lu.A[2][2] = a(0, 0) - b(1, 1); lu.A[2][3] = a(0, 1);
    lu.A[3][0] = T(0); lu.A[3][1] = -b(0, 1); lu.A[3][2] = a(1, 0); lu.A[3][3] = a(1, 1) - b(1, 1);
  }
  lu.factor();
  for (int i = 0; i < lu.N; ++i) X[i] = C[i];
  lu.solve(X);
  return lu.singular;
}

template <class T> inline void swap22(const Mat<T>& R, int i, const Mat<T>& Q) {  // src/schursort.jl:307-350
  const int n = R.n;
  const T A[4] = {R(i, i), R(i + 1, i), R(i, i + 1), R(i + 1, i + 1)};
  const T B[4] = {R(i + 2, i + 2), R(i + 3, i + 2), R(i + 2, i + 3), R(i + 3, i + 3)};
  const T C[4] = {R(i, i + 2), R(i + 1, i + 2), R(i, i + 3), R(i + 1, i + 3)};
  T X[4];
  if (sylv(A, 2, B, 2, C, X)) return;
  auto x = [&](int r, int c) { return X[r + 2 * c]; };
  // swap22_rotations, :222-239
  double c1, c2, c3, c4; T s1, s2, s3, s4, n1, n2, n3, n4;
  givens(T(-x(1, 0)), T(1), c1, s1, n1);
  givens(T(-x(0, 0)), n1, c2, s2, n2);
  T X22 = c1 * -x(1, 1);
  const T X32 = -conj_(s1) * -x(1, 1);
  X22 = -conj_(s2) * -x(0, 1) + c2 * X22;
  givens(X32, T(1), c3, s3, n3);
  givens(X22, n3, c4, s4, n4);
  Rot3<T> G1{c1, s1, c2, s2, i}, G2{c3, s3, c4, s4, i + 1};
  lmul(G1, R, i, n - 1); rmul(R, G1, 0, i + 3);
  lmul(G2, R, i, n - 1); rmul(R, G2, 0, i + 3);
  R(i + 2, i) = T(0); R(i + 3, i) = T(0); R(i + 2, i + 1) = T(0); R(i + 3, i + 1) = T(0);
  rmul(Q, G1); rmul(Q, G2);
}

template <class T> inline void swap21(const Mat<T>& R, int i, const Mat<T>& Q) {  // src/schursort.jl:365-401
  const int n = R.n;
  const T A[4] = {R(i, i), R(i + 1, i), R(i, i + 1), R(i + 1, i + 1)};
  const T B[1] = {R(i + 2, i + 2)};
  const T C[2] = {R(i, i + 2), R(i + 1, i + 2)};
  T X[2];
  if (sylv(A, 2, B, 1, C, X)) return;
  double c1, c2; T s1, s2, n1, n2;  // swap21_rotations, :287-291
  givens(T(-X[1]), T(1), c1, s1, n1);
  givens(T(-X[0]), n1, c2, s2, n2);
  Rot3<T> G1{c1, s1, c2, s2, i};
  lmul(G1, R, i, n - 1); rmul(R, G1, 0, i + 2);
  R(i + 1, i) = T(0); R(i + 2, i) = T(0);
  rmul(Q, G1);
}

template <class T> inline void swap12(const Mat<T>& R, int i, const Mat<T>& Q) {  // src/schursort.jl:419-458
  const int n = R.n;
  const T A[1] = {R(i, i)};
  const T B[4] = {R(i + 1, i + 1), R(i + 2, i + 1), R(i + 1, i + 2), R(i + 2, i + 2)};
  const T C[2] = {R(i, i + 1), R(i, i + 2)};
  T X[2];
  if (sylv(A, 1, B, 2, C, X)) return;
  double c1, c2; T s1, s2, n1, n2;  // swap12_rotations, :258-270
  givens(T(-X[0]), T(1), c1, s1, n1);
  const T X22 = -conj_(s1) * -X[1];
  givens(X22, T(1), c2, s2, n2);
  Rot2<T> G1{c1, s1, i}, G2{c2, s2, i + 1};
  lmul(G1, R, i, n - 1); rmul(R, G1, 0, i + 2);
  lmul(G2, R, i, n - 1); rmul(R, G2, 0, i + 2);
  R(i + 2, i) = T(0); R(i + 2, i + 1) = T(0);
  rmul(Q, G1); rmul(Q, G2);
}

template <class T> inline void swap11(const Mat<T>& R, int i, const Mat<T>& Q) {  // src/schursort.jl:460-482
  const int n = R.n;
  const T R11 = R(i, i), R12 = R(i, i + 1), R22 = R(i + 1, i + 1);
  T nrm;
  auto G = get_rotation(R12, T(R22 - R11), i, nrm);
  lmul(G, R, i + 2, n - 1);
  rmul(R, G, 0, i - 1);
  R(i, i) = R22; R(i + 1, i + 1) = R11;
  rmul(Q, G);
}

template <class T> inline void swap_blocks(const Mat<T>& R, int i, bool curr_11, bool next_11, const Mat<T>& Q) {
  if (curr_11) { if (next_11) swap11(R, i, Q); else swap12(R, i, Q); }  // src/schursort.jl:489-503
  else { if (next_11) swap21(R, i, Q); else swap22(R, i, Q); }
}

template <class T> inline void rotate_right(const Mat<T>& R, int from, int to, const Mat<T>& Q) {  // :19-32
  int i = to;
  while (i > from) {
    const bool curr_11 = is_start_of_11_block(R, i);
    const bool prev_11 = is_end_of_11_block(R, i - 1);
    const int j = prev_11 ? i - 1 : i - 2;
    swap_blocks(R, j, prev_11, curr_11, Q);
    i = j;
  }
}

// src/run.jl:394-457
template <class T> inline void partition_schur_three_way(const Mat<T>& R, const Mat<T>& Q, const int* groups, int ng) {
  int hi = 0, mi = 0, lo = 0;
  while (hi < ng) {
    const int group = groups[hi];
    const int bs = is_start_of_11_block(R, hi) ? 1 : 2;
    if (group == 3) {
      hi += bs;
    } else if (group == 2) {
      rotate_right(R, mi, hi, Q);
      hi += bs; mi += bs;
    } else {
      rotate_right(R, lo, hi, Q);
      hi += bs; mi += bs; lo += bs;
    }
  }
}

// sortschur!, src/run.jl:465-502
template <class T> inline void sortschur(const Mat<T>& R, const Mat<T>& Q, int to, const Ordering& o) {
  if (to <= 1) return;
  int next_idx = 0;
  while (next_idx <= to - 1) {
    int curr_idx = next_idx;
    const int curr_size = is_start_of_11_block(R, curr_idx) ? 1 : 2;
    const cplx curr_lam = eigenvalue(R, curr_idx);
    while (curr_idx > 0) {
      const int prev_size = is_end_of_11_block(R, curr_idx - 1) ? 1 : 2;
      const int prev_idx = curr_idx - prev_size;
      const cplx prev_lam = eigenvalue(R, prev_idx);
      if (!o.lt(curr_lam, prev_lam)) break;
      swap_blocks(R, prev_idx, prev_size == 1, curr_size == 1, Q);
      curr_idx -= prev_size;
    }
    next_idx += curr_size;
  }
}

// ---------------------------------------------------------------------------------------------
// Restoring the Hessenberg form, src/restore_hessenberg.jl
// ---------------------------------------------------------------------------------------------
// reflector!(y, k) -> tau (LAPACK clarfg-like), :16-45.  k = length; pivot y[k-1].
template <class T> inline T reflector(T* y, int k) {
  double xnrm = 0.0;
  for (int i = 0; i < k - 1; ++i) xnrm += abs2_(y[i]);
  T alpha = y[k - 1];
  if (xnrm == 0.0 && imag_(alpha) == 0.0) return T(0);
  xnrm = std::sqrt(xnrm);
  const double beta = -std::copysign(std::hypot(abs_(alpha), xnrm), real_(alpha));
  const T tau = (T(beta) - alpha) / beta;
  alpha = T(1) / (alpha - T(beta));
  for (int i = 0; i < k - 1; ++i) y[i] *= alpha;
  y[k - 1] = T(beta);
  return conj_(tau);
}

template <class T> struct Reflector {  // :47-65
  std::vector<T> vec;
  int offset = 0, len = 0;
  T tau = T(0);
  explicit Reflector(int max_len) : vec(max_len) {}
};

template <class T> inline void lmul(const Reflector<T>& G, const Mat<T>& H, int from, int to) {  // :138-159
  if (iszero_(G.tau)) return;
  const int len = G.len, off = G.offset;
  const T* z = G.vec.data();
  for (int col = from; col <= to; ++col) {
    T dot = T(0);
    for (int i = 0; i < len - 1; ++i) dot += conj_(z[i]) * H(i + off, col);
    dot += H(len - 1 + off, col);
    dot *= G.tau;
    for (int i = 0; i < len - 1; ++i) H(i + off, col) -= dot * z[i];
    H(len - 1 + off, col) -= dot;
  }
}
template <class T> inline void rmul(const Mat<T>& H, const Reflector<T>& G, int from, int to) {  // :161-182
  if (iszero_(G.tau)) return;
  const int len = G.len, off = G.offset;
  const T* z = G.vec.data();
  const T tc = conj_(G.tau);
  for (int row = from; row <= to; ++row) {
    T dot = T(0);
    for (int i = 0; i < len - 1; ++i) dot += H(row, i + off) * z[i];
    dot += H(row, off + len - 1);
    dot *= tc;
    for (int i = 0; i < len - 1; ++i) H(row, i + off) -= dot * conj_(z[i]);
    H(row, off + len - 1) -= dot;
  }
}

// restore_arnoldi!(H, from, to, Q, G), :75-134.  H full (maxdim+1) x maxdim, Q maxdim x maxdim.
template <class T> inline void restore_arnoldi(const Mat<T>& H, int from, int to, const Mat<T>& Q, Reflector<T>& G) {
  if (!(from < to)) return;
  const int m = H.m, n = H.n;
  T nrm = Q(n - 1, from);
  for (int i = from; i <= to - 1; ++i) {
    double c; T s;
    givens(Q(n - 1, i + 1), nrm, c, s, nrm);
    Rot2<T> rot{c, -s, i};
    rmul(H, rot, 0, std::min(i + 2, to));
    lmul(rot, H, 0, to);
    rmul(Q, rot, 0, n - 1);
  }
  H(to + 1, to) = Q(n - 1, to) * H(m - 1, n - 1);
  G.offset = from;
  for (int i = to - from; i >= 2; --i) {
    G.len = i;
    const int row = from + i;
    for (int j = 0; j < i; ++j) G.vec[j] = conj_(H(row, j + from));
    G.tau = reflector(G.vec.data(), i);
    rmul(H, G, 0, row - 1);
    for (int j = 0; j < i - 1; ++j) H(row, j + from) = T(0);
    H(row, i - 1 + from) = conj_(G.vec[i - 1]);
    lmul(G, H, from, to);
    rmul(Q, G, 0, n - 1);
  }
}

}  // namespace ks
