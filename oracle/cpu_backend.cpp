// ORACLE / TEST INFRASTRUCTURE -- never linked into the product library.
//
// A CPU backend for the product's host driver (arnoldimethod.jl_amd/csrc/ks_driver.hpp) that issues
// the SAME un-fused operation sequence the reference issues per Arnoldi step:
//     mul!(w, A, v)            CSR SpMV                       src/expansion.jl:121
//     norm(w)                  nrm2                           src/expansion.jl:81
//     mul!(h, Vprev', w)       gemv 'T'/'C'                   src/expansion.jl:84
//     mul!(w, Vprev, h, -1, 1) gemv 'N'                       src/expansion.jl:85
//     norm(w)                  nrm2                           src/expansion.jl:88
//     [second DGKS pass]                                       src/expansion.jl:91-97
//     w ./= wnorm              scal                           src/expansion.jl:106
// and per restart   mul!(V_tmp, V, Q) ; copyto!(V, V_tmp) ; copyto!(V[:,k+1], V[:,maxdim+1])
//                                                              src/run.jl:363-365
// with OpenMP over rows.  Two uses:
//   1. tests: run the product's C++ driver + small dense kernels end-to-end on a CPU-only box and
//      compare with the Python oracle (tests/test_host_driver_cpu.py);
//   2. bench.py `cpu_baseline` (kind "port"): the reference's op sequence timed on the GPU box's
//      host cores.  `spmv_threads = 1` reproduces Julia's serial SparseMatrixCSC mul!.
//
// Build: make -C oracle   (g++ -O3 -fopenmp; output oracle/_build/libkschur_cpuref.so)
#include <omp.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../arnoldimethod.jl_amd/csrc/ks_driver.hpp"

using ks::cplx;

namespace {

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double uniform_hash(uint64_t seed, uint64_t idx) {
  return (double)(splitmix64(seed ^ idx) >> 11) * (1.0 / 9007199254740992.0);
}

template <class T> struct CpuBackend : ks::Backend<T> {
  int64_t n;
  int maxdim;
  const int32_t* rowptr;
  const int32_t* colidx;
  const T* val;
  std::vector<T> V, Vtmp;
  uint64_t seed = 20240917ull, rng_count = 0;
  int spmv_threads = 0;  // 0 = all
  double t_spmv = 0, t_orth = 0, t_rot = 0;

  CpuBackend(int64_t n_, int maxdim_, const int32_t* rp, const int32_t* ci, const T* v)
      : n(n_), maxdim(maxdim_), rowptr(rp), colidx(ci), val(v), V((size_t)n_ * (maxdim_ + 1)),
        Vtmp((size_t)n_ * (maxdim_ + 1)) {}

  T* col(int j) { return V.data() + (size_t)j * n; }
  int64_t n_global() const override { return n; }

  void spmv(const T* x, T* y) {
    const int nt = spmv_threads > 0 ? spmv_threads : omp_get_max_threads();
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t r = 0; r < n; ++r) {
      T s = T(0);
      for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) s += val[p] * x[colidx[p]];
      y[r] = s;
    }
  }
  double nrm2(const T* v) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += ks::abs2_(v[i]);
    return std::sqrt(s);
  }
  void gemv_t(int j, const T* v, T* h) {  // h = V[:,0:j)^H v
    for (int c0 = 0; c0 < j; c0 += 8) {
      const int nc = std::min(8, j - c0);
      double re[8] = {0}, im[8] = {0};
#pragma omp parallel for reduction(+ : re[:8], im[:8]) schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        const T vi = v[i];
        for (int c = 0; c < nc; ++c) {
          const T t = ks::conj_(V[(size_t)(c0 + c) * n + i]) * vi;
          re[c] += ks::real_(t);
          im[c] += ks::imag_(t);
        }
      }
      for (int c = 0; c < nc; ++c) {
        if constexpr (ks::is_real_v<T>) h[c0 + c] = re[c]; else h[c0 + c] = cplx(re[c], im[c]);
      }
    }
  }
  void gemv_n_sub(int j, T* v, const T* h) {  // v -= V[:,0:j) h
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      T s = T(0);
      for (int c = 0; c < j; ++c) s += V[(size_t)c * n + i] * h[c];
      v[i] -= s;
    }
  }
  void scal(T* v, double f) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) v[i] *= f;
  }
  void fill_uniform(T* v) {
    const uint64_t s = seed + rng_count * 0x9E3779B97F4A7C15ull;
    rng_count++;
    for (int64_t i = 0; i < n; ++i) {
      if constexpr (ks::is_real_v<T>) v[i] = uniform_hash(s, (uint64_t)i);
      else v[i] = cplx(uniform_hash(s, 2 * (uint64_t)i), uniform_hash(s, 2 * (uint64_t)i + 1));
    }
  }

  // orthogonalize!, src/expansion.jl:69-109
  bool orthogonalize(int j, const ks::Mat<T>& H, ks::ExpandStats& st) {
    constexpr double eta = 0.70710678118654752440;
    T* v = col(j);
    std::vector<T> h(j), corr(j);
    double rnorm = nrm2(v);
    gemv_t(j, v, h.data());
    gemv_n_sub(j, v, h.data());
    double wnorm = nrm2(v);
    if (wnorm < eta * rnorm) {
      rnorm = wnorm;
      gemv_t(j, v, corr.data());
      gemv_n_sub(j, v, corr.data());
      for (int c = 0; c < j; ++c) h[c] += corr[c];
      wnorm = nrm2(v);
      st.reorth++;
    }
    for (int c = 0; c < j; ++c) H(c, j - 1) = h[c];
    if (wnorm <= eta * rnorm) {
      H(j, j - 1) = T(0);
      return false;
    }
    H(j, j - 1) = T(wnorm);
    scal(v, 1.0 / wnorm);
    return true;
  }

  void iterate_arnoldi(int from, int to, const ks::Mat<T>& H, ks::ExpandStats& st) override {
    for (int j = from; j <= to; ++j) {
      double t0 = ks::now_s();
      spmv(col(j - 1), col(j));
      double t1 = ks::now_s();
      t_spmv += t1 - t0;
      st.steps++;
      const bool ok = orthogonalize(j, H, st);
      if (!ok && (int64_t)j != n) {
        reinitialize(j, nullptr);
        st.breakdowns++;
      }
      t_orth += ks::now_s() - t1;
    }
  }

  // reinitialize!, src/expansion.jl:12-59
  bool reinitialize(int j, const T* v1) override {
    constexpr double eta = 0.70710678118654752440;
    T* v = col(j);
    if (v1) std::memcpy(v, v1, (size_t)n * sizeof(T)); else fill_uniform(v);
    double rnorm = nrm2(v);
    if (j == 0) { scal(v, 1.0 / rnorm); return true; }
    std::vector<T> h(j);
    gemv_t(j, v, h.data());
    gemv_n_sub(j, v, h.data());
    double wnorm = nrm2(v);
    if (wnorm < eta * rnorm) {
      rnorm = wnorm;
      gemv_t(j, v, h.data());
      gemv_n_sub(j, v, h.data());
      wnorm = nrm2(v);
    }
    if (wnorm <= eta * rnorm) return false;
    scal(v, 1.0 / wnorm);
    return true;
  }

  // mul!(V_tmp, V, Q) + copyto!, src/run.jl:363-364 / :382-383
  void rotate(int c0, int c, int r, const ks::Mat<T>& Q) override {
    double t0 = ks::now_s();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      for (int jj = 0; jj < r; ++jj) {
        T s = T(0);
        for (int cc = 0; cc < c; ++cc) s += V[(size_t)(c0 + cc) * n + i] * Q(c0 + cc, c0 + jj);
        Vtmp[(size_t)(c0 + jj) * n + i] = s;
      }
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
      for (int jj = 0; jj < r; ++jj) V[(size_t)(c0 + jj) * n + i] = Vtmp[(size_t)(c0 + jj) * n + i];
    t_rot += ks::now_s() - t0;
  }
  void col_copy(int dst, int src) override {
    if (dst != src) std::memcpy(col(dst), col(src), (size_t)n * sizeof(T));
  }
};

thread_local std::string g_err;

template <class T>
int run_partialschur(int64_t n, const int32_t* rowptr, const int32_t* colidx, const void* val, int nev, int which,
                     double tol, int mindim, int maxdim, int restarts, const void* v1, uint64_t seed, int spmv_threads,
                     void* Hout, void* Vout, double* eig_c64, int32_t* hist_i, double* hist_d) {
  ks::Params p{nev, which, tol, mindim, maxdim, restarts, 1, 1};
  std::string msg;
  if (ks::check_params(n, maxdim + 1, p, msg)) { g_err = msg; return 1; }
  CpuBackend<T> be(n, maxdim, rowptr, colidx, static_cast<const T*>(val));
  be.seed = seed;
  be.spmv_threads = spmv_threads;
  std::vector<T> H((size_t)(maxdim + 1) * maxdim, T(0)), Q((size_t)maxdim * maxdim, T(0));
  ks::Mat<T> Hm(H.data(), maxdim + 1, maxdim, maxdim + 1), Qm(Q.data(), maxdim, maxdim, maxdim);
  be.reinitialize(0, static_cast<const T*>(v1));
  std::vector<cplx> lams(maxdim);
  ks::History h;
  try {
    h = ks::partialschur_driver<T>(be, Hm, Qm, p, 0, lams.data());
  } catch (const std::exception& e) {
    g_err = e.what();
    return 5;
  }
  if (Hout) std::memcpy(Hout, H.data(), H.size() * sizeof(T));
  if (Vout) std::memcpy(Vout, be.V.data(), (size_t)n * (maxdim + 1) * sizeof(T));
  for (int i = 0; i < h.nconverged; ++i) { eig_c64[2 * i] = lams[i].real(); eig_c64[2 * i + 1] = lams[i].imag(); }
  hist_i[0] = h.mvproducts; hist_i[1] = h.nconverged; hist_i[2] = h.converged; hist_i[3] = h.nev;
  hist_i[4] = h.restarts; hist_i[5] = h.reorth; hist_i[6] = h.breakdowns;
  hist_d[0] = be.t_spmv; hist_d[1] = be.t_orth; hist_d[2] = be.t_rot; hist_d[3] = h.seconds_host;
  return 0;
}

}  // namespace

extern "C" {

const char* ksref_last_error(void) { return g_err.c_str(); }
int ksref_num_threads(void) { return omp_get_max_threads(); }

// dtype 0 = f64, 1 = c64.  CSR int32 0-based.  H: (maxdim+1) x maxdim, V: n x (maxdim+1), column-major.
int ksref_partialschur_csr(int dtype, int64_t n, const int32_t* rowptr, const int32_t* colidx, const void* val, int nev,
                           int which, double tol, int mindim, int maxdim, int restarts, const void* v1, uint64_t seed,
                           int spmv_threads, void* Hout, void* Vout, double* eig_c64, int32_t* hist_i, double* hist_d) {
  if (dtype == 0)
    return run_partialschur<double>(n, rowptr, colidx, val, nev, which, tol, mindim, maxdim, restarts, v1, seed,
                                    spmv_threads, Hout, Vout, eig_c64, hist_i, hist_d);
  return run_partialschur<cplx>(n, rowptr, colidx, val, nev, which, tol, mindim, maxdim, restarts, v1, seed,
                                spmv_threads, Hout, Vout, eig_c64, hist_i, hist_d);
}

// Timed fixed-work sample for bench.py's cpu_baseline: `cycles` restart cycles (expansion k+1..maxdim +
// host Schur + rotation) regardless of convergence, after the initial expansion to mindim.
// out_d: [0] seconds total (timed region), [1] spmv s, [2] orth s, [3] rotation s, [4] host s; out_i: [0] steps timed.
int ksref_timed_cycles_csr(int64_t n, const int32_t* rowptr, const int32_t* colidx, const double* val, int nev, int which,
                           int mindim, int maxdim, int cycles, uint64_t seed, int spmv_threads, double* out_d,
                           int32_t* out_i) {
  CpuBackend<double> be(n, maxdim, rowptr, colidx, val);
  be.seed = seed;
  be.spmv_threads = spmv_threads;
  std::vector<double> H((size_t)(maxdim + 1) * maxdim, 0.0), Q((size_t)maxdim * maxdim, 0.0);
  ks::Mat<double> Hm(H.data(), maxdim + 1, maxdim, maxdim + 1), Qm(Q.data(), maxdim, maxdim, maxdim);
  ks::ExpandStats st;
  ks::RestartScratch<double> scratch(maxdim);
  be.reinitialize(0, nullptr);
  be.iterate_arnoldi(1, mindim, Hm, st);
  int k = mindim, active = 0;
  be.t_spmv = be.t_orth = be.t_rot = 0;
  double t_host = 0;
  int steps = 0;
  const double t0 = ks::now_s();
  for (int it = 0; it < cycles; ++it) {
    be.iterate_arnoldi(k + 1, maxdim, Hm, st);
    steps += maxdim - k;
    const double th = ks::now_s();
    // tol = 0: nothing ever locks, so every cycle does the same amount of work
    const ks::RestartResult r = ks::restart_host_step(Hm, Qm, maxdim, mindim, nev, ks::Ordering{which}, 0.0, active, scratch);
    t_host += ks::now_s() - th;
    k = r.k;
    be.rotate(r.purge, maxdim - r.purge, k - r.purge, Qm);
    be.col_copy(k, maxdim);
    active = r.nlock;
  }
  out_d[0] = ks::now_s() - t0;
  out_d[1] = be.t_spmv; out_d[2] = be.t_orth; out_d[3] = be.t_rot; out_d[4] = t_host;
  out_i[0] = steps;
  out_i[1] = st.reorth;
  return 0;
}

}  // extern "C"
