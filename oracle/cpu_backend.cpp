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
#include <memory>
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
  uint64_t seed = 20240917ull, rng_count = 0;
  int spmv_threads = 0;  // 0 = all
  double t_spmv = 0, t_orth = 0, t_rot = 0;

  CpuBackend(int64_t n_, int maxdim_, const int32_t* rp, const int32_t* ci, const T* v)
      : n(n_), maxdim(maxdim_), rowptr(rp), colidx(ci), val(v) {
    // first-touch in parallel with the same static row partition every kernel uses, so each thread's
    // rows live on its own NUMA node (std::vector's serial value-initialisation would put all of V on one)
    Vbuf.reset(new T[(size_t)n * (maxdim + 1)]);
    Tbuf.reset(new T[(size_t)n * (maxdim + 1)]);
    V = Vbuf.get();
    Vtmp = Tbuf.get();
#pragma omp parallel
    {
      int64_t a, b;
      my_rows(a, b);
      for (int c = 0; c <= maxdim; ++c)
        for (int64_t i = a; i < b; ++i) { V[(size_t)c * n + i] = T(0); Vtmp[(size_t)c * n + i] = T(0); }
    }
  }

  std::unique_ptr<T[]> Vbuf, Tbuf;
  T* V = nullptr;
  T* Vtmp = nullptr;

  // contiguous row range of the calling thread
  void my_rows(int64_t& a, int64_t& b) const {
    const int nt = omp_get_num_threads(), t = omp_get_thread_num();
    const int64_t per = (n + nt - 1) / nt;
    a = std::min<int64_t>(n, t * per);
    b = std::min<int64_t>(n, a + per);
  }

  T* col(int j) { return V + (size_t)j * n; }
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
  // h = V[:,0:j)^H v   (gemv 'T'/'C'): every thread streams its row block column by column (its slice of v
  // stays in cache), partial results are summed in thread order
  void gemv_t(int j, const T* v, T* h) {
    const int ntmax = omp_get_max_threads();
    std::vector<T> part((size_t)ntmax * j, T(0));
    int used = 1;
#pragma omp parallel
    {
      int64_t a, b;
      my_rows(a, b);
      const int t = omp_get_thread_num();
#pragma omp single
      used = omp_get_num_threads();
      for (int c = 0; c < j; ++c) {
        const T* vc = V + (size_t)c * n;
        double sr = 0.0, si = 0.0;
#pragma omp simd reduction(+ : sr, si)
        for (int64_t i = a; i < b; ++i) {
          const T t2 = ks::conj_(vc[i]) * v[i];
          sr += ks::real_(t2);
          si += ks::imag_(t2);
        }
        if constexpr (ks::is_real_v<T>) part[(size_t)t * j + c] = sr; else part[(size_t)t * j + c] = cplx(sr, si);
      }
    }
    for (int c = 0; c < j; ++c) {
      T s = T(0);
      for (int t = 0; t < used; ++t) s += part[(size_t)t * j + c];
      h[c] = s;
    }
  }
  // v -= V[:,0:j) h   (gemv 'N' with alpha = -1, beta = 1)
  void gemv_n_sub(int j, T* v, const T* h) {
#pragma omp parallel
    {
      int64_t a, b;
      my_rows(a, b);
      for (int c = 0; c < j; ++c) {
        const T* vc = V + (size_t)c * n;
        const T hc = h[c];
#pragma omp simd
        for (int64_t i = a; i < b; ++i) v[i] -= vc[i] * hc;
      }
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
#pragma omp parallel
    {
      int64_t a, b;
      my_rows(a, b);
      constexpr int64_t BS = 256;  // row tile kept in cache while looping over the small Q
      for (int64_t i0 = a; i0 < b; i0 += BS) {
        const int64_t i1 = std::min(b, i0 + BS);
        for (int jj = 0; jj < r; ++jj) {
          T* out = Vtmp + (size_t)(c0 + jj) * n;
          for (int64_t i = i0; i < i1; ++i) out[i] = T(0);
          for (int cc = 0; cc < c; ++cc) {
            const T q = Q(c0 + cc, c0 + jj);
            const T* vc = V + (size_t)(c0 + cc) * n;
#pragma omp simd
            for (int64_t i = i0; i < i1; ++i) out[i] += vc[i] * q;
          }
        }
      }
#pragma omp barrier
      for (int jj = 0; jj < r; ++jj) {  // copyto!(V, V_tmp)
        const T* in = Vtmp + (size_t)(c0 + jj) * n;
        T* out = V + (size_t)(c0 + jj) * n;
        for (int64_t i = a; i < b; ++i) out[i] = in[i];
      }
    }
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
  if (Vout) std::memcpy(Vout, be.V, (size_t)n * (maxdim + 1) * sizeof(T));
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

// STREAM-triad bandwidth of the host (a[i] = b[i] + s * c[i] over three arrays of `n` doubles, first-touched by the
// threads that stream them, best of `reps`): the ceiling the bandwidth-bound CPU baseline can be judged against.
// Returns GB/s counting 24 bytes per element.
double ksref_stream_triad_gbs(int64_t n, int reps) {
  std::unique_ptr<double[]> a(new double[n]), b(new double[n]), c(new double[n]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    const double t0 = ks::now_s();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) a[i] = b[i] + 3.0 * c[i];
    best = std::min(best, ks::now_s() - t0);
  }
  return 24.0 * (double)n / best / 1e9 + 0.0 * a[n / 2];
}

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
