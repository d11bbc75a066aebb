// Where the host restart step (src/run.jl:278-360) spends its time: stage timers of restart_host_step on a sequence of
// REAL restarts (CPU backend of the oracle is not needed: a symmetric tridiagonal operator applied on the host drives the
// same H the 3-D Laplacian produces).  Build + run:  g++ -O3 -std=c++17 -DKS_TIME_STAGES tools/host_step_profile.cpp -o tools/_build/hsp && tools/_build/hsp
#include <cstdio>
#include <random>
#include <vector>

#include "../arnoldimethod.jl_amd/csrc/ks_driver.hpp"

using namespace ks;

// minimal host backend: dense-free 3-D Laplacian (20 x 21 x 22), classical Gram-Schmidt with DGKS like the oracle
struct HostBackend : Backend<double> {
  int mx = 20, my = 21, mz = 22;
  int64_t n;
  int maxdim;
  std::vector<double> V;
  HostBackend(int md) : n((int64_t)20 * 21 * 22), maxdim(md), V((size_t)n * (md + 1)) {}
  int64_t n_global() const override { return n; }
  double* col(int j) { return V.data() + (size_t)j * n; }
  void apply(const double* x, double* y) {
    for (int z = 0; z < mz; ++z)
      for (int yy = 0; yy < my; ++yy)
        for (int xx = 0; xx < mx; ++xx) {
          const int64_t i = xx + mx * (yy + (int64_t)my * z);
          double s = 6 * x[i];
          if (xx > 0) s -= x[i - 1];
          if (xx < mx - 1) s -= x[i + 1];
          if (yy > 0) s -= x[i - mx];
          if (yy < my - 1) s -= x[i + mx];
          if (z > 0) s -= x[i - mx * my];
          if (z < mz - 1) s -= x[i + mx * my];
          y[i] = s;
        }
  }
  void iterate_arnoldi(int from, int to, const Mat<double>& H, ExpandStats& st) override {
    for (int j = from; j <= to; ++j) {
      double* w = col(j);
      apply(col(j - 1), w);
      for (int pass = 0; pass < 2; ++pass)
        for (int c = 0; c < j; ++c) {
          double h = 0;
          for (int64_t i = 0; i < n; ++i) h += col(c)[i] * w[i];
          for (int64_t i = 0; i < n; ++i) w[i] -= h * col(c)[i];
          H(c, j - 1) = pass == 0 ? h : H(c, j - 1) + h;
        }
      double nr = 0;
      for (int64_t i = 0; i < n; ++i) nr += w[i] * w[i];
      nr = std::sqrt(nr);
      H(j, j - 1) = nr;
      for (int64_t i = 0; i < n; ++i) w[i] /= nr;
      st.steps++;
    }
  }
  bool reinitialize(int j, const double*) override {
    std::mt19937_64 g(7);
    double nr = 0;
    for (int64_t i = 0; i < n; ++i) { col(j)[i] = (double)(g() >> 11) / 9007199254740992.0; nr += col(j)[i] * col(j)[i]; }
    nr = std::sqrt(nr);
    for (int64_t i = 0; i < n; ++i) col(j)[i] /= nr;
    return true;
  }
  void rotate(int c0, int c, int r, const Mat<double>& Q) override {
    std::vector<double> tmp((size_t)n * r, 0.0);
    for (int jj = 0; jj < r; ++jj)
      for (int ii = 0; ii < c; ++ii) {
        const double q = Q(c0 + ii, c0 + jj);
        for (int64_t i = 0; i < n; ++i) tmp[(size_t)jj * n + i] += q * col(c0 + ii)[i];
      }
    for (int jj = 0; jj < r; ++jj) std::copy(tmp.begin() + (size_t)jj * n, tmp.begin() + (size_t)(jj + 1) * n, col(c0 + jj));
  }
  void col_copy(int dst, int src) override { std::copy(col(src), col(src) + n, col(dst)); }
};

int main() {
  const int maxdim = 40, mindim = 20, nev = 20;
  std::vector<double> Hs((size_t)(maxdim + 1) * maxdim, 0.0), Qs((size_t)maxdim * maxdim, 0.0);
  Mat<double> H(Hs.data(), maxdim + 1, maxdim, maxdim + 1), Q(Qs.data(), maxdim, maxdim, maxdim);
  HostBackend be(maxdim);
  be.reinitialize(0, nullptr);
  Params p{nev, /*SR*/ 2, 1.5e-8, mindim, maxdim, 40, 1, 0};
  std::vector<cplx> lam(maxdim);
  const double t0 = now_s();
  History h = partialschur_driver<double>(be, H, Q, p, 0, lam.data());
  const double total = now_s() - t0;
  const char* names[5] = {"local_schurfact (Q <- I, Francis QR)", "eigenvalues + residuals (eigenvectors of R)", "ordering + grouping", "partition_schur_three_way (reordering)", "restore_arnoldi (Householder)"};
  double sum = 0;
  for (int i = 0; i < 5; ++i) sum += g_stage_s[i];
  std::printf("%d restarts, host step total %.1f us per restart (driver says %.1f us), whole solve %.2f s\n", h.restarts, 1e6 * sum / h.restarts,
              1e6 * h.seconds_host / h.restarts, total);
  for (int i = 0; i < 5; ++i) std::printf("  %-48s %8.1f us  %5.1f %%\n", names[i], 1e6 * g_stage_s[i] / h.restarts, 100 * g_stage_s[i] / sum);
  return 0;
}
