/* C restatement of the ballooning hot path.  TEST INFRASTRUCTURE ONLY (CPU checker and the
 * cpu_baseline leg of bench.py); never linked into or called from the product path.
 *
 * Parity status: pinned.  tests/test_oracle_c.py checks it against oracle/ballooning_oracle.py,
 * which is pinned to golden vectors captured from the reference (tests/golden/make_golden.py).
 *
 * Algorithm ("best CPU" formulation of SURVEY.md 8d, deliberately different from the HIP path):
 *   coefficients / tridiagonal assembly   utils.py:1556-1592
 *   lam_max by LAPACK-dstebz-style Sturm bisection (division form, pivmin guard)
 *   eigenvector by inverse iteration with partial-pivoting tridiagonal LU (dlagtf/dlagts style)
 *   X, dX, Simpson Rayleigh quotient      utils.py:1601-1621
 * All functions cite the reference lines they follow (paths relative to /root/reference).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* number of eigenvalues of (T, F) above lam; d[n], e[n+1] (e[r] couples rows r-1, r), f[n] */
static int count_above(int n, const double* d, const double* e, const double* f, double lam, double pivmin) {
  int cnt = 0;
  double q = d[0] - lam * f[0];
  if (fabs(q) < pivmin) q = -pivmin;
  if (q > 0) cnt++;
  for (int r = 1; r < n; ++r) {
    q = (d[r] - lam * f[r]) - e[r] * e[r] / q;
    if (fabs(q) < pivmin) q = -pivmin;
    if (q > 0) cnt++;
  }
  return cnt;
}

/* solve (T - lam F) y = b in place (b -> y), partial pivoting; a tiny pivot is replaced (dlagts) */
static void solve_shifted(int n, const double* d, const double* e, const double* f, double lam, double* b, double* w) {
  double* dl = w; double* dd = w + n; double* du = w + 2 * n; double* du2 = w + 3 * n;
  double tiny = 0;
  for (int r = 0; r < n; ++r) {
    dd[r] = d[r] - lam * f[r];
    dl[r] = (r > 0) ? e[r] : 0.0;          /* sub-diagonal entry of row r */
    du[r] = (r < n - 1) ? e[r + 1] : 0.0;  /* super-diagonal entry of row r */
    du2[r] = 0.0;
    double s = fabs(dd[r]) + fabs(dl[r]) + fabs(du[r]);
    if (s > tiny) tiny = s;
  }
  tiny *= DBL_EPSILON;
  for (int r = 0; r < n - 1; ++r) {
    if (fabs(dd[r]) >= fabs(dl[r + 1])) {
      double piv = dd[r];
      if (fabs(piv) < tiny) piv = (piv < 0) ? -tiny : tiny;
      double m = dl[r + 1] / piv;
      dd[r] = piv;
      dd[r + 1] -= m * du[r];
      b[r + 1] -= m * b[r];
    } else { /* swap rows r, r+1 */
      double m = dd[r] / dl[r + 1];
      double t0 = dd[r + 1], t1 = du[r + 1];
      dd[r] = dl[r + 1];
      double u0 = du[r];
      du[r] = t0; du2[r] = t1;
      dd[r + 1] = u0 - m * t0;
      du[r + 1] = -m * t1;
      double tb = b[r]; b[r] = b[r + 1]; b[r + 1] = tb - m * b[r + 1];
    }
  }
  if (fabs(dd[n - 1]) < tiny) dd[n - 1] = (dd[n - 1] < 0) ? -tiny : tiny;
  b[n - 1] /= dd[n - 1];
  if (n > 1) b[n - 2] = (b[n - 2] - du[n - 2] * b[n - 1]) / dd[n - 2];
  for (int r = n - 3; r >= 0; --r) b[r] = (b[r] - du[r] * b[r + 1] - du2[r] * b[r + 2]) / dd[r];
}

/* workspace: 12*N doubles */
int ibs_oracle_solve_gcf(int N, double h, const double* g, const double* c, const double* f,
                         double* lam_out, double* gam_out, double* X, double* dX, double* work) {
  const int n = N - 2;
  if (N < 7 || (N & 1) == 0) return -1;
  double* d = work; double* e = work + N; double* fd = work + 2 * N; double* x = work + 3 * N; double* w = work + 4 * N;
  double* Xl = work + 8 * N; double* dXl = work + 9 * N;
  const double ih2 = 1.0 / (h * h);
  /* utils.py:1574-1592: half-grid g, h^2-scaled tridiagonal */
  for (int k = 0; k < N - 1; ++k) e[k] = 0.5 * (g[k] + g[k + 1]) * ih2;
  double hi = -DBL_MAX, lo = -DBL_MAX, nrm = 0;
  for (int r = 0; r < n; ++r) {
    d[r] = -(e[r] + e[r + 1]) + c[r + 1];
    fd[r] = f[r + 1];
    double v = c[r + 1] / fd[r];
    if (v > hi) hi = v;
    v = d[r] / fd[r];
    if (v > lo) lo = v;
    v = (fabs(d[r]) + e[r] + e[r + 1]) / fd[r];
    if (v > nrm) nrm = v;
  }
  double pivmin = DBL_MIN * 1e16;
  hi += 4 * DBL_EPSILON * nrm; lo -= 4 * DBL_EPSILON * nrm;
  /* bisection on count_above: lam_max = sup{lam : count_above(lam) >= 1} */
  for (int it = 0; it < 200 && hi - lo > 2 * DBL_EPSILON * fmax(fabs(lo), fabs(hi)) + 4 * DBL_MIN; ++it) {
    double mid = 0.5 * (lo + hi);
    if (count_above(n, d, e, fd, mid, pivmin) >= 1) lo = mid; else hi = mid;
  }
  double lam = 0.5 * (lo + hi);
  /* inverse iteration (generalised: rhs = F x) */
  for (int r = 0; r < n; ++r) x[r] = 1.0;
  for (int it = 0; it < 4; ++it) {
    double mx = 0;
    for (int r = 0; r < n; ++r) x[r] *= fd[r];
    solve_shifted(n, d, e, fd, lam, x, w);
    for (int r = 0; r < n; ++r) if (fabs(x[r]) > mx) mx = fabs(x[r]);
    for (int r = 0; r < n; ++r) x[r] /= mx;
  }
  /* utils.py:1602-1608: normalise by max |x| (sign made positive at the maximum) */
  int imax = 0;
  for (int r = 1; r < n; ++r) if (fabs(x[r]) > fabs(x[imax])) imax = r;
  const double sc = x[imax];
  Xl[0] = 0; Xl[N - 1] = 0;
  for (int r = 0; r < n; ++r) Xl[r + 1] = x[r] / sc;
  /* utils.py:1610-1616 */
  dXl[0] = (-1.5 * Xl[0] + 2 * Xl[1] - 0.5 * Xl[2]) / h;
  dXl[1] = (Xl[2] - Xl[0]) / (2 * h);
  dXl[N - 2] = (Xl[N - 1] - Xl[N - 3]) / (2 * h);
  dXl[N - 1] = (0.5 * Xl[N - 3] - 2 * Xl[N - 2] + 1.5 * 0.0) / h;
  for (int j = 2; j <= N - 3; ++j) dXl[j] = 2 / (3 * h) * (Xl[j + 1] - Xl[j - 1]) - (Xl[j + 2] - Xl[j - 2]) / (12 * h);
  /* utils.py:1618-1621: composite Simpson, unit spacing, odd N */
  double y0 = 0, y1 = 0;
  for (int j = 0; j < N; ++j) {
    double wgt = (j == 0 || j == N - 1) ? 1.0 : ((j & 1) ? 4.0 : 2.0);
    y0 += wgt * (-g[j] * dXl[j] * dXl[j] + c[j] * Xl[j] * Xl[j]);
    y1 += wgt * (f[j] * Xl[j] * Xl[j]);
  }
  if (lam_out) *lam_out = lam;
  if (gam_out) *gam_out = (y0 / 3.0) / (y1 / 3.0);
  if (X) memcpy(X, Xl, sizeof(double) * N);
  if (dX) memcpy(dX, dXl, sizeof(double) * N);
  return 0;
}

/* batched raw systems, OpenMP over systems.  Returns threads used. */
int ibs_oracle_solve_gcf_batch(long n_sys, int N, double h, const double* g, const double* c, const double* f,
                               long ld, double* lam, double* gam, int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  used = omp_get_max_threads();
#endif
#pragma omp parallel
  {
    double* work = (double*)malloc(sizeof(double) * 12 * (size_t)N);
#pragma omp for schedule(dynamic, 4)
    for (long s = 0; s < n_sys; ++s)
      ibs_oracle_solve_gcf(N, h, g + s * ld, c + s * ld, f + s * ld, lam ? lam + s : 0, gam ? gam + s : 0, 0, 0, work);
    free(work);
  }
  return used;
}

/* geometry-fed theta0 scan (ball_scan.py:248-275 inner loops + utils.py:1556-1624), OpenMP over (line, theta0).
 * geometry arrays [n_lines][ld]; outputs [n_lines][n_theta0].  Returns threads used. */
int ibs_oracle_gamma_scan(int n_lines, int n_theta0, int N, double h, const double* bmag, const double* gradpar,
                          const double* cvdrift, const double* cvdrift0, const double* gds2, const double* gds21,
                          const double* gds22, long ld, const double* dPdrho, const double* theta0, double* gam,
                          double* lam, int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  used = omp_get_max_threads();
#endif
  const long n_sys = (long)n_lines * n_theta0;
#pragma omp parallel
  {
    double* work = (double*)malloc(sizeof(double) * 15 * (size_t)N);
    double* g = work + 12 * (size_t)N; double* c = g + N; double* f = c + N;
#pragma omp for schedule(dynamic, 2)
    for (long s = 0; s < n_sys; ++s) {
      const int line = (int)(s / n_theta0), it = (int)(s % n_theta0);
      const long o = (long)line * ld;
      const double t0 = theta0[it], dP = dPdrho[line];
      for (int j = 0; j < N; ++j) {
        const double cv = cvdrift[o + j] + t0 * cvdrift0[o + j];                          /* ball_scan.py:267 */
        const double gd = gds2[o + j] + 2 * t0 * gds21[o + j] + t0 * t0 * gds22[o + j];   /* ball_scan.py:268 */
        const double gp = fabs(gradpar[o + j]), B = bmag[o + j];
        g[j] = gp * gd / B;                          /* utils.py:1560 */
        c[j] = -1 * dP * cv * 1 / (gp * B);          /* utils.py:1561 */
        f[j] = gd / (B * B) * 1 / (gp * B);          /* utils.py:1562 */
      }
      ibs_oracle_solve_gcf(N, h, g, c, f, lam ? lam + s : 0, gam ? gam + s : 0, 0, 0, work);
    }
    free(work);
  }
  return used;
}

/* lam_max alone (the bisection of ibs_oracle_solve_gcf without the eigenvector stage), OpenMP over systems: the arbiter of the
 * 10^6-system eigenvalue campaigns (tests/tools/reclose_campaign.py).  Returns threads used. */
int ibs_oracle_lam_batch(long n_sys, int N, double h, const double* g, const double* c, const double* f, long ld, double* lam,
                         int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  used = omp_get_max_threads();
#endif
  const int n = N - 2;
  const double ih2 = 1.0 / (h * h);
#pragma omp parallel
  {
    double* d = (double*)malloc(sizeof(double) * 3 * (size_t)N);
    double* e = d + N; double* fd = e + N;
#pragma omp for schedule(dynamic, 16)
    for (long s = 0; s < n_sys; ++s) {
      const double* gs = g + s * ld; const double* cs = c + s * ld; const double* fs = f + s * ld;
      for (int k = 0; k < N - 1; ++k) e[k] = 0.5 * (gs[k] + gs[k + 1]) * ih2;          /* utils.py:1574-1576 */
      double hi = -DBL_MAX, lo = -DBL_MAX, nrm = 0;
      for (int r = 0; r < n; ++r) {
        d[r] = -(e[r] + e[r + 1]) + cs[r + 1];                                          /* utils.py:1584-1592 */
        fd[r] = fs[r + 1];
        double v = cs[r + 1] / fd[r]; if (v > hi) hi = v;
        v = d[r] / fd[r]; if (v > lo) lo = v;
        v = (fabs(d[r]) + e[r] + e[r + 1]) / fd[r]; if (v > nrm) nrm = v;
      }
      const double pivmin = DBL_MIN * 1e16;
      hi += 4 * DBL_EPSILON * nrm; lo -= 4 * DBL_EPSILON * nrm;
      for (int it = 0; it < 200 && hi - lo > 2 * DBL_EPSILON * fmax(fabs(lo), fabs(hi)) + 4 * DBL_MIN; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (count_above(n, d, e, fd, mid, pivmin) >= 1) lo = mid; else hi = mid;
      }
      lam[s] = 0.5 * (lo + hi);
    }
    free(d);
  }
  return used;
}

/* division-form Sturm count (eigenvalues of (T, F) above shift[s]) of every system: the arbiter for counts the product-form
 * GPU sweeps get wrong next to an eigenvalue.  Returns threads used. */
int ibs_oracle_count_above_batch(long n_sys, int N, double h, const double* g, const double* c, const double* f, long ld,
                                 const double* shift, int* count, int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  used = omp_get_max_threads();
#endif
  const int n = N - 2;
  const double ih2 = 1.0 / (h * h);
#pragma omp parallel
  {
    double* d = (double*)malloc(sizeof(double) * 3 * (size_t)N);
    double* e = d + N; double* fd = e + N;
#pragma omp for schedule(dynamic, 16)
    for (long s = 0; s < n_sys; ++s) {
      const double* gs = g + s * ld; const double* cs = c + s * ld; const double* fs = f + s * ld;
      for (int k = 0; k < N - 1; ++k) e[k] = 0.5 * (gs[k] + gs[k + 1]) * ih2;
      for (int r = 0; r < n; ++r) { d[r] = -(e[r] + e[r + 1]) + cs[r + 1]; fd[r] = fs[r + 1]; }
      count[s] = count_above(n, d, e, fd, shift[s], DBL_MIN * 1e16);
    }
    free(d);
  }
  return used;
}
