// Host side of the geometry producer: the radial step.  Plain C++ (g++), threaded; no GPU involved.
//
// The reference builds one InterpolatedUnivariateSpline per Fourier mode and array (vmec_splines, utils.py:58-119) and
// evaluates each at the surface (utils.py:311-357): 4,588 scalar spline evaluations per surface.  Cubic interpolating
// splines are linear in the data and every mode shares the radial mesh, so the evaluation at the requested surfaces is
// four weight matrices (value / derivative on VMEC's full / half mesh; built once by the Python layer by splining the
// identity with the same FITPACK-type not-a-knot spline) applied to the (mode, ns) tables.  One optimizer step hands
// over totalndofs + 1 = 73 equilibria (sims_runner_NCSX.py:151-276): 115 MB of tables that are each read exactly once
// here, by as many threads as the host offers.
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../include/ibs.h"

namespace {

// sum_j w[j] * x[j] with 8 partial sums in a fixed order (results do not depend on the thread count; the compiler turns
// the partial sums into vector lanes)
__attribute__((always_inline)) inline double dot(const double* __restrict__ w, const double* __restrict__ x, int n) {
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int j = 0;
  for (; j + 8 <= n; j += 8)
    for (int l = 0; l < 8; ++l) a[l] += w[j + l] * x[j + l];
  for (; j < n; ++j) a[0] += w[j] * x[j];
  return ((a[0] + a[4]) + (a[1] + a[5])) + ((a[2] + a[6]) + (a[3] + a[7]));
}

struct Family { int src; int nw; int w[2]; int dst[2]; bool nyq; };
// source arrays (order of the `tabs` pointer table): rmnc zmns lmns gmnc bmnc bsupvmnc bsubsmns bsubumnc bsubvmnc
// weights: 0 = full mesh value, 1 = full mesh derivative, 2 = half mesh value, 3 = half mesh derivative
// destinations: tab_mn planes rmnc zmns lmns d_rmnc_d_s d_zmns_d_s d_lmns_d_s; tab_nyq planes gmnc bmnc d_bmnc_d_s
// bsupvmnc bsubsmns bsubumnc bsubvmnc (include/ibs.h: ibs_fieldline_geometry_f64)
constexpr Family kFam[9] = {
    {0, 2, {0, 1}, {0, 3}, false}, {1, 2, {0, 1}, {1, 4}, false}, {2, 2, {2, 3}, {2, 5}, false},
    {3, 1, {2, 0}, {0, 0}, true},  {4, 2, {2, 3}, {1, 2}, true},  {5, 1, {2, 0}, {3, 0}, true},
    {6, 1, {0, 0}, {4, 0}, true},  {7, 1, {2, 0}, {5, 0}, true},  {8, 1, {2, 0}, {6, 0}, true}};

// one array of one equilibrium: dst_k[s * stride + m] = sum_j W_k[s][j] * src[m][j], k < nw.  Compiled for AVX-512 / AVX2
// as well as the baseline ISA and picked at load time (the library is built once and travels to other hosts).
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
void apply_family(const double* __restrict__ src, int nm, int ns, int n_s, int nw, const double* __restrict__ w0,
                  const double* __restrict__ w1, double* __restrict__ d0, double* __restrict__ d1, size_t stride) {
  for (int m = 0; m < nm; ++m) {
    const double* row = src + (size_t)m * ns;
    for (int s = 0; s < n_s; ++s) d0[(size_t)s * stride + m] = dot(w0 + (size_t)s * ns, row, ns);
    if (nw > 1)
      for (int s = 0; s < n_s; ++s) d1[(size_t)s * stride + m] = dot(w1 + (size_t)s * ns, row, ns);
  }
}

}  // namespace

extern "C" int ibs_surface_tables_f64(int32_t n_eq, int32_t ns, int32_t n_s, int32_t mnmax, int32_t mnmax_nyq,
                                      const double* const* tabs, const double* w_full, const double* w_full_d,
                                      const double* w_half, const double* w_half_d, double* tab_mn, double* tab_nyq,
                                      int32_t n_threads) {
  if (n_eq < 0 || ns < 2 || n_s <= 0 || mnmax <= 0 || mnmax_nyq <= 0 || !tabs || !w_full || !w_full_d || !w_half ||
      !w_half_d || !tab_mn || !tab_nyq)
    return IBS_ERR_ARG;
  for (long i = 0; i < 9L * n_eq; ++i) if (!tabs[i]) return IBS_ERR_ARG;
  const double* W[4] = {w_full, w_full_d, w_half, w_half_d};       // each [n_s][ns]; the half-mesh ones with a zero column 0
  // work items: (equilibrium, family)
  const long n_items = 9L * n_eq;
  auto work = [&](long i0, long i1) {
    for (long it = i0; it < i1; ++it) {
      const int q = (int)(it / 9);
      const Family& f = kFam[it % 9];
      const int nm = f.nyq ? mnmax_nyq : mnmax, planes = f.nyq ? 7 : 6;
      double* dst = (f.nyq ? tab_nyq : tab_mn) + (size_t)q * n_s * planes * nm;
      apply_family(tabs[it], nm, ns, n_s, f.nw, W[f.w[0]], W[f.w[1]], dst + (size_t)f.dst[0] * nm, dst + (size_t)f.dst[1] * nm,
                   (size_t)planes * nm);
    }
  };
  // default: up to 16 threads (one GPU's share of a multi-GPU host; measured on the 256-thread MI355X host: 73 equilibria in
  // 10.2 ms with 1 thread, 2.5 with 8, 2.2 with 16, 2.3 with 32 -- and 10 ms again with one thread per hardware thread)
  int nt = n_threads > 0 ? n_threads : std::min(16, (int)std::thread::hardware_concurrency());
  nt = (int)std::max<long>(1, std::min<long>(nt, n_items));
  if (nt == 1) { work(0, n_items); return 0; }
  std::vector<std::thread> th;
  th.reserve(nt);
  for (int t = 0; t < nt; ++t) th.emplace_back(work, n_items * t / nt, n_items * (t + 1) / nt);
  for (auto& t : th) t.join();
  return 0;
}
