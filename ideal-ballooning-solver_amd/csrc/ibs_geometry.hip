// Field-line geometry on the device: the producer of the eight arrays the
// ballooning kernels consume, so that geometry never leaves HBM between the equilibrium tables and
// the eigen-solve.  Restates the arithmetic of the reference's vmec_fieldlines (utils.py:359-720);
// the radial splines (utils.py:37-158, 311-357) stay on the host and hand over per-surface Fourier
// coefficient vectors.
//
// One thread per grid point: theta_pest -> theta_vmec by a secant solve on
//   theta_vmec + sum_mn lmns sin(m theta_vmec - n phi) = theta_pest            (utils.py:391-416)
// then two Fourier syntheses (mnmax and Nyquist mode sets, utils.py:420-468) and the metric algebra
// (utils.py:474-720).  Mode coefficients are wave-uniform (a block works on one field line), so they
// arrive through scalar loads; the work is FP64 sin/cos evaluation, i.e. compute bound.
#include <hip/hip_runtime.h>
#include "ibs_launch.hpp"

namespace ibs {

struct GeoArgs {     // must match the declaration in ibs_api.hip
  int n_surf, mnmax, mnmax_nyq, n_lines, N;
  const double *xm, *xn, *xm_nyq, *xn_nyq;
  const double* tab_mn;    // [n_surf][6][mnmax]      rmnc zmns lmns d_rmnc_d_s d_zmns_d_s d_lmns_d_s
  const double* tab_nyq;   // [n_surf][7][mnmax_nyq]  gmnc bmnc d_bmnc_d_s bsupvmnc bsubsmns bsubumnc bsubvmnc
  const double* scal;      // [n_surf][6]             s iota d_iota_d_s d_pressure_d_s phiedge Aminor_p
  const int* line_surf; const double* line_alpha; const double* theta;
  long ld;
  double* geo;             // [8][n_lines][ld]  bmag gradpar cvdrift cvdrift0 gds2 gds21 gds22 gbdrift
  double* dPdrho;          // [n_lines]
};

__global__ void __launch_bounds__(256) k_fieldline_geometry(GeoArgs a) {
  const int line = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.N) return;
  const int js = a.line_surf[line];
  const double* sc = a.scal + 6 * js;
  const double s = sc[0], iota = sc[1], diota = sc[2], dp = sc[3], phiedge = sc[4], L = sc[5];
  const double alpha = a.line_alpha[line];
  const double tp = a.theta[j];
  const double phi = (tp - alpha) / iota;                                    // utils.py:373 (phi_center = 0)
  const double* rmnc = a.tab_mn + (size_t)js * 6 * a.mnmax;
  const double* zmns = rmnc + a.mnmax; const double* lmns = zmns + a.mnmax;
  const double* drmnc = lmns + a.mnmax; const double* dzmns = drmnc + a.mnmax; const double* dlmns = dzmns + a.mnmax;
  // ---- theta_vmec: secant from (theta_p, theta_p + 0.1)                     utils.py:391-416
  auto resid = [&](double tv) {
    double acc = 0.0;
    for (int k = 0; k < a.mnmax; ++k) acc += lmns[k] * sin(a.xm[k] * tv - a.xn[k] * phi);
    return tp - (tv + acc);
  };
  double p0 = tp, p1 = tp + 0.1;
  double q0 = resid(p0), q1 = resid(p1);
  for (int it = 0; it < 60; ++it) {
    const double den = q1 - q0;
    if (den == 0.0) break;
    const double step = q1 * (p1 - p0) / den;
    p0 = p1; q0 = q1;
    p1 = p1 - step;
    if (fabs(step) <= 1e-15 * fmax(1.0, fabs(p1))) break;
    q1 = resid(p1);
  }
  const double tv = p1;
  // ---- Fourier synthesis, non-Nyquist set                                   utils.py:420-444
  double R = 0, R_s = 0, R_t = 0, R_p = 0, Z_s = 0, Z_t = 0, Z_p = 0, l_s = 0, l_t = 0, l_p = 0;
  for (int k = 0; k < a.mnmax; ++k) {
    const double m = a.xm[k], n = a.xn[k];
    double sa, ca;
    sincos(m * tv - n * phi, &sa, &ca);
    R += rmnc[k] * ca; R_s += drmnc[k] * ca; R_t -= rmnc[k] * m * sa; R_p += rmnc[k] * n * sa;
    Z_s += dzmns[k] * sa; Z_t += zmns[k] * m * ca; Z_p -= zmns[k] * n * ca;
    l_s += dlmns[k] * sa; l_t += lmns[k] * m * ca; l_p -= lmns[k] * n * ca;
  }
  // ---- Nyquist set                                                          utils.py:447-468
  const double* gmnc = a.tab_nyq + (size_t)js * 7 * a.mnmax_nyq;
  const double* bmnc = gmnc + a.mnmax_nyq; const double* dbmnc = bmnc + a.mnmax_nyq;
  const double* bsupv = dbmnc + a.mnmax_nyq; const double* bsubs = bsupv + a.mnmax_nyq;
  const double* bsubu = bsubs + a.mnmax_nyq; const double* bsubv = bsubu + a.mnmax_nyq;
  double sqg = 0, modB = 0, B_s = 0, B_t = 0, B_p = 0, Bsup_phi = 0, Bsub_s = 0, Bsub_t = 0, Bsub_p = 0;
  for (int k = 0; k < a.mnmax_nyq; ++k) {
    const double m = a.xm_nyq[k], n = a.xn_nyq[k];
    double sa, ca;
    sincos(m * tv - n * phi, &sa, &ca);
    sqg += gmnc[k] * ca; modB += bmnc[k] * ca; B_s += dbmnc[k] * ca;
    B_t -= bmnc[k] * m * sa; B_p += bmnc[k] * n * sa;
    Bsup_phi += bsupv[k] * ca; Bsub_s += bsubs[k] * sa; Bsub_t += bsubu[k] * ca; Bsub_p += bsubv[k] * ca;
  }
  // ---- metric algebra                                                       utils.py:474-720
  const double etf = -phiedge / (2 * M_PI);
  double sp, cp;
  sincos(phi, &sp, &cp);
  const double X_t = R_t * cp, X_p = R_p * cp - R * sp, X_s = R_s * cp;
  const double Y_t = R_t * sp, Y_p = R_p * sp + R * cp, Y_s = R_s * sp;
  const double isg = 1.0 / sqg;
  const double gsx = (Y_t * Z_p - Z_t * Y_p) * isg, gsy = (Z_t * X_p - X_t * Z_p) * isg, gsz = (X_t * Y_p - Y_t * X_p) * isg;
  const double gtx = (Y_p * Z_s - Z_p * Y_s) * isg, gty = (Z_p * X_s - X_p * Z_s) * isg, gtz = (X_p * Y_s - Y_p * X_s) * isg;
  const double gpx = (Y_s * Z_t - Z_s * Y_t) * isg, gpy = (Z_s * X_t - X_s * Z_t) * isg, gpz = (X_s * Y_t - Y_s * X_t) * isg;
  const double ls = l_s - phi * diota;
  const double c1 = 1 + l_t, c2 = -iota + l_p;
  const double gax = ls * gsx + c1 * gtx + c2 * gpx, gay = ls * gsy + c1 * gty + c2 * gpy, gaz = ls * gsz + c1 * gtz + c2 * gpz;
  const double psx = gsx * etf, psy = gsy * etf, psz = gsz * etf;
  const double BxgB_alpha = (Bsub_s * B_t * (l_p - iota) + Bsub_t * B_p * ls + Bsub_p * B_s * c1
                             - Bsub_p * B_t * ls - Bsub_t * B_s * (l_p - iota) - Bsub_s * B_p * c1) * isg;
  const double BxgB_psi = (Bsub_t * B_p - Bsub_p * B_t) * isg * etf;
  const double Bref = 2 * fabs(etf) / (L * L);
  const double sgn = etf > 0 ? 1.0 : (etf < 0 ? -1.0 : 0.0);
  const double sq = sqrt(s);
  const double shat = (-2 * s / iota) * diota;                                 // utils.py:316
  const double B3 = modB * modB * modB;
  const double bmag = modB / Bref;
  const double gradpar = L * (iota * Bsup_phi) / modB;
  const double gds2 = (gax * gax + gay * gay + gaz * gaz) * L * L * s;
  const double gds21 = (gax * psx + gay * psy + gaz * psz) * shat / Bref;
  const double gds22 = (psx * psx + psy * psy + psz * psz) * shat * shat / (L * L * Bref * Bref * s);
  const double gbdrift = -1.0 * 2 * Bref * L * L * sq * BxgB_alpha / B3 * sgn;
  const double gbdrift0 = -1.0 * BxgB_psi * 2 * shat / (B3 * sq) * sgn;
  const double mu0 = 4 * M_PI * 1.0e-7;
  const double cvdrift = gbdrift - 2 * Bref * L * L * sq * mu0 * dp * sgn / (etf * modB * modB);
  const size_t plane = (size_t)a.n_lines * a.ld, o = (size_t)line * a.ld + j;
  a.geo[o] = bmag; a.geo[plane + o] = gradpar; a.geo[2 * plane + o] = cvdrift; a.geo[3 * plane + o] = gbdrift0;
  a.geo[4 * plane + o] = gds2; a.geo[5 * plane + o] = gds21; a.geo[6 * plane + o] = gds22; a.geo[7 * plane + o] = gbdrift;
}

// dPdrho of each line: -0.5 mean((cvdrift - gbdrift) bmag^2)   (ball_scan.py:262)
__global__ void __launch_bounds__(256) k_line_dPdrho(int n_lines, int N, long ld, const double* geo, double* dPdrho) {
  __shared__ double part[4];
  const int line = blockIdx.x;
  const size_t plane = (size_t)n_lines * ld, o = (size_t)line * ld;
  double s = 0.0;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const double B = geo[o + j];
    s += (geo[2 * plane + o + j] - geo[7 * plane + o + j]) * B * B;
  }
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dPdrho[line] = -0.5 * (part[0] + part[1] + part[2] + part[3]) / N;
}

hipError_t launch_geometry(const GeoArgs& a, hipStream_t st) {
  dim3 grid((a.N + 255) / 256, a.n_lines);
  hipLaunchKernelGGL(k_fieldline_geometry, grid, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.dPdrho) {
    hipLaunchKernelGGL(k_line_dPdrho, dim3(a.n_lines), dim3(256), 0, st, a.n_lines, a.N, a.ld, a.geo, a.dPdrho);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace ibs
