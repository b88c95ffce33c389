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
#include <cstdlib>

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
  // optional row structure of the mode lists (VMEC order: modes grouped by m, consecutive n inside a group):
  // rows_*[r] = {first mode index, number of modes}; the angle then advances by -dphi_n per mode and only
  // one sincos per row is needed.  nrows_* = 0 selects the generic one-sincos-per-mode kernel.
  int nrows_mn, nrows_nyq;
  const int* rows_mn;      // [nrows_mn][2]
  const int* rows_nyq;     // [nrows_nyq][2]
  double dn_mn, dn_nyq;    // common n-spacing inside the rows of each set (rows with another spacing are split by the host)
};

// metric algebra shared by both kernels: utils.py:474 (flux sign), :480-508 (dual relations), :515-538
// (grad psi, grad alpha), :603-618 / :646-650 (B x grad B . grad alpha / psi), :654-720 (GS2 normalisation)
#define GEO_TAIL \
  const double etf = -phiedge / (2 * M_PI); \
  double sp, cp; \
  sincos(phi, &sp, &cp); \
  const double X_t = R_t * cp, X_p = R_p * cp - R * sp, X_s = R_s * cp; \
  const double Y_t = R_t * sp, Y_p = R_p * sp + R * cp, Y_s = R_s * sp; \
  const double isg = 1.0 / sqg; \
  const double gsx = (Y_t * Z_p - Z_t * Y_p) * isg, gsy = (Z_t * X_p - X_t * Z_p) * isg, gsz = (X_t * Y_p - Y_t * X_p) * isg; \
  const double gtx = (Y_p * Z_s - Z_p * Y_s) * isg, gty = (Z_p * X_s - X_p * Z_s) * isg, gtz = (X_p * Y_s - Y_p * X_s) * isg; \
  const double gpx = (Y_s * Z_t - Z_s * Y_t) * isg, gpy = (Z_s * X_t - X_s * Z_t) * isg, gpz = (X_s * Y_t - Y_s * X_t) * isg; \
  const double ls = l_s - phi * diota; \
  const double c1 = 1 + l_t, c2 = -iota + l_p; \
  const double gax = ls * gsx + c1 * gtx + c2 * gpx, gay = ls * gsy + c1 * gty + c2 * gpy, gaz = ls * gsz + c1 * gtz + c2 * gpz; \
  const double psx = gsx * etf, psy = gsy * etf, psz = gsz * etf; \
  const double BxgB_alpha = (Bsub_s * B_t * (l_p - iota) + Bsub_t * B_p * ls + Bsub_p * B_s * c1 \
                             - Bsub_p * B_t * ls - Bsub_t * B_s * (l_p - iota) - Bsub_s * B_p * c1) * isg; \
  const double BxgB_psi = (Bsub_t * B_p - Bsub_p * B_t) * isg * etf; \
  const double Bref = 2 * fabs(etf) / (L * L); \
  const double sgn = etf > 0 ? 1.0 : (etf < 0 ? -1.0 : 0.0); \
  const double sq = sqrt(s); \
  const double shat = (-2 * s / iota) * diota;  \
  const double B3 = modB * modB * modB; \
  const double bmag = modB / Bref; \
  const double gradpar = L * (iota * Bsup_phi) / modB; \
  const double gds2 = (gax * gax + gay * gay + gaz * gaz) * L * L * s; \
  const double gds21 = (gax * psx + gay * psy + gaz * psz) * shat / Bref; \
  const double gds22 = (psx * psx + psy * psy + psz * psz) * shat * shat / (L * L * Bref * Bref * s); \
  const double gbdrift = -1.0 * 2 * Bref * L * L * sq * BxgB_alpha / B3 * sgn; \
  const double gbdrift0 = -1.0 * BxgB_psi * 2 * shat / (B3 * sq) * sgn; \
  const double mu0 = 4 * M_PI * 1.0e-7; \
  const double cvdrift = gbdrift - 2 * Bref * L * L * sq * mu0 * dp * sgn / (etf * modB * modB); \
  const size_t plane = (size_t)a.n_lines * a.ld, o = (size_t)line * a.ld + j; \
  a.geo[o] = bmag; a.geo[plane + o] = gradpar; a.geo[2 * plane + o] = cvdrift; a.geo[3 * plane + o] = gbdrift0; \
  a.geo[4 * plane + o] = gds2; a.geo[5 * plane + o] = gds21; a.geo[6 * plane + o] = gds22; a.geo[7 * plane + o] = gbdrift; \

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
  // secant iteration (superlinear): once a step is below 1e-9 the next one lands at rounding level, so
  // exactly one more update is taken and the loop ends (a test on the rounding-level step never fires)
  double p0 = tp, p1 = tp + 0.1;
  double q0 = resid(p0), q1 = resid(p1);
  bool last = false;
  for (int it = 0; it < 40; ++it) {
    const double den = q1 - q0;
    if (den == 0.0) break;
    const double step = q1 * (p1 - p0) / den;
    p0 = p1; q0 = q1;
    p1 = p1 - step;
    if (last) break;
    last = fabs(step) <= 1e-9 * fmax(1.0, fabs(p1));
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
  GEO_TAIL
}


// (cos, sin) of the first angle of a row, m*tv - n0*phi, from running (cos, sin)(m_cur*tv) and (n0_cur*phi):
// rows come sorted by m (VMEC order), so m advances by plane rotations; n0 takes one or two values.
struct RowStart {
  double tv, phi, ctv, stv;
  double m_cur, cm, sm;       // cos/sin(m_cur * tv)
  double n_cur, cn, sn;       // cos/sin(n_cur * phi)
  __device__ __forceinline__ void init(double tv_, double phi_) {
    tv = tv_; phi = phi_;
    sincos(tv, &stv, &ctv);
    m_cur = 0.0; cm = 1.0; sm = 0.0;
    n_cur = 0.0; cn = 1.0; sn = 0.0;
  }
  __device__ __forceinline__ void start(double m, double n0, double& ca, double& sa) {
    if (m != m_cur) {
      const double dm = m - m_cur;
      if (dm > 0.0 && dm <= 4.0 && dm == floor(dm)) {
        for (double q = 0.0; q < dm; q += 1.0) { const double c2 = cm * ctv - sm * stv, s2 = sm * ctv + cm * stv; cm = c2; sm = s2; }
      } else {
        sincos(m * tv, &sm, &cm);
      }
      m_cur = m;
    }
    if (n0 != n_cur) { sincos(n0 * phi, &sn, &cn); n_cur = n0; }
    ca = cm * cn + sm * sn;      // cos(m tv - n0 phi)
    sa = sm * cn - cm * sn;      // sin(m tv - n0 phi)
  }
};

// Same arithmetic with the mode lists walked row by row (all modes of one m): inside a row the angle
// m theta - n phi decreases by a constant dn*phi per mode, so (cos, sin) advance by one plane rotation
// (4 flops) instead of a sincos.  Rows hold <= 2 ntor + 1 <= ~30 modes: the rotation error stays ~1e-15.
// LPP lanes cooperate on one grid point: lane `sub` takes modes sub, sub+LPP, ... of every row and the partial
// sums are combined by an in-register butterfly.  LPP = 8 turns the long per-point dependency chain of small
// batches into 8x more, 8x shorter threads; LPP = 1 is the throughput form for large batches.
template <int LPP>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (LPP >= 2)
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if constexpr (LPP >= 4)
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if constexpr (LPP >= 8)
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x141, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

template <int LPP>
__global__ void __launch_bounds__(256) k_fieldline_geometry_rows(GeoArgs a) {
  // the surface's mode tables are staged once per block in LDS (all lanes read the same entry: broadcast,
  // conflict-free); from global memory each entry would be a dependent scalar-cache miss
  extern __shared__ __align__(16) unsigned char geo_smem[];
  double* sm = reinterpret_cast<double*>(geo_smem);
  const int line = blockIdx.y;
  const int js = a.line_surf[line];
  double* xm_s = sm; double* xn_s = xm_s + a.mnmax;
  double* tmn = xn_s + a.mnmax;                 // 6 x mnmax
  double* xmq_s = tmn + 6 * a.mnmax; double* xnq_s = xmq_s + a.mnmax_nyq;
  double* tnq = xnq_s + a.mnmax_nyq;            // 7 x mnmax_nyq
  {
    const double* g_mn = a.tab_mn + (size_t)js * 6 * a.mnmax;
    const double* g_nq = a.tab_nyq + (size_t)js * 7 * a.mnmax_nyq;
    for (int k = threadIdx.x; k < a.mnmax; k += blockDim.x) { xm_s[k] = a.xm[k]; xn_s[k] = a.xn[k]; }
    for (int k = threadIdx.x; k < 6 * a.mnmax; k += blockDim.x) tmn[k] = g_mn[k];
    for (int k = threadIdx.x; k < a.mnmax_nyq; k += blockDim.x) { xmq_s[k] = a.xm_nyq[k]; xnq_s[k] = a.xn_nyq[k]; }
    for (int k = threadIdx.x; k < 7 * a.mnmax_nyq; k += blockDim.x) tnq[k] = g_nq[k];
  }
  __syncthreads();
  const int sub = threadIdx.x % LPP;
  const int j_raw = (blockIdx.x * blockDim.x + threadIdx.x) / LPP;
  const bool live = j_raw < a.N;           // lanes past the end keep computing (on the last point) so that the
  const int j = live ? j_raw : a.N - 1;    // butterflies stay full; they just do not store
  const double* sc = a.scal + 6 * js;
  const double s = sc[0], iota = sc[1], diota = sc[2], dp = sc[3], phiedge = sc[4], L = sc[5];
  const double alpha = a.line_alpha[line];
  const double tp = a.theta[j];
  const double phi = (tp - alpha) / iota;
  const double* rmnc = tmn;
  const double* zmns = rmnc + a.mnmax; const double* lmns = zmns + a.mnmax;
  const double* drmnc = lmns + a.mnmax; const double* dzmns = drmnc + a.mnmax; const double* dlmns = dzmns + a.mnmax;
  // the per-mode step dn (= nfp for VMEC tables) is the same in every row: its rotation is set up once
  const double dn_mn = a.dn_mn, dn_nyq = a.dn_nyq;
  double sd, cd, ss0, cs0;
  sincos(LPP * dn_mn * phi, &sd, &cd);          // per-lane step: LPP modes
  sincos(sub * dn_mn * phi, &ss0, &cs0);         // this lane's offset inside a row
  auto resid = [&](double tv) {
    double acc = 0.0;
    RowStart rs;
    rs.init(tv, phi);
    for (int r = 0; r < a.nrows_mn; ++r) {
      const int k0 = a.rows_mn[2 * r], cnt = a.rows_mn[2 * r + 1];
      double sa, ca;
      rs.start(xm_s[k0], xn_s[k0], ca, sa);
      { const double c2 = ca * cs0 + sa * ss0, s2 = sa * cs0 - ca * ss0; ca = c2; sa = s2; }
      for (int k = k0 + sub; k < k0 + cnt; k += LPP) {
        acc += lmns[k] * sa;
        const double c2 = ca * cd + sa * sd, s2 = sa * cd - ca * sd;   // angle -= dn*phi
        ca = c2; sa = s2;
      }
    }
    return tp - (tv + group_sum<LPP>(acc));
  };
  // secant iteration (superlinear): once a step is below 1e-9 the next one lands at rounding level, so
  // exactly one more update is taken and the loop ends (a test on the rounding-level step never fires)
  // second secant point: one fixed-point step theta_p + resid(theta_p) (already O(lambda^2) close) instead
  // of the reference's theta_p + 0.1; the root is the same, two evaluations fewer on average
  double p0 = tp;
  double q0 = resid(p0);
  double p1 = tp + q0;
  double q1 = resid(p1);
  bool last = false;
  for (int it = 0; it < 40; ++it) {
    const double den = q1 - q0;
    if (den == 0.0) break;
    const double step = q1 * (p1 - p0) / den;
    p0 = p1; q0 = q1;
    p1 = p1 - step;
    if (last) break;
    last = fabs(step) <= 1e-9 * fmax(1.0, fabs(p1));
    q1 = resid(p1);
  }
  const double tv = p1;
  double R = 0, R_s = 0, R_t = 0, R_p = 0, Z_s = 0, Z_t = 0, Z_p = 0, l_s = 0, l_t = 0, l_p = 0;
  RowStart rs;
  rs.init(tv, phi);
  for (int r = 0; r < a.nrows_mn; ++r) {
    const int k0 = a.rows_mn[2 * r], cnt = a.rows_mn[2 * r + 1];
    double sa, ca;
    rs.start(xm_s[k0], xn_s[k0], ca, sa);
    { const double c2 = ca * cs0 + sa * ss0, s2 = sa * cs0 - ca * ss0; ca = c2; sa = s2; }
    for (int k = k0 + sub; k < k0 + cnt; k += LPP) {
      const double m = xm_s[k], n = xn_s[k];
      R += rmnc[k] * ca; R_s += drmnc[k] * ca; R_t -= rmnc[k] * m * sa; R_p += rmnc[k] * n * sa;
      Z_s += dzmns[k] * sa; Z_t += zmns[k] * m * ca; Z_p -= zmns[k] * n * ca;
      l_s += dlmns[k] * sa; l_t += lmns[k] * m * ca; l_p -= lmns[k] * n * ca;
      const double c2 = ca * cd + sa * sd, s2 = sa * cd - ca * sd;
      ca = c2; sa = s2;
    }
  }
  const double* gmnc = tnq;
  const double* bmnc = gmnc + a.mnmax_nyq; const double* dbmnc = bmnc + a.mnmax_nyq;
  const double* bsupv = dbmnc + a.mnmax_nyq; const double* bsubs = bsupv + a.mnmax_nyq;
  const double* bsubu = bsubs + a.mnmax_nyq; const double* bsubv = bsubu + a.mnmax_nyq;
  double sqg = 0, modB = 0, B_s = 0, B_t = 0, B_p = 0, Bsup_phi = 0, Bsub_s = 0, Bsub_t = 0, Bsub_p = 0;
  if (dn_nyq != dn_mn) { sincos(LPP * dn_nyq * phi, &sd, &cd); sincos(sub * dn_nyq * phi, &ss0, &cs0); }
  R = group_sum<LPP>(R); R_s = group_sum<LPP>(R_s); R_t = group_sum<LPP>(R_t); R_p = group_sum<LPP>(R_p);
  Z_s = group_sum<LPP>(Z_s); Z_t = group_sum<LPP>(Z_t); Z_p = group_sum<LPP>(Z_p);
  l_s = group_sum<LPP>(l_s); l_t = group_sum<LPP>(l_t); l_p = group_sum<LPP>(l_p);
  rs.init(tv, phi);
  for (int r = 0; r < a.nrows_nyq; ++r) {
    const int k0 = a.rows_nyq[2 * r], cnt = a.rows_nyq[2 * r + 1];
    double sa, ca;
    rs.start(xmq_s[k0], xnq_s[k0], ca, sa);
    { const double c2 = ca * cs0 + sa * ss0, s2 = sa * cs0 - ca * ss0; ca = c2; sa = s2; }
    for (int k = k0 + sub; k < k0 + cnt; k += LPP) {
      const double m = xmq_s[k], n = xnq_s[k];
      sqg += gmnc[k] * ca; modB += bmnc[k] * ca; B_s += dbmnc[k] * ca;
      B_t -= bmnc[k] * m * sa; B_p += bmnc[k] * n * sa;
      Bsup_phi += bsupv[k] * ca; Bsub_s += bsubs[k] * sa; Bsub_t += bsubu[k] * ca; Bsub_p += bsubv[k] * ca;
      const double c2 = ca * cd + sa * sd, s2 = sa * cd - ca * sd;
      ca = c2; sa = s2;
    }
  }
  sqg = group_sum<LPP>(sqg); modB = group_sum<LPP>(modB); B_s = group_sum<LPP>(B_s); B_t = group_sum<LPP>(B_t);
  B_p = group_sum<LPP>(B_p); Bsup_phi = group_sum<LPP>(Bsup_phi); Bsub_s = group_sum<LPP>(Bsub_s);
  Bsub_t = group_sum<LPP>(Bsub_t); Bsub_p = group_sum<LPP>(Bsub_p);
  if (!live || sub != 0) return;
  GEO_TAIL
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
  // lanes per point: measured (tools/bench_geo.py) 1 is best or within 15 % of best from 128 x 513 points up:
  // the kernel is VALU bound, splitting a point over lanes only adds row-start work.  IBS_GEO_LPP overrides.
  const size_t lds = (size_t)(8 * a.mnmax + 9 * a.mnmax_nyq) * sizeof(double);
  if (a.nrows_mn > 0 && a.nrows_nyq > 0 && lds <= 150 * 1024) {
    int lpp = 1;
    if (const char* e = getenv("IBS_GEO_LPP")) lpp = atoi(e);
    auto go = [&](auto kern, int l) {
      hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e1 != hipSuccess) return e1;
      hipLaunchKernelGGL(kern, dim3((a.N * l + 255) / 256, a.n_lines), dim3(256), lds, st, a);
      return hipSuccess;
    };
    hipError_t e2 = lpp == 8 ? go(k_fieldline_geometry_rows<8>, 8) : lpp == 4 ? go(k_fieldline_geometry_rows<4>, 4)
                  : lpp == 2 ? go(k_fieldline_geometry_rows<2>, 2) : go(k_fieldline_geometry_rows<1>, 1);
    if (e2 != hipSuccess) return e2;
  } else {
    hipLaunchKernelGGL(k_fieldline_geometry, grid, dim3(256), 0, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.dPdrho) {
    hipLaunchKernelGGL(k_line_dPdrho, dim3(a.n_lines), dim3(256), 0, st, a.n_lines, a.N, a.ld, a.geo, a.dPdrho);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace ibs
