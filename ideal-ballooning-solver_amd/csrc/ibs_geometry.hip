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
  const int js = min(max(a.line_surf[line], 0), a.n_surf - 1);   // (device-resident indices are not range-checked by the C ABI)
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
  double n_keep, cn_keep, sn_keep;   // the last non-zero first-n seen: phi is fixed per point, so this survives init(tv)
  __device__ __forceinline__ void init_phi(double phi_) { phi = phi_; n_keep = 0.0; cn_keep = 1.0; sn_keep = 0.0; }
  __device__ __forceinline__ void init(double tv_) {
    tv = tv_;
    sincos(tv, &stv, &ctv);
    m_cur = 0.0; cm = 1.0; sm = 0.0;
    n_cur = 0.0; cn = 1.0; sn = 0.0;
  }
  __device__ __forceinline__ void init(double tv_, double phi_) { init_phi(phi_); init(tv_); }
  __device__ __forceinline__ void start(double m, double n0, double& ca, double& sa) {
    if (m != m_cur) {
      if (m - m_cur == 1.0) {          // VMEC order: the next row is the next m -> one plane rotation
        const double c2 = cm * ctv - sm * stv, s2 = sm * ctv + cm * stv; cm = c2; sm = s2;
      } else {
        sincos(m * tv, &sm, &cm);
      }
      m_cur = m;
    }
    if (n0 != n_cur) {
      if (n0 == 0.0) { cn = 1.0; sn = 0.0; }
      else {
        if (n0 != n_keep) { sincos(n0 * phi, &sn_keep, &cn_keep); n_keep = n0; }
        cn = cn_keep; sn = sn_keep;
      }
      n_cur = n0;
    }
    ca = cm * cn + sm * sn;      // cos(m tv - n0 phi)
    sa = sm * cn - cm * sn;      // sin(m tv - n0 phi)
  }
};

// Same arithmetic with the mode lists walked row by row (all modes of one m): inside a row the angle
// m theta - n phi decreases by a constant D = dn*phi per mode, so cos and sin follow the three-term recurrence
//   t[k+1] = 2 cos(D) t[k] - t[k-1]        (one fma each; rows hold <= ~60 modes: error growth ~k^2 eps)
// instead of a sincos per mode.  One thread per grid point, one field line per block row.
//
// The surface's tables are staged once per block in LDS (every lane reads the same entry: broadcast,
// conflict-free) with each row padded by zero coefficients to a multiple of 4 modes, so that the inner loops
// are unrolled with all LDS reads of a group issued before its arithmetic (an un-unrolled loop exposes the
// full LDS latency per mode: measured 2x slower).  Layout, mode numbers already multiplied in:
//   lm   [P]      lmns (root solve)
//   amn  [P][10]  rmnc d_rmnc m*rmnc n*rmnc | d_zmns m*zmns n*zmns | d_lmns m*lmns n*lmns
//   anq  [Q][10]  gmnc bmnc d_bmnc m*bmnc n*bmnc bsupv bsubs bsubu bsubv (pad)
// so a mode costs five 16-byte LDS reads and one fma per accumulated quantity.
constexpr int kGeoBlock = 512;      // 8 waves share one staged table set; 2 blocks per CU -> 4 waves per SIMD
constexpr int kGeoMaxRows = 128;

__device__ __forceinline__ int geo_pad4(int n) { return (n + 3) & ~3; }
__host__ __device__ inline int geo_cap(int nmodes, int nrows) { return (nmodes + 3 * nrows + 4 + 3) & ~3; }

// padded offset table of one mode set (off[r] .. off[r+1]) and the (m, first n) of each row
__device__ __forceinline__ void geo_row_tables(int t, int nrows, const int* rows, const double* xm, const double* xn,
                                               int* off, double* rm, double* rn) {
  if (t >= 0 && t <= nrows) {
    int o = 0;
    for (int q = 0; q < t; ++q) o += geo_pad4(rows[2 * q + 1]);
    off[t] = o;
    if (t < nrows) { const int k0 = rows[2 * t]; rm[t] = xm[k0]; rn[t] = xn[k0]; }
  }
}
// padded position -> source mode (or -1 for padding)
__device__ __forceinline__ int geo_src_mode(int idx, int nrows, const int* off, const int* rows) {
  int r = 0;
  while (r < nrows && off[r + 1] <= idx) ++r;
  if (r >= nrows) return -1;
  const int i = idx - off[r];
  return i < rows[2 * r + 1] ? rows[2 * r] + i : -1;
}

// sum over the LPP adjacent lanes that share one grid point (in-register butterfly, quad_perm DPP)
template <int LPP>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (LPP >= 2)
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if constexpr (LPP >= 4)
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if constexpr (LPP >= 8)      // every lane of a quad now holds the quad's sum: the mirror image within 8 lanes is the other quad
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x141, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}
template <int LPP>
__device__ __forceinline__ int row_int(int v) {
  if constexpr (LPP == 1) return __builtin_amdgcn_readfirstlane(v); else return v;
}

// LPP = 1: one lane per grid point (throughput form).  LPP = 2, 4, 8: the rows (m values) of both mode sets are
// dealt round-robin to LPP adjacent lanes and the partial sums combined by a butterfly: the per-point dependency
// chain gets LPP times shorter, which is what bounds small batches (too few waves to hide it).
template <int LPP>
__device__ __forceinline__ void geo_rows_body(const GeoArgs& a) {
  extern __shared__ __align__(16) unsigned char geo_smem[];
  const int line = blockIdx.y;
  const int js = min(max(a.line_surf[line], 0), a.n_surf - 1);   // (device-resident indices are not range-checked by the C ABI)
  const int nr1 = a.nrows_mn, nr2 = a.nrows_nyq;
  const int P = geo_cap(a.mnmax, nr1), Q = geo_cap(a.mnmax_nyq, nr2);
  double* lm_s = reinterpret_cast<double*>(geo_smem);
  double* amn = lm_s + P;
  double* anq = amn + 10 * P;
  double* rm1 = anq + 10 * Q; double* rn1 = rm1 + nr1;
  double* rm2 = rn1 + nr1; double* rn2 = rm2 + nr2;
  int* off1 = reinterpret_cast<int*>(rn2 + nr2); int* off2 = off1 + nr1 + 1;
  {
    const int t = threadIdx.x;
    geo_row_tables(t, nr1, a.rows_mn, a.xm, a.xn, off1, rm1, rn1);
    geo_row_tables(t - 256, nr2, a.rows_nyq, a.xm_nyq, a.xn_nyq, off2, rm2, rn2);     // threads 256.. (t - 256 >= 0)
    __syncthreads();
    const double* g_mn = a.tab_mn + (size_t)js * 6 * a.mnmax;
    const double* g_nq = a.tab_nyq + (size_t)js * 7 * a.mnmax_nyq;
    const int n1 = a.mnmax, n2 = a.mnmax_nyq;
    for (int idx = t; idx < P; idx += kGeoBlock) {
      const int k = geo_src_mode(idx, nr1, off1, a.rows_mn);
      double* q = amn + 10 * idx;
      if (k >= 0) {
        const double m = a.xm[k], n = a.xn[k];
        const double rm = g_mn[k], zm = g_mn[n1 + k], lm = g_mn[2 * n1 + k];
        lm_s[idx] = lm;
        q[0] = rm; q[1] = g_mn[3 * n1 + k]; q[2] = m * rm; q[3] = n * rm;
        q[4] = g_mn[4 * n1 + k]; q[5] = m * zm; q[6] = n * zm;
        q[7] = g_mn[5 * n1 + k]; q[8] = m * lm; q[9] = n * lm;
      } else {
        lm_s[idx] = 0.0;
        for (int c = 0; c < 10; ++c) q[c] = 0.0;
      }
    }
    for (int idx = t; idx < Q; idx += kGeoBlock) {
      const int k = geo_src_mode(idx, nr2, off2, a.rows_nyq);
      double* q = anq + 10 * idx;
      if (k >= 0) {
        const double m = a.xm_nyq[k], n = a.xn_nyq[k];
        const double bm = g_nq[n2 + k];
        q[0] = g_nq[k]; q[1] = bm; q[2] = g_nq[2 * n2 + k]; q[3] = m * bm; q[4] = n * bm;
        q[5] = g_nq[3 * n2 + k]; q[6] = g_nq[4 * n2 + k]; q[7] = g_nq[5 * n2 + k]; q[8] = g_nq[6 * n2 + k]; q[9] = 0.0;
      } else {
        for (int c = 0; c < 10; ++c) q[c] = 0.0;
      }
    }
  }
  __syncthreads();
  const int sub = threadIdx.x % LPP;
  const int j_raw = (blockIdx.x * kGeoBlock + threadIdx.x) / LPP;
  if ((int)(blockIdx.x * kGeoBlock + (threadIdx.x & ~63u)) / LPP >= a.N) return;   // whole wave past the end of the line (no barrier follows)
  const bool live = j_raw < a.N;
  const int j = live ? j_raw : a.N - 1;
  const double* sc = a.scal + 6 * js;
  const double s = sc[0], iota = sc[1], diota = sc[2], dp = sc[3], phiedge = sc[4], L = sc[5];
  const double alpha = a.line_alpha[line];
  const double tp = a.theta[j];
  const double phi = (tp - alpha) / iota;
  double sD, cD;
  sincos(a.dn_mn * phi, &sD, &cD);
  double two_cD = 2.0 * cD;
  RowStart rs;
  rs.init_phi(phi);
  auto resid = [&](double tv) {
    double acc0 = 0.0, acc1 = 0.0;
    rs.init(tv);
    int o_nx = off1[sub < nr1 ? sub : 0], e_nx = off1[sub < nr1 ? sub + 1 : 0];
    double m_nx = rm1[sub < nr1 ? sub : 0], n_nx = rn1[sub < nr1 ? sub : 0];
    for (int r = sub; r < nr1; r += LPP) {
      // this row's table entries were fetched one row ahead (the LDS latency hides behind the previous row)
      const int o = row_int<LPP>(o_nx);
      const int ng = (row_int<LPP>(e_nx) - o) >> 2;
      const double m_r = m_nx, n_r = n_nx;
      { const int rn = r + LPP < nr1 ? r + LPP : r; o_nx = off1[rn]; e_nx = off1[rn + 1]; m_nx = rm1[rn]; n_nx = rn1[rn]; }
      double s0, c0;
      rs.start(m_r, n_r, c0, s0);
      double sm1 = s0 * cD + c0 * sD;              // sin of the (virtual) previous mode: angle + D
      const double2* Lp = reinterpret_cast<const double2*>(lm_s + o);
      for (int g = 0; g < ng; ++g) {
        const double2 u = Lp[2 * g], v = Lp[2 * g + 1];
        const double s1 = fma(two_cD, s0, -sm1);
        const double s2 = fma(two_cD, s1, -s0);
        const double s3 = fma(two_cD, s2, -s1);
        acc0 = fma(u.x, s0, acc0); acc1 = fma(u.y, s1, acc1);
        acc0 = fma(v.x, s2, acc0); acc1 = fma(v.y, s3, acc1);
        sm1 = s3; s0 = fma(two_cD, s3, -s2);
      }
    }
    return tp - (tv + group_sum<LPP>(acc0 + acc1));
  };
  // secant iteration (superlinear, e_{n+1} ~ C e_n e_{n-1}): it stops, like the reference's scipy secant
  // (tol 1.48e-8, utils.py:391-416), on the step size; a step below 1e-9 means the point it leaves was 1e-9 from
  // the root, so the point it lands on is at ~1e-13 or better and is not evaluated again (one residual
  // evaluation = 242 modes saved per point).
  // Second point: one fixed-point step theta_p + resid(theta_p) instead of the reference's theta_p + 0.1
  // (same root, about two evaluations fewer).
  double p0 = tp;
  double q0 = resid(p0);
  double p1 = tp + q0;
  double q1 = resid(p1);
  for (int it = 0; it < 40; ++it) {
    const double den = q1 - q0;
    if (den == 0.0) break;
    const double step = q1 * (p1 - p0) / den;
    p0 = p1; q0 = q1;
    p1 = p1 - step;
    if (fabs(step) <= 1e-9 * fmax(1.0, fabs(p1))) break;
    q1 = resid(p1);
  }
  const double tv = p1;
  double R = 0, R_s = 0, R_t = 0, R_p = 0, Z_s = 0, Z_t = 0, Z_p = 0, l_s = 0, l_t = 0, l_p = 0;
  rs.init(tv);
  int o_nx = off1[sub < nr1 ? sub : 0], e_nx = off1[sub < nr1 ? sub + 1 : 0];
  double m_nx = rm1[sub < nr1 ? sub : 0], n_nx = rn1[sub < nr1 ? sub : 0];
  for (int r = sub; r < nr1; r += LPP) {
    const int o = row_int<LPP>(o_nx);
    const int ng = (row_int<LPP>(e_nx) - o) >> 1;
    const double m_r = m_nx, n_r = n_nx;
    { const int rn = r + LPP < nr1 ? r + LPP : r; o_nx = off1[rn]; e_nx = off1[rn + 1]; m_nx = rm1[rn]; n_nx = rn1[rn]; }
    double sa, ca;
    rs.start(m_r, n_r, ca, sa);
    double sm1 = sa * cD + ca * sD, cm1 = ca * cD - sa * sD;
    const double2* q = reinterpret_cast<const double2*>(amn + 10 * o);
    for (int g = 0; g < ng; ++g, q += 10) {
      const double2 q0_ = q[0], q1_ = q[1], q2_ = q[2], q3_ = q[3], q4_ = q[4];
      const double2 q5_ = q[5], q6_ = q[6], q7_ = q[7], q8_ = q[8], q9_ = q[9];
      const double sb = fma(two_cD, sa, -sm1), cb = fma(two_cD, ca, -cm1);
      R = fma(q0_.x, ca, R); R_s = fma(q0_.y, ca, R_s); R_t = fma(-q1_.x, sa, R_t); R_p = fma(q1_.y, sa, R_p);
      Z_s = fma(q2_.x, sa, Z_s); Z_t = fma(q2_.y, ca, Z_t); Z_p = fma(-q3_.x, ca, Z_p);
      l_s = fma(q3_.y, sa, l_s); l_t = fma(q4_.x, ca, l_t); l_p = fma(-q4_.y, ca, l_p);
      R = fma(q5_.x, cb, R); R_s = fma(q5_.y, cb, R_s); R_t = fma(-q6_.x, sb, R_t); R_p = fma(q6_.y, sb, R_p);
      Z_s = fma(q7_.x, sb, Z_s); Z_t = fma(q7_.y, cb, Z_t); Z_p = fma(-q8_.x, cb, Z_p);
      l_s = fma(q8_.y, sb, l_s); l_t = fma(q9_.x, cb, l_t); l_p = fma(-q9_.y, cb, l_p);
      sm1 = sb; cm1 = cb;
      sa = fma(two_cD, sb, -sa); ca = fma(two_cD, cb, -ca);
    }
  }
  if (a.dn_nyq != a.dn_mn) {
    sincos(a.dn_nyq * phi, &sD, &cD);
    two_cD = 2.0 * cD;
  }
  if constexpr (LPP > 1) {
    R = group_sum<LPP>(R); R_s = group_sum<LPP>(R_s); R_t = group_sum<LPP>(R_t); R_p = group_sum<LPP>(R_p);
    Z_s = group_sum<LPP>(Z_s); Z_t = group_sum<LPP>(Z_t); Z_p = group_sum<LPP>(Z_p);
    l_s = group_sum<LPP>(l_s); l_t = group_sum<LPP>(l_t); l_p = group_sum<LPP>(l_p);
  }
  double sqg = 0, modB = 0, B_s = 0, B_t = 0, B_p = 0, Bsup_phi = 0, Bsub_s = 0, Bsub_t = 0, Bsub_p = 0;
  rs.init(tv);
  o_nx = off2[sub < nr2 ? sub : 0]; e_nx = off2[sub < nr2 ? sub + 1 : 0];
  m_nx = rm2[sub < nr2 ? sub : 0]; n_nx = rn2[sub < nr2 ? sub : 0];
  for (int r = sub; r < nr2; r += LPP) {
    const int o = row_int<LPP>(o_nx);
    const int ng = (row_int<LPP>(e_nx) - o) >> 1;
    const double m_r = m_nx, n_r = n_nx;
    { const int rn = r + LPP < nr2 ? r + LPP : r; o_nx = off2[rn]; e_nx = off2[rn + 1]; m_nx = rm2[rn]; n_nx = rn2[rn]; }
    double sa, ca;
    rs.start(m_r, n_r, ca, sa);
    double sm1 = sa * cD + ca * sD, cm1 = ca * cD - sa * sD;
    const double2* q = reinterpret_cast<const double2*>(anq + 10 * o);
    for (int g = 0; g < ng; ++g, q += 10) {
      const double2 q0_ = q[0], q1_ = q[1], q2_ = q[2], q3_ = q[3], q4_ = q[4];
      const double2 q5_ = q[5], q6_ = q[6], q7_ = q[7], q8_ = q[8], q9_ = q[9];
      const double sb = fma(two_cD, sa, -sm1), cb = fma(two_cD, ca, -cm1);
      sqg = fma(q0_.x, ca, sqg); modB = fma(q0_.y, ca, modB); B_s = fma(q1_.x, ca, B_s);
      B_t = fma(-q1_.y, sa, B_t); B_p = fma(q2_.x, sa, B_p);
      Bsup_phi = fma(q2_.y, ca, Bsup_phi); Bsub_s = fma(q3_.x, sa, Bsub_s); Bsub_t = fma(q3_.y, ca, Bsub_t);
      Bsub_p = fma(q4_.x, ca, Bsub_p);
      sqg = fma(q5_.x, cb, sqg); modB = fma(q5_.y, cb, modB); B_s = fma(q6_.x, cb, B_s);
      B_t = fma(-q6_.y, sb, B_t); B_p = fma(q7_.x, sb, B_p);
      Bsup_phi = fma(q7_.y, cb, Bsup_phi); Bsub_s = fma(q8_.x, sb, Bsub_s); Bsub_t = fma(q8_.y, cb, Bsub_t);
      Bsub_p = fma(q9_.x, cb, Bsub_p);
      sm1 = sb; cm1 = cb;
      sa = fma(two_cD, sb, -sa); ca = fma(two_cD, cb, -ca);
    }
  }
  if constexpr (LPP > 1) {
    sqg = group_sum<LPP>(sqg); modB = group_sum<LPP>(modB); B_s = group_sum<LPP>(B_s); B_t = group_sum<LPP>(B_t);
    B_p = group_sum<LPP>(B_p); Bsup_phi = group_sum<LPP>(Bsup_phi); Bsub_s = group_sum<LPP>(Bsub_s);
    Bsub_t = group_sum<LPP>(Bsub_t); Bsub_p = group_sum<LPP>(Bsub_p);
  }
  if (!live || sub != 0) return;
  GEO_TAIL
}

// ---- two grid points per lane (round 2) -------------------------------------------------------------------------
// The one-point form is bound by the LDS pipe, not by the VALU: every lane of a wave reads the same table entry, and a
// ds_read_b128 occupies the LDS array for 4 cycles per wave whether or not its lanes broadcast, so a group of two modes
// costs 10 reads = 40 LDS cycles per wave, 160 per CU (four SIMDs share one array), against 24 FMAs = 96 VALU cycles
// per SIMD.  With two points per lane every table entry read feeds twice the arithmetic: 160 LDS cycles against 192
// VALU cycles per group -- the VALU binds, as it should.  256-thread blocks x 2 points keep the block's 512 points
// and its staged tables; the registers (two sets of accumulators) allow two waves per SIMD, which is what two
// blocks per CU provide.
constexpr int kGeoBlock2 = 256;

struct RowStart2 {
  double tv[2], phi[2], ctv[2], stv[2];
  double cm[2], sm[2], cn[2], sn[2], cn_keep[2], sn_keep[2];
  double m_cur, n_cur, n_keep;        // (the same for every point: the row sequence is)
  __device__ __forceinline__ void init_phi(double p0, double p1) {
    phi[0] = p0; phi[1] = p1; n_keep = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p) { cn_keep[p] = 1.0; sn_keep[p] = 0.0; }
  }
  __device__ __forceinline__ void init(const double (&tv_)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      tv[p] = tv_[p];
      sincos(tv[p], &stv[p], &ctv[p]);
      cm[p] = 1.0; sm[p] = 0.0; cn[p] = 1.0; sn[p] = 0.0;
    }
    m_cur = 0.0; n_cur = 0.0;
  }
  __device__ __forceinline__ void start(double m, double n0, double (&ca)[2], double (&sa)[2]) {
    if (m != m_cur) {
      if (m - m_cur == 1.0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const double c2 = cm[p] * ctv[p] - sm[p] * stv[p], s2 = sm[p] * ctv[p] + cm[p] * stv[p];
          cm[p] = c2; sm[p] = s2;
        }
      } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) sincos(m * tv[p], &sm[p], &cm[p]);
      }
      m_cur = m;
    }
    if (n0 != n_cur) {
      if (n0 == 0.0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) { cn[p] = 1.0; sn[p] = 0.0; }
      } else {
        if (n0 != n_keep) {
#pragma unroll
          for (int p = 0; p < 2; ++p) sincos(n0 * phi[p], &sn_keep[p], &cn_keep[p]);
          n_keep = n0;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) { cn[p] = cn_keep[p]; sn[p] = sn_keep[p]; }
      }
      n_cur = n0;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ca[p] = cm[p] * cn[p] + sm[p] * sn[p];
      sa[p] = sm[p] * cn[p] - cm[p] * sn[p];
    }
  }
};

// metric algebra and stores of one point (GEO_TAIL on named values)
struct GeoSums {
  double R, R_s, R_t, R_p, Z_s, Z_t, Z_p, l_s, l_t, l_p;
  double sqg, modB, B_s, B_t, B_p, Bsup_phi, Bsub_s, Bsub_t, Bsub_p;
};
__device__ __forceinline__ void geo_tail(const GeoArgs& a, int line, int j, double phi, double s, double iota, double diota,
                                         double dp, double phiedge, double L, const GeoSums& q) {
  const double R = q.R, R_s = q.R_s, R_t = q.R_t, R_p = q.R_p, Z_s = q.Z_s, Z_t = q.Z_t, Z_p = q.Z_p;
  const double l_s = q.l_s, l_t = q.l_t, l_p = q.l_p;
  const double sqg = q.sqg, modB = q.modB, B_s = q.B_s, B_t = q.B_t, B_p = q.B_p, Bsup_phi = q.Bsup_phi;
  const double Bsub_s = q.Bsub_s, Bsub_t = q.Bsub_t, Bsub_p = q.Bsub_p;
  GEO_TAIL
}

__global__ void __launch_bounds__(kGeoBlock2) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_fieldline_geometry_rows2(GeoArgs a) {
  extern __shared__ __align__(16) unsigned char geo_smem[];
  const int line = blockIdx.y;
  const int js = min(max(a.line_surf[line], 0), a.n_surf - 1);
  const int nr1 = a.nrows_mn, nr2 = a.nrows_nyq;
  const int P = geo_cap(a.mnmax, nr1), Q = geo_cap(a.mnmax_nyq, nr2);
  double* lm_s = reinterpret_cast<double*>(geo_smem);
  double* amn = lm_s + P;
  double* anq = amn + 10 * P;
  double* rm1 = anq + 10 * Q; double* rn1 = rm1 + nr1;
  double* rm2 = rn1 + nr1; double* rn2 = rm2 + nr2;
  int* off1 = reinterpret_cast<int*>(rn2 + nr2); int* off2 = off1 + nr1 + 1;
  {
    const int t = threadIdx.x;
    for (int q = t; q <= nr1; q += kGeoBlock2) geo_row_tables(q, nr1, a.rows_mn, a.xm, a.xn, off1, rm1, rn1);
    for (int q = t; q <= nr2; q += kGeoBlock2) geo_row_tables(q, nr2, a.rows_nyq, a.xm_nyq, a.xn_nyq, off2, rm2, rn2);
    __syncthreads();
    const double* g_mn = a.tab_mn + (size_t)js * 6 * a.mnmax;
    const double* g_nq = a.tab_nyq + (size_t)js * 7 * a.mnmax_nyq;
    const int n1 = a.mnmax, n2 = a.mnmax_nyq;
    for (int idx = t; idx < P; idx += kGeoBlock2) {
      const int k = geo_src_mode(idx, nr1, off1, a.rows_mn);
      double* q = amn + 10 * idx;
      if (k >= 0) {
        const double m = a.xm[k], n = a.xn[k];
        const double rm = g_mn[k], zm = g_mn[n1 + k], lm = g_mn[2 * n1 + k];
        lm_s[idx] = lm;
        q[0] = rm; q[1] = g_mn[3 * n1 + k]; q[2] = m * rm; q[3] = n * rm;
        q[4] = g_mn[4 * n1 + k]; q[5] = m * zm; q[6] = n * zm;
        q[7] = g_mn[5 * n1 + k]; q[8] = m * lm; q[9] = n * lm;
      } else {
        lm_s[idx] = 0.0;
        for (int c = 0; c < 10; ++c) q[c] = 0.0;
      }
    }
    for (int idx = t; idx < Q; idx += kGeoBlock2) {
      const int k = geo_src_mode(idx, nr2, off2, a.rows_nyq);
      double* q = anq + 10 * idx;
      if (k >= 0) {
        const double m = a.xm_nyq[k], n = a.xn_nyq[k];
        const double bm = g_nq[n2 + k];
        q[0] = g_nq[k]; q[1] = bm; q[2] = g_nq[2 * n2 + k]; q[3] = m * bm; q[4] = n * bm;
        q[5] = g_nq[3 * n2 + k]; q[6] = g_nq[4 * n2 + k]; q[7] = g_nq[5 * n2 + k]; q[8] = g_nq[6 * n2 + k]; q[9] = 0.0;
      } else {
        for (int c = 0; c < 10; ++c) q[c] = 0.0;
      }
    }
  }
  __syncthreads();
  // points of this lane: j0 = base + t, j1 = base + 256 + t (coalesced stores per point set)
  const int base = blockIdx.x * (2 * kGeoBlock2);
  if (base + (int)(threadIdx.x & ~63u) >= a.j_end) return;      // whole wave past the end of the range (no barrier follows)
  int j[2]; bool live[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int jr = base + p * kGeoBlock2 + (int)threadIdx.x;
    live[p] = jr < a.j_end;
    j[p] = live[p] ? jr : a.j_end - 1;
  }
  const double* sc = a.scal + 6 * js;
  const double s = sc[0], iota = sc[1], diota = sc[2], dp = sc[3], phiedge = sc[4], L = sc[5];
  const double alpha = a.line_alpha[line];
  double tp[2], phi[2], sD[2], cD[2], two_cD[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    tp[p] = a.theta[j[p]];
    phi[p] = (tp[p] - alpha) / iota;
    sincos(a.dn_mn * phi[p], &sD[p], &cD[p]);
    two_cD[p] = 2.0 * cD[p];
  }
  RowStart2 rs;
  rs.init_phi(phi[0], phi[1]);
  auto resid = [&](const double (&tv)[2], double (&out)[2]) {
    double acc0[2] = {0.0, 0.0}, acc1[2] = {0.0, 0.0};
    rs.init(tv);
    int o_nx = off1[0], e_nx = off1[nr1 > 0 ? 1 : 0];
    double m_nx = rm1[0], n_nx = rn1[0];
    for (int r = 0; r < nr1; ++r) {
      const int o = __builtin_amdgcn_readfirstlane(o_nx);
      const int ng = (__builtin_amdgcn_readfirstlane(e_nx) - o) >> 2;
      const double m_r = m_nx, n_r = n_nx;
      { const int rn = r + 1 < nr1 ? r + 1 : r; o_nx = off1[rn]; e_nx = off1[rn + 1]; m_nx = rm1[rn]; n_nx = rn1[rn]; }
      double s0[2], c0[2], sm1[2];
      rs.start(m_r, n_r, c0, s0);
#pragma unroll
      for (int p = 0; p < 2; ++p) sm1[p] = s0[p] * cD[p] + c0[p] * sD[p];
      const double2* Lp = reinterpret_cast<const double2*>(lm_s + o);
      for (int g = 0; g < ng; ++g) {
        const double2 u = Lp[2 * g], v = Lp[2 * g + 1];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const double s1 = fma(two_cD[p], s0[p], -sm1[p]);
          const double s2 = fma(two_cD[p], s1, -s0[p]);
          const double s3 = fma(two_cD[p], s2, -s1);
          acc0[p] = fma(u.x, s0[p], acc0[p]); acc1[p] = fma(u.y, s1, acc1[p]);
          acc0[p] = fma(v.x, s2, acc0[p]); acc1[p] = fma(v.y, s3, acc1[p]);
          sm1[p] = s3; s0[p] = fma(two_cD[p], s3, -s2);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) out[p] = tp[p] - (tv[p] + (acc0[p] + acc1[p]));
  };
  // the secant iteration of the one-point form, both points in step: a point that has converged keeps its value
  double p0[2] = {tp[0], tp[1]}, q0[2], p1[2], q1[2];
  resid(p0, q0);
#pragma unroll
  for (int p = 0; p < 2; ++p) p1[p] = tp[p] + q0[p];
  resid(p1, q1);
  bool fin[2] = {false, false};
  for (int it = 0; it < 40; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const double den = q1[p] - q0[p];
      if (!fin[p]) {
        if (den == 0.0) fin[p] = true;
        else {
          const double step = q1[p] * (p1[p] - p0[p]) / den;
          p0[p] = p1[p]; q0[p] = q1[p];
          p1[p] = p1[p] - step;
          if (fabs(step) <= 1e-9 * fmax(1.0, fabs(p1[p]))) fin[p] = true;
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(!(fin[0] && fin[1])) == 0ull) break;      // every point of the wave is done
    double qn[2];
    resid(p1, qn);
#pragma unroll
    for (int p = 0; p < 2; ++p) if (!fin[p]) q1[p] = qn[p];
  }
  double tv[2] = {p1[0], p1[1]};
  GeoSums S[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    S[p].R = S[p].R_s = S[p].R_t = S[p].R_p = S[p].Z_s = S[p].Z_t = S[p].Z_p = S[p].l_s = S[p].l_t = S[p].l_p = 0.0;
    S[p].sqg = S[p].modB = S[p].B_s = S[p].B_t = S[p].B_p = S[p].Bsup_phi = S[p].Bsub_s = S[p].Bsub_t = S[p].Bsub_p = 0.0;
  }
  rs.init(tv);
  {
    int o_nx = off1[0], e_nx = off1[nr1 > 0 ? 1 : 0];
    double m_nx = rm1[0], n_nx = rn1[0];
    for (int r = 0; r < nr1; ++r) {
      const int o = __builtin_amdgcn_readfirstlane(o_nx);
      const int ng = (__builtin_amdgcn_readfirstlane(e_nx) - o) >> 1;
      const double m_r = m_nx, n_r = n_nx;
      { const int rn = r + 1 < nr1 ? r + 1 : r; o_nx = off1[rn]; e_nx = off1[rn + 1]; m_nx = rm1[rn]; n_nx = rn1[rn]; }
      double sa[2], ca[2], sm1[2], cm1[2];
      rs.start(m_r, n_r, ca, sa);
#pragma unroll
      for (int p = 0; p < 2; ++p) { sm1[p] = sa[p] * cD[p] + ca[p] * sD[p]; cm1[p] = ca[p] * cD[p] - sa[p] * sD[p]; }
      const double2* q = reinterpret_cast<const double2*>(amn + 10 * o);
      for (int g = 0; g < ng; ++g, q += 10) {
        const double2 q0_ = q[0], q1_ = q[1], q2_ = q[2], q3_ = q[3], q4_ = q[4];
        const double2 q5_ = q[5], q6_ = q[6], q7_ = q[7], q8_ = q[8], q9_ = q[9];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          GeoSums& A = S[p];
          const double sa_ = sa[p], ca_ = ca[p];
          const double sb = fma(two_cD[p], sa_, -sm1[p]), cb = fma(two_cD[p], ca_, -cm1[p]);
          A.R = fma(q0_.x, ca_, A.R); A.R_s = fma(q0_.y, ca_, A.R_s); A.R_t = fma(-q1_.x, sa_, A.R_t); A.R_p = fma(q1_.y, sa_, A.R_p);
          A.Z_s = fma(q2_.x, sa_, A.Z_s); A.Z_t = fma(q2_.y, ca_, A.Z_t); A.Z_p = fma(-q3_.x, ca_, A.Z_p);
          A.l_s = fma(q3_.y, sa_, A.l_s); A.l_t = fma(q4_.x, ca_, A.l_t); A.l_p = fma(-q4_.y, ca_, A.l_p);
          A.R = fma(q5_.x, cb, A.R); A.R_s = fma(q5_.y, cb, A.R_s); A.R_t = fma(-q6_.x, sb, A.R_t); A.R_p = fma(q6_.y, sb, A.R_p);
          A.Z_s = fma(q7_.x, sb, A.Z_s); A.Z_t = fma(q7_.y, cb, A.Z_t); A.Z_p = fma(-q8_.x, cb, A.Z_p);
          A.l_s = fma(q8_.y, sb, A.l_s); A.l_t = fma(q9_.x, cb, A.l_t); A.l_p = fma(-q9_.y, cb, A.l_p);
          sm1[p] = sb; cm1[p] = cb;
          sa[p] = fma(two_cD[p], sb, -sa_); ca[p] = fma(two_cD[p], cb, -ca_);
        }
      }
    }
  }
  if (a.dn_nyq != a.dn_mn) {
#pragma unroll
    for (int p = 0; p < 2; ++p) { sincos(a.dn_nyq * phi[p], &sD[p], &cD[p]); two_cD[p] = 2.0 * cD[p]; }
  }
  rs.init(tv);
  {
    int o_nx = off2[0], e_nx = off2[nr2 > 0 ? 1 : 0];
    double m_nx = rm2[0], n_nx = rn2[0];
    for (int r = 0; r < nr2; ++r) {
      const int o = __builtin_amdgcn_readfirstlane(o_nx);
      const int ng = (__builtin_amdgcn_readfirstlane(e_nx) - o) >> 1;
      const double m_r = m_nx, n_r = n_nx;
      { const int rn = r + 1 < nr2 ? r + 1 : r; o_nx = off2[rn]; e_nx = off2[rn + 1]; m_nx = rm2[rn]; n_nx = rn2[rn]; }
      double sa[2], ca[2], sm1[2], cm1[2];
      rs.start(m_r, n_r, ca, sa);
#pragma unroll
      for (int p = 0; p < 2; ++p) { sm1[p] = sa[p] * cD[p] + ca[p] * sD[p]; cm1[p] = ca[p] * cD[p] - sa[p] * sD[p]; }
      const double2* q = reinterpret_cast<const double2*>(anq + 10 * o);
      for (int g = 0; g < ng; ++g, q += 10) {
        const double2 q0_ = q[0], q1_ = q[1], q2_ = q[2], q3_ = q[3], q4_ = q[4];
        const double2 q5_ = q[5], q6_ = q[6], q7_ = q[7], q8_ = q[8], q9_ = q[9];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          GeoSums& A = S[p];
          const double sa_ = sa[p], ca_ = ca[p];
          const double sb = fma(two_cD[p], sa_, -sm1[p]), cb = fma(two_cD[p], ca_, -cm1[p]);
          A.sqg = fma(q0_.x, ca_, A.sqg); A.modB = fma(q0_.y, ca_, A.modB); A.B_s = fma(q1_.x, ca_, A.B_s);
          A.B_t = fma(-q1_.y, sa_, A.B_t); A.B_p = fma(q2_.x, sa_, A.B_p);
          A.Bsup_phi = fma(q2_.y, ca_, A.Bsup_phi); A.Bsub_s = fma(q3_.x, sa_, A.Bsub_s); A.Bsub_t = fma(q3_.y, ca_, A.Bsub_t);
          A.Bsub_p = fma(q4_.x, ca_, A.Bsub_p);
          A.sqg = fma(q5_.x, cb, A.sqg); A.modB = fma(q5_.y, cb, A.modB); A.B_s = fma(q6_.x, cb, A.B_s);
          A.B_t = fma(-q6_.y, sb, A.B_t); A.B_p = fma(q7_.x, sb, A.B_p);
          A.Bsup_phi = fma(q7_.y, cb, A.Bsup_phi); A.Bsub_s = fma(q8_.x, sb, A.Bsub_s); A.Bsub_t = fma(q8_.y, cb, A.Bsub_t);
          A.Bsub_p = fma(q9_.x, cb, A.Bsub_p);
          sm1[p] = sb; cm1[p] = cb;
          sa[p] = fma(two_cD[p], sb, -sa_); ca[p] = fma(two_cD[p], cb, -ca_);
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
    if (live[p]) geo_tail(a, line, j[p], phi[p], s, iota, diota, dp, phiedge, L, S[p]);
}

// ---- one grid point per WAVE: the few points a line has beyond a multiple of the block's 512 ---------------------
// N = N_zeta + 1 = 2^k + 1 leaves ONE point per line for a third block, which then holds a CU's table slot (68 KB of
// LDS) for a full wave's duration with one lane working: a quarter of the resident-wave slots of the configs[2] shape.
// Here the 64 lanes share the modes of one point instead (one sincos per mode, no tables staged: the coefficients
// come straight from L2), and wave reductions close the sums.
__device__ __forceinline__ double geo_wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__global__ void __launch_bounds__(256) k_fieldline_geometry_tail(GeoArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = a.j_end - a.j_begin;
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= (long)a.n_lines * r) return;
  const int line = (int)(w / r);
  const int j = a.j_begin + (int)(w % r);
  const int js = min(max(a.line_surf[line], 0), a.n_surf - 1);
  const double* sc = a.scal + 6 * js;
  const double s = sc[0], iota = sc[1], diota = sc[2], dp = sc[3], phiedge = sc[4], L = sc[5];
  const double alpha = a.line_alpha[line];
  const double tp = a.theta[j];
  const double phi = (tp - alpha) / iota;
  const double* rmnc = a.tab_mn + (size_t)js * 6 * a.mnmax;
  const double* zmns = rmnc + a.mnmax; const double* lmns = zmns + a.mnmax;
  const double* drmnc = lmns + a.mnmax; const double* dzmns = drmnc + a.mnmax; const double* dlmns = dzmns + a.mnmax;
  auto resid = [&](double tv) {
    double acc = 0.0;
    for (int k = lane; k < a.mnmax; k += 64) acc += lmns[k] * sin(a.xm[k] * tv - a.xn[k] * phi);
    return tp - (tv + geo_wave_sum(acc));
  };
  // the secant iteration of the row kernels (start: one fixed-point step; stop on the step size)
  double p0 = tp, q0 = resid(p0);
  double p1 = tp + q0, q1 = resid(p1);
  for (int it = 0; it < 40; ++it) {
    const double den = q1 - q0;
    if (den == 0.0) break;
    const double step = q1 * (p1 - p0) / den;
    p0 = p1; q0 = q1;
    p1 = p1 - step;
    if (fabs(step) <= 1e-9 * fmax(1.0, fabs(p1))) break;
    q1 = resid(p1);
  }
  const double tv = p1;
  GeoSums S;
  S.R = S.R_s = S.R_t = S.R_p = S.Z_s = S.Z_t = S.Z_p = S.l_s = S.l_t = S.l_p = 0.0;
  S.sqg = S.modB = S.B_s = S.B_t = S.B_p = S.Bsup_phi = S.Bsub_s = S.Bsub_t = S.Bsub_p = 0.0;
  for (int k = lane; k < a.mnmax; k += 64) {
    const double m = a.xm[k], n = a.xn[k];
    double sa, ca;
    sincos(m * tv - n * phi, &sa, &ca);
    S.R += rmnc[k] * ca; S.R_s += drmnc[k] * ca; S.R_t -= rmnc[k] * m * sa; S.R_p += rmnc[k] * n * sa;
    S.Z_s += dzmns[k] * sa; S.Z_t += zmns[k] * m * ca; S.Z_p -= zmns[k] * n * ca;
    S.l_s += dlmns[k] * sa; S.l_t += lmns[k] * m * ca; S.l_p -= lmns[k] * n * ca;
  }
  const double* gmnc = a.tab_nyq + (size_t)js * 7 * a.mnmax_nyq;
  const double* bmnc = gmnc + a.mnmax_nyq; const double* dbmnc = bmnc + a.mnmax_nyq;
  const double* bsupv = dbmnc + a.mnmax_nyq; const double* bsubs = bsupv + a.mnmax_nyq;
  const double* bsubu = bsubs + a.mnmax_nyq; const double* bsubv = bsubu + a.mnmax_nyq;
  for (int k = lane; k < a.mnmax_nyq; k += 64) {
    const double m = a.xm_nyq[k], n = a.xn_nyq[k];
    double sa, ca;
    sincos(m * tv - n * phi, &sa, &ca);
    S.sqg += gmnc[k] * ca; S.modB += bmnc[k] * ca; S.B_s += dbmnc[k] * ca;
    S.B_t -= bmnc[k] * m * sa; S.B_p += bmnc[k] * n * sa;
    S.Bsup_phi += bsupv[k] * ca; S.Bsub_s += bsubs[k] * sa; S.Bsub_t += bsubu[k] * ca; S.Bsub_p += bsubv[k] * ca;
  }
  S.R = geo_wave_sum(S.R); S.R_s = geo_wave_sum(S.R_s); S.R_t = geo_wave_sum(S.R_t); S.R_p = geo_wave_sum(S.R_p);
  S.Z_s = geo_wave_sum(S.Z_s); S.Z_t = geo_wave_sum(S.Z_t); S.Z_p = geo_wave_sum(S.Z_p);
  S.l_s = geo_wave_sum(S.l_s); S.l_t = geo_wave_sum(S.l_t); S.l_p = geo_wave_sum(S.l_p);
  S.sqg = geo_wave_sum(S.sqg); S.modB = geo_wave_sum(S.modB); S.B_s = geo_wave_sum(S.B_s); S.B_t = geo_wave_sum(S.B_t);
  S.B_p = geo_wave_sum(S.B_p); S.Bsup_phi = geo_wave_sum(S.Bsup_phi); S.Bsub_s = geo_wave_sum(S.Bsub_s);
  S.Bsub_t = geo_wave_sum(S.Bsub_t); S.Bsub_p = geo_wave_sum(S.Bsub_p);
  if (lane == 0) geo_tail(a, line, j, phi, s, iota, diota, dp, phiedge, L, S);
}

// register budget: the throughput form is held to 128 VGPRs (4 waves per SIMD hide the LDS latency); the
// latency forms run with few waves anyway and take what they need
__global__ void __launch_bounds__(kGeoBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_fieldline_geometry_rows(GeoArgs a) { geo_rows_body<1>(a); }
template <int LPP>
__global__ void __launch_bounds__(kGeoBlock) k_fieldline_geometry_rows_split(GeoArgs a) { geo_rows_body<LPP>(a); }

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
  const size_t lds = (size_t)(11 * geo_cap(a.mnmax, a.nrows_mn) + 10 * geo_cap(a.mnmax_nyq, a.nrows_nyq) + 2 * (a.nrows_mn + a.nrows_nyq)) * sizeof(double)
                     + (size_t)(a.nrows_mn + a.nrows_nyq + 2) * sizeof(int);
  if (a.nrows_mn > 0 && a.nrows_nyq > 0 && a.nrows_mn <= kGeoMaxRows && a.nrows_nyq <= kGeoMaxRows && lds <= 150 * 1024) {
    // lanes per point: 1.  The split forms (IBS_GEO_LPP = 2, 4: rows dealt to adjacent lanes) were measured
    // slower at every shape tried (128 x 513: 120 / 144 / 135 us; 2048 x 1025: 1.74 / 2.17 / 2.71 ms): each
    // block stages its own 68 KB of tables and only two blocks fit a CU, so more, smaller blocks do not shorten
    // the critical path.  Kept as an experiment switch only.
    // ... except for batches that leave most of the chip idle (the refinement rounds of ibs_refine_f64: 3 lines per
    // point): there the launch is the latency of one thread's ~630 modes, and splitting a point over 4 (2) lanes
    // shortens it (5 surfaces, N = 969: 3.3 -> 2.2 ms for the whole refinement).  IBS_GEO_LPP=1|2|4 overrides.
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
      else n_cu = 256;
    }
    const long blocks1 = (long)((a.N + kGeoBlock - 1) / kGeoBlock) * a.n_lines;
    // (8 lanes per point, geo_lpp = 8, shortens a 15-line call from 52 to 46 us, but it is not dispatched by itself: the
    //  refinement's evaluation count reacts chaotically to the summation order -- the reference batch of bench.py went
    //  from 10 to 29 evaluations on one surface for a 2e-15 change of its maximum -- so nothing is gained on average)
    int lpp = blocks1 * 4 <= n_cu ? 4 : (blocks1 * 2 <= n_cu ? 2 : 1);
    if (a.lpp == 1 || a.lpp == 2 || a.lpp == 4 || a.lpp == 8) lpp = a.lpp;
    // Batches of at least one full block per CU: two grid points per lane (the LDS pipe no longer holds the synthesis
    // up), and the few points a line has beyond a multiple of 512 go to the one-point-per-wave kernel instead of a
    // block of their own.  tools/geo_bench.py, lines x 1,025 points, one point per lane -> this form: 2,048 lines
    // 1.45 -> 1.05 ms, 1,024: 0.75 -> 0.55, 256: 0.23 -> 0.17, 128: 0.158 -> 0.134; 8,760 x 969: 4.95 -> 4.35 ms; smaller
    // batches are one block's latency, which two points per lane double (128 x 513: 0.10 -> 0.13 ms).
    // geo_lpp = -2 forces this form, 1 | 2 | 4 the others.
    const long n_points = (long)a.n_lines * a.N;
    if (a.lpp == -2 || (a.lpp == 0 && n_points >= (long)n_cu * kGeoBlock)) {
      GeoArgs b = a;
      const int per = 2 * kGeoBlock2;
      const int rem = a.N % per;
      b.j_begin = 0;
      b.j_end = (rem > 0 && rem <= 16 && a.N > per) ? a.N - rem : a.N;
      hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fieldline_geometry_rows2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e1 != hipSuccess) return e1;
      hipLaunchKernelGGL(k_fieldline_geometry_rows2, dim3((b.j_end + per - 1) / per, a.n_lines), dim3(kGeoBlock2), lds, st, b);
      if (b.j_end < a.N) {
        GeoArgs c = a;
        c.j_begin = b.j_end; c.j_end = a.N;
        const long waves = (long)a.n_lines * (c.j_end - c.j_begin);
        hipLaunchKernelGGL(k_fieldline_geometry_tail, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, c);
      }
      lpp = 0;
    }
    auto go = [&](auto kern, int l) {
      hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e1 != hipSuccess) return e1;
      hipLaunchKernelGGL(kern, dim3((a.N * l + kGeoBlock - 1) / kGeoBlock, a.n_lines), dim3(kGeoBlock), lds, st, a);
      return hipSuccess;
    };
    hipError_t e2 = lpp == 0 ? hipSuccess : lpp == 8 ? go(k_fieldline_geometry_rows_split<8>, 8)
                  : lpp == 4 ? go(k_fieldline_geometry_rows_split<4>, 4)
                  : lpp == 2 ? go(k_fieldline_geometry_rows_split<2>, 2) : go(k_fieldline_geometry_rows, 1);
    if (e2 != hipSuccess) return e2;
  } else {
    hipLaunchKernelGGL(k_fieldline_geometry, dim3((a.N + 255) / 256, a.n_lines), dim3(256), 0, st, a);
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
