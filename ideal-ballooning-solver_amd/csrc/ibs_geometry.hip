// Field-line geometry on the device: the producer of the eight arrays the
// ballooning kernels consume, so that geometry never leaves HBM between the equilibrium tables and
// the eigen-solve.  Restates the arithmetic of the reference's vmec_fieldlines (utils.py:359-720);
// the radial splines (utils.py:37-158, 311-357) stay on the host and hand over per-surface Fourier
// coefficient vectors.
//
// Per grid point: theta_pest -> theta_vmec by a secant solve on
//   theta_vmec + sum_mn lmns sin(m theta_vmec - n phi) = theta_pest            (utils.py:391-416)
// then two Fourier syntheses (mnmax and Nyquist mode sets, utils.py:420-468) and the metric algebra
// (utils.py:474-720).  k_fieldline_geometry below is the plain form (one thread per point, one sincos per
// mode: any mode ordering, the fallback); the row kernels further down are what runs (FP64-issue bound).
#include <hip/hip_runtime.h>
#include "ibs_launch.hpp"
#include <cstdlib>
#include <type_traits>

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
  const size_t plane = a.plane, o = (size_t)line * a.ld + j; \
  a.geo[o] = bmag; a.geo[plane + o] = gradpar; a.geo[2 * plane + o] = cvdrift; a.geo[3 * plane + o] = gbdrift0; \
  a.geo[4 * plane + o] = gds2; a.geo[5 * plane + o] = gds21; a.geo[6 * plane + o] = gds22; a.geo[7 * plane + o] = gbdrift; \

__global__ void __launch_bounds__(256) k_fieldline_geometry(GeoArgs a) {
  const int line = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.N || (a.n_lines_dev && line >= *a.n_lines_dev)) return;
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



// =====================================================================================================================
// Row kernels (round 3).  The mode lists are walked row by row (a row = all modes of one m whose n advance by the common
// step dn): inside a row the angle m theta - n phi changes by the constant D = dn*phi per mode, so cos and sin follow
//   t[k+1] = 2 cos(D) t[k] - t[k-1]        (one fma each; rows hold <= ~60 modes: error growth ~k^2 eps)
// instead of a sincos per mode, and cos / sin(m theta) advance from row to row by one plane rotation.
//
// Against the round-2 kernels (k_fieldline_geometry_rows / _rows2 / _split):
//  (1) PREPARED TABLE IMAGES.  The per-surface tables a block works from -- array-of-structures, mode numbers multiplied
//      in, rows zero-padded, two modes per 160-byte group -- are built ONCE per call (k_geo_prepare) in global memory in
//      exactly the layout the LDS wants, so staging is a straight coalesced copy instead of a search + gather per
//      element in every block.
//  (2) FACTORISED ROOT SOLVE.  phi is fixed while theta_vmec is iterated (utils.py:391-416), and
//        sum_mn l_mn sin(m tv - n phi) = sum_m [ sin(m tv) P_m - cos(m tv) Q_m ],
//        P_m = sum_n l_mn cos(n phi),  Q_m = sum_n l_mn sin(n phi)
//      so ONE pass over the 242 modes gives (P_m, Q_m) -- kept in registers, MAXR rows at most -- and each of the ~6
//      secant evaluations costs a sincos and 6 flops per ROW instead of 2 per MODE.
//  (3) PERSISTENT BLOCKS.  One 512-thread block per CU (two waves per SIMD) owns a contiguous range of lines and re-stages its
//      table image only when the surface changes; between re-stagings its waves CLAIM wave-items one at a time.
//  (4) TWO FAMILIES OF FORMS.  A wave-item is 64*PPL/LPP consecutive grid points of one line.
//      One lane per point (LPP = 1; PPL = 2 points per lane for batches that fill the chip, 1 for medium ones) -- round 4: the
//      rows in N-SYMMETRIC form (a pair of modes n_c +- t dn costs one multiply-add per sum and point, cos / sin(t dn phi) by
//      one recurrence per row) with the table coefficients BROADCAST BY 64-BIT DPP (fmac_bc below: one ds_read_b64 per
//      sixteen coefficients, so the table reads are off the LDS pipe, which a ds_read_b128 occupies as long as a v_fma_f64
//      occupies the FP64 pipe whether or not its lanes broadcast).
//      Several lanes per point (LPP = 2 / 4 / 8) for small batches (the last rounds of ibs_refine_f64: a handful of lines):
//      every lane takes an equally long n-segment of EVERY row; the image stores each lane's groups of two modes of all
//      rows back to back (lane stride odd in 16-byte units: the LPP addresses of a read fall on distinct banks), so the
//      synthesis is ONE flat loop over the lane's groups with the next group's ten 16-byte reads in flight while the current
//      one is consumed -- rows of two groups (LPP = 8) would otherwise expose the LDS latency 26 + 11 times per point.
constexpr int kGeoBlock = 512;
constexpr int kGeoMaxRows = 128;

// phase timestamps (100 MHz wall clock) of wave 0 of the first 256 blocks: debug builds only (-DGEO_PROBE, tools/geo_probe.py)
#ifdef GEO_PROBE
__device__ long long geo_probe_buf[256 * 16];
#define GEO_PROBE_AT(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) geo_probe_buf[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define GEO_PROBE_AT(k) do {} while (0)
#endif

struct GeoImgLayout {      // offsets (in doubles) inside one surface's image: a function of the table sizes and LPP only
  int T1c, T2c;            // capacities (groups per lane) of the two lists, even
  int o_ri1, o_ri2;        // row info [nr][4]: m, first n, n advance per lane, groups per lane (LPP = 1: pairs)
  int o_int;               // ints: goff1[nr1 + 1], goff2[nr2 + 1], code1[nr1], code2[nr2] (+ LPP = 1: soff1[nr1 + 1], soff2[nr2 + 1])
  int o_lm, o_amn, o_anq;  // lane-major lists (LPP > 1)
  int s_lm, s_amn, s_anq;  // lane strides
  int o_c0mn, o_c0nq;      // LPP = 1, n-symmetric rows: centre coefficients [nr][10] (slot 9: the centre n)
  int o_symn;              //   pair tables in 16-double blocks (one lane of a 16-lane row each, see fmac_bc): mn [pairs][32] = (sum, difference) of
                           //   9 columns + pad, then nyq [pairs][16] = 8 columns (offset: header slot 14), then the lmns pairs of ALL rows by
                           //   pair index [t][16 NB] for the root solve (offset: header slot 18; NB = 2, or 3 with more than 12 rows)
  int total;               // capacity; LPP = 1: the used length is header slot 13
};
__host__ __device__ inline GeoImgLayout geo_layout(int mnmax, int nr1, int mnmax_nyq, int nr2, int lpp) {
  GeoImgLayout L;
  // a row of c modes gives every lane ceil(c / (2 lpp)) groups of two modes; + 1 keeps the total even
  L.T1c = (mnmax / (2 * lpp) + nr1 + 2) & ~1;
  L.T2c = (mnmax_nyq / (2 * lpp) + nr2 + 2) & ~1;
  L.o_ri1 = 20;            // header: s iota d_iota_d_s phiedge Aminor_p + the per-surface factors of the metric algebra + list sizes
  L.o_ri2 = L.o_ri1 + 4 * nr1;
  L.o_int = L.o_ri2 + 4 * nr2;
  const int n_int = ((nr1 + 1) + (nr2 + 1)) * (lpp == 1 ? 2 : 1) + nr1 + nr2;
  L.o_lm = (L.o_int + ((n_int + 1) >> 1) + 1) & ~1;
  if (lpp == 1) {
    // a row of c modes centred at k0 has max(k0, c - 1 - k0) < c pairs: the pair tables never hold more than one entry per mode
    L.s_lm = L.s_amn = L.s_anq = 0; L.o_amn = L.o_anq = L.o_lm;
    L.o_c0mn = L.o_lm;
    L.o_c0nq = L.o_c0mn + 10 * nr1;
    L.o_symn = L.o_c0nq + 10 * nr2;
    L.total = L.o_symn + 32 * mnmax + 16 * mnmax_nyq + kGeoMaxPairs * 16 * (nr1 > 12 ? 3 : 2);
    return L;
  }
  L.s_lm = 2 * (L.T1c + 1);
  L.o_amn = L.o_lm + lpp * L.s_lm;
  L.s_amn = 2 * (10 * L.T1c + 1);
  L.o_anq = L.o_amn + lpp * L.s_amn;
  L.s_anq = 2 * (10 * L.T2c + 1);
  L.o_c0mn = L.o_c0nq = L.o_symn = 0;
  L.total = L.o_anq + lpp * L.s_anq;
  return L;
}
int geo_lpp_index(int lpp) { return lpp == 1 ? 0 : (lpp == 2 ? 1 : (lpp == 4 ? 2 : 3)); }
size_t geo_image_doubles(const GeoArgs& a, int lpp) {
  return (size_t)geo_layout(a.mnmax, a.nrows_mn, a.mnmax_nyq, a.nrows_nyq, lpp).total;
}

// One block per surface.  Group = two modes x 10 coefficients (mode numbers multiplied in):
//   mn  : rmnc d_rmnc m*rmnc n*rmnc | d_zmns m*zmns n*zmns | d_lmns m*lmns n*lmns
//   nyq : gmnc bmnc d_bmnc m*bmnc n*bmnc bsupv bsubs bsubu bsubv (pad)
// so a mode costs five 16-byte LDS reads and one fma per accumulated quantity; lm holds the lmns pairs of the same groups
// (P / Q pass of the root solve).  Row codes (uniform over the block, so the kernels branch on scalars):
//   bits 0-1: 0 = same m as the row before (or m = 0 in the first row), 1 = m advanced by one (plane rotation), 2 = any
//             other m (sincos);   bits 2-3: 0 = first n is 0 (one lane per point), 1 = the same first n as the last row that
//             computed cos / sin(n phi), 2 = compute them.
__global__ void k_geo_mark(int n_lines, int n_surf, const int* line_surf, int* used) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_lines) used[min(max(line_surf[i], 0), n_surf - 1)] = 1;
}

__global__ void __launch_bounds__(256) k_geo_prepare(GeoArgs a, int lpp, double* img) {
  __shared__ int goff1[kGeoMaxRows + 1], goff2[kGeoMaxRows + 1], cnt1[kGeoMaxRows], cnt2[kGeoMaxRows];
  __shared__ double rm_s[2][kGeoMaxRows], rn_s[2][kGeoMaxRows];
  const int js = blockIdx.x, t = threadIdx.x;
  if (a.surf_used && a.surf_used[js] == 0) return;          // (block-uniform) no line of this call lies on the surface
  const int nr1 = a.nrows_mn, nr2 = a.nrows_nyq;
  const GeoImgLayout L = geo_layout(a.mnmax, nr1, a.mnmax_nyq, nr2, lpp);
  double* I = img + (size_t)js * L.total;
  // (the row tables come in with one load per thread; the serial passes below run on the LDS copies)
  for (int r = t; r < nr1; r += blockDim.x) { cnt1[r] = a.rows_mn[2 * r + 1]; const int k0 = a.rows_mn[2 * r]; rm_s[0][r] = a.xm[k0]; rn_s[0][r] = a.xn[k0]; }
  for (int r = t; r < nr2; r += blockDim.x) { cnt2[r] = a.rows_nyq[2 * r + 1]; const int k0 = a.rows_nyq[2 * r]; rm_s[1][r] = a.xm_nyq[k0]; rn_s[1][r] = a.xn_nyq[k0]; }
  __syncthreads();
  int* ints = reinterpret_cast<int*>(I + L.o_int);
  int* igoff1 = ints; int* igoff2 = igoff1 + nr1 + 1; int* code1 = igoff2 + nr2 + 1; int* code2 = code1 + nr1;
  if (t < 2) {                                     // groups per lane and row, running offsets (totals made even), row codes
    int* goff = t ? goff2 : goff1; const int* cnt = t ? cnt2 : cnt1; const int nr = t ? nr2 : nr1;
    const double dn = t ? a.dn_nyq : a.dn_mn;
    int* code = t ? code2 : code1; int* igoff = t ? igoff2 : igoff1; double* ri = I + (t ? L.o_ri2 : L.o_ri1);
    int o = 0;
    for (int r = 0; r < nr; ++r) { goff[r] = o; o += (cnt[r] + 2 * lpp - 1) / (2 * lpp); }
    if (o & 1) ++o;                                // (the last row gets one group of zeros more)
    goff[nr] = o;
    double m_prev = 0.0, keep_n = 0.0, keep_adv = 0.0;
    bool keep = false;
    for (int r = 0; r < nr; ++r) {
      const int gs = goff[r + 1] - goff[r];
      const double m = rm_s[t][r], n0 = rn_s[t][r], adv = 2.0 * gs * dn;
      ri[4 * r] = m; ri[4 * r + 1] = n0; ri[4 * r + 2] = adv; ri[4 * r + 3] = gs;
      const int mc = (m == m_prev) ? 0 : ((m - m_prev == 1.0) ? 1 : 2);
      int nc;
      if (lpp == 1 && n0 == 0.0) nc = 0;
      else if (keep && n0 == keep_n && adv == keep_adv) nc = 1;
      else { nc = 2; keep = true; keep_n = n0; keep_adv = adv; }
      code[r] = mc | (nc << 2);
      igoff[r] = goff[r];
      m_prev = m;
    }
    igoff[nr] = goff[nr];
  }
  __syncthreads();
  const int T1 = goff1[nr1], T2 = goff2[nr2];
  if (t == 0) {
    // header: scalars of the surface and the factors of the metric algebra that do not depend on the grid point
    // (utils.py:474, 654-720: Psi' = -phiedge / 2 pi, Bref = 2 |Psi'| / L^2, shat = -2 s iota' / iota, ...)
    const double* sc = a.scal + 6 * js;
    const double s_ = sc[0], iota = sc[1], diota = sc[2], dp = sc[3], phiedge = sc[4], Lm = sc[5];
    const double etf = -phiedge / (2 * M_PI);
    const double Bref = 2 * fabs(etf) / (Lm * Lm);
    const double sgn = etf > 0 ? 1.0 : (etf < 0 ? -1.0 : 0.0);
    const double sq = sqrt(s_);
    const double shat = (-2 * s_ / iota) * diota;
    const double mu0 = 4 * M_PI * 1.0e-7;
    I[0] = s_; I[1] = iota; I[2] = diota; I[3] = etf; I[4] = Lm; I[5] = 1.0 / Bref;
    I[6] = Lm * Lm * s_;                                  // gds2  = |grad alpha|^2 L^2 s
    I[7] = shat / Bref;                                   // gds21 = (grad alpha . grad psi) shat / Bref
    I[8] = shat * shat / (Lm * Lm * Bref * Bref * s_);     // gds22 = |grad psi|^2 shat^2 / (L^2 Bref^2 s)
    I[9] = -1.0 * 2 * Bref * Lm * Lm * sq * sgn;          // gbdrift  = I9 BxgB_alpha / B^3
    I[10] = -1.0 * 2 * shat / sq * sgn;                   // gbdrift0 = I10 BxgB_psi / B^3
    I[11] = 2 * Bref * Lm * Lm * sq * mu0 * dp * sgn / etf;   // cvdrift = gbdrift - I11 / B^2
    I[12] = Lm * iota;                                    // gradpar = I12 B^phi / B
    if (lpp != 1) { I[13] = (double)T1; I[14] = (double)T2; I[15] = -1.0; I[16] = I[17] = 0.0; }
  }
  const double* g_mn = a.tab_mn + (size_t)js * 6 * a.mnmax;
  const double* g_nq = a.tab_nyq + (size_t)js * 7 * a.mnmax_nyq;
  const int n1 = a.mnmax, n2 = a.mnmax_nyq;
  // ---- one lane per point: rows in n-symmetric form.  With the row's n = n_c + t dn, t = -k0 .. cnt-1-k0 (n_c = 0 whenever
  // the row holds n = 0: VMEC's rows run n = -ntor .. ntor, or 0 .. ntor for m = 0), beta = m theta - n_c phi, D = dn phi:
  //   sum_t v_t cos(beta - t D) = cos(beta) P + sin(beta) Q,    sum_t v_t sin(beta - t D) = sin(beta) P - cos(beta) Q,
  //   P = v_0 + sum_{t>0} (v_t + v_-t) cos(t D),                Q = sum_{t>0} (v_t - v_-t) sin(t D)
  // so a PAIR of modes costs one fma per accumulated quantity and point instead of two, cos / sin(t D) are the same for every
  // row, and the (P_m, Q_m) of the root solve are the lmns column of the same tables.
  if (lpp == 1) {
    __shared__ int k0_s[2][kGeoMaxRows], tm_s[2][kGeoMaxRows], so_s[2][kGeoMaxRows + 1], any_s[2], tmax_s;
    if (t == 0) { any_s[0] = any_s[1] = 0; tmax_s = 0; }
    __syncthreads();
    for (int e = t; e < nr1 + nr2; e += blockDim.x) {
      const int w = e >= nr1 ? 1 : 0, r = w ? e - nr1 : e;
      const int cnt = w ? cnt2[r] : cnt1[r];
      const double dn = w ? a.dn_nyq : a.dn_mn, n0 = rn_s[w][r];
      int k0 = 0; double nc = n0;
      if (dn != 0.0) {                                           // (the host check geo_rows_fit_host guards the same division)
        const double kz = -n0 / dn;
        const int kr = (int)(kz + (kz >= 0 ? 0.5 : -0.5));
        if (kr >= 0 && kr < cnt && fabs(n0 + kr * dn) <= 1e-9 * fabs(dn)) { k0 = kr; nc = 0.0; }
      }
      k0_s[w][r] = k0;
      const int tm = k0 > cnt - 1 - k0 ? k0 : cnt - 1 - k0;
      tm_s[w][r] = tm;
      (I + (w ? L.o_c0nq : L.o_c0mn))[10 * r + 9] = nc;
      if (w) (I + L.o_c0nq)[10 * r + 8] = 0.0;
      if (nc != 0.0) atomicMax(&any_s[w], 1);                  // some row is centred away from n = 0
      if (!w) atomicMax(&tmax_s, tm);
    }
    __syncthreads();
    if (t < 2) {
      const int nr = t ? nr2 : nr1;
      int* isoff = code2 + nr2 + (t ? nr1 + 1 : 0);
      double* ri = I + (t ? L.o_ri2 : L.o_ri1);
      int o = 0;
      for (int r = 0; r < nr; ++r) { so_s[t][r] = o; isoff[r] = o; ri[4 * r + 3] = tm_s[t][r]; o += tm_s[t][r]; }
      so_s[t][nr] = o; isoff[nr] = o;
    }
    __syncthreads();
    const int S1 = so_s[0][nr1], S2 = so_s[1][nr2];
    const int NB = nr1 > 12 ? 3 : 2, Ts = tmax_s;
    const int o_synq = L.o_symn + 32 * S1, o_pq = o_synq + 16 * S2;
    // The image reserves the root solve's (P, Q) tables for kGeoMaxPairs pair indices (geo_layout).  A row with more -- over
    // 129 modes centred on n = 0, or over 65 uncentred: not a VMEC table, but rows are caller data at the C ABI -- would run
    // past the image and past the LDS copy of it.  The host entry points check the rows (geo_rows_fit) and send such tables to
    // k_fieldline_geometry; as a second line of defence the surface is flagged here (slot 15 = -2, used length = the header)
    // and k_geo_rows writes NaN for its lines.
    // (the same flag when the rows together claim more pair entries than the image holds one-per-mode: overlapping rows -- refused by
    //  the host check, but device-resident rows are only checked once per table set)
    const bool too_long = Ts > kGeoMaxPairs || S1 > n1 || S2 > n2;
    if (t == 0) {
      I[13] = too_long ? (double)L.o_ri1 : (double)(o_pq + 16 * NB * Ts); I[14] = (double)o_synq; I[15] = too_long ? -2.0 : (double)Ts;
      I[16] = (double)any_s[0]; I[17] = (double)any_s[1];
      I[18] = (double)o_pq;
    }
    if (too_long) return;
    // value of column c of mode k:  mn  rmnc d_rmnc n*rmnc | zmns d_zmns n*zmns | lmns d_lmns n*lmns
    //                               nyq gmnc bmnc d_bmnc n*bmnc bsupv bsubs bsubu bsubv
    auto val_mn = [&](int k, int c) {
      const int fam = c / 3, kind = c - 3 * fam;
      const double v = g_mn[(size_t)(kind == 1 ? 3 + fam : fam) * n1 + k];
      return kind == 2 ? a.xn[k] * v : v;
    };
    auto val_nq = [&](int k, int c) {
      const int pl = c <= 2 ? c : (c == 3 ? 1 : c - 1);
      const double v = g_nq[(size_t)pl * n2 + k];
      return c == 3 ? a.xn_nyq[k] * v : v;
    };
    for (int e = t; e < 9 * nr1; e += blockDim.x) { const int r = e / 9, c = e - 9 * r; (I + L.o_c0mn)[10 * r + c] = val_mn(a.rows_mn[2 * r] + k0_s[0][r], c); }
    for (int e = t; e < 8 * nr2; e += blockDim.x) { const int r = e / 8, c = e - 8 * r; (I + L.o_c0nq)[10 * r + c] = val_nq(a.rows_nyq[2 * r] + k0_s[1][r], c); }
    auto pair_row = [](int tt, int nr, const int* so) { int r = 0; while (r + 1 < nr && so[r + 1] <= tt) ++r; return r; };
    double* symn = I + L.o_symn;
    for (int e = t; e < 9 * S1; e += blockDim.x) {
      const int tt = e / 9, c = e - 9 * tt, r = pair_row(tt, nr1, so_s[0]);
      const int dt = tt - so_s[0][r] + 1, k0 = k0_s[0][r], kp = k0 + dt, km = k0 - dt, kb = a.rows_mn[2 * r];
      const double vp = kp < cnt1[r] ? val_mn(kb + kp, c) : 0.0, vm = km >= 0 ? val_mn(kb + km, c) : 0.0;
      symn[32 * tt + 2 * c] = vp + vm; symn[32 * tt + 2 * c + 1] = vp - vm;
    }
    double* synq = I + o_synq;
    for (int e = t; e < 8 * S2; e += blockDim.x) {
      const int tt = e >> 3, c = e & 7, r = pair_row(tt, nr2, so_s[1]);
      const int dt = tt - so_s[1][r] + 1, k0 = k0_s[1][r], kp = k0 + dt, km = k0 - dt, kb = a.rows_nyq[2 * r];
      const double vp = kp < cnt2[r] ? val_nq(kb + kp, c) : 0.0, vm = km >= 0 ? val_nq(kb + km, c) : 0.0;
      synq[2 * e] = vp + vm; synq[2 * e + 1] = vp - vm;
    }
    double* pq = I + o_pq;
    const double* lmn = g_mn + 2 * (size_t)n1;
    for (int e = t; e < 8 * NB * Ts; e += blockDim.x) {           // (rows with fewer pairs, and the slots beyond the rows: zeros)
      const int tt = e / (8 * NB), r = e - tt * 8 * NB;
      double vp = 0.0, vm = 0.0;
      if (r < nr1) {
        const int k0 = k0_s[0][r], kp = k0 + tt + 1, km = k0 - tt - 1, kb = a.rows_mn[2 * r];
        if (kp < cnt1[r]) vp = lmn[kb + kp];
        if (km >= 0) vm = lmn[kb + km];
      }
      pq[16 * NB * tt + 2 * r] = vp + vm; pq[16 * NB * tt + 2 * r + 1] = vp - vm;
    }
    return;
  }
  auto row_of = [](int g, int nr, const int* goff) { int r = 0; while (r + 1 < nr && goff[r + 1] <= g) ++r; return r; };
  // source mode of slot u (0 | 1) of group g of lane sub, or -1 (padding)
  auto src_mode = [&](int sub, int g, int u, int nr, const int* goff, const int* cnt, const int* rows) {
    const int r = row_of(g, nr, goff);
    const int gs = goff[r + 1] - goff[r];
    const int i = (sub * gs + (g - goff[r])) * 2 + u;
    return i < cnt[r] ? rows[2 * r] + i : -1;
  };
  for (int e = t; e < lpp * T1 * 2; e += blockDim.x) {
    const int sub = e / (T1 * 2), rem = e - sub * T1 * 2, g = rem >> 1, u = rem & 1;
    const int k = src_mode(sub, g, u, nr1, goff1, cnt1, a.rows_mn);
    double* q = I + L.o_amn + (size_t)sub * L.s_amn + 20 * g + 10 * u;
    double* lq = I + L.o_lm + (size_t)sub * L.s_lm + 2 * g + u;
    if (k >= 0) {
      const double m = a.xm[k], n = a.xn[k];
      const double rm = g_mn[k], zm = g_mn[n1 + k], l_ = g_mn[2 * n1 + k];
      *lq = l_;
      q[0] = rm; q[1] = g_mn[3 * n1 + k]; q[2] = m * rm; q[3] = n * rm;
      q[4] = g_mn[4 * n1 + k]; q[5] = m * zm; q[6] = n * zm;
      q[7] = g_mn[5 * n1 + k]; q[8] = m * l_; q[9] = n * l_;
    } else {
      *lq = 0.0;
      for (int c = 0; c < 10; ++c) q[c] = 0.0;
    }
  }
  for (int e = t; e < lpp * T2 * 2; e += blockDim.x) {
    const int sub = e / (T2 * 2), rem = e - sub * T2 * 2, g = rem >> 1, u = rem & 1;
    const int k = src_mode(sub, g, u, nr2, goff2, cnt2, a.rows_nyq);
    double* q = I + L.o_anq + (size_t)sub * L.s_anq + 20 * g + 10 * u;
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

// sin and cos for |x| < ~1e5 (the angles here stay below ~2e3): Cody-Waite reduction by pi/2 in three 33-bit pieces (k * piece
// exact for |k| < 2^20, the scheme of fdlibm's medium range) and the fdlibm kernels on [-pi/4, pi/4]; ~1 ulp.  A third of the
// instructions of the general-range library sincos, which is what the root solve's evaluations mostly consisted of.
__device__ __forceinline__ void geo_sincos(double x, double* sn, double* cs) {
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, x);
  r = fma(-k, 6.07710050650619224932e-11, r);
  r = fma(-k, 2.02226624879595063154e-21, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06); ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03); ps = fma(z, ps, -1.66666666666666324348e-01);
  const double s = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07); pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03); pc = fma(z, pc, 4.16666666666666019037e-02);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k;
  const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// sum over the LPP adjacent lanes that share one grid point (in-register butterfly, quad_perm / row_half_mirror DPP);
// every lane of the group ends up with the same bits
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

// (cos, sin) of a row's first angle for PPL points of a lane.  cos / sin(m tv) advance with the rows (rows come sorted by
// m: one plane rotation per row), cos / sin(n phi) of the lane's first n of a row are kept from the last row that computed
// them (phi is fixed per point, and all rows but the m = 0 one start at the same n).  The row codes of the image say which
// case a row is, uniformly for the block.
template <int PPL>
struct RowTrig {
  double phi[PPL], ctv[PPL], stv[PPL], tv[PPL];
  double cm[PPL], sm[PPL];              // cos / sin(m_cur tv)
  double ck[PPL], sk[PPL];              // cos / sin(n phi) of the last row with code 2
  __device__ __forceinline__ void init_phi(const double (&phi_)[PPL]) {
#pragma unroll
    for (int p = 0; p < PPL; ++p) { phi[p] = phi_[p]; ck[p] = 1.0; sk[p] = 0.0; }
  }
  __device__ __forceinline__ void set_tv(const double (&tv_)[PPL]) {
#pragma unroll
    for (int p = 0; p < PPL; ++p) { tv[p] = tv_[p]; geo_sincos(tv[p], &stv[p], &ctv[p]); }
  }
  __device__ __forceinline__ void rewind() {         // back to m = 0 for the next pass over the rows (same tv)
#pragma unroll
    for (int p = 0; p < PPL; ++p) { cm[p] = 1.0; sm[p] = 0.0; }
  }
  __device__ __forceinline__ void advance_m(int mcode, double m) {
    if (mcode == 1) {                  // VMEC order: the next row is the next m -> one plane rotation
#pragma unroll
      for (int p = 0; p < PPL; ++p) {
        const double c2 = cm[p] * ctv[p] - sm[p] * stv[p], s2 = sm[p] * ctv[p] + cm[p] * stv[p];
        cm[p] = c2; sm[p] = s2;
      }
    } else if (mcode == 2) {
#pragma unroll
      for (int p = 0; p < PPL; ++p) geo_sincos(m * tv[p], &sm[p], &cm[p]);
    }
  }
  // cos / sin(n phi) of this lane's first n of a row
  __device__ __forceinline__ void n_part(int ncode, double n0, double (&cn)[PPL], double (&sn)[PPL]) {
    if (ncode == 2) {
#pragma unroll
      for (int p = 0; p < PPL; ++p) geo_sincos(n0 * phi[p], &sk[p], &ck[p]);
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) { cn[p] = ncode == 0 ? 1.0 : ck[p]; sn[p] = ncode == 0 ? 0.0 : sk[p]; }
  }
  __device__ __forceinline__ void start(int code, double m, double n0, double (&ca)[PPL], double (&sa)[PPL]) {
    advance_m(code & 3, m);
    double cn[PPL], sn[PPL];
    n_part(code >> 2, n0, cn, sn);
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      ca[p] = cm[p] * cn[p] + sm[p] * sn[p];      // cos(m tv - n0 phi)
      sa[p] = sm[p] * cn[p] - cm[p] * sn[p];      // sin(m tv - n0 phi)
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
// The same algebra (utils.py:474-720) with the factors that depend on the surface alone taken from the image header H
// (k_geo_prepare): two divisions per grid point (1 / sqrt(g), 1 / |B|) instead of nine and no square root.
__device__ __forceinline__ void geo_tail_h(const GeoArgs& a, const double* H, int line, int j, double phi, double sp, double cp,
                                           const GeoSums& q) {
  const double iota = H[1], diota = H[2], etf = H[3];
  const double X_t = q.R_t * cp, X_p = q.R_p * cp - q.R * sp, X_s = q.R_s * cp;
  const double Y_t = q.R_t * sp, Y_p = q.R_p * sp + q.R * cp, Y_s = q.R_s * sp;
  const double Z_t = q.Z_t, Z_p = q.Z_p, Z_s = q.Z_s;
  const double isg = 1.0 / q.sqg;
  const double gsx = (Y_t * Z_p - Z_t * Y_p) * isg, gsy = (Z_t * X_p - X_t * Z_p) * isg, gsz = (X_t * Y_p - Y_t * X_p) * isg;     // grad s   (utils.py:480-508)
  const double gtx = (Y_p * Z_s - Z_p * Y_s) * isg, gty = (Z_p * X_s - X_p * Z_s) * isg, gtz = (X_p * Y_s - Y_p * X_s) * isg;     // grad theta
  const double gpx = (Y_s * Z_t - Z_s * Y_t) * isg, gpy = (Z_s * X_t - X_s * Z_t) * isg, gpz = (X_s * Y_t - Y_s * X_t) * isg;     // grad phi
  const double ls = q.l_s - phi * diota;
  const double c1 = 1 + q.l_t, c2 = -iota + q.l_p;
  const double gax = ls * gsx + c1 * gtx + c2 * gpx, gay = ls * gsy + c1 * gty + c2 * gpy, gaz = ls * gsz + c1 * gtz + c2 * gpz;  // grad alpha (utils.py:515-538)
  const double psx = gsx * etf, psy = gsy * etf, psz = gsz * etf;                                                                 // grad psi
  const double BxgB_alpha = (q.Bsub_s * q.B_t * c2 + q.Bsub_t * q.B_p * ls + q.Bsub_p * q.B_s * c1
                             - q.Bsub_p * q.B_t * ls - q.Bsub_t * q.B_s * c2 - q.Bsub_s * q.B_p * c1) * isg;                      // utils.py:603-618
  const double BxgB_psi = (q.Bsub_t * q.B_p - q.Bsub_p * q.B_t) * isg * etf;                                                      // utils.py:646-650
  const double iB = 1.0 / q.modB, iB2 = iB * iB, iB3 = iB2 * iB;
  const double bmag = q.modB * H[5];
  const double gradpar = H[12] * q.Bsup_phi * iB;
  const double gds2 = (gax * gax + gay * gay + gaz * gaz) * H[6];
  const double gds21 = (gax * psx + gay * psy + gaz * psz) * H[7];
  const double gds22 = (psx * psx + psy * psy + psz * psz) * H[8];
  const double gbdrift = H[9] * BxgB_alpha * iB3;
  const double gbdrift0 = H[10] * BxgB_psi * iB3;
  const double cvdrift = gbdrift - H[11] * iB2;
  const size_t plane = a.plane, o = (size_t)line * a.ld + j;
  a.geo[o] = bmag; a.geo[plane + o] = gradpar; a.geo[2 * plane + o] = cvdrift; a.geo[3 * plane + o] = gbdrift0;
  a.geo[4 * plane + o] = gds2; a.geo[5 * plane + o] = gds21; a.geo[6 * plane + o] = gds22; a.geo[7 * plane + o] = gbdrift;
}

// one group (two modes) of the non-Nyquist / Nyquist synthesis for one point; advances the angle recurrence by two modes
__device__ __forceinline__ void geo_group_mn(GeoSums& A, const double2 (&q)[10], double two_cD, double& sa, double& ca,
                                             double& sm1, double& cm1) {
  const double sa_ = sa, ca_ = ca;
  const double sb = fma(two_cD, sa_, -sm1), cb = fma(two_cD, ca_, -cm1);
  A.R = fma(q[0].x, ca_, A.R); A.R_s = fma(q[0].y, ca_, A.R_s); A.R_t = fma(-q[1].x, sa_, A.R_t); A.R_p = fma(q[1].y, sa_, A.R_p);
  A.Z_s = fma(q[2].x, sa_, A.Z_s); A.Z_t = fma(q[2].y, ca_, A.Z_t); A.Z_p = fma(-q[3].x, ca_, A.Z_p);
  A.l_s = fma(q[3].y, sa_, A.l_s); A.l_t = fma(q[4].x, ca_, A.l_t); A.l_p = fma(-q[4].y, ca_, A.l_p);
  A.R = fma(q[5].x, cb, A.R); A.R_s = fma(q[5].y, cb, A.R_s); A.R_t = fma(-q[6].x, sb, A.R_t); A.R_p = fma(q[6].y, sb, A.R_p);
  A.Z_s = fma(q[7].x, sb, A.Z_s); A.Z_t = fma(q[7].y, cb, A.Z_t); A.Z_p = fma(-q[8].x, cb, A.Z_p);
  A.l_s = fma(q[8].y, sb, A.l_s); A.l_t = fma(q[9].x, cb, A.l_t); A.l_p = fma(-q[9].y, cb, A.l_p);
  sm1 = sb; cm1 = cb;
  sa = fma(two_cD, sb, -sa_); ca = fma(two_cD, cb, -ca_);
}
__device__ __forceinline__ void geo_group_nyq(GeoSums& A, const double2 (&q)[10], double two_cD, double& sa, double& ca,
                                              double& sm1, double& cm1) {
  const double sa_ = sa, ca_ = ca;
  const double sb = fma(two_cD, sa_, -sm1), cb = fma(two_cD, ca_, -cm1);
  A.sqg = fma(q[0].x, ca_, A.sqg); A.modB = fma(q[0].y, ca_, A.modB); A.B_s = fma(q[1].x, ca_, A.B_s);
  A.B_t = fma(-q[1].y, sa_, A.B_t); A.B_p = fma(q[2].x, sa_, A.B_p);
  A.Bsup_phi = fma(q[2].y, ca_, A.Bsup_phi); A.Bsub_s = fma(q[3].x, sa_, A.Bsub_s); A.Bsub_t = fma(q[3].y, ca_, A.Bsub_t);
  A.Bsub_p = fma(q[4].x, ca_, A.Bsub_p);
  A.sqg = fma(q[5].x, cb, A.sqg); A.modB = fma(q[5].y, cb, A.modB); A.B_s = fma(q[6].x, cb, A.B_s);
  A.B_t = fma(-q[6].y, sb, A.B_t); A.B_p = fma(q[7].x, sb, A.B_p);
  A.Bsup_phi = fma(q[7].y, cb, A.Bsup_phi); A.Bsub_s = fma(q[8].x, sb, A.Bsub_s); A.Bsub_t = fma(q[8].y, cb, A.Bsub_t);
  A.Bsub_p = fma(q[9].x, cb, A.Bsub_p);
  sm1 = sb; cm1 = cb;
  sa = fma(two_cD, sb, -sa_); ca = fma(two_cD, cb, -ca_);
}

// reciprocal for well-scaled operands (hardware seed + two Newton steps): steers the secant step only
__device__ __forceinline__ double geo_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
}

// acc += tab[lane K of this lane's 16-lane row] * x: the 64-bit DPP form of v_fmac_f64 (row_newbcast, the one DPP control the
// FP64 pipe takes; full rate, tools/dpp64_probe.hip).  ONE ds_read_b64 -- every lane its (lane & 15)-th entry of a 16-double
// block -- so feeds sixteen multiply-adds per point instead of one 16-byte broadcast read per two: the table reads leave the
// LDS pipe (which a ds_read_b128 occupies as long as a v_fma_f64 occupies the FP64 pipe) and the 36 registers they landed in.
// `tab` must come straight from the LDS read (a VALU write needs two wait states before a DPP read: the blocks are loaded a
// whole step before they are used).
template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, double tab, double x) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x), "n"(K));
}
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}

// One wave-item: grid points [jfirst, jfirst + 64 PPL / LPP) of `line`, tables of its surface staged in LDS at I.
template <int PPL, int LPP, int MAXR>
__device__ __forceinline__ void geo_item(const GeoArgs& a, const double* I, const GeoImgLayout& L, int line, int jfirst) {
  static_assert(PPL == 1 || LPP == 1, "either several points per lane or several lanes per point");
  const int lane = threadIdx.x & 63;
  const int sub = LPP > 1 ? lane % LPP : 0;
  const int nr1 = a.nrows_mn, nr2 = a.nrows_nyq;
  const double* ri1 = I + L.o_ri1; const double* ri2 = I + L.o_ri2;
  const int* goff1 = reinterpret_cast<const int*>(I + L.o_int); const int* goff2 = goff1 + nr1 + 1;
  const int* code1 = goff2 + nr2 + 1; const int* code2 = code1 + nr1;
  const double iota = I[1];
  int j[PPL]; bool live[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    const int jr = LPP > 1 ? jfirst + lane / LPP : jfirst + p * 64 + lane;
    live[p] = jr < a.j_end;
    j[p] = live[p] ? jr : a.j_end - 1;
  }
  const double alpha = a.line_alpha[line];
  double tp[PPL], phi[PPL], sD[PPL], cD[PPL], two_cD[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    tp[p] = a.theta[j[p]];
    phi[p] = (tp[p] - alpha) / iota;                                          // utils.py:373 (phi_center = 0)
    geo_sincos(a.dn_mn * phi[p], &sD[p], &cD[p]);
    two_cD[p] = 2.0 * cD[p];
  }
  RowTrig<PPL> rs;
  rs.init_phi(phi);
  GEO_PROBE_AT(2);
  // ---- (P_m, Q_m) of the root solve
  double Pm[PPL][MAXR], Qm[PPL][MAXR];
  const int* soff1 = code2 + nr2; const int* soff2 = soff1 + nr1 + 1;         // (LPP = 1 images only)
  if constexpr (LPP == 1) {
    // n-symmetric rows (k_geo_prepare): one recurrence for cos / sin(t D), two multiply-adds per row and pair index, all rows'
    // (sum, difference) of one pair index in NB blocks of sixteen
    constexpr int NB = (2 * MAXR + 15) / 16;
    const double* pq = I + __builtin_amdgcn_readfirstlane((int)I[18]);
    const double* c0 = I + L.o_c0mn;
    const int Tsym = __builtin_amdgcn_readfirstlane((int)I[15]);
    const int l16 = lane & 15;
    if (Tsym > 0) {
      double cx[PPL], sx[PPL], cy[PPL], sy[PPL];
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {                                          // first pair: starts the sums (uniform reads)
        const double2 u = *reinterpret_cast<const double2*>(pq + 2 * r);
        const double l0 = r < nr1 ? c0[10 * (r < nr1 ? r : 0) + 6] : 0.0;
#pragma unroll
        for (int p = 0; p < PPL; ++p) { Pm[p][r] = fma(u.x, cD[p], l0); Qm[p][r] = u.y * sD[p]; }
      }
#pragma unroll
      for (int p = 0; p < PPL; ++p) { cx[p] = cD[p]; sx[p] = sD[p]; cy[p] = fma(two_cD[p], cD[p], -1.0); sy[p] = two_cD[p] * sD[p]; }
      auto pq_step = [&](const double* blk, const double (&c_)[PPL], const double (&s_)[PPL]) {
        double tb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) tb[b] = blk[16 * b + l16];
        static_for<0, MAXR>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
#pragma unroll
          for (int p = 0; p < PPL; ++p) { fmac_bc<(2 * r) & 15>(Pm[p][r], tb[(2 * r) >> 4], c_[p]); fmac_bc<(2 * r + 1) & 15>(Qm[p][r], tb[(2 * r) >> 4], s_[p]); }
        });
      };
      int t = 1;
      for (; t + 1 < Tsym; t += 2) {
        pq_step(pq + 16 * NB * t, cy, sy);
#pragma unroll
        for (int p = 0; p < PPL; ++p) { cx[p] = fma(two_cD[p], cy[p], -cx[p]); sx[p] = fma(two_cD[p], sy[p], -sx[p]); }
        pq_step(pq + 16 * NB * (t + 1), cx, sx);
#pragma unroll
        for (int p = 0; p < PPL; ++p) { cy[p] = fma(two_cD[p], cx[p], -cy[p]); sy[p] = fma(two_cD[p], sx[p], -sy[p]); }
      }
      if (t < Tsym) pq_step(pq + 16 * NB * t, cy, sy);
    } else {
#pragma unroll
      for (int r = 0; r < MAXR; ++r)
#pragma unroll
        for (int p = 0; p < PPL; ++p) { Pm[p][r] = r < nr1 ? c0[10 * (r < nr1 ? r : 0) + 6] : 0.0; Qm[p][r] = 0.0; }
    }
    if (__builtin_amdgcn_readfirstlane((int)I[16]) != 0) {                      // rows centred away from n = 0 (not VMEC's)
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        if (r < nr1) {
          const double nc = c0[10 * r + 9];
#pragma unroll
          for (int p = 0; p < PPL; ++p) {
            double sn, cn;
            geo_sincos(nc * phi[p], &sn, &cn);
            const double P_ = cn * Pm[p][r] - sn * Qm[p][r], Q_ = sn * Pm[p][r] + cn * Qm[p][r];
            Pm[p][r] = P_; Qm[p][r] = Q_;
          }
        }
      }
    }
  } else {
    // one pass over lmns with the n-recurrence alone (several lanes per point: every lane its n-segment of every row)
    const double2* Lp = reinterpret_cast<const double2*>(I + L.o_lm + (size_t)sub * L.s_lm);
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
#pragma unroll
      for (int p = 0; p < PPL; ++p) { Pm[p][r] = 0.0; Qm[p][r] = 0.0; }
      if (r < nr1) {                                                          // (block-uniform)
        const int g0 = __builtin_amdgcn_readfirstlane(goff1[r]), g1 = __builtin_amdgcn_readfirstlane(goff1[r + 1]);
        const int code = __builtin_amdgcn_readfirstlane(code1[r]);
        const double n0 = ri1[4 * r + 1] + (LPP > 1 ? (double)sub * ri1[4 * r + 2] : 0.0);
        double c0[PPL], s0[PPL], cm1[PPL], sm1[PPL], aP[PPL], aQ[PPL], bP[PPL], bQ[PPL];
        rs.n_part(code >> 2, n0, c0, s0);
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          cm1[p] = c0[p] * cD[p] + s0[p] * sD[p]; sm1[p] = s0[p] * cD[p] - c0[p] * sD[p];     // the (virtual) previous n: angle - D
          aP[p] = aQ[p] = bP[p] = bQ[p] = 0.0;
        }
        for (int g = g0; g < g1; ++g) {
          const double2 u = Lp[g];
#pragma unroll
          for (int p = 0; p < PPL; ++p) {
            const double c1 = fma(two_cD[p], c0[p], -cm1[p]), s1 = fma(two_cD[p], s0[p], -sm1[p]);
            aP[p] = fma(u.x, c0[p], aP[p]); aQ[p] = fma(u.x, s0[p], aQ[p]);
            bP[p] = fma(u.y, c1, bP[p]); bQ[p] = fma(u.y, s1, bQ[p]);
            cm1[p] = c1; sm1[p] = s1;
            c0[p] = fma(two_cD[p], c1, -c0[p]); s0[p] = fma(two_cD[p], s1, -s0[p]);
          }
        }
#pragma unroll
        for (int p = 0; p < PPL; ++p) { Pm[p][r] = group_sum<LPP>(aP[p] + bP[p]); Qm[p][r] = group_sum<LPP>(aQ[p] + bQ[p]); }
      }
    }
  }
  GEO_PROBE_AT(3);
  // ---- theta_vmec: tv + sum_m [sin(m tv) P_m - cos(m tv) Q_m] = theta_pest                 utils.py:391-416
  int mcode[MAXR];                                                            // (scalar registers)
#pragma unroll
  for (int r = 0; r < MAXR; ++r) mcode[r] = r < nr1 ? (__builtin_amdgcn_readfirstlane(code1[r]) & 3) : 0;
  auto resid = [&](const double (&tv)[PPL], double (&out)[PPL]) {
    double acc[PPL], cmv[PPL], smv[PPL], ct[PPL], st[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) { geo_sincos(tv[p], &st[p], &ct[p]); cmv[p] = 1.0; smv[p] = 0.0; acc[p] = 0.0; }
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      if (r < nr1) {
        if (mcode[r] == 1) {
#pragma unroll
          for (int p = 0; p < PPL; ++p) {
            const double c2 = cmv[p] * ct[p] - smv[p] * st[p], s2 = smv[p] * ct[p] + cmv[p] * st[p];
            cmv[p] = c2; smv[p] = s2;
          }
        } else if (mcode[r] == 2) {
          const double m = ri1[4 * r];
#pragma unroll
          for (int p = 0; p < PPL; ++p) geo_sincos(m * tv[p], &smv[p], &cmv[p]);
        }
#pragma unroll
        for (int p = 0; p < PPL; ++p) acc[p] = fma(smv[p], Pm[p][r], fma(-cmv[p], Qm[p][r], acc[p]));
      }
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) out[p] = tp[p] - (tv[p] + acc[p]);
  };
  // Secant iteration (superlinear, e_{n+1} ~ C e_n e_{n-1}): it stops, like the reference's scipy secant (tol 1.48e-8,
  // utils.py:391-416), on the step size; a step below 1e-9 means the point it leaves was 1e-9 from the root, so the point
  // it lands on is at ~1e-13 or better and is not evaluated again.  Second point: one fixed-point step
  // theta_p + resid(theta_p) instead of the reference's theta_p + 0.1 (same root, about two evaluations fewer).
  // All points of a wave iterate in step: a point that has converged keeps its value.
  double p0[PPL], q0[PPL], p1[PPL], q1[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) p0[p] = tp[p];
  resid(p0, q0);
#pragma unroll
  for (int p = 0; p < PPL; ++p) p1[p] = tp[p] + q0[p];
  resid(p1, q1);
  bool fin[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) fin[p] = false;
  for (int it = 0; it < 40; ++it) {
    bool all = true;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const double den = q1[p] - q0[p];
      if (!fin[p]) {
        if (den == 0.0) fin[p] = true;
        else {
          const double step = q1[p] * (p1[p] - p0[p]) * geo_rcp(den);
          p0[p] = p1[p]; q0[p] = q1[p];
          p1[p] = p1[p] - step;
          if (fabs(step) <= 1e-9 * fmax(1.0, fabs(p1[p]))) fin[p] = true;
        }
      }
      all = all && fin[p];
    }
    if (__builtin_amdgcn_ballot_w64(!all) == 0ull) break;      // every point of the wave is done
    double qn[PPL];
    resid(p1, qn);
#pragma unroll
    for (int p = 0; p < PPL; ++p) if (!fin[p]) q1[p] = qn[p];
  }
  GEO_PROBE_AT(4);
  // ---- Fourier synthesis: non-Nyquist set (utils.py:420-444), then Nyquist set (utils.py:447-468)
  GeoSums S[PPL];
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    S[p].R = S[p].R_s = S[p].R_t = S[p].R_p = S[p].Z_s = S[p].Z_t = S[p].Z_p = S[p].l_s = S[p].l_t = S[p].l_p = 0.0;
    S[p].sqg = S[p].modB = S[p].B_s = S[p].B_t = S[p].B_p = S[p].Bsup_phi = S[p].Bsub_s = S[p].Bsub_t = S[p].Bsub_p = 0.0;
  }
  rs.set_tv(p1);
  auto run_set = [&](auto is_nyq, const double* list, const int* goff, const int* code, const double* ri, int nr) {
    constexpr bool NYQ = decltype(is_nyq)::value;
    const double2* base = reinterpret_cast<const double2*>(list);
    rs.rewind();
    // this row's table entries are fetched one row ahead (the LDS latency hides behind the previous row)
    int g_nx = goff[0], e_nx = goff[nr > 0 ? 1 : 0], c_nx = code[0];
    double m_nx = ri[0], n_nx = ri[1], a_nx = ri[2];
    for (int r = 0; r < nr; ++r) {
      const int g0 = __builtin_amdgcn_readfirstlane(g_nx), g1 = __builtin_amdgcn_readfirstlane(e_nx);
      const int cd = __builtin_amdgcn_readfirstlane(c_nx);
      const double m_r = m_nx, n_r = n_nx + (LPP > 1 ? (double)sub * a_nx : 0.0);
      { const int rn = r + 1 < nr ? r + 1 : r; g_nx = goff[rn]; e_nx = goff[rn + 1]; c_nx = code[rn]; m_nx = ri[4 * rn]; n_nx = ri[4 * rn + 1]; a_nx = ri[4 * rn + 2]; }
      double sa[PPL], ca[PPL], sm1[PPL], cm1[PPL];
      rs.start(cd, m_r, n_r, ca, sa);
#pragma unroll
      for (int p = 0; p < PPL; ++p) { sm1[p] = sa[p] * cD[p] + ca[p] * sD[p]; cm1[p] = ca[p] * cD[p] - sa[p] * sD[p]; }   // angle + D
      const double2* q = base + 10 * g0;
      for (int g = g0; g < g1; ++g, q += 10) {
        double2 v[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = q[i];
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          if constexpr (NYQ) geo_group_nyq(S[p], v, two_cD[p], sa[p], ca[p], sm1[p], cm1[p]);
          else geo_group_mn(S[p], v, two_cD[p], sa[p], ca[p], sm1[p], cm1[p]);
        }
      }
    }
  };
  // one lane per point: a row's (P, Q) of every column by ONE recurrence over the pairs, then the row's cos / sin(beta)
  // distribute them over the sums (k_geo_prepare)
  auto run_sym = [&](auto is_nyq, const double* c0t, const double* tab, const int* soff, const int* code, const double* ri, int nr, bool any_nc) {
    constexpr bool NYQ = decltype(is_nyq)::value;
    constexpr int A = NYQ ? 8 : 9, W = NYQ ? 16 : 32;                           // columns; doubles per pair in the table
    const int l16 = lane & 15;
    rs.rewind();
    for (int r = 0; r < nr; ++r) {
      const int t0 = __builtin_amdgcn_readfirstlane(soff[r]), t1 = __builtin_amdgcn_readfirstlane(soff[r + 1]);
      const double m_r = ri[4 * r];
      rs.advance_m(__builtin_amdgcn_readfirstlane(code[r]) & 3, m_r);
      const double* c0 = c0t + 10 * r;
      // cos / sin(t D) live in two register sets that take turns (x = t, y = t + 1, then x <- 2 cos(D) y - x = t + 2, ...):
      // no register moves in the loop
      double P[PPL][A], Q[PPL][A], cx[PPL], sx[PPL], cy[PPL], sy[PPL];
      // (a pair's table block is read while the pair before it is consumed: the read latency stays off the FP64 pipe's path)
      auto load_blk = [&](const double* blk, double (&tb)[2]) {
        tb[0] = blk[l16];
        if constexpr (!NYQ) tb[1] = blk[16 + l16];                             // (lanes 0, 1: column 8)
      };
      auto pair_step = [&](const double (&tb)[2], const double (&c_)[PPL], const double (&s_)[PPL]) {
        static_for<0, 8>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
#pragma unroll
          for (int p = 0; p < PPL; ++p) { fmac_bc<2 * c>(P[p][c], tb[0], c_[p]); fmac_bc<2 * c + 1>(Q[p][c], tb[0], s_[p]); }
        });
        if constexpr (!NYQ) {
#pragma unroll
          for (int p = 0; p < PPL; ++p) { fmac_bc<0>(P[p][A - 1], tb[1], c_[p]); fmac_bc<1>(Q[p][A - 1], tb[1], s_[p]); }
        }
      };
      const double* q = tab + (size_t)W * t0;
      if (t0 < t1) {                                                           // first pair: starts the sums (uniform reads)
        double ty[2], tx[2];
        load_blk(q + W, ty);                                                   // (beyond a row's last pair: the next row's, unused)
#pragma unroll
        for (int c = 0; c < A; ++c) {
          const double2 u = *reinterpret_cast<const double2*>(q + 2 * c);
#pragma unroll
          for (int p = 0; p < PPL; ++p) { P[p][c] = fma(u.x, cD[p], c0[c]); Q[p][c] = u.y * sD[p]; }
        }
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          cx[p] = cD[p]; sx[p] = sD[p];
          cy[p] = fma(two_cD[p], cD[p], -1.0); sy[p] = two_cD[p] * sD[p];         // t = 2
        }
        q += W;
        int t = t0 + 1;
        for (; t + 1 < t1; t += 2, q += 2 * W) {
          load_blk(q + W, tx);
          __builtin_amdgcn_sched_barrier(0);                                   // (the scheduler would sink the read to its first use)
          pair_step(ty, cy, sy);
#pragma unroll
          for (int p = 0; p < PPL; ++p) { cx[p] = fma(two_cD[p], cy[p], -cx[p]); sx[p] = fma(two_cD[p], sy[p], -sx[p]); }
          load_blk(q + 2 * W, ty);
          __builtin_amdgcn_sched_barrier(0);
          pair_step(tx, cx, sx);
#pragma unroll
          for (int p = 0; p < PPL; ++p) { cy[p] = fma(two_cD[p], cx[p], -cy[p]); sy[p] = fma(two_cD[p], sx[p], -sy[p]); }
        }
        if (t < t1) pair_step(ty, cy, sy);
      } else {
#pragma unroll
        for (int p = 0; p < PPL; ++p)
#pragma unroll
          for (int c = 0; c < A; ++c) { P[p][c] = c0[c]; Q[p][c] = 0.0; }
      }
      double cb[PPL], sb[PPL];
#pragma unroll
      for (int p = 0; p < PPL; ++p) { cb[p] = rs.cm[p]; sb[p] = rs.sm[p]; }
      if (any_nc) {                                                            // (block-uniform; not VMEC's rows)
        const double nc = c0[9];
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          double sn, cn;
          geo_sincos(nc * phi[p], &sn, &cn);
          cb[p] = rs.cm[p] * cn + rs.sm[p] * sn; sb[p] = rs.sm[p] * cn - rs.cm[p] * sn;      // cos / sin(m tv - n_c phi)
        }
      }
#pragma unroll
      for (int p = 0; p < PPL; ++p) {
        GeoSums& X = S[p];
        const double c_ = cb[p], s_ = sb[p], mc = m_r * c_, ms = m_r * s_;
        // acc +- (x P + y Q): two multiply-adds per sum
        auto add = [&](double& acc, double x, double y, int c) { acc = fma(x, P[p][c], fma(y, Q[p][c], acc)); };
        if constexpr (!NYQ) {
          add(X.R, c_, s_, 0); add(X.R_s, c_, s_, 1); add(X.R_t, -ms, mc, 0); add(X.R_p, s_, -c_, 2);         // sum v cos = c P + s Q, sum v sin = s P - c Q
          add(X.Z_s, s_, -c_, 4); add(X.Z_t, mc, ms, 3); add(X.Z_p, -c_, -s_, 5);
          add(X.l_s, s_, -c_, 7); add(X.l_t, mc, ms, 6); add(X.l_p, -c_, -s_, 8);
        } else {
          add(X.sqg, c_, s_, 0); add(X.modB, c_, s_, 1); add(X.B_s, c_, s_, 2); add(X.B_t, -ms, mc, 1); add(X.B_p, s_, -c_, 3);
          add(X.Bsup_phi, c_, s_, 4); add(X.Bsub_s, s_, -c_, 5); add(X.Bsub_t, c_, s_, 6); add(X.Bsub_p, c_, s_, 7);
        }
      }
    }
  };
  if constexpr (LPP == 1) {
    run_sym(std::false_type{}, I + L.o_c0mn, I + L.o_symn, soff1, code1, ri1, nr1,
            __builtin_amdgcn_readfirstlane((int)I[16]) != 0);
  } else {
    run_set(std::false_type{}, I + L.o_amn + (size_t)sub * L.s_amn, goff1, code1, ri1, nr1);
  }
  GEO_PROBE_AT(5);
  if (a.dn_nyq != a.dn_mn) {
#pragma unroll
    for (int p = 0; p < PPL; ++p) { geo_sincos(a.dn_nyq * phi[p], &sD[p], &cD[p]); two_cD[p] = 2.0 * cD[p]; }
  }
  if constexpr (LPP == 1) {
    run_sym(std::true_type{}, I + L.o_c0nq, I + __builtin_amdgcn_readfirstlane((int)I[14]), soff2, code2, ri2, nr2,
            __builtin_amdgcn_readfirstlane((int)I[17]) != 0);
  } else {
    run_set(std::true_type{}, I + L.o_anq + (size_t)sub * L.s_anq, goff2, code2, ri2, nr2);
  }
  GEO_PROBE_AT(6);
  if constexpr (LPP > 1) {
    GeoSums& A = S[0];
    A.R = group_sum<LPP>(A.R); A.R_s = group_sum<LPP>(A.R_s); A.R_t = group_sum<LPP>(A.R_t); A.R_p = group_sum<LPP>(A.R_p);
    A.Z_s = group_sum<LPP>(A.Z_s); A.Z_t = group_sum<LPP>(A.Z_t); A.Z_p = group_sum<LPP>(A.Z_p);
    A.l_s = group_sum<LPP>(A.l_s); A.l_t = group_sum<LPP>(A.l_t); A.l_p = group_sum<LPP>(A.l_p);
    A.sqg = group_sum<LPP>(A.sqg); A.modB = group_sum<LPP>(A.modB); A.B_s = group_sum<LPP>(A.B_s); A.B_t = group_sum<LPP>(A.B_t);
    A.B_p = group_sum<LPP>(A.B_p); A.Bsup_phi = group_sum<LPP>(A.Bsup_phi); A.Bsub_s = group_sum<LPP>(A.Bsub_s);
    A.Bsub_t = group_sum<LPP>(A.Bsub_t); A.Bsub_p = group_sum<LPP>(A.Bsub_p);
  }
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    if (live[p] && sub == 0) {
      double sp, cp;
      geo_sincos(phi[p], &sp, &cp);
      geo_tail_h(a, I, line, j[p], phi[p], sp, cp, S[p]);
    }
  }
  GEO_PROBE_AT(7);
}

// Persistent blocks.  A UNIT is eight consecutive wave-items of ONE line (one per wave of the block), so a unit never
// straddles two surfaces; block b owns the units [b U / B, (b + 1) U / B), U = n_lines * units_per_line (line-major).  The
// table image of a surface is copied into LDS when a unit needs a surface other than the staged one (lines usually come
// sorted by surface: once per surface and block) -- the only block barriers of the kernel: between re-stagings the waves
// run through their items independently, so a SIMD always has both of its waves to pick instructions from (with a
// barrier per unit the older wave of each SIMD, which the issue arbiter favours, finished its item ~30 % earlier and
// left the younger one alone at half the issue rate).
template <int PPL, int LPP, int MAXR>
__global__ void __launch_bounds__(kGeoBlock) k_geo_rows(GeoArgs a, const double* __restrict__ img) {
  extern __shared__ __align__(16) unsigned char geo_smem[];
  double* I = reinterpret_cast<double*>(geo_smem);
  const GeoImgLayout L = geo_layout(a.mnmax, a.nrows_mn, a.mnmax_nyq, a.nrows_nyq, LPP);
  const int n_lines = a.n_lines_dev ? min(*a.n_lines_dev, a.n_lines) : a.n_lines;
  constexpr int PTS = 64 * PPL / LPP;
  constexpr int WPB = kGeoBlock / 64;
  const int ipl = (a.j_end + PTS - 1) / PTS;             // wave-items per line
  const int upl = (ipl + WPB - 1) / WPB;                 // units per line
  const long U = (long)n_lines * upl;
  const long u_begin = U * blockIdx.x / gridDim.x, u_end = U * (blockIdx.x + 1) / gridDim.x;
  const int wave = threadIdx.x >> 6;
  __shared__ int claim;                                  // next unclaimed wave-item of the segment
  int cur = -1;
  GEO_PROBE_AT(0);
  auto surf_of = [&](long u) { return min(max(__builtin_amdgcn_readfirstlane(a.line_surf[(int)(u / upl)]), 0), a.n_surf - 1); };   // (device-resident indices are not range-checked by the C ABI)
  for (long u = u_begin; u < u_end;) {
    // segment: the units from u on that lie on one surface (block-uniform)
    const int js = surf_of(u);
    long v = u + 1;
    while (v < u_end && surf_of(v) == js) ++v;
    __syncthreads();                                     // every wave is done with the staged image and with the claim counter
    if (threadIdx.x == 0) claim = 0;
    if (js != cur) {
      // straight copy of the prepared image, four 16-byte loads in flight per thread
      const double2* src = reinterpret_cast<const double2*>(img + (size_t)js * L.total);
      double2* dst = reinterpret_cast<double2*>(I);
      // (lengths are even; an LPP = 1 image is used up to its header slot 13)
      const int n2 = (LPP == 1 ? __builtin_amdgcn_readfirstlane((int)img[(size_t)js * L.total + 13]) : L.total) >> 1;
      for (int k = threadIdx.x; k < n2; k += 4 * kGeoBlock) {
        double2 v4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int kk = k + q * kGeoBlock; v4[q] = kk < n2 ? src[kk] : double2{0.0, 0.0}; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int kk = k + q * kGeoBlock; if (kk < n2) dst[kk] = v4[q]; }
      }
      cur = js;
    }
    __syncthreads();
    // The waves CLAIM their items: the older wave of each SIMD, which the issue arbiter favours whatever s_setprio says, gets
    // through an item in ~0.65 of its mate's time; with a fixed item per wave and unit it sat at the next re-staging barrier
    // (or had left the kernel) while the mate finished alone at the lone-wave issue rate.
    const int n_items = (int)(v - u) * WPB;
    // (LPP = 1) a surface whose rows are too long for the image (k_geo_prepare)
    const bool general = LPP == 1 && __builtin_amdgcn_readfirstlane((int)I[15]) < -1;
    for (;;) {
      int k = 0;
      if ((threadIdx.x & 63) == 0) k = atomicAdd(&claim, 1);
      k = __builtin_amdgcn_readfirstlane(k);
      if (k >= n_items) break;
      const long uu = u + k / WPB;
      const int line = (int)(uu / upl);
      const int item = (int)(uu - (long)line * upl) * WPB + (k % WPB);
      if (item < ipl) {
        GEO_PROBE_AT(1);
        if constexpr (LPP == 1) {
          if (general) {
            // (the host entry points send such tables to k_fieldline_geometry; only device-resident rows rewritten in place
            // behind the library's back get here: NaN, like every other invalid input -- never an out-of-bounds access)
            for (int jj = item * PTS + (threadIdx.x & 63); jj < min(item * PTS + PTS, a.j_end); jj += 64)
              for (int q = 0; q < 8; ++q) a.geo[q * a.plane + (size_t)line * a.ld + jj] = __builtin_nan("");
            continue;
          }
        }
        geo_item<PPL, LPP, MAXR>(a, I, L, line, item * PTS);
      }
    }
    u = v;
  }
#ifdef GEO_PROBE
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && wave < 8) geo_probe_buf[blockIdx.x * 16 + 8 + wave] = wall_clock64();   // when each wave is done
#endif
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
// grid point j of `line` by ONE WAVE, straight from the global tables (any mode ordering, any row lengths)
__device__ __forceinline__ void geo_point_wave(const GeoArgs& a, int line, int j) {
  const int lane = threadIdx.x & 63;
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
    for (int k = lane; k < a.mnmax; k += 64) {
      double sa, ca;
      geo_sincos(a.xm[k] * tv - a.xn[k] * phi, &sa, &ca);           // (|angle| stays far below the 1e5 the fast form covers)
      acc += lmns[k] * sa;
    }
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
    geo_sincos(m * tv - n * phi, &sa, &ca);
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
    geo_sincos(m * tv - n * phi, &sa, &ca);
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
__global__ void __launch_bounds__(256) k_fieldline_geometry_tail(GeoArgs a) {
  const int r = a.j_end - a.j_begin;
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= (long)(a.n_lines_dev ? min(*a.n_lines_dev, a.n_lines) : a.n_lines) * r) return;
  geo_point_wave(a, (int)(w / r), a.j_begin + (int)(w % r));
}

// dPdrho of each line: -0.5 mean((cvdrift - gbdrift) bmag^2)   (ball_scan.py:262)
__global__ void __launch_bounds__(256) k_line_dPdrho(int n_lines, int N, long ld, size_t plane, const double* geo, double* dPdrho) {
  __shared__ double part[4];
  const int line = blockIdx.x;
  const size_t o = (size_t)line * ld;
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

hipError_t launch_line_dPdrho(int n_lines, int N, long ld, size_t plane, const double* geo, double* dPdrho, hipStream_t st) {
  hipLaunchKernelGGL(k_line_dPdrho, dim3((unsigned)n_lines), dim3(256), 0, st, n_lines, N, ld, plane, geo, dPdrho);
  return hipGetLastError();
}

// Which form serves a batch of n_lines x N grid points on a chip of n_cu CUs: the persistent kernel has 8 n_cu wave slots
// (two waves per SIMD); the forms with more points per wave-item cost less per point (one lane per point: rows in n-symmetric
// form fed by DPP broadcasts), so the choice is the SMALLEST item that still gives no wave slot a second item
// (tools/geo_form_sweep.py: 30 / 54 / 84 / 201 lines of 969 points -> 4 / 2 / 1 lanes per point / two points per lane).
//   lpp_opt: 0 = by batch size, 1 | 2 | 4 | 8 = lanes per point, -2 = two points per lane.
GeoForm geo_pick_form(long n_lines, int N, int n_cu, int lpp_opt) {
  if (lpp_opt == -2) return GeoForm{2, 1};
  if (lpp_opt == 1 || lpp_opt == 2 || lpp_opt == 4 || lpp_opt == 8) return GeoForm{1, lpp_opt};
  const long pts = n_lines * (long)N, slots = 8L * n_cu;
  if (pts > slots * 64) return GeoForm{2, 1};
  if (pts > slots * 32) return GeoForm{1, 1};
  if (pts > slots * 16) return GeoForm{1, 2};
  if (pts > slots * 8) return GeoForm{1, 4};
  return GeoForm{1, 8};
}
bool geo_rows_usable(const GeoArgs& a, int lpp) {
  if (a.nrows_mn <= 0 || a.nrows_nyq <= 0 || a.nrows_mn > kGeoMaxRows || a.nrows_nyq > kGeoMaxRows) return false;
  if (a.nrows_mn > 24 || (a.nrows_mn > 12 && lpp != 1)) return false;      // (P_m, Q_m) live in registers: MAXR instantiations below
  return geo_image_doubles(a, lpp) * sizeof(double) <= 150 * 1024;
}

// The form for a.n_lines-like batches of THESE tables: an image in the one-lane-per-point layout is sized for the worst case (one
// pair per mode), so big mode sets (W7-X-like: ~400 + ~600 modes) do not fit the LDS there; they run with two (four, eight)
// lanes per point, whose per-lane lists are half (...) as long, before anything falls back to the one-sincos-per-mode kernel.
GeoForm geo_pick_usable(const GeoArgs& a, long n_lines, int N, int n_cu) {
  GeoForm f = geo_pick_form(n_lines, N, n_cu, a.lpp);
  while (f.lpp < 8 && !geo_rows_usable(a, f.lpp)) f = GeoForm{1, f.lpp * 2};
  return f;
}

// a.form (set by the caller through geo_pick_usable, or {0, 0} = pick here), a.img[geo_lpp_index(lpp)] = room for
// n_surf * geo_image_doubles(a, lpp) doubles, a.img_ready = bit per image already built from these tables (updated).
hipError_t launch_geometry(GeoArgs& a, hipStream_t st, int n_cu) {
  GeoForm f = a.form;
  if (f.ppl == 0) f = geo_pick_usable(a, a.n_lines, a.N, n_cu);
  const size_t plane = a.plane ? a.plane : (size_t)a.n_lines * a.ld;
  a.plane = plane;
  const int li = geo_lpp_index(f.lpp);
  if (geo_rows_usable(a, f.lpp) && a.img[li]) {
    if (!(a.img_ready & (1u << li))) {
      if (a.surf_used) {                                    // images of the surfaces this call's lines lie on only
        (void)hipMemsetAsync(a.surf_used, 0, (size_t)a.n_surf * sizeof(int), st);
        hipLaunchKernelGGL(k_geo_mark, dim3((a.n_lines + 255) / 256), dim3(256), 0, st, a.n_lines, a.n_surf, a.line_surf, a.surf_used);
      }
      hipLaunchKernelGGL(k_geo_prepare, dim3(a.n_surf), dim3(256), 0, st, a, f.lpp, a.img[li]);
      a.img_ready |= 1u << li;
    }
    const size_t lds = geo_image_doubles(a, f.lpp) * sizeof(double);
    const int pts = 64 * f.ppl / f.lpp;
    GeoArgs b = a;
    const int rem = a.N % pts;
    // the few points a line has beyond a multiple of the wave-item (N = 2^k + 1: one) go to the one-point-per-wave
    // kernel instead of a wave-item of their own
    b.j_begin = 0;
    b.j_end = (f.lpp == 1 && rem > 0 && rem <= 16 && a.N > pts) ? a.N - rem : a.N;
    const long units = (long)a.n_lines * (((b.j_end + pts - 1) / pts + 7) / 8);     // eight wave-items of one line each
    long nblk = units;
    if (nblk > n_cu) nblk = n_cu;
    if (nblk < 1) nblk = 1;
    auto go = [&](auto kern, int ppl, int lpp, int maxr) {
      hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e1 != hipSuccess) return e1;
      hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kGeoBlock), lds, st, b, (const double*)a.img[li]);
      note_launch(nblk, kGeoBlock, "ibs::k_geo_rows<%d, %d, %d>", ppl, lpp, maxr);
      return hipSuccess;
    };
    hipError_t e2;
    if (a.nrows_mn > 12) e2 = go(k_geo_rows<1, 1, 24>, 1, 1, 24);
    else if (f.ppl == 2) e2 = go(k_geo_rows<2, 1, 12>, 2, 1, 12);
    else if (f.lpp == 1) e2 = go(k_geo_rows<1, 1, 12>, 1, 1, 12);
    else if (f.lpp == 2) e2 = go(k_geo_rows<1, 2, 12>, 1, 2, 12);
    else if (f.lpp == 4) e2 = go(k_geo_rows<1, 4, 12>, 1, 4, 12);
    else e2 = go(k_geo_rows<1, 8, 12>, 1, 8, 12);
    if (e2 != hipSuccess) return e2;
    if (b.j_end < a.N) {
      GeoArgs c = a;
      c.j_begin = b.j_end; c.j_end = a.N;
      const long waves = (long)a.n_lines * (c.j_end - c.j_begin);
      hipLaunchKernelGGL(k_fieldline_geometry_tail, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, c);
    }
  } else {
    hipLaunchKernelGGL(k_fieldline_geometry, dim3((a.N + 255) / 256, a.n_lines), dim3(256), 0, st, a);
    note_launch((long)((a.N + 255) / 256) * a.n_lines, 256, "ibs::k_fieldline_geometry");
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.dPdrho) {
    hipLaunchKernelGGL(k_line_dPdrho, dim3(a.n_lines), dim3(256), 0, st, a.n_lines, a.N, a.ld, plane, a.geo, a.dPdrho);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace ibs

#ifdef GEO_PROBE
extern "C" int ibs_geo_probe_clear() {
  static long long zeros[256 * 16];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(ibs::geo_probe_buf), zeros, sizeof(zeros));
}
extern "C" int ibs_geo_probe_read(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ibs::geo_probe_buf), sizeof(long long) * 256 * 16);
}
#endif
