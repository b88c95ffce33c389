// Wave-per-field-line ideal-ballooning eigen-solver core for gfx950 (CDNA4, wave64).
//
// What it computes (reference: /root/reference/utils.py:1550-1624 gamma_ball_full):
//   the largest eigenvalue lam and eigenvector x of the tridiagonal pencil T x = lam F x built
//   from (g, c, f) on N grid points (utils.py:1574-1592), then the reference's growth rate
//   gam = Simpson(-g dX^2 + c X^2) / Simpson(f X^2) with X = x/max|x| and the 2nd/4th-order
//   finite-difference dX (utils.py:1601-1621).
//
// How (MI355X-first, not the reference's dense LU + ARPACK):
//   * one 64-lane wavefront owns one system; lane L keeps a contiguous chunk of <= M rows in
//     registers in a symmetric diagonal scaling x = s z that makes every in-chunk off-diagonal
//     exactly 1, so the three-term recurrence is ONE fma per row:  z[i+1] = -(D[i]-sig*Ph[i]) z[i] - z[i-1]
//   * a sweep at shift sig = per-lane 2x2 transfer matrices -> wave-wide Kogge-Stone scan of
//     2x2 products (DPP row_shr / row_bcast, power-of-two renormalised, exponent carried as int)
//     -> per-lane replay from the incoming vector.  A forward sweep yields the Sturm count (sign changes of
//     the forward solution = eigenvalues above sig) and the shooting value u_{n+1}(sig) (the characteristic
//     polynomial up to a constant, mantissa + exponent).
//   * the shift iteration (solve()) brackets lam_max with counts only and proposes shifts by Brent-style
//     Muller/secant interpolation of the shooting value once the bracket has isolated lam_max; it needs no
//     start vector and cannot land on a wrong eigenvalue.
//   * ONE backward sweep (from the right end) at the converged shift gives the second solution w; u and w give
//     the twisted (double-sweep) eigenvector  x = u/u_k (r<=k), w/w_k (r>=k)  with k = argmax |u_r w_r|
//     (discrete Wronskian: gamma_r = W/(u_r w_r)) and the Rayleigh-quotient polish rho = sig + gamma_k / sum(f x^2).
//   * no MFMA: there is no dense contraction here; the binding resource is FP64 VALU issue.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace ibs {

constexpr int kWave = 64;

#ifdef IBS_TRACE_ALL
__device__ double ibs_trace_all[404];
#endif
// phase timestamps for tools/phase_probe.hip (debug builds only; the library is built without IBS_PROBE)
#ifdef IBS_PROBE
__device__ long long ibs_probe_buf[16 * 4096];
#define IBS_PROBE_AT(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) ibs_probe_buf[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define IBS_PROBE_AT(k) do {} while (0)
#endif



// ---------------------------------------------------------------- scalar helpers (f64 / f32)
__device__ __forceinline__ int fexp(double x) { return __builtin_amdgcn_frexp_exp(x); }
__device__ __forceinline__ int fexp(float x) { return __builtin_amdgcn_frexp_expf(x); }
__device__ __forceinline__ double xldexp(double x, int e) { return __builtin_ldexp(x, e); }
__device__ __forceinline__ float xldexp(float x, int e) { return __builtin_ldexpf(x, e); }
__device__ __forceinline__ double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double xabs(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float xabs(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double xmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float xmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double xmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float xmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ bool signbit_of(double a) { return __double2hiint(a) < 0; }
__device__ __forceinline__ bool signbit_of(float a) { return __float_as_int(a) < 0; }
// sign bits differ: one xor of the high words + one 32-bit compare (the empty asm keeps the compiler from widening
// the test to a 64-bit xor + 64-bit compare)
__device__ __forceinline__ bool sign_differs(double a, double b) {
  int x = __double2hiint(a) ^ __double2hiint(b);
  asm("" : "+v"(x));
  return x < 0;
}
__device__ __forceinline__ bool sign_differs(float a, float b) { return (__float_as_int(a) ^ __float_as_int(b)) < 0; }
// the 32-bit word that carries the sign bit (bit 31)
__device__ __forceinline__ unsigned sign_word(double a) { return (unsigned)__double2hiint(a); }
__device__ __forceinline__ unsigned sign_word(float a) { return (unsigned)__float_as_int(a); }
__device__ __forceinline__ bool finite_of(double a) { return xabs(a) <= 1.7976931348623157e308; }
__device__ __forceinline__ bool finite_of(float a) { return xabs(a) <= 3.4028234e38f; }

// reciprocal for well-scaled operands (no denormal / overflow handling): hardware seed + two Newton steps.
// Used only where the result steers the iteration (Newton step length, vector normalisation), never in the
// certified quantities.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}

// cheaper variants for quantities that only PROPOSE a shift (relative error ~1e-8 is harmless there):
// hardware seed + one Newton step
__device__ __forceinline__ double approx_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float approx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double approx_sqrt(double x) {      // x >= 0, well scaled
  const double y = __builtin_amdgcn_rsq(x);
  const double s = x * y;
  return x > 0.0 ? __builtin_fma(0.5 * y, __builtin_fma(-s, s, x), s) : 0.0;
}
__device__ __forceinline__ float approx_sqrt(float x) { return __builtin_sqrtf(x); }

// sin and cos of an angle in [0, pi] (the trial vector of WaveSolver::setup): Cody-Waite reduction by pi/2 and the fdlibm
// kernels, ~1 ulp; the float form uses the fast hardware functions (the trial vector only steers the iteration)
__device__ __forceinline__ void trial_sincos(double x, double& sn, double& cs) {
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-k, 6.07710050650619224932e-11, r);
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06); ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03); ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double s_ = __builtin_fma(z * r, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07); pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03); pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double c_ = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const int q = (int)k;
  const double s1 = (q & 1) ? c_ : s_, c1 = (q & 1) ? s_ : c_;
  sn = (q & 2) ? -s1 : s1;
  cs = ((q + 1) & 2) ? -c1 : c1;
}
__device__ __forceinline__ void trial_sincos(float x, float& sn, float& cs) { sn = __sinf(x); cs = __cosf(x); }

// integer-built exponent helpers (v_frexp_exp / v_ldexp are slow-rate FP64 instructions):
// expo_of: e with |x| in [2^e, 2^(e+1)) for normal x, a large negative number for 0 / denormals
__device__ __forceinline__ int expo_of(double x) { const int eb = (__double2hiint(x) >> 20) & 0x7ff; return eb ? eb - 1023 : -(1 << 28); }
__device__ __forceinline__ int expo_of(float x) { const int eb = (__float_as_int(x) >> 23) & 0xff; return eb ? eb - 127 : -(1 << 28); }
// pow2_of<T>(k) = 2^k for k within the normal range (callers clamp)
template <typename T> __device__ __forceinline__ T pow2_of(int k);
template <> __device__ __forceinline__ double pow2_of<double>(int k) { return __hiloint2double((k + 1023) << 20, 0); }
template <> __device__ __forceinline__ float pow2_of<float>(int k) { return __int_as_float((k + 127) << 23); }
template <typename T> struct ExpLim;
template <> struct ExpLim<double> { static constexpr int v = 1000; };
template <> struct ExpLim<float> { static constexpr int v = 120; };

template <typename T> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; };

// ---------------------------------------------------------------- cross-lane moves
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xF, false);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_t(double old, double src) {
  int lo = dpp_i<CTRL, ROWMASK>(__double2loint(old), __double2loint(src));
  int hi = dpp_i<CTRL, ROWMASK>(__double2hiint(old), __double2hiint(src));
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_t(float old, float src) {
  return __int_as_float(dpp_i<CTRL, ROWMASK>(__float_as_int(old), __float_as_int(src)));
}

__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double readlane_t(double v, int l) {
  return __hiloint2double(readlane_i(__double2hiint(v), l), readlane_i(__double2loint(v), l));
}
__device__ __forceinline__ float readlane_t(float v, int l) { return __int_as_float(readlane_i(__float_as_int(v), l)); }

// fetch with zero fill (bound_ctrl): a lane whose source lies outside its row reads 0; with all rows enabled no previous value
// has to be set up in the destination (the form with an `old` operand costs a v_mov per dword and step)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dppz_i(int src) { return __builtin_amdgcn_update_dpp(0, src, CTRL, ROWMASK, 0xF, true); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dppz_t(double src) {
  return __hiloint2double(dppz_i<CTRL, ROWMASK>(__double2hiint(src)), dppz_i<CTRL, ROWMASK>(__double2loint(src)));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dppz_t(float src) { return __int_as_float(dppz_i<CTRL, ROWMASK>(__float_as_int(src))); }

// wave-wide reductions on the VALU (DPP row_shr / row_bcast: no LDS round trips); the result is
// read from lane 63 into SGPRs, i.e. it is wave-uniform.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  v += dppz_t<0x111, 0xF>(v);
  v += dppz_t<0x112, 0xF>(v);
  v += dppz_t<0x114, 0xF>(v);
  v += dppz_t<0x118, 0xF>(v);
  v += dpp_t<0x142, 0xA>(T(0), v);
  v += dpp_t<0x143, 0xC>(T(0), v);
  return readlane_t(v, 63);
}
__device__ __forceinline__ int wave_sum_i(int v) {
  v += dppz_i<0x111, 0xF>(v);
  v += dppz_i<0x112, 0xF>(v);
  v += dppz_i<0x114, 0xF>(v);
  v += dppz_i<0x118, 0xF>(v);
  v += dpp_i<0x142, 0xA>(0, v);
  v += dpp_i<0x143, 0xC>(0, v);
  return readlane_i(v, 63);
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {  // identity for missing lanes: the value itself
  v = xmax(v, dpp_t<0x111, 0xF>(v, v));
  v = xmax(v, dpp_t<0x112, 0xF>(v, v));
  v = xmax(v, dpp_t<0x114, 0xF>(v, v));
  v = xmax(v, dpp_t<0x118, 0xF>(v, v));
  v = xmax(v, dpp_t<0x142, 0xA>(v, v));
  v = xmax(v, dpp_t<0x143, 0xC>(v, v));
  return readlane_t(v, 63);
}
template <typename T>
__device__ __forceinline__ T uniform(T v) {  // value is already identical in all lanes: make it scalar
  return readlane_t(v, 0);
}

// ---------------------------------------------------------------- 2x2 transfer matrices with exponent
template <typename T>
struct M2 {
  T a, b, c, d;  // [[a b],[c d]]
  int e;         // true matrix = this * 2^e
};

// Power-of-two renormalisation to max-entry in [0.5, 1).  The exponent is taken from the bits of the largest
// entry and applied as a MULTIPLY by an exactly representable 2^-ex (integer-built): v_ldexp_f64 /
// v_frexp_exp are not full-rate FP64 instructions, v_mul_f64 is.
__device__ __forceinline__ double pow2_scale_of(double m, int& ex_out) {
  const int eb = (__double2hiint(m) >> 20) & 0x7ff;        // biased exponent of m >= 0 (0 for m = 0)
  ex_out = eb - 1022;                                      // m = mant * 2^ex_out, mant in [0.5, 1)
  return __hiloint2double((2045 - eb) << 20, 0);           // 2^-(ex_out)
}
__device__ __forceinline__ float pow2_scale_of(float m, int& ex_out) {
  const int eb = (__float_as_int(m) >> 23) & 0xff;
  ex_out = eb - 126;
  return __int_as_float((253 - eb) << 23);
}
template <typename T>
__device__ __forceinline__ void renorm(M2<T>& R) {
  const T m = xmax(xmax(xabs(R.a), xabs(R.b)), xmax(xabs(R.c), xabs(R.d)));
  int ex;
  const T sc = pow2_scale_of(m, ex);
  R.a *= sc; R.b *= sc; R.c *= sc; R.d *= sc;
  R.e += ex;
}
// A <- A * B (A applied after B), in place.  NORM: renormalise to max-entry in [0.5, 1).
// Written so that each row needs one temporary only (the scan steps run under an exec mask and every
// extra temporary costs a masked register copy).
template <typename T, bool NORM>
__device__ __forceinline__ void mul_inplace(M2<T>& A, const M2<T>& B) {
  const T na = xfma(A.a, B.a, A.b * B.c);
  A.b = xfma(A.a, B.b, A.b * B.d);
  A.a = na;
  const T nc = xfma(A.c, B.a, A.d * B.c);
  A.d = xfma(A.c, B.b, A.d * B.d);
  A.c = nc;
  A.e += B.e;
  if (NORM) renorm(A);
}
template <typename T, bool NORM>
__device__ __forceinline__ M2<T> mul(const M2<T>& A, const M2<T>& B) {
  M2<T> R = A;
  mul_inplace<T, NORM>(R, B);
  return R;
}
// neighbour fetch for the scans: lanes without a valid source receive garbage/zero and are masked
// out by the caller (the product is applied under `if (valid)`), so no identity has to be built.
template <typename T, int CTRL, int ROWMASK>
__device__ __forceinline__ M2<T> dpp_fetch(const M2<T>& s) {
  M2<T> r;
  r.a = dppz_t<CTRL, ROWMASK>(s.a); r.b = dppz_t<CTRL, ROWMASK>(s.b);
  r.c = dppz_t<CTRL, ROWMASK>(s.c); r.d = dppz_t<CTRL, ROWMASK>(s.d);
  r.e = dppz_i<CTRL, ROWMASK>(s.e);
  return r;
}
template <typename T>
__device__ __forceinline__ M2<T> lane_bcast(const M2<T>& s, int src_lane) {  // src_lane is a constant
  M2<T> r;
  r.a = readlane_t(s.a, src_lane); r.b = readlane_t(s.b, src_lane);
  r.c = readlane_t(s.c, src_lane); r.d = readlane_t(s.d, src_lane);
  r.e = readlane_i(s.e, src_lane);
  return r;
}

// inclusive prefix product over lanes:  P_L = A_L * A_{L-1} * ... * A_0.
// Entries stay in range without renormalising every step: inputs have max-entry < 1, a product of two such matrices
// is < 2, of four < 8; renormalise every second step.  (Dropping the renormalisation for FP64 altogether -- the upper
// bound of a 64-fold product is 2^63 -- was tried in round 2: sweeps per solve went from 15.3 to 18.9 and the count
// certificates of the rough config-5 systems failed; a product of normalised matrices has no LOWER bound.)
// Only the FIRST COLUMN (a, c) and the exponent of the result are defined on return: every consumer takes the prefix as
// "the solution that starts with (1, 0) at the left end" (incoming pair of the next lane, shooting value), so the last
// step -- whose fetched factor is already a complete prefix -- fetches and forms the first column only.
template <typename T>
struct ScanNorm { static constexpr bool v = true; };
template <typename T>
__device__ __forceinline__ M2<T> scan_fwd(M2<T> P, int lane) {
  constexpr bool RN = ScanNorm<T>::v;
  const int l16 = lane & 15, row = lane >> 4;
  { const M2<T> F = dpp_fetch<T, 0x111, 0xF>(P); if (l16 >= 1) mul_inplace<T, false>(P, F); }   // row_shr:1
  { const M2<T> F = dpp_fetch<T, 0x112, 0xF>(P); if (l16 >= 2) mul_inplace<T, RN>(P, F); }      // row_shr:2
  { const M2<T> F = dpp_fetch<T, 0x114, 0xF>(P); if (l16 >= 4) mul_inplace<T, false>(P, F); }   // row_shr:4
  { const M2<T> F = dpp_fetch<T, 0x118, 0xF>(P); if (l16 >= 8) mul_inplace<T, RN>(P, F); }      // row_shr:8
  { const M2<T> F = dpp_fetch<T, 0x142, 0xA>(P); if (row & 1) mul_inplace<T, false>(P, F); }    // row_bcast:15 -> rows 1,3
  {                                                                                             // row_bcast:31 -> rows 2,3
    const T Fa = dppz_t<0x143, 0xC>(P.a), Fc = dppz_t<0x143, 0xC>(P.c);
    const int Fe = dppz_i<0x143, 0xC>(P.e);
    if (row >= 2) {
      const T na = xfma(P.a, Fa, P.b * Fc);
      P.c = xfma(P.c, Fa, P.d * Fc);
      P.a = na;
      P.e += Fe;
      if (RN) {                            // (first column only)
        const T m = xmax(xabs(P.a), xabs(P.c));
        int ex;
        const T sc = pow2_scale_of(m, ex);
        P.a *= sc; P.c *= sc; P.e += ex;
      }
    }
  }
  return P;
}
// inclusive suffix product over lanes:  Q_L = B_L * B_{L+1} * ... * B_63
template <typename T>
__device__ __forceinline__ M2<T> scan_bwd(M2<T> Q, int lane) {
  const int l16 = lane & 15, row = lane >> 4;
  { const M2<T> F = dpp_fetch<T, 0x101, 0xF>(Q); if (l16 < 15) mul_inplace<T, false>(Q, F); }   // row_shl:1
  { const M2<T> F = dpp_fetch<T, 0x102, 0xF>(Q); if (l16 < 14) mul_inplace<T, true>(Q, F); }    // row_shl:2
  { const M2<T> F = dpp_fetch<T, 0x104, 0xF>(Q); if (l16 < 12) mul_inplace<T, false>(Q, F); }   // row_shl:4
  { const M2<T> F = dpp_fetch<T, 0x108, 0xF>(Q); if (l16 < 8) mul_inplace<T, true>(Q, F); }     // row_shl:8
  // rows 0,2 <- first lane of rows 1,3 (their row totals); then rows 0,1 <- lane 32 (rows 2+3)
  {
    const M2<T> t16 = lane_bcast(Q, 16), t48 = lane_bcast(Q, 48);
    M2<T> F;
    F.a = row == 0 ? t16.a : t48.a; F.b = row == 0 ? t16.b : t48.b;
    F.c = row == 0 ? t16.c : t48.c; F.d = row == 0 ? t16.d : t48.d; F.e = row == 0 ? t16.e : t48.e;
    if ((row & 1) == 0) mul_inplace<T, false>(Q, F);
  }
  { const M2<T> F = lane_bcast(Q, 32); if (row < 2) mul_inplace<T, true>(Q, F); }
  return Q;
}

// ---------------------------------------------------------------- division-form re-close (round 6)
// The matrix whose top eigenvalue is meant is utils.py:1584-1597; the recurrence is SURVEY.md Appendix A's / LAPACK dstebz's
//   q_0 = d_0 - sig f_1,   q_r = (d_r - sig f_{r+1}) - e_r^2 / q_{r-1},   |q| < pivmin -> -pivmin,
// #(q_r > 0) = eigenvalues of (T, F) above sig.  Every step is one rounded operation on the ORIGINAL entries, so the count is exact
// for a pencil whose entries differ by a few ulp each: its error in sig is a few eps ||A||, independent of N (the prefix-product
// sweeps of WaveSolver are exact for a perturbation that can reach ~N^2 eps ||A||).  It is serial in the rows, so the wave runs it
// for 64 SHIFTS at once: lane L walks ALL rows of the one system at its own shift (the loads are wave-uniform: one request each).
// Src: g(j), c(j), f(j) [and gh(k)] for grid point j of that system, the same for every lane.
template <typename T, class Src>
__device__ __forceinline__ int count_above_div(const Src& src, int N, T ih2, T sig) {
  constexpr T pivmin = T(2.2250738585072014e-292);      // DBL_MIN * 1e16
  const int n = N - 2;
  T gcur = T(0), e_lo;
  if constexpr (Src::kHasGh) e_lo = src.gh(0) * ih2;
  else { const T g0 = src.g(0); gcur = src.g(1); e_lo = T(0.5) * (g0 + gcur) * ih2; }
  T q = T(1);
  int cnt = 0;
#pragma unroll 4
  for (int r = 0; r < n; ++r) {          // (unrolled: the loads of the next rows are in flight while the divisions of these retire)
    const int j = r + 1;
    T e_hi;
    if constexpr (Src::kHasGh) e_hi = src.gh(j) * ih2;
    else { const T gnext = src.g(j + 1); e_hi = T(0.5) * (gcur + gnext) * ih2; gcur = gnext; }
    const T d = src.c(j) - (e_lo + e_hi);                // the diagonal exactly as WaveSolver::setup forms it (utils.py:1584-1592)
    const T a = xfma(-sig, src.f(j), d);
    q = r == 0 ? a : a - (e_lo * e_lo) / q;
    q = xabs(q) < pivmin ? -pivmin : q;
    cnt += q > T(0) ? 1 : 0;
    e_lo = e_hi;
  }
  return cnt;
}

// The same count from rows prepared once in LDS (k_fix_gcf): d_r, e_r^2 (e_0^2 unused), f_r for r = 0 .. n-1, read by all lanes at the
// same address (broadcast).  The division is the hardware reciprocal + two Newton steps (a few ulp: the count is then exact for a
// pencil a few more ulp away): the recurrence is one dependent chain per lane, ~55 clocks per row instead of ~95 with the IEEE
// division, and nothing waits for global memory.
template <typename T>
__device__ __forceinline__ int count_above_rows(const T* d, const T* e2, const T* f, int n, T sig) {
  constexpr T pivmin = T(2.2250738585072014e-292);
  T q = xfma(-sig, f[0], d[0]);
  q = xabs(q) < pivmin ? -pivmin : q;
  int cnt = q > T(0) ? 1 : 0;
#pragma unroll 8
  for (int r = 1; r < n; ++r) {
    const T a = xfma(-sig, f[r], d[r]);
    q = xfma(-e2[r], fast_rcp(q), a);
    q = xabs(q) < pivmin ? -pivmin : q;
    cnt += q > T(0) ? 1 : 0;
  }
  return cnt;
}

// lam_max by 64-way multisection on a division-form count `count(sig)`, from the bracket [lo, hi]: each pass runs 64 shifts (lane 0
// at lo, lane 63 at hi) and keeps the interval between the highest shift with an eigenvalue above it and the next one; a bracket
// that does not hold lam_max (its end shifts say so) is moved and widened 64-fold instead.  Ends at width <= stop_eps eps ||A||.
// All 64 lanes of the wave take part; the result is wave-uniform.
// passes: sweeps used (each = n dependent divisions per lane);  returns false if 24 passes did not close (non-finite data).
template <typename T, class CountF>
__device__ __forceinline__ bool multisect(CountF&& count, T lo, T hi, T normA, T stop_eps, int lane, T& lam, int& passes) {
  const T epsA = Eps<T>::v * normA;
  bool ok = false;
  passes = 0;
  while (passes < 24) {
    ++passes;
    const T w = hi - lo;
    const T sig = lane == kWave - 1 ? hi : xfma(T(lane) * T(1.0 / 63.0), w, lo);
    const int cnt = count(sig);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(cnt >= 1);
    if (!(m & 1ull)) { hi = lo; lo = lo - T(64) * w; continue; }          // lam_max < lo
    const int top = 63 - __builtin_clzll(m);                              // the highest shift with an eigenvalue above it
    if (top == kWave - 1) { lo = hi; hi = hi + T(64) * w; continue; }     // lam_max >= hi
    lo = readlane_t(sig, top); hi = readlane_t(sig, top + 1);
    if (!(hi - lo > stop_eps * epsA)) { ok = true; break; }
  }
  lam = T(0.5) * (lo + hi);
  return ok;
}
// ... on the rows of `src` in memory, to 2 eps ||A||
template <typename T, class Src>
__device__ __forceinline__ bool multisect_division(const Src& src, int N, T h, T lo, T hi, T normA, int lane, T& lam, int& passes) {
  const T ih2 = T(1) / (h * h);
  return multisect<T>([&](T sig) { return count_above_div<T, Src>(src, N, ih2, sig); }, lo, hi, normA, T(2), lane, lam, passes);
}
// ... from a bracket of half width 2^11 eps ||A|| about `center` (the re-close of a suspect solve about its Rayleigh polish, which
// was within 100 eps ||A|| of lam_max on every suspect of the 10 x 2^20-system campaign: two passes take a valid bracket to
// 1.03 eps ||A||; a polish further off costs the passes that move the bracket)
template <typename T, class Src>
__device__ __forceinline__ bool reclose_division(const Src& src, int N, T h, T center, T normA, int lane, T& lam, int& passes) {
  const T epsA = Eps<T>::v * normA;
  return multisect_division<T, Src>(src, N, h, center - T(2048) * epsA, center + T(2048) * epsA, normA, lane, lam, passes);
}

// ---------------------------------------------------------------- per-wave solver state
struct SolveInfo {
  int iters;    // fused double sweeps used
  int status;   // 0 ok, bit0 = iteration cap hit, bit1 = non-finite data / non-positive g or f
};

template <typename T, int M>
struct WaveSolver {
  // chunk of this lane, symmetric scaling (see header comment).  Register budget: during the shift iteration only D and
  // Ph (4M VGPRs) are live; the backward solution zw is stored once, after convergence; the forward solution is never
  // stored but replayed from its incoming pair (M fma per replay), and the diagonal scaling s is rebuilt from g when the
  // eigenvector is assembled -- so the peak is 6M + temporaries instead of 10M (256 VGPRs + scratch at M = 16 before).
  T D[M], Ph[M];
  T kap, ikap;
  bool has_last;   // this lane owns M rows (else M-1)
  int lane;
  // sweep products
  T zw[M];
  T u0_in, zu_m1;  // (z_0, z_{-1}) of the forward solution: the pair the lane's rows are replayed from
  T zw_p1;         // z_{cnt} of the backward solution (local scaling)
  int Eu, Ew;      // power-of-two exponents of this lane's forward / backward solutions
  T sig_vec;       // shift of the last twisted() call (assemble() replays the forward solution at it)
  // bounds
  T lo, hi, normA;
  // Rayleigh quotient and residual bound of the trial vector x_j = sin(pi j / (N - 1)) (setup<Src, true>): lam_max >= rho,
  // and some eigenvalue lies within del of rho
  T trial_rho, trial_del;
  T trial_mrg;     // allowance for the rounding of rho's sums (N terms): (8 + N / 2) eps |A|
  // shooting value of the last forward sweep (mantissa-like, power-of-two exponent)
  T shoot_m; int shoot_e;

  static constexpr int kSetupRows = 4;
  // rows of this lane: [start, start+cnt)
  __device__ __forceinline__ static int rows_start(int lane, int n) {
    const int rem = n - kWave * (M - 1);
    return lane * (M - 1) + (lane < rem ? lane : rem);
  }

  // Src provides g(j), c(j), f(j) for grid point j in [0, N) (LDS-backed)
  // TRIAL: also forms the Rayleigh quotient rho = x'Tx / x'Fx and the residual bound del = |Tx - rho Fx|_{F^-1} / |x|_F of the
  // fundamental mode of the grid, x_j = sin(pi j / (N - 1)) (zero at both ends like the eigenfunctions; = cos(theta / 8) on the
  // reference's [-4 pi, 4 pi] grids, the long-wavelength factor of its own start vector, ball_scan.py:209).  rho is a
  // rigorous lower bound of lam_max (Courant-Fischer) that needs no count, and [rho, rho + del] is where trial_guess() lets
  // the shift iteration start: on the NCSX lines rho sits 2-4 eigenvalue gaps below lam_max where the Gershgorin /
  // diagonal-quotient bracket is ~500 gaps wide (10.8 instead of 15.3 forward sweeps per solve, 14 instead of 19 at the
  // worst; numpy model of the iteration in tools/sim_solve.py).  ~0.6 of a sweep: one fma recurrence for x, 7 flops per row, three
  // wave sums.
  template <class Src, bool TRIAL = false>
  __device__ __forceinline__ bool setup(const Src& src, int N, T h) {
    lane = threadIdx.x & (kWave - 1);
    const int n = N - 2;
    const int rem = n - kWave * (M - 1);
    has_last = lane < rem;
    const int a = rows_start(lane, n);
    const T ih2 = T(1) / (h * h);
    T sc = T(1);
    // half-grid g between grid points k and k+1 (utils.py:1574-1576): the mean of the neighbours on a uniform
    // grid, or a caller-supplied array when the input grid was regridded (Src::kHasGh)
    T gcur = T(0);
    T e_lo;
    bool bad = false;                            // non-finite data, g <= 0 or f <= 0 anywhere in this lane's rows
    if constexpr (Src::kHasGh) {
      e_lo = src.gh(a) * ih2;
      bad = !(src.g(a) > T(0));
    } else {
      const T g0 = src.g(a);
      gcur = src.g(a + 1);
      e_lo = T(0.5) * (g0 + gcur) * ih2;  // e_a
      bad = !(g0 > T(0)) || !(gcur > T(0));
    }
    T vhi = -T(1e300), vlo = -T(1e300), vna = T(0), sum_c = T(0), sum_f = T(0);
    const T e_first = e_lo;
    T ts_prev = T(0), ts_cur = T(0), two_cd = T(0), tA = T(0), tB = T(0), tC = T(0);
    if constexpr (TRIAL) {
      const T dl = T(3.14159265358979323846) / T(N - 1);
      T s0, c0, sd, cd;
      trial_sincos(T(a) * dl, s0, c0);                 // grid point a = the left neighbour of this lane's first row
      trial_sincos(dl, sd, cd);
      ts_prev = s0; ts_cur = xfma(s0, cd, c0 * sd); two_cd = T(2) * cd;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const bool act = (i < M - 1) || has_last;
      if (act) {
        const int j = a + i + 1;  // grid point of row a+i
        T gnext = T(0), e_hi;                           // e_{a+i+1}
        if constexpr (Src::kHasGh) { e_hi = src.gh(j) * ih2; bad = bad || !(src.g(j + 1) > T(0)); }
        else { gnext = src.g(j + 1); e_hi = T(0.5) * (gcur + gnext) * ih2; bad = bad || !(gnext > T(0)); }
        const T cj = src.c(j), fj = src.f(j);
        const T d = cj - (e_lo + e_hi);
        const T s2 = sc * sc;
        D[i] = d * s2; Ph[i] = fj * s2;
        const T rf = fast_rcp(fj);          // bounds only (margins added below)
        if constexpr (TRIAL) {
          const T ts_next = xfma(two_cd, ts_cur, -ts_prev);
          const T Tx = xfma(e_lo, ts_prev, xfma(d, ts_cur, e_hi * ts_next));
          tA = xfma(ts_cur, Tx, tA); tB = xfma(fj * ts_cur, ts_cur, tB); tC = xfma(Tx * rf, Tx, tC);
          ts_prev = ts_cur; ts_cur = ts_next;
        }
        vhi = xmax(vhi, cj * rf);
        vlo = xmax(vlo, d * rf);
        vna = xmax(vna, (xabs(d) + e_lo + e_hi) * rf);
        sum_c += cj; sum_f += fj;
        bad = bad || !(fj > T(0)) || !(e_hi > T(0)) || !finite_of(cj);
        sc = fast_rcp(e_hi * sc);           // e s_i s_{i+1} = 1 to rounding (two Newton steps on the hardware seed)
        gcur = gnext; e_lo = e_hi;
      } else {
        D[i] = T(0); Ph[i] = T(0);
      }
      // the rows hang on one dependent chain (sc), so the scheduler would put EVERY row's LDS reads in flight first
      // (3 to 7 values per row: > 400 VGPRs at M = 32); at most kSetupRows rows at a time
      if constexpr (M > 8) { if ((i % kSetupRows) == kSetupRows - 1) __builtin_amdgcn_sched_barrier(0); }
    }
    IBS_PROBE_AT(8);
    kap = sc; ikap = fast_rcp(sc);
    bad = bad || !(e_first > T(0));
    // wave-level bounds:  lam_max <= max c/f (Gershgorin);  lam_max >= any Rayleigh quotient
    const T e0 = readlane_t(e_first, 0);
    const T en = readlane_t(e_lo, kWave - 1);
    const T sc_all = wave_sum(sum_c), sf_all = wave_sum(sum_f);
    normA = uniform(wave_max(vna));
    hi = uniform(wave_max(vhi));
    lo = uniform(xmax(wave_max(vlo), (sc_all - e0 - en) / sf_all));
    hi += T(8) * Eps<T>::v * normA;
    lo -= T(8) * Eps<T>::v * normA;
    if constexpr (TRIAL) {
      const T A = wave_sum(tA), B = wave_sum(tB), C = wave_sum(tC);
      const T rho = A / B;
      trial_rho = rho;
      trial_del = approx_sqrt(xmax(xfma(-rho, A, C), T(0)) / B);       // |r|^2 = x'T F^-1 T x - rho x'Tx  (rho = x'Tx / x'Fx)
    } else {
      trial_rho = T(0); trial_del = T(-1);
    }
    trial_mrg = T(8 + N / 2) * Eps<T>::v * normA;
    chk_slack = T(N < 256 ? 2 * N : 512) * Eps<T>::v * normA;        // min(8 tol, 2 N eps ||A||), tol = 64 eps ||A||  (solve<true>)
    IBS_PROBE_AT(9);
    return __any(bad) != 0;
  }

  // Start of a cold solve from the trial vector's bracket (setup<Src, true>): raises lo to rho (no count needed) and
  // returns the (guess, width) = (rho, del / 4) that makes solve()'s warm path test rho + del / 4 first and walk up from
  // there; when the bracket is not worth it (del not well below the Gershgorin width: a localized mode the trial vector
  // does not resemble) the guess is NaN, which solve() treats as a cold start.
  __device__ __forceinline__ void trial_guess(T& guess, T& width) {
    const bool use = U(finite_of(trial_rho) && trial_del > T(0) && trial_del < T(0.25) * (hi - lo));
    if (use) lo = xmax(lo, trial_rho - trial_mrg);      // (the computed quotient is a lower bound of lam_max up to its own rounding)
    guess = use ? trial_rho : T(__builtin_nanf(""));
    width = T(0.25) * trial_del;
  }

  // forward sweep at shift sig (solution from the left end).  Returns the Sturm count (eigenvalues > sig).
  __device__ __forceinline__ int sweep_fwd(T sig) {
    T fA = T(1), fAp = T(0), fB = T(0), fBp = T(1);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const T tf = xfma(-sig, Ph[i], D[i]);
      if ((i < M - 1) || has_last) {
        const T nA = xfma(-tf, fA, -fAp), nB = xfma(-tf, fB, -fBp);
        fAp = fA; fA = nA; fBp = fB; fB = nB;
      }
    }
    M2<T> F;
    // (z_0, z_-1) -> (kap z_cnt, z_{cnt-1}/kap)   [hand-over into the next lane's scaling]
    F.a = fA * kap; F.b = fB * kap; F.c = fAp * ikap; F.d = fBp * ikap; F.e = 0;
    renorm(F);
    const M2<T> P = scan_fwd(F, lane);
    // shooting value: the forward solution one row past the end, u_{n+1}(sig) = kap_63 * P_63.a * 2^P_63.e, is
    // the characteristic polynomial of the pencil up to a sig-independent positive factor
    shoot_m = readlane_t(P.a, kWave - 1); shoot_e = readlane_i(P.e, kWave - 1);
    // incoming vector: first column of the left neighbour's inclusive product; (1,0) at the left end
    const T u0 = dpp_t<0x138, 0xF>(T(1), P.a);   // wave_shr:1
    const T um = dpp_t<0x138, 0xF>(T(0), P.c);
    Eu = dpp_i<0x138, 0xF>(0, P.e);
    zu_m1 = um; u0_in = u0;
    T zc = u0, zp = um;
    // Every sign change is counted between two values that ONE lane derives from ONE incoming pair: first
    // (u_{a-1}, u_a) as the scan delivers them, then the lane's own rows.  The step from a lane's last row into the
    // next lane's first row is counted by that next lane: if each lane used "its" version of that value (replayed
    // here, scanned there), a value near zero could carry different signs in the two lanes and a crossing would be
    // counted twice or not at all (FP32: 1e-4 of the config-5 systems locked onto lam_2 that way).  Only the last
    // lane also counts its final step (into u_n, the shooting value).
    const int ncount = (has_last ? M : M - 1) - (lane == kWave - 1 ? 0 : 1);
    int count = __popcll(__ballot(sign_differs(u0, um)));      // lane 0: (1, 0), no change
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const T t = xfma(-sig, Ph[i], D[i]);
      const bool act = (i < M - 1) || has_last;
      const T zn = xfma(-t, zc, -zp);
      const bool flip = (i < M - 2 || i < ncount) && sign_differs(zn, zc);
      count += __popcll(__ballot(flip));
      if (act) { zp = zc; zc = zn; }
    }
    return count;
  }

  // backward sweep at shift sig (solution from the right end); fills zw.  Returns the number of sign
  // changes of the backward solution w_{n-1}, ..., w_0, w_{-1} = the Sturm count (eigenvalues > sig) taken from the
  // other end of the matrix (the same theorem applied to the reversed index order).
  __device__ __forceinline__ int sweep_bwd(T sig) {
    T bA = T(1), bAn = T(0), bB = T(0), bBn = T(1);
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const T tb = xfma(-sig, Ph[i], D[i]);
      if ((i < M - 1) || has_last) {
        const T nA = xfma(-tb, bA, -bAn), nB = xfma(-tb, bB, -bBn);
        bAn = bA; bA = nA; bBn = bB; bB = nB;
      }
    }
    M2<T> B;
    // incoming (p, q) from lane L+1 is first rescaled into this lane's (z_{cnt-1}, z_cnt) = (kap p, q/kap)
    B.a = bA * kap; B.b = bB * ikap; B.c = bAn * kap; B.d = bBn * ikap; B.e = 0;
    renorm(B);
    const M2<T> Q = scan_bwd(B, lane);
    const T wp = dpp_t<0x130, 0xF>(T(1), Q.a);   // wave_shl:1   (p, q) = (z_-1, z_0) of lane L+1
    const T wq = dpp_t<0x130, 0xF>(T(0), Q.c);
    Ew = dpp_i<0x130, 0xF>(0, Q.e);
    T wc = wp * kap, wn = wq * ikap;   // (z_{cnt-1}, z_cnt)
    zw_p1 = wn;
    int count = 0;
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const T t = xfma(-sig, Ph[i], D[i]);
      const bool act = (i < M - 1) || has_last;
      zw[i] = act ? wc : wn;           // the unused last slot holds z_cnt (needed as "i+1" neighbour)
      const T z2 = xfma(-t, wc, -wn);
      count += __popcll(__ballot(act && sign_differs(z2, wc)));
      if (act) { wn = wc; wc = z2; }
    }
    return count;
  }

  __device__ __forceinline__ int sweep(T sig) {
    const int c = sweep_fwd(sig);
    sweep_bwd(sig);
    return c;
  }

  // twisted estimate from the last forward sweep (replayed) and the stored backward solution.  Returns rho
  // (Rayleigh/Newton update of sig); fills the per-lane normalisation (fu, fw, thr) that assemble() uses.
  //   rho = sig + gamma_k / sum_r f_r x_r^2,   gamma_k = [(T - sig F) x]_k,   x = u / u_k (r <= k), w / w_k (r >= k)
  // In the scaled variables (x = s z, Ph = f s^2) the scaling of row k cancels:
  //   rho = sig + num / (z_u,k z_w,k sum_r Ph_r zhat_r^2),  num = row-k residual of (z_u,k z_w,k) zhat,  zhat = z / z_k.
  T fu, fw;
  int thr;
  // CNT (solve<true>): also counts the eigenvalues above sig OTHER than the one the twist row belongs to, the robust way -- the
  // twisted factorisation N_k D_k N_k^T has the inertia of T - sig F, and its pivots are the sign changes of the forward solution on
  // the rows up to k and of the backward solution on the rows from k on, i.e. of each solution in the direction in which it GROWS:
  //   eigenvalues above sig = #changes(u; rows <= k) + #changes(w; rows >= k) + [gamma_k > 0].
  // `extra_above` = the first two terms.  It must be 0 when the shift iteration closes on lam_max; 1 or more says that the counts
  // of the prefix-product sweeps lost an eigenvalue (a shift that landed within their noise band of lam_2 read 0 instead of 1 and
  // became the bracket's upper end: one system of 10^6 at N_zeta = 1024 returned lam_2, 1.5e-10 ||A|| below lam_max: golden G10).
  int extra_above;
  template <bool CNT = false>
  __device__ __forceinline__ T twisted(T sig) {
    // pass A: replay the forward solution; per-lane candidate for the twist row k = argmax f |u w| (any row with a large
    // product will do: discrete Wronskian, gamma_r = W / (u_r w_r)) together with the entries around it, so that no
    // dynamically indexed access is needed afterwards
    T best = T(0);
    int bi = 0;
    T zu_b = T(1), zum_b = T(0), zw_b = T(1), zwp_b = T(0), t_b = T(0);
    {
      T zc = u0_in, zp = zu_m1;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const T t = xfma(-sig, Ph[i], D[i]);
        const bool act = (i < M - 1) || has_last;
        const T a = act ? xabs(Ph[i] * (zc * zw[i])) : T(0);
        const bool better = a > best;
        best = better ? a : best; bi = better ? i : bi;
        zu_b = better ? zc : zu_b; zum_b = better ? zp : zum_b; t_b = better ? t : t_b;
        zw_b = better ? zw[i] : zw_b; zwp_b = better ? (i == M - 1 ? zw_p1 : zw[i < M - 1 ? i + 1 : M - 1]) : zwp_b;
        const T zn = xfma(-t, zc, -zp);
        if (act) { zp = zc; zc = zn; }
      }
    }
    // wave argmax in (exponent, mantissa) form; the lane id rides in the low 6 mantissa bits
    // key = (exponent of the product incl. the lanes' scan exponents) + mantissa in [1, 2): integer-built
    const int ex = expo_of(best);
    constexpr int L = ExpLim<T>::v;
    const int exc = ex < -L ? -L : (ex > L ? L : ex);
    T key = (best > T(0) && finite_of(best) && ex > -(1 << 27)) ? T(ex + Eu + Ew) + T(0.5) * (best * pow2_of<T>(-exc)) : -T(1e30);
    int Lk;
    if constexpr (sizeof(T) == 8) {
      key = __hiloint2double(__double2hiint(key), (__double2loint(key) & ~63) | lane);
      Lk = __double2loint(wave_max(key)) & 63;
    } else {
      key = __int_as_float((__float_as_int(key) & ~63) | lane);
      Lk = __float_as_int(wave_max(key)) & 63;
    }
    const int ik = readlane_i(bi, Lk);
    const T zu_k = readlane_t(zu_b, Lk), zw_k = readlane_t(zw_b, Lk), t_k = readlane_t(t_b, Lk);
    const T um1 = readlane_t(zum_b, Lk), wp1 = readlane_t(zwp_b, Lk);
    const T uw = zu_k * zw_k;
    const T num = xfma(um1, zw_k, xfma(t_k, uw, wp1 * zu_k));   // row-k residual of the twisted vector (x u_k w_k)
    const int Euk = readlane_i(Eu, Lk), Ewk = readlane_i(Ew, Lk);
    int du = Eu - Euk, dw = Ew - Ewk;
    du = du > 1000 ? 1000 : (du < -2000 ? -2000 : du);
    dw = dw > 1000 ? 1000 : (dw < -2000 ? -2000 : dw);
    fu = xldexp(fast_rcp(zu_k), du);
    fw = xldexp(fast_rcp(zw_k), dw);
    thr = (lane < Lk) ? M : ((lane > Lk) ? -1 : ik);
    // pass B: sum f x^2 over the twisted vector (forward solution replayed again)
    T acc = T(0);
    using SignWord = typename std::conditional<(M + 2 <= 32), unsigned, unsigned long long>::type;
    SignWord su = sign_word(zu_m1) >> 31, sw = 0;      // CNT: sign bits of (u_-1, u_0 .. u_{M-1}) and of (w_0 .. w_{M-1})
    constexpr int MH = M / 2;                          // (M > 30: collected in two 32-bit words, rows below / from MH, joined after the loop)
    unsigned su_b = 0, sw_a = 0, sw_b = 0, su_a = sign_word(zu_m1) >> 31;
    {
      T zc = u0_in, zp = zu_m1;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const T t = xfma(-sig, Ph[i], D[i]);
        const bool act = (i < M - 1) || has_last;
        const T xu = zc * fu, xw = zw[i] * fw;
        const T x = (i <= thr) ? xu : xw;
        if (act) acc = xfma(Ph[i] * x, x, acc);
        if constexpr (CNT) {
          if constexpr (sizeof(SignWord) == 4) {
            su = (unsigned)__builtin_amdgcn_alignbit(su, sign_word(zc), 31);          // (su << 1) | sign(u_i)
            sw = (unsigned)__builtin_amdgcn_alignbit(sw, sign_word(zw[i]), 31);
          } else if (i < MH) {
            su_a = (unsigned)__builtin_amdgcn_alignbit(su_a, sign_word(zc), 31);
            sw_a = (unsigned)__builtin_amdgcn_alignbit(sw_a, sign_word(zw[i]), 31);
          } else {
            su_b = (unsigned)__builtin_amdgcn_alignbit(su_b, sign_word(zc), 31);
            sw_b = (unsigned)__builtin_amdgcn_alignbit(sw_b, sign_word(zw[i]), 31);
          }
        }
        const T zn = xfma(-t, zc, -zp);
        if (act) { zp = zc; zc = zn; }
      }
    }
    if constexpr (CNT) {
      // transition i of u = (u_{i-1} -> u_i) sits at bit M-1-i of su ^ (su >> 1), counted for the rows i <= thr this lane owns (each
      // lane from ITS incoming pair, like sweep_fwd); transition i of w = (w_{i+1} -> w_i), w_M = zw_p1, at bit M-i of sw2 ^ (sw2 << 1),
      // counted for the owned rows i >= thr
      if constexpr (sizeof(SignWord) == 8) {
        su = ((SignWord)su_a << (M - MH)) | (SignWord)su_b;
        sw = ((SignWord)sw_a << (M - MH)) | (SignWord)sw_b;
      }
      const int last = has_last ? M - 1 : M - 2;
      const int iu = thr < last ? thr : last;                                   // u: i = 0 .. iu   (thr = -1: none)
      const SignWord one = 1;
      const SignWord mu = iu >= 0 ? (((one << (iu + 1)) - one) << (M - 1 - iu)) : SignWord(0);
      const int iw = thr > 0 ? thr : 0;                                         // w: i = iw .. last   (thr = M: none)
      const SignWord mw = iw <= last ? (((one << (last - iw + 1)) - one) << (M - last)) : SignWord(0);
      const SignWord sw2 = (sw << 1) | (SignWord)(sign_word(zw_p1) >> 31);
      const SignWord fl_u = (su ^ (su >> 1)) & mu, fl_w = (sw2 ^ (sw2 << 1)) & mw;
      int c;
      if constexpr (sizeof(SignWord) == 4) c = __builtin_popcount(fl_u) + __builtin_popcount(fl_w);
      else c = __builtin_popcountll(fl_u) + __builtin_popcountll(fl_w);
      extra_above = wave_sum_i(c);
    }
    sig_vec = sig;
    const T tot = wave_sum(acc);
    return sig + num * fast_rcp(uw * tot);
  }

  // eigenvector entries of this lane's rows up to a common factor (normalised by the caller).  The diagonal scaling s is
  // rebuilt here from g with the arithmetic of setup() (s_0 = 1, s_{i+1} = 1 / (e_{i+1} s_i)), the forward solution is
  // replayed at the shift of the last twisted() call.  D, Ph and zw are consumed row by row: x[] replaces them.
  // gh: the caller-supplied half-grid g of setup() (Src::kHasGh), or null = mean of the neighbouring g.
  template <class Src>
  __device__ __forceinline__ void assemble(const Src& src, const T* gh, int N, T h, T (&x)[M]) {
    int a = rows_start(lane, N - 2);
    asm volatile("" : "+v"(a));          // (keeps the compiler from carrying setup()'s LDS addresses across the iteration)
    const T ih2 = T(1) / (h * h);
    T sc = T(1);
    T gcur = src.g(a + 1);
    T zc = u0_in, zp = zu_m1;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const bool act = (i < M - 1) || has_last;
      const T xu = sc * zc * fu, xw = sc * zw[i] * fw;
      x[i] = act ? ((i <= thr) ? xu : xw) : T(0);
      if (act) {
        T e_hi;
        if (gh) e_hi = gh[a + i + 1] * ih2;                    // (wave-uniform branch)
        else { const T gnext = src.g(a + i + 2); e_hi = T(0.5) * (gcur + gnext) * ih2; gcur = gnext; }
        sc = fast_rcp(e_hi * sc);
        const T zn = xfma(-xfma(-sig_vec, Ph[i], D[i]), zc, -zp);
        zp = zc; zc = zn;
      }
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most 4 rows of LDS reads in flight (registers)
    }
  }

  // Root finder on the shift.  Every forward sweep returns the Sturm count C(sig) AND the shooting value
  // p(sig) = u_{n+1}(sig) (characteristic polynomial up to a constant).  The bracket [lo, hi] is moved by counts
  // ONLY (count(lo) >= 1, count(hi) == 0: backward-stable certificates).  Until a shift with C == 1 has been
  // seen the bracket is bisected; from then on lo lies between lam_2 and lam_max, p has the single root lam_max
  // in the bracket, and the next shift is proposed Brent-style from the shooting values: the root of the
  // parabola through the two bracket ends and the most recently replaced end (Muller; secant with two points),
  // accepted only if it lies in the 3/4 of the bracket next to the end with the smaller |p| and the step is
  // less than half the previous one (else bisection), so the far regime (hi many eigenvalue gaps above
  // lam_max, where interpolation crawls) costs no more than bisection; a step that does not even halve |p| is
  // followed by a bisection.  A proposal within 4096 tol of that end that follows an interpolation step which
  // cut |p| by >= 16x (the convergent regime) is trusted to ~tol and the missing count certificate is placed
  // tol beyond it (the offset doubles whenever such a certificate fails), so that two counts close the bracket.
  // The solve ends when the certified bracket is narrower than 4 tol; ONE backward sweep + twisted
  // factorisation at the last shift then gives the eigenvector (assemble()) and the Rayleigh-quotient polish.
  // Cost model (tools/sweep_bench.hip, 1 wave per SIMD): forward sweep 0.66 us, backward sweep + twisted 1.3 us;
  // the earlier twisted-Newton iteration needed ~15 forward and ~10 backward/twisted per D3D-shape system,
  // this one ~15.5 forward and one backward/twisted.
#ifdef IBS_TRACE
  T* trace = nullptr;   // debug builds only (tools/trace_solve.hip): 6 values per iteration
#endif
#ifdef IBS_TRACE_ALL
  // debug builds only (tools/probe_direct.hip): EVERY sweep of the FP64 solve of block 0 / wave 0 -- shift, count, bracket before it
#define IBS_TRACE_SWEEP(it_, sig_, C_, lo_, hi_) do { if (sizeof(T) == 8 && blockIdx.x == 0 && threadIdx.x == 0 && (it_) <= 100) { \
    double* q_ = ibs_trace_all + 4 * ((it_) - 1); q_[0] = (double)(sig_); q_[1] = (double)(C_); q_[2] = (double)(lo_); q_[3] = (double)(hi_); ibs_trace_all[400] = (double)(it_); } } while (0)
#else
#define IBS_TRACE_SWEEP(it_, sig_, C_, lo_, hi_) do {} while (0)
#endif
  struct Pt { T x, m; int e; };      // shift, shooting value m * 2^e
  // All quantities of the shift iteration are wave-uniform but live in VGPRs (there is no scalar FP64), and a
  // branch on a VALU compare is compiled as a divergent one (exec masking, copies of every live value).  U()
  // turns such a compare into a scalar condition (ballot != 0: all lanes agree), so the control flow of the
  // iteration is scalar branches.
  __device__ __forceinline__ static bool U(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
  __device__ __forceinline__ static int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ static int lg2_of(const Pt& p) { const int e = expo_of(p.m); return e < -(1 << 27) ? e : p.e + e; }
  // Root of the parabola through (o, a, b) -- or of the secant through (a, b) when !use_o or the parabola has
  // no root in the bracket -- that lies in (lo_, hi_) nearest b.  Branch-free, one reciprocal level deep:
  // with h1 = a.x - o.x, h2 = b.x - a.x (both scaled by a power of two so that |h2| is in [1, 2)) and
  //   QA = (f2-f1) h1 - (f1-f0) h2,  QB = QA h2 + (f2-f1) h1 (h1+h2),  H = h1 h2 (h1+h2)
  // the parabola a z^2 + b z + f2 in z = x - b.x has a = QA/H, b = QB/H, so its roots are
  //   z = -2 f2 H / (QB +- sqrt(QB^2 - 4 QA f2 H))  and  z = -(QB +- sqrt(..)) / (2 QA).
  __device__ __forceinline__ static bool interpolate(const Pt& o, bool use_o, const Pt& a, const Pt& b,
                                                     T lo_, T hi_, T& rho) {
    constexpr int L = ExpLim<T>::v;
    int emax = a.e > b.e ? a.e : b.e;
    if (use_o && o.e > emax) emax = o.e;
    auto val = [&](const Pt& p) { int d = p.e - emax; d = d < -L ? -L : d; return p.m * pow2_of<T>(d); };
    const T x2 = b.x, f1 = val(a), f2 = val(b), f0 = val(o);
    const T h2r = x2 - a.x;
    int k = expo_of(h2r);
    k = k < -L ? -L : (k > L ? L : k);
    const T sc = pow2_of<T>(-k), back = pow2_of<T>(k);
    const T h2 = h2r * sc, h1 = (a.x - o.x) * sc;
    const T df21 = f2 - f1;
    // secant
    const T z_s = -f2 * h2 * approx_rcp(df21);
    const T r_s = xfma(z_s, back, x2);
    const bool in_s = finite_of(r_s) && r_s > lo_ && r_s < hi_;
    // parabola
    const T S = h1 + h2;
    const T QA = xfma(df21, h1, -(f1 - f0) * h2);
    const T QB = xfma(QA, h2, df21 * h1 * S);
    const T H = h1 * h2 * S;
    const T disc = xfma(QB, QB, -T(4) * QA * f2 * H);
    const T sq = approx_sqrt(xmax(disc, T(0)));
    const T den = QB >= T(0) ? QB + sq : QB - sq;          // the denominator of larger magnitude
    const T z1 = -T(2) * f2 * H * approx_rcp(den);          // root nearer to b
    const T z2 = -den * approx_rcp(T(2) * QA);              // the other root
    const T r1 = xfma(z1, back, x2), r2 = xfma(z2, back, x2);
    const bool par = use_o && disc >= T(0) && H != T(0);
    const bool in1 = par && finite_of(r1) && r1 > lo_ && r1 < hi_;
    const bool in2 = par && finite_of(r2) && r2 > lo_ && r2 < hi_;
    const bool pick1 = in1 && (!in2 || xabs(z1) <= xabs(z2));
    rho = pick1 ? r1 : (in2 ? r2 : r_s);
    return in1 || in2 || in_s;
  }
  // The same for lanes that hold DIFFERENT problems (GroupSolver::solve): the parabola's root nearer to b is formed first; it is
  // the root of smaller magnitude, so the form above takes it whenever it lies in the bracket, and the other root and the secant
  // are only evaluated when some lane that wants a proposal (`need`) has that root outside.  Same values as interpolate().
  __device__ __forceinline__ static bool interpolate_lazy(const Pt& o, bool use_o, const Pt& a, const Pt& b,
                                                          T lo_, T hi_, bool need, T& rho) {
    constexpr int L = ExpLim<T>::v;
    int emax = a.e > b.e ? a.e : b.e;
    if (use_o && o.e > emax) emax = o.e;
    auto val = [&](const Pt& p) { int d = p.e - emax; d = d < -L ? -L : d; return p.m * pow2_of<T>(d); };
    const T x2 = b.x, f1 = val(a), f2 = val(b), f0 = val(o);
    const T h2r = x2 - a.x;
    int k = expo_of(h2r);
    k = k < -L ? -L : (k > L ? L : k);
    const T sc = pow2_of<T>(-k), back = pow2_of<T>(k);
    const T h2 = h2r * sc, h1 = (a.x - o.x) * sc;
    const T df21 = f2 - f1;
    const T S = h1 + h2;
    const T QA = xfma(df21, h1, -(f1 - f0) * h2);
    const T QB = xfma(QA, h2, df21 * h1 * S);
    const T H = h1 * h2 * S;
    const T disc = xfma(QB, QB, -T(4) * QA * f2 * H);
    const T sq = approx_sqrt(xmax(disc, T(0)));
    const T den = QB >= T(0) ? QB + sq : QB - sq;          // the denominator of larger magnitude
    const T z1 = -T(2) * f2 * H * approx_rcp(den);          // root nearer to b
    const T r1 = xfma(z1, back, x2);
    const bool par = use_o && disc >= T(0) && H != T(0);
    const bool in1 = par && finite_of(r1) && r1 > lo_ && r1 < hi_;
    rho = r1;
    bool got = in1;
    if (__builtin_amdgcn_ballot_w64(need && !in1) != 0ull) {
      const T z2 = -den * approx_rcp(T(2) * QA);            // the other root
      const T r2 = xfma(z2, back, x2);
      const bool in2 = par && finite_of(r2) && r2 > lo_ && r2 < hi_;
      const T z_s = -f2 * h2 * approx_rcp(df21);            // secant
      const T r_s = xfma(z_s, back, x2);
      const bool in_s = finite_of(r_s) && r_s > lo_ && r_s < hi_;
      const bool pick1 = in1 && (!in2 || xabs(z1) <= xabs(z2));
      rho = pick1 ? r1 : (in2 ? r2 : r_s);
      got = in1 || in2 || in_s;
    }
    return got;
  }
  // Optional warm start: `guess` is an estimate of lam_max from a nearby problem (previous optimizer
  // iteration, DOF-perturbed equilibrium: sims_runner_NCSX.py:151-276 re-scans 72 perturbed equilibria),
  // `width` its expected error.  The first shift is guess + width; if the count says lam_max is still
  // above, the shift walks up geometrically until the count certifies an upper bound; the next shift is
  // guess - width (expected count 1).
  // CHK (round 6, the raw-system kernels): a consistency check of the closing bracket that costs nothing.  The counts of the
  // prefix-product sweeps are exact for a matrix perturbed by up to ~N^2 eps ||A|| in the worst case (iid-random coefficients: a few
  // systems per million close 2e-12 .. 1.5e-10 ||A|| away from lam_max), while the twisted factorisation at the last shift uses
  // each solution only in its GROWING direction: its Rayleigh polish rho = sig + gamma_k / sum f x^2 (sign(gamma_k) is the twisted
  // factorisation's verdict on which side of sig the eigenvalue lies) stays good where the eigenvector is localised -- exactly
  // where the counts go wrong.  Measured on 10 x 2^20 systems against division-form bisection (tests/tools/reclose_campaign.py,
  // profiles/r06_reclose_*): EVERY system beyond 4 N eps ||A|| had its polish 32 .. 2048 tol outside the certified bracket, and
  // no system whose polish lay within 8 tol of it was off by more than 2e-13 ||A||.  (The other free check -- the backward sweep's
  // count against the forward one at the last shift -- fired on 0.4-4 % of the rough systems and on none of the bad ones: dropped.)
  // So: a polish within chk_slack = min(8 tol, 2 N eps ||A||) of the bracket is clamped into it (whichever of the two is right,
  // the result is within max(4 tol, chk_slack) of lam_max); beyond that the solve is SUSPECT and the kernel closes it again by
  // division-form multisection on the original rows (reclose_division above).  When the eigenvector is wanted as well, the kernel
  // REPEATS set-up, one sweep pair at the re-closed eigenvalue and the growth-rate stage for that system after its regular
  // outputs are written, i.e. where nothing of this solver is live any more: sweeping again with D / Ph still live (+16-24
  // registers), jumping back into the iteration (+80) or looping back to set-up (+100 and scratch: the loop-invariant row loads
  // get hoisted) all cost the raw kernels occupancy.
  bool suspect, closed;
  T rho_last;
  // mark-only mode (diagnostics, cold): floor(log2(distance of the polish from the final bracket [lo, hi] / tol)) + 8 in bits 0..5
  // (0 = inside), bit 6 = extra_above != 0
  __device__ __forceinline__ int why() const {
    if (!closed) return 0;
    const T tol = T(64) * Eps<T>::v * normA, rho = rho_last;
    const T dist = xmax(xmax(lo - rho, rho - hi), T(0));
    int bk = dist > T(0) ? expo_of(dist) - expo_of(tol) + 8 : 0;
    bk = uniform_i(finite_of(rho) ? (bk < 1 ? (dist > T(0) ? 1 : 0) : (bk > 63 ? 63 : bk)) : 63);
    return bk | (extra_above != 0 ? 64 : 0);
  }
  T chk_slack;     // (set by setup())
  template <bool CHK = false>
  __device__ __forceinline__ T solve(SolveInfo& inf, bool warm = false, T guess = T(0), T width = T(0)) {
    // f64: 64 ulp of ||A||;  f32: 8 ulp (the counts themselves are only good to ~eps32*||A||)
    const T tol = (sizeof(T) == 8 ? T(64) : T(8)) * Eps<T>::v * normA;
    T sig = T(0.5) * (lo + hi), lam = hi;
    bool expand = false, try_below = false;
    T wstep = T(0);
    if (warm && U(finite_of(guess) && width > T(0) && guess + width < hi && guess + width > lo)) {
      sig = guess + width; expand = true; wstep = T(4) * width; try_below = true;
    }
    T off_up = tol, off_dn = tol;   // how far beyond the estimate the next certificate is placed
    T rho_trust = hi;
    int aimed = 0;                  // +1 / -1: the last proposal was an upper / lower certificate attempt
    bool lo1 = false;               // count(lo) == 1 is known: lo lies between lam_2 and lam_max
    bool hi_f = false, old_ok = false, was_interp = false;
    bool conv = false, force_bis = false;   // the last interpolation step cut |p| >= 16x / did not even halve it
    int lg_prev = 0;                        // log2 of the smaller |p| at the bracket ends when it was proposed
    Pt Plo{lo, T(0), 0}, Phi{hi, T(0), 0}, Pold{hi, T(0), 0};
    T sig_prev = sig;
    int it = 0;
    bool done = false;
    constexpr int kMaxIt = 200;
#ifdef IBS_PROBE
    long long t_sweep = 0, t_full = 0; int n_full = 0;
#endif
    while (!done && it < kMaxIt) {
#ifdef IBS_PROBE
      const long long tp0 = wall_clock64();
#endif
      const int C = sweep_fwd(sig);
#ifdef IBS_PROBE
      const long long tp1 = wall_clock64(); t_sweep += tp1 - tp0;
#endif
      ++it;
      IBS_TRACE_SWEEP(it, sig, C, lo, hi);
      // every shift lies strictly inside (lo, hi), so each count moves one end of the bracket
      if (C == 0) {
        if (hi_f) { Pold = Phi; old_ok = true; }
        hi = sig; Phi = Pt{sig, shoot_m, shoot_e}; hi_f = true;
      } else {
        if (lo1) { Pold = Plo; old_ok = true; }
        lo = sig; lo1 = (C == 1); Plo = Pt{sig, shoot_m, shoot_e};
      }
      const T prevstep = xabs(sig - sig_prev);
      sig_prev = sig;
      if (expand) {                       // warm start: walk up until the count certifies an upper bound
        if (C != 0 && U(sig + wstep < hi)) { sig += wstep; wstep *= T(4); continue; }
        expand = false;
      }
#ifdef IBS_EXP_BISECT
      if (true) {
#else
      if (!lo1) {                         // locate phase: bisection on the count alone (short path)
#endif
        if (U((hi - lo) <= T(4) * tol)) { done = true; break; }
        if (try_below && U(guess - width > lo && guess - width < hi)) sig = guess - width;
        else sig = T(0.5) * (lo + hi);
        try_below = false;
        continue;
      }
      // did the interpolation step that produced this shift pay off?  (integer log2 of the shooting values)
      if (was_interp) {
        const int e_now = expo_of(shoot_m);
        const int red = lg_prev - (e_now < -(1 << 27) ? e_now : shoot_e + e_now);
        conv = red >= 4;
        force_bis = red < 1;
      } else if (aimed == 0) conv = false;
      // a failed certificate attempt means the estimate is off by more than the offset: widen it
      const bool cert = (aimed != 0);     // this sweep was a certificate attempt around rho_trust
      if (aimed > 0 && C != 0) off_up *= T(2);
      if (aimed < 0 && C == 0) off_dn *= T(2);
      aimed = 0;
      if (U((hi - lo) <= T(4) * tol)) { done = true; break; }
      T rho = sig;
      bool ok = false, near = false;
      if (cert) { rho = rho_trust; ok = true; near = true; }
      else if (hi_f && !force_bis) {
        const int lg_lo = lg2_of(Plo), lg_hi = lg2_of(Phi);
        const bool b_is_lo = lg_lo <= lg_hi;                 // b = the end with the smaller |p|
        lg_prev = b_is_lo ? lg_lo : lg_hi;
        const Pt b = b_is_lo ? Plo : Phi;
        const Pt a = b_is_lo ? Phi : Plo;
        const bool use_o = old_ok && Pold.x != a.x && Pold.x != b.x;
        T r;
        const bool got = interpolate(Pold, use_o, a, b, lo, hi, r);
        const T q = T(0.25) * (T(3) * a.x + b.x);
        const bool inside = r >= xmin(q, b.x) && r <= xmax(q, b.x);
        const T stepb = xabs(r - b.x);
        const bool nr = was_interp && conv && stepb <= T(4096) * tol;
        const bool acc = got && inside && (nr || (stepb < T(0.5) * prevstep && stepb >= T(9.5367431640625e-07) * prevstep));
        if (U(acc)) { rho = r; ok = true; near = U(nr); }
      }
#ifdef IBS_TRACE
      if (lane == 0 && trace && it <= 64) { T* q = trace + 6 * (it - 1); q[0] = sig; q[1] = T(C); q[2] = ok ? rho : T(-999); q[3] = lo; q[4] = hi; q[5] = T(wall_clock64() % 100000000); }
#endif
      bool moved = false, interp_now = false;
      if (ok) {
        if (near) {
          // rho is trusted to ~tol: place the missing count certificate tol beyond it
          if (!cert && U(xabs(rho - rho_trust) > T(4096) * tol)) { off_up = tol; off_dn = tol; }   // fresh estimate
          rho_trust = rho;
          const T up = xmax(rho, lo), dn = xmin(rho, hi);
          T nxt = rho;
          if (U(hi > up + T(2) * off_up)) { nxt = up + off_up; aimed = 1; }
          else if (U(lo < dn - T(2) * off_dn)) { nxt = dn - off_dn; aimed = -1; }
          if (aimed != 0 && U(nxt > lo && nxt < hi)) { sig = nxt; moved = true; }
          else aimed = 0;
        } else {
          sig = rho; moved = true; interp_now = true;     // interpolate() returns points inside (lo, hi) only
        }
      }
      force_bis = false;
      if (!moved) sig = T(0.5) * (lo + hi);
      was_interp = interp_now;
#ifdef IBS_PROBE
      t_full += wall_clock64() - tp1; ++n_full;
#endif
    }
#ifdef IBS_PROBE
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) { long long* q = &ibs_probe_buf[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16]; q[5] = t_sweep; q[6] = t_full; q[7] = n_full; }
#endif
    inf.iters = it;
    inf.status = done ? 0 : 1;
    // eigenvector and Rayleigh-quotient polish at the last shift (inside the final bracket when done)
    IBS_PROBE_AT(10);
    sweep_bwd(sig);
    IBS_PROBE_AT(11);
    const T rho = twisted<CHK>(sig);
    IBS_PROBE_AT(12);
    if (done) lam = (finite_of(rho) && rho >= lo && rho <= hi) ? rho : T(0.5) * (lo + hi);
    else lam = sig;
    if constexpr (CHK) {
      const bool rho_in = U(finite_of(rho) && rho >= lo - chk_slack && rho <= hi + chk_slack);
      suspect = done && (!rho_in || extra_above != 0);
      rho_last = rho;
      if (done && rho_in) lam = xmin(xmax(rho, lo), hi);
      closed = done;
    }
#ifdef IBS_TRACE
    if (lane == 0 && trace && it < 64) { T* q = trace + 6 * it; q[0] = sig; q[1] = T(-1); q[2] = rho; q[3] = lo; q[4] = hi; q[5] = T(wall_clock64() % 100000000); }
#endif
    return lam;
  }
};

// ---------------------------------------------------------------- growth-rate stage (utils.py:1601-1621)
// X (normalised eigenfunction incl. the two zero end points) is in LDS.  N odd.
// Optional Hellmann-Feynman sums for up to NP tangent coefficient sets (utils.py:1676-1680).
// LDS rows are private to a wave: ordering inside the wave is all that is needed
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// position of element j in a padded LDS row (see lds_pitch in ibs_launch.hpp)
__device__ __forceinline__ int lpos(int j) { return j + (j >> 3); }
// dX of utils.py:1610-1616 on a padded row (the branch-free form used by the wave kernels is in finish())
template <typename T>
__device__ __forceinline__ T fd_derivative_p(const T* X, int j, int N, T ih) {
  if (j == 0) return (T(-1.5) * X[lpos(0)] + T(2) * X[lpos(1)] - T(0.5) * X[lpos(2)]) * ih;
  if (j == 1) return (X[lpos(2)] - X[lpos(0)]) * (T(0.5) * ih);
  if (j == N - 2) return (X[lpos(N - 1)] - X[lpos(N - 3)]) * (T(0.5) * ih);
  if (j == N - 1) return (T(0.5) * X[lpos(N - 3)] - T(2) * X[lpos(N - 2)]) * ih;
  return (T(2) / T(3)) * ih * (X[lpos(j + 1)] - X[lpos(j - 1)]) - (X[lpos(j + 2)] - X[lpos(j - 2)]) * (ih / T(12));
}
__device__ __forceinline__ int simpson_w(int j, int N) { return (j == 0 || j == N - 1) ? 1 : ((j & 1) ? 4 : 2); }

}  // namespace ibs
