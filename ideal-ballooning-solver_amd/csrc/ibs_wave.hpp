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
//     -> per-lane replay from the incoming vector.  Forward (from the left end) and backward (from
//     the right end) sweeps run fused; sign changes of the forward solution are the Sturm count.
//   * the forward solution u and the backward solution w give the twisted (double-sweep)
//     eigenvector estimate  x = u/u_k (r<=k), w/w_k (r>=k)  with k = argmax |u_r w_r| (discrete
//     Wronskian: gamma_r = W/(u_r w_r)), and the Rayleigh-quotient/Newton update
//     rho = sig + gamma_k / sum(f x^2).  rho is always a lower bound of lam_max, the Sturm count
//     says on which side of lam_max (and of lam_2) sig is, so the iteration is a safeguarded
//     Newton/bisection hybrid that needs no start vector and cannot land on a wrong eigenvalue.
//   * no MFMA: there is no dense contraction here; the binding resource is FP64 VALU issue.
#pragma once
#include <hip/hip_runtime.h>

namespace ibs {

constexpr int kWave = 64;

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
__device__ __forceinline__ bool finite_of(double a) { return xabs(a) <= 1.7976931348623157e308; }
__device__ __forceinline__ bool finite_of(float a) { return xabs(a) <= 3.4028234e38f; }

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

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = xmax(v, __shfl_xor(v, d));
  return v;
}
template <typename T>
__device__ __forceinline__ T uniform(T v) {  // value is already identical in all lanes: make it scalar
  if constexpr (sizeof(T) == 8) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
  } else {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
  }
}

// ---------------------------------------------------------------- 2x2 transfer matrices with exponent
template <typename T>
struct M2 {
  T a, b, c, d;  // [[a b],[c d]]
  int e;         // true matrix = this * 2^e
};

template <typename T>
__device__ __forceinline__ void renorm(M2<T>& R) {
  T m = xmax(xmax(xabs(R.a), xabs(R.b)), xmax(xabs(R.c), xabs(R.d)));
  int ex = fexp(m);
  R.a = xldexp(R.a, -ex); R.b = xldexp(R.b, -ex); R.c = xldexp(R.c, -ex); R.d = xldexp(R.d, -ex);
  R.e += ex;
}
// R = A * B (A applied after B), renormalised
template <typename T>
__device__ __forceinline__ M2<T> mul(const M2<T>& A, const M2<T>& B) {
  M2<T> R;
  R.a = xfma(A.a, B.a, A.b * B.c);
  R.b = xfma(A.a, B.b, A.b * B.d);
  R.c = xfma(A.c, B.a, A.d * B.c);
  R.d = xfma(A.c, B.b, A.d * B.d);
  R.e = A.e + B.e;
  renorm(R);
  return R;
}
template <typename T, int CTRL, int ROWMASK>
__device__ __forceinline__ M2<T> dpp_fetch(const M2<T>& s) {  // lanes without a source get the identity
  M2<T> r;
  r.a = dpp_t<CTRL, ROWMASK>(T(1), s.a);
  r.b = dpp_t<CTRL, ROWMASK>(T(0), s.b);
  r.c = dpp_t<CTRL, ROWMASK>(T(0), s.c);
  r.d = dpp_t<CTRL, ROWMASK>(T(1), s.d);
  r.e = dpp_i<CTRL, ROWMASK>(0, s.e);
  return r;
}
template <typename T>
__device__ __forceinline__ M2<T> lane_fetch(const M2<T>& s, int src_lane, bool take) {
  M2<T> r;
  T a = __shfl(s.a, src_lane), b = __shfl(s.b, src_lane), c = __shfl(s.c, src_lane), d = __shfl(s.d, src_lane);
  int e = __shfl(s.e, src_lane);
  r.a = take ? a : T(1); r.b = take ? b : T(0); r.c = take ? c : T(0); r.d = take ? d : T(1); r.e = take ? e : 0;
  return r;
}

// inclusive prefix product over lanes:  P_L = A_L * A_{L-1} * ... * A_0
template <typename T>
__device__ __forceinline__ M2<T> scan_fwd(M2<T> P) {
  P = mul(P, dpp_fetch<T, 0x111, 0xF>(P));  // row_shr:1
  P = mul(P, dpp_fetch<T, 0x112, 0xF>(P));  // row_shr:2
  P = mul(P, dpp_fetch<T, 0x114, 0xF>(P));  // row_shr:4
  P = mul(P, dpp_fetch<T, 0x118, 0xF>(P));  // row_shr:8
  P = mul(P, dpp_fetch<T, 0x142, 0xA>(P));  // row_bcast:15 -> rows 1,3
  P = mul(P, dpp_fetch<T, 0x143, 0xC>(P));  // row_bcast:31 -> rows 2,3
  return P;
}
// inclusive suffix product over lanes:  Q_L = B_L * B_{L+1} * ... * B_63
template <typename T>
__device__ __forceinline__ M2<T> scan_bwd(M2<T> Q, int lane) {
  Q = mul(Q, dpp_fetch<T, 0x101, 0xF>(Q));  // row_shl:1
  Q = mul(Q, dpp_fetch<T, 0x102, 0xF>(Q));  // row_shl:2
  Q = mul(Q, dpp_fetch<T, 0x104, 0xF>(Q));  // row_shl:4
  Q = mul(Q, dpp_fetch<T, 0x108, 0xF>(Q));  // row_shl:8
  const int row = lane >> 4;
  Q = mul(Q, lane_fetch(Q, (row + 1) << 4, (row & 1) == 0));  // rows 0,2 <- first lane of rows 1,3
  Q = mul(Q, lane_fetch(Q, 32, row < 2));                      // rows 0,1 <- lane 32 (rows 2+3)
  return Q;
}

// ---------------------------------------------------------------- per-wave solver state
struct SolveInfo {
  int iters;    // fused double sweeps used
  int status;   // 0 ok, bit0 = iteration cap hit, bit1 = non-finite data / non-positive g or f
};

template <typename T, int M>
struct WaveSolver {
  // chunk of this lane, symmetric scaling (see header comment)
  T D[M], Ph[M], S[M];
  T kap, ikap;
  bool has_last;   // this lane owns M rows (else M-1)
  int lane;
  // sweep products
  T zu[M], zw[M];
  T zu_m1, zw_p1;  // z_{-1} of the forward solution, z_{cnt} of the backward solution (local scaling)
  int Eu, Ew;      // power-of-two exponents of this lane's forward / backward solutions
  // bounds
  T lo, hi, normA;

  // rows of this lane: [start, start+cnt)
  __device__ __forceinline__ static int rows_start(int lane, int n) {
    const int rem = n - kWave * (M - 1);
    return lane * (M - 1) + (lane < rem ? lane : rem);
  }

  // Src provides g(j), c(j), f(j) for grid point j in [0, N) (LDS-backed)
  template <class Src>
  __device__ __forceinline__ bool setup(const Src& src, int N, T h) {
    lane = threadIdx.x & (kWave - 1);
    const int n = N - 2;
    const int rem = n - kWave * (M - 1);
    has_last = lane < rem;
    const int a = rows_start(lane, n);
    const T ih2 = T(1) / (h * h);
    T sc = T(1);
    T gprev = src.g(a);
    T gcur = src.g(a + 1);
    T e_lo = T(0.5) * (gprev + gcur) * ih2;  // e_a
    T vhi = -T(1e300), vlo = -T(1e300), vna = T(0), sum_c = T(0), sum_f = T(0);
    bool bad = false;
    const T e_first = e_lo;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const bool act = (i < M - 1) || has_last;
      if (act) {
        const int j = a + i + 1;  // grid point of row a+i
        const T gnext = src.g(j + 1);
        const T e_hi = T(0.5) * (gcur + gnext) * ih2;  // e_{a+i+1}
        const T cj = src.c(j), fj = src.f(j);
        const T d = cj - (e_lo + e_hi);
        const T s2 = sc * sc;
        S[i] = sc; D[i] = d * s2; Ph[i] = fj * s2;
        const T rf = T(1) / fj;
        vhi = xmax(vhi, cj * rf);
        vlo = xmax(vlo, d * rf);
        vna = xmax(vna, (xabs(d) + e_lo + e_hi) * rf);
        sum_c += cj; sum_f += fj;
        bad = bad || !(fj > T(0)) || !(e_hi > T(0)) || !finite_of(cj);
        sc = T(1) / (e_hi * sc);
        gcur = gnext; e_lo = e_hi;
      } else {
        S[i] = T(0); D[i] = T(0); Ph[i] = T(0);
      }
    }
    kap = sc; ikap = T(1) / sc;
    bad = bad || !(e_first > T(0));
    // wave-level bounds:  lam_max <= max c/f (Gershgorin);  lam_max >= any Rayleigh quotient
    const T e0 = __shfl(e_first, 0);
    const T en = __shfl(e_lo, kWave - 1);
    const T sc_all = wave_sum(sum_c), sf_all = wave_sum(sum_f);
    normA = uniform(wave_max(vna));
    hi = uniform(wave_max(vhi));
    lo = uniform(xmax(wave_max(vlo), (sc_all - e0 - en) / sf_all));
    hi += T(8) * Eps<T>::v * normA;
    lo -= T(8) * Eps<T>::v * normA;
    return __any(bad) != 0;
  }

  // one fused forward+backward sweep at shift sig.  Returns the Sturm count (eigenvalues > sig).
  __device__ __forceinline__ int sweep(T sig) {
    // ---- pass 1: chunk transfer matrices (two columns each direction)
    T fA = T(1), fAp = T(0), fB = T(0), fBp = T(1);
    T bA = T(1), bAn = T(0), bB = T(0), bBn = T(1);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int ib = M - 1 - i;
      const T tf = xfma(-sig, Ph[i], D[i]);
      const T tb = xfma(-sig, Ph[ib], D[ib]);
      if ((i < M - 1) || has_last) {
        const T nA = xfma(-tf, fA, -fAp), nB = xfma(-tf, fB, -fBp);
        fAp = fA; fA = nA; fBp = fB; fB = nB;
      }
      if ((ib < M - 1) || has_last) {
        const T nA = xfma(-tb, bA, -bAn), nB = xfma(-tb, bB, -bBn);
        bAn = bA; bA = nA; bBn = bB; bB = nB;
      }
    }
    M2<T> F, B;
    // forward: (z_0, z_-1) -> (kap z_cnt, z_{cnt-1}/kap)   [hand-over into the next lane's scaling]
    F.a = fA * kap; F.b = fB * kap; F.c = fAp * ikap; F.d = fBp * ikap; F.e = 0;
    renorm(F);
    // backward: incoming (p, q) from lane L+1 is first rescaled into this lane's (z_{cnt-1}, z_cnt) = (kap p, q/kap)
    B.a = bA * kap; B.b = bB * ikap; B.c = bAn * kap; B.d = bBn * ikap; B.e = 0;
    renorm(B);
    // ---- wave scans
    const M2<T> P = scan_fwd(F);
    const M2<T> Q = scan_bwd(B, lane);
    // incoming vectors: first column of the neighbour's inclusive product; (1,0) at the two ends
    T u0 = dpp_t<0x138, 0xF>(T(1), P.a);   // wave_shr:1
    T um = dpp_t<0x138, 0xF>(T(0), P.c);
    Eu = dpp_i<0x138, 0xF>(0, P.e);
    T wp = dpp_t<0x130, 0xF>(T(1), Q.a);   // wave_shl:1   (p, q) = (z_-1, z_0) of lane L+1
    T wq = dpp_t<0x130, 0xF>(T(0), Q.c);
    Ew = dpp_i<0x130, 0xF>(0, Q.e);
    // ---- pass 2: replay, keep the solutions, count sign changes of the forward solution
    zu_m1 = um;
    T zc = u0, zp = um;
    int count = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const T t = xfma(-sig, Ph[i], D[i]);
      const bool act = (i < M - 1) || has_last;
      zu[i] = act ? zc : T(0);
      const T zn = xfma(-t, zc, -zp);
      const bool flip = act && (signbit_of(zn) != signbit_of(zc));
      count += __popcll(__ballot(flip));
      if (act) { zp = zc; zc = zn; }
    }
    T wc = wp * kap, wn = wq * ikap;   // (z_{cnt-1}, z_cnt)
    zw_p1 = wn;
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const T t = xfma(-sig, Ph[i], D[i]);
      const bool act = (i < M - 1) || has_last;
      zw[i] = act ? wc : wn;           // the unused last slot holds z_cnt (needed as "i+1" neighbour)
      const T z2 = xfma(-t, wc, -wn);
      if (act) { wn = wc; wc = z2; }
    }
    return count;
  }

  // twisted estimate from the last sweep.  Returns rho (Rayleigh/Newton update of sig); fills the
  // per-lane normalisation (fu, fw, thr) that assemble() uses.
  T fu, fw;
  int thr;
  __device__ __forceinline__ T twisted(T sig) {
    T best = T(0), bnum = T(0), bu = T(1), bw = T(1);
    int bi = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const T t = xfma(-sig, Ph[i], D[i]);
      const T uw = zu[i] * zw[i];
      const T pr = S[i] * S[i] * uw;
      const T um1 = (i == 0) ? zu_m1 : zu[i > 0 ? i - 1 : 0];
      const T wp1 = (i == M - 1) ? zw_p1 : zw[i < M - 1 ? i + 1 : M - 1];
      const T num = xfma(um1, zw[i], xfma(t, uw, wp1 * zu[i]));
      const bool better = xabs(pr) > xabs(best);
      best = better ? pr : best; bnum = better ? num : bnum; bi = better ? i : bi;
      bu = better ? S[i] * zu[i] : bu; bw = better ? S[i] * zw[i] : bw;
    }
    // wave argmax of |u_k w_k| in (exponent, mantissa) form
    const T ab = xabs(best);
    const int ex = fexp(ab);
    T key = (ab > T(0) && finite_of(ab)) ? T(ex + Eu + Ew) + xldexp(ab, -ex) : -T(1e30);
    int kl = lane;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const T k2 = __shfl_xor(key, d);
      const int l2 = __shfl_xor(kl, d);
      const bool take = (k2 > key) || (k2 == key && l2 < kl);
      key = take ? k2 : key; kl = take ? l2 : kl;
    }
    const int Lk = __builtin_amdgcn_readfirstlane(kl);
    const int ik = __shfl(bi, Lk);
    const T gam_k = __shfl(bnum, Lk) / __shfl(best, Lk);
    const T uk = __shfl(bu, Lk), wk = __shfl(bw, Lk);
    const int Euk = __shfl(Eu, Lk), Ewk = __shfl(Ew, Lk);
    int du = Eu - Euk, dw = Ew - Ewk;
    du = du > 1000 ? 1000 : (du < -2000 ? -2000 : du);
    dw = dw > 1000 ? 1000 : (dw < -2000 ? -2000 : dw);
    fu = xldexp(T(1) / uk, du);
    fw = xldexp(T(1) / wk, dw);
    thr = (lane < Lk) ? M : ((lane > Lk) ? -1 : ik);
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const T xu = zu[i] * fu, xw = zw[i] * fw;
      const T x = (i <= thr) ? xu : xw;
      if ((i < M - 1) || has_last) acc = xfma(Ph[i] * x, x, acc);
    }
    const T tot = wave_sum(acc);
    return uniform(sig + gam_k / tot);
  }

  // eigenvector entries of this lane's rows (twisted, x_k = 1) from the last sweep/twisted() call
  __device__ __forceinline__ void assemble(T (&x)[M]) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const T xu = S[i] * zu[i] * fu, xw = S[i] * zw[i] * fw;
      x[i] = ((i < M - 1) || has_last) ? ((i <= thr) ? xu : xw) : T(0);
    }
  }

  // safeguarded Newton / bisection on the shift.  On return the last sweep was taken at a shift
  // whose distance to lam_max is below the tolerance, so assemble() gives the eigenvector.
  __device__ __forceinline__ T solve(SolveInfo& inf) {
    const T tol = T(64) * Eps<T>::v * normA;
    T sig = hi, rej = -T(1), lam = hi;
    int it = 0;
    bool done = false;
    constexpr int kMaxIt = 160;
    while (!done && it < kMaxIt) {
      const int C = sweep(sig);
      ++it;
      if (C == 0) hi = xmin(hi, sig); else lo = xmax(lo, sig);
      const T rho = twisted(sig);
      const bool ok = finite_of(rho);
      const bool tryn = ok && (C == 1 || (C == 0 && (rej < T(0) || (hi - lo) <= T(0.125) * rej)));
      bool moved = false;
      if (tryn) {
        const bool acc = (rho > lo) && (rho < hi);
        lo = xmax(lo, rho - tol);
        if (xabs(rho - sig) <= tol) { lam = rho; done = true; }
        else if (acc) { sig = rho; moved = true; }
        else if (C == 0) rej = hi - lo;
      }
      if (!done && !moved) {
        if (hi - lo <= tol) {
          lam = T(0.5) * (lo + hi);
          sweep(lam); twisted(lam); ++it;
          done = true;
        } else {
          sig = T(0.5) * (lo + hi);
        }
      }
    }
    inf.iters = it;
    inf.status = done ? 0 : 1;
    if (!done) lam = sig;
    return lam;
  }
};

// ---------------------------------------------------------------- growth-rate stage (utils.py:1601-1621)
// X (normalised eigenfunction incl. the two zero end points) is in LDS.  N odd.
// Optional Hellmann-Feynman sums for up to NP tangent coefficient sets (utils.py:1676-1680).
template <typename T>
__device__ __forceinline__ T fd_derivative(const T* X, int j, int N, T ih) {
  // utils.py:1610-1616
  if (j == 0) return (T(-1.5) * X[0] + T(2) * X[1] - T(0.5) * X[2]) * ih;
  if (j == 1) return (X[2] - X[0]) * (T(0.5) * ih);
  if (j == N - 2) return (X[N - 1] - X[N - 3]) * (T(0.5) * ih);
  if (j == N - 1) return (T(0.5) * X[N - 3] - T(2) * X[N - 2]) * ih;
  return (T(2) / T(3)) * ih * (X[j + 1] - X[j - 1]) - (X[j + 2] - X[j - 2]) * (ih / T(12));
}
__device__ __forceinline__ int simpson_w(int j, int N) { return (j == 0 || j == N - 1) ? 1 : ((j & 1) ? 4 : 2); }

}  // namespace ibs
