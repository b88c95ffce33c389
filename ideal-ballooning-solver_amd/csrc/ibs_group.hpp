// Sub-wave variant of the field-line solver: G = 64/P systems share one wavefront, P lanes each
// (P = 32 or 16).  Same mathematics as ibs_wave.hpp (read its header first); what changes:
//   * the Kogge-Stone scan needs log2(P) steps and is amortised over G systems -- the scan is ~70 %
//     of a sweep at P = 64, so large batches of short grids (N <= 514 for P = 32, N <= 258 for
//     P = 16) get ~2x / ~3x more solves per issued instruction;
//   * everything that is wave-uniform in WaveSolver (shift, bracket, counts, k, ...) is only
//     GROUP-uniform here and lives in VGPRs, replicated over the lanes of the group; the Newton /
//     bisection state machine is written branch-free per lane and the wave loops until every
//     group is done (a finished group keeps re-evaluating its frozen shift, so its last sweep --
//     needed for the eigenvector -- stays intact).
// Small batches keep P = 64 (one wave per SIMD is the lowest latency); the C-ABI layer picks P.
#pragma once
#include "ibs_wave.hpp"

namespace ibs {

template <int P>
struct Grp {
  static_assert(P == 16 || P == 32 || P == 64, "group size");
  static constexpr int G = 64 / P;

  // value held by the LAST lane of each group, broadcast to the whole group
  template <typename T>
  __device__ __forceinline__ static T bcast_last(T v, int lane) {
    if constexpr (P == 64) return readlane_t(v, 63);
    else if constexpr (P == 32) { const T a = readlane_t(v, 31), b = readlane_t(v, 63); return lane < 32 ? a : b; }
    else return dpp_t<0x15F, 0xF>(v, v);   // row_newbcast:15
  }
  __device__ __forceinline__ static int bcast_last_i(int v, int lane) {
    if constexpr (P == 64) return readlane_i(v, 63);
    else if constexpr (P == 32) { const int a = readlane_i(v, 31), b = readlane_i(v, 63); return lane < 32 ? a : b; }
    else return dpp_i<0x15F, 0xF>(v, v);
  }
  template <typename T>
  __device__ __forceinline__ static T sum(T v, int lane) {
    v += dppz_t<0x111, 0xF>(v);
    v += dppz_t<0x112, 0xF>(v);
    v += dppz_t<0x114, 0xF>(v);
    v += dppz_t<0x118, 0xF>(v);
    if constexpr (P >= 32) v += dpp_t<0x142, 0xA>(T(0), v);
    if constexpr (P == 64) v += dpp_t<0x143, 0xC>(T(0), v);
    return bcast_last(v, lane);
  }
  __device__ __forceinline__ static int sum_i(int v, int lane) {
    v += dppz_i<0x111, 0xF>(v);
    v += dppz_i<0x112, 0xF>(v);
    v += dppz_i<0x114, 0xF>(v);
    v += dppz_i<0x118, 0xF>(v);
    if constexpr (P >= 32) v += dpp_i<0x142, 0xA>(0, v);
    if constexpr (P == 64) v += dpp_i<0x143, 0xC>(0, v);
    return bcast_last_i(v, lane);
  }
  template <typename T>
  __device__ __forceinline__ static T max(T v, int lane) {
    v = xmax(v, dpp_t<0x111, 0xF>(v, v));
    v = xmax(v, dpp_t<0x112, 0xF>(v, v));
    v = xmax(v, dpp_t<0x114, 0xF>(v, v));
    v = xmax(v, dpp_t<0x118, 0xF>(v, v));
    if constexpr (P >= 32) v = xmax(v, dpp_t<0x142, 0xA>(v, v));
    if constexpr (P == 64) v = xmax(v, dpp_t<0x143, 0xC>(v, v));
    return bcast_last(v, lane);
  }
  // inclusive prefix product inside each group (range and first-column remarks: scan_fwd in ibs_wave.hpp -- FP64 needs no
  // renormalisation inside the scan; only (a, c, e) of the result are defined, the last step forms the first column only)
  template <typename T>
  __device__ __forceinline__ static M2<T> scan_fwd(M2<T> Q, int lane) {
    constexpr bool RN = ScanNorm<T>::v;
    const int l16 = lane & 15, row = lane >> 4;
    auto last_step = [&](const T Fa, const T Fc, const int Fe, const bool on) {
      if (on) {
        const T na = xfma(Q.a, Fa, Q.b * Fc);
        Q.c = xfma(Q.c, Fa, Q.d * Fc);
        Q.a = na;
        Q.e += Fe;
        if (RN) { const T m = xmax(xabs(Q.a), xabs(Q.c)); int ex; const T sc = pow2_scale_of(m, ex); Q.a *= sc; Q.c *= sc; Q.e += ex; }
      }
    };
    { const M2<T> F = dpp_fetch<T, 0x111, 0xF>(Q); if (l16 >= 1) Q = mul<T, false>(Q, F); }
    { const M2<T> F = dpp_fetch<T, 0x112, 0xF>(Q); if (l16 >= 2) Q = mul<T, RN>(Q, F); }
    { const M2<T> F = dpp_fetch<T, 0x114, 0xF>(Q); if (l16 >= 4) Q = mul<T, false>(Q, F); }
    if constexpr (P == 16) {
      last_step(dppz_t<0x118, 0xF>(Q.a), dppz_t<0x118, 0xF>(Q.c), dppz_i<0x118, 0xF>(Q.e), l16 >= 8);
    } else {
      { const M2<T> F = dpp_fetch<T, 0x118, 0xF>(Q); if (l16 >= 8) Q = mul<T, RN>(Q, F); }
      if constexpr (P == 32) {
        last_step(dppz_t<0x142, 0xA>(Q.a), dppz_t<0x142, 0xA>(Q.c), dppz_i<0x142, 0xA>(Q.e), (row & 1) != 0);
      } else {
        { const M2<T> F = dpp_fetch<T, 0x142, 0xA>(Q); if (row & 1) Q = mul<T, false>(Q, F); }
        last_step(dppz_t<0x143, 0xC>(Q.a), dppz_t<0x143, 0xC>(Q.c), dppz_i<0x143, 0xC>(Q.e), row >= 2);
      }
    }
    return Q;
  }
  // inclusive suffix product inside each group
  template <typename T>
  __device__ __forceinline__ static M2<T> scan_bwd(M2<T> Q, int lane) {
    const int l16 = lane & 15, row = lane >> 4;
    { const M2<T> F = dpp_fetch<T, 0x101, 0xF>(Q); if (l16 < 15) Q = mul<T, false>(Q, F); }
    { const M2<T> F = dpp_fetch<T, 0x102, 0xF>(Q); if (l16 < 14) Q = mul<T, true>(Q, F); }
    { const M2<T> F = dpp_fetch<T, 0x104, 0xF>(Q); if (l16 < 12) Q = mul<T, false>(Q, F); }
    { const M2<T> F = dpp_fetch<T, 0x108, 0xF>(Q); if (l16 < 8) Q = mul<T, true>(Q, F); }
    if constexpr (P >= 32) {
      const M2<T> t16 = lane_bcast(Q, 16), t48 = lane_bcast(Q, 48);
      M2<T> F;
      F.a = row == 0 ? t16.a : t48.a; F.b = row == 0 ? t16.b : t48.b;
      F.c = row == 0 ? t16.c : t48.c; F.d = row == 0 ? t16.d : t48.d; F.e = row == 0 ? t16.e : t48.e;
      if ((row & 1) == 0) Q = mul<T, P == 32>(Q, F);
    }
    if constexpr (P == 64) { const M2<T> F = lane_bcast(Q, 32); if (row < 2) Q = mul<T, true>(Q, F); }
    return Q;
  }
};

template <typename T, int M, int P>
struct GroupSolver {
  using GP = Grp<P>;
  static constexpr int G = 64 / P;
  T D[M], Ph[M];            // the scaling s itself is not kept: it cancels in the iteration (see twisted())
  T kap, ikap;
  bool has_last;
  int lane, lg, gid;
  T sig_vec;                // shift of the last twisted() call (assemble() replays the forward solution there)
  T zw[M];                  // backward solution of the last sweep_bwd; the forward one is replayed from (u0_in, zu_m1)
  T u0_in;
  T zu_m1, zw_p1;
  int Eu, Ew;
  T lo, hi, normA;          // group-replicated
  T trial_rho, trial_del;   // Rayleigh quotient / residual bound of the trial vector (setup<Src, true>; see WaveSolver::setup)
  T trial_mrg;              // allowance for the rounding of rho's sums: (8 + N / 2) eps |A|
  T shoot_m; int shoot_e;   // shooting value of the last forward sweep (group-replicated), see WaveSolver
  T fu, fw;
  int thr;

  __device__ __forceinline__ static int rows_start(int lg_, int n) {
    const int rem = n - P * (M - 1);
    return lg_ * (M - 1) + (lg_ < rem ? lg_ : rem);
  }

  // Src provides g(j), c(j), f(j) of THIS lane's system
  template <class Src, bool TRIAL = false>
  __device__ __forceinline__ bool setup(const Src& src, int N, T h) {
    lane = threadIdx.x & 63;
    lg = lane & (P - 1);
    gid = lane / P;
    const int n = N - 2;
    const int rem = n - P * (M - 1);
    has_last = lg < rem;
    const int a = rows_start(lg, n);
    const T ih2 = T(1) / (h * h);
    T sc = T(1);
    const T g0 = src.g(a);
    T gcur = src.g(a + 1);
    T e_lo = T(0.5) * (g0 + gcur) * ih2;
    T vhi = -T(1e300), vlo = -T(1e300), vna = T(0), sum_c = T(0), sum_f = T(0);
    bool bad = !(g0 > T(0)) || !(gcur > T(0));   // non-finite data, g <= 0 or f <= 0 anywhere in this lane's rows
    const T e_first = e_lo;
    T ts_prev = T(0), ts_cur = T(0), two_cd = T(0), tA = T(0), tB = T(0), tC = T(0);
    if constexpr (TRIAL) {               // trial vector x_j = sin(pi j / (N - 1)): WaveSolver::setup
      const T dl = T(3.14159265358979323846) / T(N - 1);
      T s0, c0, sd, cd;
      trial_sincos(T(a) * dl, s0, c0);
      trial_sincos(dl, sd, cd);
      ts_prev = s0; ts_cur = xfma(s0, cd, c0 * sd); two_cd = T(2) * cd;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const bool act = (i < M - 1) || has_last;
      if (act) {
        const int j = a + i + 1;
        const T gnext = src.g(j + 1);
        bad = bad || !(gnext > T(0));
        const T e_hi = T(0.5) * (gcur + gnext) * ih2;
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
        sc = fast_rcp(e_hi * sc);           // e s_i s_{i+1} = 1 to rounding
        gcur = gnext; e_lo = e_hi;
      } else {
        D[i] = T(0); Ph[i] = T(0);
      }
    }
    kap = sc; ikap = fast_rcp(sc);
    bad = bad || !(e_first > T(0));
    // e_0 lives in the group's first lane, e_n in its last lane
    const T ends = GP::sum((lg == 0 ? e_first : T(0)) + (lg == P - 1 ? e_lo : T(0)), lane);   // e_0 + e_n
    const T sc_all = GP::sum(sum_c, lane), sf_all = GP::sum(sum_f, lane);
    normA = GP::max(vna, lane);
    hi = GP::max(vhi, lane);
    lo = xmax(GP::max(vlo, lane), (sc_all - ends) / sf_all);
    hi += T(8) * Eps<T>::v * normA;
    lo -= T(8) * Eps<T>::v * normA;
    if constexpr (TRIAL) {
      const T A = GP::sum(tA, lane), B = GP::sum(tB, lane), C = GP::sum(tC, lane);
      const T rho = A / B;
      trial_rho = rho;
      trial_del = approx_sqrt(xmax(xfma(-rho, A, C), T(0)) / B);
    } else {
      trial_rho = T(0); trial_del = T(-1);
    }
    trial_mrg = T(8 + N / 2) * Eps<T>::v * normA;
    chk_slack = T(N < 256 ? 2 * N : 512) * Eps<T>::v * normA;        // min(8 tol, 2 N eps ||A||)   (solve<true>)
    return GP::sum_i(bad ? 1 : 0, lane) != 0;   // per group
  }

  // per-group start of a cold solve from the trial vector's bracket (WaveSolver::trial_guess, written with selects: the
  // groups of a wave decide independently); groups with take == false keep their (guess, width, warm)
  __device__ __forceinline__ void trial_guess(bool take, T& guess, T& width, bool& warm) {
    const bool use = take && finite_of(trial_rho) && trial_del > T(0) && trial_del < T(0.25) * (hi - lo);
    lo = use ? xmax(lo, trial_rho - trial_mrg) : lo;
    guess = use ? trial_rho : guess;
    width = use ? T(0.25) * trial_del : width;
    warm = use ? true : warm;
  }

  // forward sweep; returns the group's Sturm count (eigenvalues > sig), group-replicated.  STORE: keep the
  // incoming vector so that twisted() / assemble() can replay the forward solution (M fma per replay: cheaper
  // in registers than an M-entry array, which is what bounds the occupancy of these kernels)
  template <bool STORE = true>
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
    F.a = fA * kap; F.b = fB * kap; F.c = fAp * ikap; F.d = fBp * ikap; F.e = 0;
    renorm(F);
    const M2<T> Q = GP::scan_fwd(F, lane);
    shoot_m = GP::bcast_last(Q.a, lane); shoot_e = GP::bcast_last_i(Q.e, lane);
    T u0 = dpp_t<0x138, 0xF>(T(1), Q.a);   // wave_shr:1
    T um = dpp_t<0x138, 0xF>(T(0), Q.c);
    int eu = dpp_i<0x138, 0xF>(0, Q.e);
    if constexpr (P < 64) { const bool first = (lg == 0); u0 = first ? T(1) : u0; um = first ? T(0) : um; eu = first ? 0 : eu; }
    Eu = eu;
    zu_m1 = um;
    if constexpr (STORE) u0_in = u0;
    T zc = u0, zp = um;
    // (sign changes counted from ONE incoming pair per lane; the step into the next lane's first row belongs to that
    //  lane: see WaveSolver::sweep_fwd)
    const int ncount = (has_last ? M : M - 1) - (lg == P - 1 ? 0 : 1);
    // The sign bits of um, u0, z_1 .. z_M are shifted into ONE word (v_alignbit: one instruction per row instead of the xor /
    // compare / add of a per-row test); the sign changes are the set bits of bits ^ (bits >> 1) under the mask of the steps
    // this lane counts: step i (z_i -> z_{i+1}) sits at bit M - 1 - i, the incoming pair (um -> u0; the first lane of a group
    // holds (0, 1): no change) at bit M.  Steps M - 2 and M - 1 count only if i < ncount.
    static_assert(M + 2 <= 32, "sign word");
    unsigned bits = (unsigned)__builtin_amdgcn_alignbit(sign_word(um) >> 31, sign_word(u0), 31);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const T t = xfma(-sig, Ph[i], D[i]);
      const bool act = (i < M - 1) || has_last;
      const T zn = xfma(-t, zc, -zp);
      bits = (unsigned)__builtin_amdgcn_alignbit(bits, sign_word(zn), 31);     // (bits << 1) | sign(zn)
      if (act) { zp = zc; zc = zn; }
    }
    const unsigned valid = ((1u << (M + 1)) - 1u) & ~((ncount < M ? 1u : 0u) | (ncount < M - 1 ? 2u : 0u));
    const int count = __builtin_popcount((bits ^ (bits >> 1)) & valid);
    return GP::sum_i(count, lane);
  }

  __device__ __forceinline__ void sweep_bwd(T sig) {
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
    B.a = bA * kap; B.b = bB * ikap; B.c = bAn * kap; B.d = bBn * ikap; B.e = 0;
    renorm(B);
    const M2<T> Q = GP::scan_bwd(B, lane);
    T wp = dpp_t<0x130, 0xF>(T(1), Q.a);   // wave_shl:1
    T wq = dpp_t<0x130, 0xF>(T(0), Q.c);
    int ew = dpp_i<0x130, 0xF>(0, Q.e);
    if constexpr (P < 64) { const bool last = (lg == P - 1); wp = last ? T(1) : wp; wq = last ? T(0) : wq; ew = last ? 0 : ew; }
    Ew = ew;
    T wc = wp * kap, wn = wq * ikap;
    zw_p1 = wn;
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      const T t = xfma(-sig, Ph[i], D[i]);
      const bool act = (i < M - 1) || has_last;
      zw[i] = act ? wc : wn;
      const T z2 = xfma(-t, wc, -wn);
      if (act) { wn = wc; wc = z2; }
    }
  }

  // twisted estimate per group (see WaveSolver::twisted); returns rho, group-replicated.  CNT: also extra_above (group-replicated)
  int extra_above;
  template <bool CNT = false>
  __device__ __forceinline__ T twisted(T sig) {
    // pass A: replay the forward solution; per-lane candidate for the twist row k = argmax f |u w| together with
    // the entries around it (so that no dynamically indexed array access is needed afterwards)
    T best = T(0);
    int bi = 0;
    T zu_b = T(1), zum_b = T(0), zw_b = T(1), zwp_b = T(0), t_b = T(0);
    {
      T zc = u0_in, zp = zu_m1;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const T t = xfma(-sig, Ph[i], D[i]);
        const bool act = (i < M - 1) || has_last;
        const T a = act ? xabs(Ph[i] * (zc * zw[i])) : T(0);   // f-weighted |u w|: any row with a large product will do
        const bool better = a > best;
        best = better ? a : best; bi = better ? i : bi;
        zu_b = better ? zc : zu_b; zum_b = better ? zp : zum_b; t_b = better ? t : t_b;
        zw_b = better ? zw[i] : zw_b; zwp_b = better ? (i == M - 1 ? zw_p1 : zw[i < M - 1 ? i + 1 : M - 1]) : zwp_b;
        const T zn = xfma(-t, zc, -zp);
        if (act) { zp = zc; zc = zn; }
      }
    }
    const int ex = fexp(best);
    T key = (best > T(0) && finite_of(best)) ? T(ex + Eu + Ew) + xldexp(best, -ex) : -T(1e30);
    int Lk;   // group-replicated lane id of the twist row's owner
    if constexpr (sizeof(T) == 8) {
      key = __hiloint2double(__double2hiint(key), (__double2loint(key) & ~63) | lane);
      Lk = __double2loint(GP::max(key, lane)) & 63;
    } else {
      key = __int_as_float((__float_as_int(key) & ~63) | lane);
      Lk = __float_as_int(GP::max(key, lane)) & 63;
    }
    // per group: fetch the owner's candidate entries through scalar reads, keep them in this group's lanes
    T zu_k = T(1), zw_k = T(1), t_k = T(0), um1 = T(0), wp1 = T(0);
    int ik = 0, Euk = 0, Ewk = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int sLk = readlane_i(Lk, g * P);
      const bool mine = (gid == g);
      const T a_zu = readlane_t(zu_b, sLk), a_zw = readlane_t(zw_b, sLk), a_t = readlane_t(t_b, sLk);
      const T a_um1 = readlane_t(zum_b, sLk), a_wp1 = readlane_t(zwp_b, sLk);
      const int a_ik = readlane_i(bi, sLk), a_Eu = readlane_i(Eu, sLk), a_Ew = readlane_i(Ew, sLk);
      zu_k = mine ? a_zu : zu_k; zw_k = mine ? a_zw : zw_k; t_k = mine ? a_t : t_k;
      um1 = mine ? a_um1 : um1; wp1 = mine ? a_wp1 : wp1;
      ik = mine ? a_ik : ik; Euk = mine ? a_Eu : Euk; Ewk = mine ? a_Ew : Ewk;
    }
    const T uw = zu_k * zw_k;
    const T num = xfma(um1, zw_k, xfma(t_k, uw, wp1 * zu_k));
    // rho = sig + gamma_k / sum(f x^2) with x = s z / (s_k z_k): the scaling s_k cancels,
    //   gamma_k = num / (s_k^2 u w),  sum f x^2 = sum(Ph zhat^2) / s_k^2,  zhat = z / z_k
    int du = Eu - Euk, dw = Ew - Ewk;
    du = du > 1000 ? 1000 : (du < -2000 ? -2000 : du);
    dw = dw > 1000 ? 1000 : (dw < -2000 ? -2000 : dw);
    fu = xldexp(fast_rcp(zu_k), du);
    fw = xldexp(fast_rcp(zw_k), dw);
    thr = (lane < Lk) ? M : ((lane > Lk) ? -1 : ik);
    // pass B: sum f x^2 over the twisted vector (forward solution replayed again)
    T acc = T(0);
    static_assert(M + 2 <= 32, "sign word");
    unsigned su = sign_word(zu_m1) >> 31, sw = 0;      // CNT: sign bits of (u_-1, u_0 .. u_{M-1}) and of (w_0 .. w_{M-1}): WaveSolver::twisted
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
          su = (unsigned)__builtin_amdgcn_alignbit(su, sign_word(zc), 31);
          sw = (unsigned)__builtin_amdgcn_alignbit(sw, sign_word(zw[i]), 31);
        }
        const T zn = xfma(-t, zc, -zp);
        if (act) { zp = zc; zc = zn; }
      }
    }
    if constexpr (CNT) {
      const int last = has_last ? M - 1 : M - 2;
      const int iu = thr < last ? thr : last;
      const unsigned mu = iu >= 0 ? (((1u << (iu + 1)) - 1u) << (M - 1 - iu)) : 0u;
      const int iw = thr > 0 ? thr : 0;
      const unsigned mw = iw <= last ? (((1u << (last - iw + 1)) - 1u) << (M - last)) : 0u;
      const unsigned sw2 = (sw << 1) | (sign_word(zw_p1) >> 31);
      extra_above = GP::sum_i(__builtin_popcount((su ^ (su >> 1)) & mu) + __builtin_popcount((sw2 ^ (sw2 << 1)) & mw), lane);
    }
    sig_vec = sig;
    const T tot = GP::sum(acc, lane);
    return sig + num * fast_rcp(uw * tot);
  }

  // eigenvector entries of this lane's rows up to a common factor per group (normalised by the caller);
  // the diagonal scaling s is rebuilt here from g (s_0 = 1, s_{i+1} = 1/(e_{i+1} s_i)), the forward solution is
  // replayed at the shift of the last twisted() call
  template <class Src>
  __device__ __forceinline__ void assemble(const Src& src, int N, T h, T (&x)[M]) {
    const int a = rows_start(lg, N - 2);
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
        const T gnext = src.g(a + i + 2);
        sc = fast_rcp(T(0.5) * (gcur + gnext) * ih2 * sc);
        gcur = gnext;
        const T zn = xfma(-xfma(-sig_vec, Ph[i], D[i]), zc, -zp);
        zp = zc; zc = zn;
      }
    }
  }

  // The shift iteration of WaveSolver::solve (read its comment first), vectorised over the groups of the wave:
  // every quantity is group-uniform and lives in VGPRs, the state machine is written with selects, and the wave
  // loops until every group is done (a finished group keeps re-evaluating its frozen shift).  Each iteration is
  // exactly ONE forward sweep for all groups, so the only divergence left is the iteration count (15 +- 2 on
  // NCSX-like systems).  `bad` groups are parked as done.  Returns lam (group-replicated).
  // Optional warm start per group (see WaveSolver::solve): `guess` = eigenvalue of a nearby problem, `width` its
  // expected error (both group-replicated).  The first shift is guess + width and walks up geometrically until a
  // count certifies an upper bound; the next one is guess - width.  Certified exactly like the cold solve.
  // CHK: the closing bracket's consistency check of WaveSolver::solve<true> (read its comment), per group
  bool suspect, closed;
  T rho_last;
  __device__ __forceinline__ int why() const {           // (mark-only mode: WaveSolver::why, per group)
    const T tol = T(64) * Eps<T>::v * normA, rho = rho_last;
    const T dist = xmax(xmax(lo - rho, rho - hi), T(0));
    int bk = dist > T(0) ? expo_of(dist) - expo_of(tol) + 8 : 0;
    bk = finite_of(rho) ? (bk < 1 ? (dist > T(0) ? 1 : 0) : (bk > 63 ? 63 : bk)) : 63;
    return closed ? (bk | (extra_above != 0 ? 64 : 0)) : 0;
  }
  T chk_slack;     // (set by setup())
  template <bool CHK = false>
  __device__ __forceinline__ T solve(bool bad, int& iters_out, int& status_out, bool warm = false, T guess = T(0),
                                     T width = T(0)) {
    using WS = WaveSolver<T, M>;
    using Pt = typename WS::Pt;
    const T tol = (sizeof(T) == 8 ? T(64) : T(8)) * Eps<T>::v * normA;
    const bool warm_ok = warm && finite_of(guess) && width > T(0) && guess + width < hi && guess + width > lo;
    bool expand = warm_ok, try_below = warm_ok;
    T wstep = T(4) * width;
    const T g_below = guess - width;
    T sig = warm_ok ? guess + width : T(0.5) * (lo + hi), sig_prev = sig, lam = hi;
    T off_up = tol, off_dn = tol, rho_trust = hi;
    int aimed = 0, it = 0, lg_prev = 0;
    bool lo1 = false, hi_f = false, old_ok = false, was_interp = false, conv = false;
    Pt Plo{lo, T(0), 0}, Phi{hi, T(0), 0}, Pold{hi, T(0), 0};
    bool done = bad;
    constexpr int kMaxIt = 200;
    int guard = 0;
    auto sel = [](bool c, const Pt& a, const Pt& b) { return Pt{c ? a.x : b.x, c ? a.m : b.m, c ? a.e : b.e}; };
    while (__any(!done) && guard < kMaxIt) {
      ++guard;
      const int C = sweep_fwd<true>(sig);      // (the incoming pair is kept: the last sweep of every group is the one its eigenvector replays)
      const bool act = !done;
      it += act ? 1 : 0;
      const Pt cur{sig, shoot_m, shoot_e};
      // every shift lies strictly inside (lo, hi): each count moves one end of the bracket
      const bool uh = act && C == 0, ul = act && C != 0;
      const bool t1 = uh && hi_f, t2 = ul && lo1;
      Pold = sel(t1, Phi, sel(t2, Plo, Pold)); old_ok = old_ok || t1 || t2;
      hi = uh ? sig : hi; Phi = sel(uh, cur, Phi); hi_f = hi_f || uh;
      lo = ul ? sig : lo; Plo = sel(ul, cur, Plo); lo1 = ul ? (C == 1) : lo1;
      const T prevstep = xabs(sig - sig_prev);
      sig_prev = sig;
      // did the interpolation step that produced this shift pay off?
      const int e_now = expo_of(shoot_m);
      const int red = lg_prev - (e_now < -(1 << 27) ? e_now : shoot_e + e_now);
      conv = was_interp ? (red >= 4) : (aimed != 0 ? conv : false);
      const bool force_bis = was_interp && red < 1;
      const bool cert = aimed != 0;
      off_up = (act && aimed > 0 && C != 0) ? T(2) * off_up : off_up;
      off_dn = (act && aimed < 0 && C == 0) ? T(2) * off_dn : off_dn;
      const bool collapsed = (hi - lo) <= T(4) * tol;
      done = done || (act && collapsed);
      const bool go = act && !collapsed;
      T rho = rho_trust;
      bool ok = go && cert, near = ok;
      const bool want_i = go && lo1 && !cert && hi_f && !force_bis;
      if (__any(want_i)) {
        const int lg_lo = WS::lg2_of(Plo), lg_hi = WS::lg2_of(Phi);
        const bool b_is_lo = lg_lo <= lg_hi;
        const Pt b = sel(b_is_lo, Plo, Phi), a = sel(b_is_lo, Phi, Plo);
        const bool use_o = old_ok && Pold.x != a.x && Pold.x != b.x;
        T r;
        const bool got = WS::interpolate_lazy(Pold, use_o, a, b, lo, hi, want_i, r);
        const T q = T(0.25) * (T(3) * a.x + b.x);
        const bool inside = r >= xmin(q, b.x) && r <= xmax(q, b.x);
        const T stepb = xabs(r - b.x);
        const bool nr = was_interp && conv && stepb <= T(4096) * tol;
        const bool acc = want_i && got && inside &&
                         (nr || (stepb < T(0.5) * prevstep && stepb >= T(9.5367431640625e-07) * prevstep));
        lg_prev = acc ? (b_is_lo ? lg_lo : lg_hi) : lg_prev;
        rho = acc ? r : rho; ok = ok || acc; near = near || (acc && nr);
      }
      // next shift
      const T mid = T(0.5) * (lo + hi);
      // (the certificate placement and the warm-start prologue are skipped while no group of the wave is in that state: the
      //  decision code between two sweeps costs these kernels as many instructions as the sweep itself)
      const bool trust = ok && near;
      bool c_up = false, aim_ok = false;
      T cand = mid;
      if (__any(trust)) {
        const bool fresh = trust && !cert && xabs(rho - rho_trust) > T(4096) * tol;
        off_up = fresh ? tol : off_up; off_dn = fresh ? tol : off_dn;
        rho_trust = trust ? rho : rho_trust;
        const T up = xmax(rho, lo), dn = xmin(rho, hi);
        c_up = hi > up + T(2) * off_up;
        const bool c_dn = lo < dn - T(2) * off_dn;
        cand = c_up ? up + off_up : dn - off_dn;
        aim_ok = trust && (c_up || c_dn) && cand > lo && cand < hi;
      }
      const bool interp_now = ok && !near;
      const T nxt = aim_ok ? cand : (interp_now ? rho : mid);
      // warm start prologue: keep walking up while the count says lam_max is still above; afterwards, while lam_max
      // is not yet isolated, try the lower end of the guessed interval once
      bool ovr = false;
      if (__any(expand || try_below)) {
        const bool exp_go = go && expand && C != 0 && sig + wstep < hi;
        const bool locate = go && !exp_go && !lo1;
        const bool tb = locate && try_below && g_below > lo && g_below < hi;
        ovr = exp_go || tb;
        sig = go ? (exp_go ? sig + wstep : (tb ? g_below : nxt)) : sig;
        wstep = exp_go ? T(4) * wstep : wstep;
        expand = exp_go;
        try_below = locate ? false : try_below;
      } else {
        sig = go ? nxt : sig;
      }
      aimed = (go && !ovr) ? (aim_ok ? (c_up ? 1 : -1) : 0) : 0;
      was_interp = go && interp_now && !ovr;
    }
    // eigenvector and Rayleigh-quotient polish at each group's last shift.  A group that is done has swept its final shift last (it
    // keeps re-evaluating the frozen shift while the wave's other groups finish), so the forward solution is already in place; only
    // a wave that left on the iteration cap holds a shift it has not swept yet.  (Until round 4 every solve paid one more forward
    // sweep here: 1 of 11.)
    // (guard == 0: every group of the wave was flagged before the first sweep -- the sweep still runs once, so that the backward
    // sweep and the outputs of such systems are built from written state and repeat bit for bit)
    if (__any(!done) || guard == 0) sweep_fwd<true>(sig);
    sweep_bwd(sig);
    const T rho = twisted<CHK>(sig);
    iters_out = it;
    status_out = bad ? 2 : (done ? 0 : 1);
    lam = done ? ((finite_of(rho) && rho >= lo && rho <= hi) ? rho : T(0.5) * (lo + hi)) : sig;
    if constexpr (CHK) {
      const bool rho_in = finite_of(rho) && rho >= lo - chk_slack && rho <= hi + chk_slack;
      suspect = done && !bad && (!rho_in || extra_above != 0);
      rho_last = rho;
      lam = (done && rho_in) ? xmin(xmax(rho, lo), hi) : lam;
      closed = done && !bad;
    }
    return lam;
  }
};

}  // namespace ibs
