// Kernels of the ballooning hot path, one translation unit per rows-per-lane value M
// (compiled with -DIBS_M=<m>; see Makefile).  Launchers are registered in a table the C-ABI
// layer (ibs_api.hip) dispatches on M = ceil((N-2)/64).
//
//  k_solve_gcf   : raw (g, c, f) systems            -> lam, gam, [X, dX]        (config 5 / C1)
//  k_gamma_scan  : field-line geometry x theta0 grid -> gam, lam, [X, dX, dgam/dtheta0]
//                  (ball_scan.py:248-275 inner loops, utils.py:1556-1624, utils.py:1666-1680)
//  k_sturm_count : inertia of T - lam F at given shifts (pins the s-alpha stability test)
#include "ibs_wave.hpp"
#include "ibs_launch.hpp"
#include "ibs_refine.hpp"
#include <type_traits>

#ifndef IBS_M
#error "compile with -DIBS_M=<rows per lane>"
#endif
// fewest rows per lane for which the read-from-global-memory forms are built: k_solve_gcf_direct (FP64 solver) pays where the
// staging limits the occupancy (N > 578; below, the sub-wave forms are faster: tools/bench_forms.py), the all-FP32 eigenvalue-only
// form k_solve_gcf_f32lam_direct on every big batch
#ifndef IBS_DIRECT_MIN_M
#define IBS_DIRECT_MIN_M 9
#endif
#ifndef IBS_F32LAM_DIRECT_MIN_M
#define IBS_F32LAM_DIRECT_MIN_M 3
#endif

namespace ibs {

// ---------------------------------------------------------------- coefficient sources (LDS-backed)
template <typename T>
struct SrcGCF {
  static constexpr bool kHasGh = false;
  const T* gs; const T* cs; const T* fs;
  __device__ __forceinline__ T g(int j) const { return gs[lpos(j)]; }
  __device__ __forceinline__ T c(int j) const { return cs[lpos(j)]; }
  __device__ __forceinline__ T f(int j) const { return fs[lpos(j)]; }
  __device__ __forceinline__ void gcf(int j, T& g_, T& c_, T& f_) const { const int q = lpos(j); g_ = gs[q]; c_ = cs[q]; f_ = fs[q]; }
};

// growth-rate stage of k_solve_gcf: g, c from the padded LDS rows, f (whose slot now holds X) from global memory
template <typename T>
struct SrcGCFG {
  static constexpr bool kHasGh = false;
  const T* gs; const T* cs; const T* fg;
  __device__ __forceinline__ T g(int j) const { return gs[lpos(j)]; }
  __device__ __forceinline__ T c(int j) const { return cs[lpos(j)]; }
  __device__ __forceinline__ T f(int j) const { return fg[j]; }
  __device__ __forceinline__ void gcf(int j, T& g_, T& c_, T& f_) const { const int q = lpos(j); g_ = gs[q]; c_ = cs[q]; f_ = fg[j]; }
};

// geometry of one field line staged in LDS as 7 derived arrays (shared by all theta0 of the line):
//   A1 = |gradpar|/B, A3 = 1/(|gradpar| B^3), C0 = -dPdrho cvdrift/(|gradpar| B),
//   C1 = -dPdrho cvdrift0/(|gradpar| B), G0 = gds2, G1 = gds21, G2 = gds22
// so that for a given theta0 (ball_scan.py:267-268, utils.py:1560-1562)
//   gds2_fth = G0 + 2 theta0 G1 + theta0^2 G2,  g = A1 gds2_fth,  f = A3 gds2_fth,  c = C0 + theta0 C1
// raw (g, c, f) with a separately supplied half-grid g (regridded non-uniform input, utils.py:1567-1576)
template <typename T>
struct SrcGCFH {
  static constexpr bool kHasGh = true;
  const T* gs; const T* cs; const T* fs; const T* ghs;   // ghs: global memory, N-1 values
  __device__ __forceinline__ T g(int j) const { return gs[lpos(j)]; }
  __device__ __forceinline__ T c(int j) const { return cs[lpos(j)]; }
  __device__ __forceinline__ T f(int j) const { return fs[lpos(j)]; }
  __device__ __forceinline__ T gh(int k) const { return ghs[k]; }
  __device__ __forceinline__ void gcf(int j, T& g_, T& c_, T& f_) const { const int q = lpos(j); g_ = gs[q]; c_ = cs[q]; f_ = fs[q]; }
};

// FP32 rows staged in LDS read as FP64 (the certificate of the all-FP32 kernel k_solve_gcf<float, M>)
template <typename TS>
struct SrcGCFF {
  static constexpr bool kHasGh = false;
  const TS* gs; const TS* cs; const TS* fs;
  __device__ __forceinline__ double g(int j) const { return (double)gs[lpos(j)]; }
  __device__ __forceinline__ double c(int j) const { return (double)cs[lpos(j)]; }
  __device__ __forceinline__ double f(int j) const { return (double)fs[lpos(j)]; }
};

template <typename T, bool SCALED = false>
struct SrcGeo {
  static constexpr bool kHasGh = false;
  const T* A1; const T* A3; const T* C0; const T* C1; const T* G0; const T* G1; const T* G2;
  T th0, two_th0, th0sq;
  T cs = T(1);       // SCALED: common factor of c (C0, C1 staged without -dPdrho: k_refine_eval)
  // (rows are padded: element j at lpos(j))
  __device__ __forceinline__ T sc(T v) const { if constexpr (SCALED) return cs * v; else return v; }
  __device__ __forceinline__ T gd(int j) const { const int q = lpos(j); return G0[q] + two_th0 * G1[q] + th0sq * G2[q]; }
  __device__ __forceinline__ T g(int j) const { return A1[lpos(j)] * gd(j); }
  __device__ __forceinline__ T c(int j) const { const int q = lpos(j); return sc(C0[q] + th0 * C1[q]); }
  __device__ __forceinline__ T f(int j) const { return A3[lpos(j)] * gd(j); }
  // d/dtheta0 tangents (utils.py:1669-1673)
  __device__ __forceinline__ T gdp(int j) const { const int q = lpos(j); return T(2) * G1[q] + two_th0 * G2[q]; }
  __device__ __forceinline__ T g_t(int j) const { return A1[lpos(j)] * gdp(j); }
  __device__ __forceinline__ T c_t(int j) const { return sc(C1[lpos(j)]); }
  __device__ __forceinline__ T f_t(int j) const { return A3[lpos(j)] * gdp(j); }
  // (g, c, f) and their theta0 tangents at one point with each LDS value read once
  __device__ __forceinline__ void gcf(int j, T& g_, T& c_, T& f_) const {
    const int q = lpos(j);
    const T d = G0[q] + two_th0 * G1[q] + th0sq * G2[q];
    g_ = A1[q] * d; c_ = sc(C0[q] + th0 * C1[q]); f_ = A3[q] * d;
  }
  __device__ __forceinline__ void gcf_with_tangent(int j, T& g_, T& c_, T& f_, T& gt, T& ct, T& ft) const {
    const int q = lpos(j);
    const T g1 = G1[q], g2 = G2[q], a1 = A1[q], a3 = A3[q], c1 = C1[q];
    const T d = G0[q] + two_th0 * g1 + th0sq * g2, dp = T(2) * g1 + two_th0 * g2;
    g_ = a1 * d; c_ = sc(C0[q] + th0 * c1); f_ = a3 * d;
    gt = a1 * dp; ct = sc(c1); ft = a3 * dp;
  }
};


struct NoTangent {};
template <typename U> struct TypeId { using type = U; };     // (keeps an output type out of template argument deduction)

// alpha-tangent of (g, c, f): central difference between the field lines at alpha +- del_alpha/2,
// each with ITS OWN dPdrho (utils.py:1683-1718); lines are read from global memory.
template <typename T>
struct AlphaTangent {
  const T* l; const T* r; long ld;   // [8][ld]: bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift
  T mdP_l, mdP_r, th0, two_th0, th0sq, inv_da;
  __device__ __forceinline__ void gcf(const T* p, T mdP, int j, T& g, T& c, T& f) const {
    const T B = p[j], gp = xabs(p[ld + j]);
    const T cv = p[2 * ld + j] + th0 * p[3 * ld + j];                                   // utils.py:1692 / 1704
    const T gd = p[4 * ld + j] + two_th0 * p[5 * ld + j] + th0sq * p[6 * ld + j];     // utils.py:1693 / 1705
    g = gp * gd / B; c = mdP * cv / (gp * B); f = gd / (B * B) / (gp * B);              // utils.py:1707-1713
  }
  __device__ __forceinline__ void at(int j, T& ga, T& ca, T& fa) const {
    T g1, c1, f1, g2, c2, f2;
    gcf(r, mdP_r, j, g1, c1, f1); gcf(l, mdP_l, j, g2, c2, f2);
    ga = (g1 - g2) * inv_da; ca = (c1 - c2) * inv_da; fa = (f1 - f2) * inv_da;         // utils.py:1716-1718
  }
};

// shared tail: eigenvector -> X in LDS -> growth rate (and optional outputs)
template <typename T, int M, class Src, bool HF, class Tan = NoTangent, typename TO = T>
__device__ __forceinline__ void finish(WaveSolver<T, M>& ws, const Src& src, int N, T h, T* Xs,
                                       T lam, const SolveInfo& inf, long sys, typename TypeId<TO>::type* lam_out,
                                       typename TypeId<TO>::type* gam_out, typename TypeId<TO>::type* X_out,
                                       typename TypeId<TO>::type* dX_out, typename TypeId<TO>::type* dth0_out, int* info_out,
                                       const Tan* tan = nullptr, typename TypeId<TO>::type* dalpha_out = nullptr,
                                       const T* gh = nullptr) {
  const int lane = ws.lane;
  const int n = N - 2;
  T x[M];
  ws.assemble(src, gh, N, h, x);
  T m = T(0);
#pragma unroll
  for (int i = 0; i < M; ++i) m = xmax(m, xabs(x[i]));
  m = uniform(wave_max(m));
  const int a = WaveSolver<T, M>::rows_start(lane, n);
  const T rm = T(1) / m;
#pragma unroll
  for (int i = 0; i < M; ++i)
    if ((i < M - 1) || ws.has_last) Xs[lpos(a + i + 1)] = x[i] * rm;      // utils.py:1605 (v / max|v|)
  if (lane == 0) { Xs[lpos(0)] = T(0); Xs[lpos(N - 1)] = T(0); }          // utils.py:1607-1608
  wave_lds_sync();                                                  // Xs is private to this wave
  IBS_PROBE_AT(13);
  // dX (utils.py:1610-1616) in one branch-free form: with the neighbour indices clamped to [0, N-1],
  //   dX = A (X[j+1] - X[j-1]) + B (X[j+2] - X[j-2]),
  //   (A, B) = (2/3, -1/12)/h inside, (1/2, 0)/h at j = 1, N-2, (2, -1/2)/h at j = 0, N-1
  // (the one-sided end formulas -3/2 X0 + 2 X1 - 1/2 X2 and 1/2 X[N-3] - 2 X[N-2] + 3/2 X[N-1] regrouped).
  const T ih = T(1) / h;
  const T A_in = (T(2) / T(3)) * ih, B_in = -ih / T(12), A_e1 = T(0.5) * ih, A_e0 = T(2) * ih, B_e0 = T(-0.5) * ih;
  bool do_hf = false;
  if constexpr (HF) do_hf = dth0_out != nullptr;
  T y0 = T(0), y1 = T(0), hc = T(0), hg = T(0), hf = T(0), ac = T(0), ag = T(0), af = T(0);
#pragma unroll 3
  for (int j0 = 0; j0 < N; j0 += kWave) {
    const int j = j0 + lane;
    const bool in = j < N;
    const int jc = in ? j : N - 1;
    const int jm1 = jc > 0 ? jc - 1 : 0, jm2 = jc > 1 ? jc - 2 : 0;
    const int jp1 = jc < N - 1 ? jc + 1 : N - 1, jp2 = jc < N - 2 ? jc + 2 : N - 1;
    const T X = Xs[lpos(jc)];
    const T d1 = Xs[lpos(jp1)] - Xs[lpos(jm1)], d2 = Xs[lpos(jp2)] - Xs[lpos(jm2)];
    const bool end0 = (jc == 0) || (jc == N - 1), end1 = (jc == 1) || (jc == N - 2);
    const T A = end0 ? A_e0 : (end1 ? A_e1 : A_in), B = end0 ? B_e0 : (end1 ? T(0) : B_in);
    const T dX = xfma(A, d1, B * d2);
    const T w = in ? (end0 ? T(1) : ((jc & 1) ? T(4) : T(2))) : T(0);      // Simpson weights (the 1/3 cancels)
    const T X2 = w * (X * X), dX2 = w * (dX * dX);
    T g_, c_, f_;
    if (do_hf) {
      if constexpr (HF) {
        T gt, ct, ft;
        src.gcf_with_tangent(jc, g_, c_, f_, gt, ct, ft);
        hc = xfma(ct, X2, hc); hg = xfma(gt, dX2, hg); hf = xfma(ft, X2, hf);
      }
    } else {
      src.gcf(jc, g_, c_, f_);
    }
    y0 += c_ * X2 - g_ * dX2;                                        // utils.py:1618
    y1 = xfma(f_, X2, y1);                                           // utils.py:1619
    if constexpr (!std::is_same<Tan, NoTangent>::value) {
      T ga, ca, fa;
      tan->at(jc, ga, ca, fa);
      ac = xfma(ca, X2, ac); ag = xfma(ga, dX2, ag); af = xfma(fa, X2, af);
    }
    if (X_out && in) X_out[sys * N + j] = (TO)X;
    if (dX_out && in) dX_out[sys * N + j] = (TO)dX;
  }
  IBS_PROBE_AT(14);
  y0 = wave_sum(y0); y1 = wave_sum(y1);
  const T gam = y0 / y1;                                             // utils.py:1621 (the 1/3 of Simpson cancels)
  if constexpr (HF) {
    if (do_hf) {
      hc = wave_sum(hc); hg = wave_sum(hg); hf = wave_sum(hf);
      const T jac = hc / y1 - hg / y1 - gam * hf / y1;               // utils.py:1676-1680
      if (lane == 0) dth0_out[sys] = (TO)jac;
    }
  }
  if constexpr (!std::is_same<Tan, NoTangent>::value) {
    ac = wave_sum(ac); ag = wave_sum(ag); af = wave_sum(af);
    const T jac = ac / y1 - ag / y1 - gam * af / y1;                 // utils.py:1721-1725
    if (lane == 0 && dalpha_out) dalpha_out[sys] = (TO)jac;
  }
  if (lane == 0) {
    if (lam_out) lam_out[sys] = (TO)lam;
    // (write-through, agent scope: the fused per-surface reduction of k_gamma_scan reads it from another CU)
    if (gam_out) __hip_atomic_store(gam_out + sys, (TO)gam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (info_out) info_out[sys] = inf.iters | (inf.status << 16);
  }
}

// one grid point of the Simpson sums of utils.py:1618-1619 (and, with WT, of the theta0-tangent sums of
// utils.py:1676-1680); w = Simpson weight
template <typename T, class Src, bool WT>
__device__ __forceinline__ void simpson_point(const Src& src, int j, T w, T X, T dX, T& y0, T& y1, T& hc, T& hg, T& hf) {
  const T X2 = w * (X * X), dX2 = w * (dX * dX);
  T g_, c_, f_;
  if constexpr (WT) {
    T gt, ct, ft;
    src.gcf_with_tangent(j, g_, c_, f_, gt, ct, ft);
    hc = xfma(ct, X2, hc); hg = xfma(gt, dX2, hg); hf = xfma(ft, X2, hf);
  } else {
    src.gcf(j, g_, c_, f_);
  }
  y0 += c_ * X2 - g_ * dX2;                                          // utils.py:1618
  y1 = xfma(f_, X2, y1);                                             // utils.py:1619
}

// Growth rate without an LDS round trip: the eigenfunction stays in the lanes' row chunks, the four stencil neighbours beyond a chunk come from the adjacent lanes (wave_shr / wave_shl),
// and every lane sums its own rows of the Simpson integrals; the two end points j = 0, N-1 (X = 0, one-sided dX)
// are added by the first and last lane.  Same per-point arithmetic as finish() (utils.py:1601-1621).  X and dX, when
// requested, go through the wave's LDS row Xs (which may alias an LDS array of `src`: it is written after the
// sums) to be stored coalesced.
// PASSES > 1 (row-streamed raw systems, k_solve_gcf_rows): the Simpson sums are separable in (g, c, f), so they are
// accumulated in PASSES sweeps over the rows, `src.begin_pass(p)` putting the p-th coefficient array into the wave's one
// LDS row in between (the other two read as zero).
template <typename T, int M, class Src, bool HF, int PASSES = 1, typename TO = T>
__device__ __forceinline__ void finish_chunk(WaveSolver<T, M>& ws, Src& src, int N, T h, T* Xs, T lam,
                                             const SolveInfo& inf, long sys, typename TypeId<TO>::type* lam_out,
                                             typename TypeId<TO>::type* gam_out, typename TypeId<TO>::type* X_out,
                                             typename TypeId<TO>::type* dX_out, typename TypeId<TO>::type* dth0_out,
                                             int* info_out, const T* gh = nullptr) {
  static_assert(M >= 3, "the halo exchange takes two rows from each neighbour lane");
  const int lane = ws.lane;
  const int n = N - 2;
  const bool hl = ws.has_last;
  T x[M];
  ws.assemble(src, gh, N, h, x);
  T m = T(0);
#pragma unroll
  for (int i = 0; i < M; ++i) m = xmax(m, xabs(x[i]));
  m = uniform(wave_max(m));
  const T rm = T(1) / m;
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] *= rm;                                   // utils.py:1605 (v / max|v|)
  int a = WaveSolver<T, M>::rows_start(lane, n);
  // The coefficient rows read below were read once already by setup(), at the same LDS addresses: unless the row
  // index is made opaque here the compiler keeps all those addresses alive (spilled to scratch: 200 VGPRs at
  // M = 16) across the whole shift iteration instead of recomputing them.
  asm volatile("" : "+v"(a));
  // xe[k] = X at grid point a + k - 1 (k = 0, 1: the previous lane's last two rows; beyond this lane's rows: the
  // next lane's first two).  The zero end points X[0], X[N-1] (and the clamped X[-1], X[N]) are the DPP fill value.
  const T lastv = hl ? x[M - 1] : x[M - 2], last2 = hl ? x[M - 2] : x[M - 3];
  const T xp1 = dpp_t<0x130, 0xF>(T(0), x[0]), xp2 = dpp_t<0x130, 0xF>(T(0), x[1]);    // wave_shl:1
  T xe[M + 4];
  xe[0] = dpp_t<0x138, 0xF>(T(0), last2); xe[1] = dpp_t<0x138, 0xF>(T(0), lastv);       // wave_shr:1
#pragma unroll
  for (int i = 0; i < M - 1; ++i) xe[i + 2] = x[i];
  xe[M + 1] = hl ? x[M - 1] : xp1; xe[M + 2] = hl ? xp1 : xp2; xe[M + 3] = xp2;
  IBS_PROBE_AT(13);
  const T ih = T(1) / h;
  const T A_in = (T(2) / T(3)) * ih, B_in = -ih / T(12), A_e1 = T(0.5) * ih, A_e0 = T(2) * ih, B_e0 = T(-0.5) * ih;
  bool do_hf = false;
  if constexpr (HF) do_hf = dth0_out != nullptr;
  T y0 = T(0), y1 = T(0), hc = T(0), hg = T(0), hf = T(0);
  const T w_even = ((a + 1) & 1) ? T(4) : T(2), w_odd = ((a + 1) & 1) ? T(2) : T(4);   // Simpson weight of slot i
  const int i_end = (lane == kWave - 1) ? (hl ? M - 1 : M - 2) : -1;                   // slot of grid point N-2
  // one-sided dX of the two end points (utils.py:1610, 1614), formed now so that the rows it needs are not kept
  // alive across the row loop
  const T dX_end = xfma(A_e0, lane == 0 ? x[0] : -lastv, B_e0 * (lane == 0 ? x[1] : -last2));
  auto all_points = [&](auto with_tangent) {
    if (lane == 0 || lane == kWave - 1)                                                // j = 0, N-1
      simpson_point<T, Src, HF && decltype(with_tangent)::value>(src, lane == 0 ? 0 : N - 1, T(1), T(0), dX_end, y0, y1, hc, hg, hf);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const bool act = (i < M - 1) || hl;
      const bool e1 = (i == 0 && lane == 0) || (i == i_end);                           // j = 1, N-2: utils.py:1611, 1613
      const T A = e1 ? A_e1 : A_in, B = e1 ? T(0) : B_in;
      const T dX = xfma(A, xe[i + 3] - xe[i + 1], B * (xe[i + 4] - xe[i]));           // utils.py:1616
      const T w = act ? ((i & 1) ? w_odd : w_even) : T(0);
      simpson_point<T, Src, HF && decltype(with_tangent)::value>(src, a + i + 1, w, xe[i + 2], dX, y0, y1, hc, hg, hf);
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most 4 rows of LDS reads in flight (registers)
    }
  };
  if constexpr (HF) {
    if (do_hf) all_points(std::true_type{});
    else all_points(std::false_type{});
  } else {
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      if constexpr (PASSES > 1) src.begin_pass(pass);
      all_points(std::false_type{});
    }
  }
  IBS_PROBE_AT(14);
  y0 = wave_sum(y0); y1 = wave_sum(y1);
  const T gam = y0 / y1;                                             // utils.py:1621 (the 1/3 of Simpson cancels)
  if constexpr (HF) {
    if (do_hf) {
      hc = wave_sum(hc); hg = wave_sum(hg); hf = wave_sum(hf);
      const T jac = hc / y1 - hg / y1 - gam * hf / y1;               // utils.py:1676-1680
      if (lane == 0) dth0_out[sys] = (TO)jac;
    }
  }
  if (lane == 0) {
    if (lam_out) lam_out[sys] = (TO)lam;
    // (write-through, agent scope: the fused per-surface reduction of k_gamma_scan reads it from another CU)
    if (gam_out) __hip_atomic_store(gam_out + sys, (TO)gam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (info_out) info_out[sys] = inf.iters | (inf.status << 16);
  }
  if (X_out || dX_out) {                   // wave-uniform
    for (int pass = 0; pass < 2; ++pass) {
      TO* out = pass ? dX_out : X_out;
      if (!out) continue;
      wave_lds_sync();                     // the row's previous content (f of k_solve_gcf, or X) has been consumed
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const bool e1 = (i == 0 && lane == 0) || (i == i_end);
        const T A = e1 ? A_e1 : A_in, B = e1 ? T(0) : B_in;
        const T v = pass ? xfma(A, xe[i + 3] - xe[i + 1], B * (xe[i + 4] - xe[i])) : xe[i + 2];
        if ((i < M - 1) || hl) Xs[lpos(a + i + 1)] = v;
      }
      if (lane == 0) Xs[lpos(0)] = pass ? dX_end : T(0);                                        // utils.py:1607, 1610
      if (lane == kWave - 1) Xs[lpos(N - 1)] = pass ? dX_end : T(0);                            // utils.py:1608, 1614
      wave_lds_sync();
      for (int j = lane; j < N; j += kWave) out[sys * N + j] = (TO)Xs[lpos(j)];
    }
  }
}

// rows of one raw system straight from global memory (k_solve_gcf_direct; the division-form re-close of every raw kernel)
template <typename T, typename TI>
struct SrcDirect {
  static constexpr bool kHasGh = false;
  const TI* gg; const TI* cg; const TI* fg;
  __device__ __forceinline__ T g(int j) const { return (T)gg[j]; }
  __device__ __forceinline__ T c(int j) const { return (T)cg[j]; }
  __device__ __forceinline__ T f(int j) const { return (T)fg[j]; }
  __device__ __forceinline__ void gcf(int j, T& g_, T& c_, T& f_) const { g_ = (T)gg[j]; c_ = (T)cg[j]; f_ = (T)fg[j]; }
};
template <typename T>
struct SrcDirectH {      // ... with a caller-supplied half-grid g (ibs_solve_gcfh_f64)
  static constexpr bool kHasGh = true;
  const T* gg; const T* cg; const T* fg; const T* ghg;
  __device__ __forceinline__ T g(int j) const { return gg[j]; }
  __device__ __forceinline__ T c(int j) const { return cg[j]; }
  __device__ __forceinline__ T f(int j) const { return fg[j]; }
  __device__ __forceinline__ T gh(int k) const { return ghg[k]; }
};

// ---------------------------------------------------------------- FP64 raw systems: re-close of a suspect solve (round 6)
// After WaveSolver::solve<true>: a solve whose closing bracket failed the consistency checks is closed again by division-form
// multisection on the original rows `rows` (reclose_division: a few eps ||A||, independent of N).  flags (GcfArgs::flags,
// ibs_set_option "reclose"): bit 0 = re-close, bit 1 = only mark (diagnostics); a suspect system carries the informational status
// bit 3 either way.  Returns true when lam was replaced: the caller then repeats the eigenvector stage at it (redo_vector_at).
template <typename T, int M, class Rows>
__device__ __forceinline__ bool reclose_suspect(const WaveSolver<T, M>& ws, SolveInfo& inf, const Rows& rows, int N, T h, int flags,
                                                T& lam) {
  if (!(flags & 1)) {                                      // mark-only (diagnostics): every polish outside the bracket, and how far
    if (flags != 0 && ws.why() != 0) inf.status |= 8 | (ws.why() << 5);
    return false;
  }
  if (!ws.suspect) return false;                           // (wave-uniform)
  inf.status |= 8;
  int passes;
  T l2;
  const T center = WaveSolver<T, M>::U(finite_of(ws.rho_last)) ? ws.rho_last : lam;
  const bool ok = reclose_division<T, Rows>(rows, N, h, center, ws.normA, ws.lane, l2, passes);
  inf.iters += passes;
  if (!ok) { inf.status |= 1; return false; }
  lam = l2;
  return true;
}

// The big-batch forms do not re-close in place: a suspect system is appended to the launch's list and k_fix_gcf (below) does the rest.
template <typename T>
__device__ __forceinline__ void list_suspect(int* cnt, long* sys_list, double* center, bool valid, bool first_lane, long sys, T rho, T lam) {
  if (cnt && valid && first_lane) {
    const int i = atomicAdd(cnt, 1);
    sys_list[i] = sys;
    center[i] = finite_of(rho) ? (double)rho : (double)lam;
  }
}

// ---------------------------------------------------------------- raw (g, c, f) systems
// block = WPB waves, one system per wave; dynamic LDS = WPB * 3N * sizeof(T): (g, c, f) staged once
// through LDS for the chunked register load; afterwards the eigenfunction X reuses f's slot and the
// growth-rate stage re-reads f (coalesced, L2-resident) from global memory.
template <typename T, int M>
__global__ void __launch_bounds__(256) k_solve_gcf(long n_sys, int N, T h, const T* __restrict__ g,
                                                   const T* __restrict__ c, const T* __restrict__ f, long ld,
                                                   T* lam_out, T* gam_out, T* X_out, T* dX_out, int* info_out,
                                                   const T* __restrict__ gh, int flags) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long sys = (long)blockIdx.x * wpb + wave;
  const bool valid = sys < n_sys;
  const long sysc = valid ? sys : (n_sys - 1);
  const int P = lds_pitch(N);
  T* gs = smem + (size_t)wave * 3 * P;
  T* cs = gs + P; T* fs = cs + P; T* Xs = fs;
  const T* gg = g + sysc * ld; const T* cg = c + sysc * ld; const T* fg = f + sysc * ld;
  const bool want_vec = gam_out || X_out || dX_out;     // (kernel-uniform)
  for (int j = lane; j < N; j += kWave) { const int q = lpos(j); gs[q] = gg[j]; cs[q] = cg[j]; fs[q] = fg[j]; }
  wave_lds_sync();   // the staging rows are private to this wave
  SrcGCF<T> src{gs, cs, fs};
  WaveSolver<T, M> ws;
  SolveInfo inf{0, 0};
  bool bad;
  // FP64 solves start from the trial vector's bracket (10 instead of 14.5 sweeps on the smooth family of config 5, 17 instead of
  // 22.6 on the rough one: there the Rayleigh quotient alone helps, as a lower bound).  The all-FP32 kernel keeps its cold start:
  // its stated outlier tolerance (tests/test_gpu_configs.py) was measured on that path.
  constexpr bool kTrial = sizeof(T) == 8;
  if (gh) { SrcGCFH<T> srch{gs, cs, fs, gh + sysc * ld}; bad = ws.template setup<SrcGCFH<T>, kTrial>(srch, N, h); }   // wave-uniform branch
  else bad = ws.template setup<SrcGCF<T>, kTrial>(src, N, h);
  wave_lds_sync();   // every lane has taken its f chunk: the (wave-private) slot can be reused for X
  T lam = T(0);
  bool redo = false;   // the solve was re-closed in division form: the eigenvector stage is repeated at the end (FP64 only)
  if (!bad) {
    if constexpr (kTrial) {
      T g_, w_;
      ws.trial_guess(g_, w_);
      lam = ws.template solve<true>(inf, true, g_, w_);
      // (a suspect solve is closed again in division form on the rows in global memory)
      if (gh) { const SrcDirectH<T> sd{gg, cg, fg, gh + sysc * ld}; redo = reclose_suspect<T, M>(ws, inf, sd, N, h, flags, lam); }
      else { const SrcDirect<T, T> sd{gg, cg, fg}; redo = reclose_suspect<T, M>(ws, inf, sd, N, h, flags, lam); }
    } else lam = ws.solve(inf);
  } else { inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); }
  if constexpr (sizeof(T) == 4) {
    // FP64 CERTIFICATE of the all-FP32 result.  FP32 Sturm counts are good to ~eps32 ||A||; between two close eigenvalues a
    // count off by one lets the bracket close on lam_2 (1e-4 of the smooth config-5 systems at N_zeta <= 512, round 3).  The
    // rows are still staged: the FP64 solver is set up on them (every float is a double) and ONE count pair at
    // lam32 +- n eps32 ||A|| says whether lam_max lies within that distance (count 0 above, >= 1 below: backward-stable).  A
    // system that fails is solved in FP64 from the trial bracket; its info word carries status bit 2 (informational).
    if (!bad) {
      WaveSolver<double, M> wd;
      const SrcGCFF<T> srcw{gs, cs, fs};
      const bool bad2 = wd.template setup<SrcGCFF<T>, true>(srcw, N, (double)h);
      const double tolc = (double)(N - 2) * (double)Eps<float>::v * wd.normA;
      bool ok = !bad2 && wd.sweep_fwd((double)lam + tolc) == 0;
      if (ok) ok = wd.sweep_fwd((double)lam - tolc) >= 1;
      if (!ok && !bad2) {                                  // (wave-uniform)
        SolveInfo inf2{0, 0};
        double g_, w_;
        wd.trial_guess(g_, w_);
        lam = (T)wd.solve(inf2, true, g_, w_);
        inf.iters += inf2.iters + 2; inf.status |= inf2.status | 4;
      } else inf.iters += 2;
    }
  }
  if (!want_vec) {                         // (kernel-uniform) eigenvalues only: no eigenvector, no Simpson sums
    if (lane == 0 && valid) {
      if (lam_out) lam_out[sysc] = lam;
      if (info_out) info_out[sysc] = inf.iters | (inf.status << 16);
    }
    return;
  }
  if constexpr (M >= 3) {                  // f is read from its LDS slot, which X / dX reuse afterwards
    finish_chunk<T, M, SrcGCF<T>, false>(ws, src, N, h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                         valid ? gam_out : nullptr, valid ? X_out : nullptr,
                                         valid ? dX_out : nullptr, nullptr, valid ? info_out : nullptr,
                                         gh ? gh + sysc * ld : nullptr);
  } else {
    const SrcGCFG<T> srcf{gs, cs, fg};     // growth-rate stage: f from global memory
    finish<T, M, SrcGCFG<T>, false>(ws, srcf, N, h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                   valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                   nullptr, valid ? info_out : nullptr, static_cast<const NoTangent*>(nullptr), nullptr,
                                   gh ? gh + sysc * ld : nullptr);
  }
  if constexpr (kTrial) {
    if (redo) {                            // (wave-uniform, rare) eigenvector stage once more, at the re-closed eigenvalue
      wave_lds_sync();
      for (int j = lane; j < N; j += kWave) fs[lpos(j)] = fg[j];     // (X / dX went through f's slot)
      wave_lds_sync();
      WaveSolver<T, M> w2;
      if (gh) { SrcGCFH<T> srch{gs, cs, fs, gh + sysc * ld}; (void)w2.template setup<SrcGCFH<T>, false>(srch, N, h); }
      else (void)w2.template setup<SrcGCF<T>, false>(src, N, h);
      wave_lds_sync();
      w2.sweep(lam); w2.twisted(lam);
      if constexpr (M >= 3) {
        finish_chunk<T, M, SrcGCF<T>, false>(w2, src, N, h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                             valid ? gam_out : nullptr, valid ? X_out : nullptr,
                                             valid ? dX_out : nullptr, nullptr, valid ? info_out : nullptr,
                                             gh ? gh + sysc * ld : nullptr);
      } else {
        const SrcGCFG<T> srcf{gs, cs, fg};
        finish<T, M, SrcGCFG<T>, false>(w2, srcf, N, h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                       valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                       nullptr, valid ? info_out : nullptr, static_cast<const NoTangent*>(nullptr), nullptr,
                                       gh ? gh + sysc * ld : nullptr);
      }
    }
  }
}

// ---------------------------------------------------------------- raw (g, c, f) systems, rows straight from global memory
// Large batches on long grids (round 5).  k_solve_gcf stages the three rows of its system in LDS: 24.6 KB per wave at
// N_zeta = 1024, i.e. FIVE waves per CU where the registers (198 at M = 16) admit eight -- the FP64 pipe ran at 0.47 of
// its issue rate.  Here every lane reads its own chunk of M consecutive rows of g, c, f directly (a lane's chunk is M x 8
// contiguous bytes; the 64 chunks of a wave tile the row, so every fetched line is used in full, by the one lane that
// owns it), once in setup() and once more in the growth-rate stage (L2 / Infinity Cache), like the sub-wave kernels do
// at 32 / 16 lanes per system.  No LDS at all unless X / dX are requested (then one row per wave).
// TI = float: FP32 in HBM, widened exactly as read, FP64 solver (the arithmetic of k_solve_gcf_wide).
template <typename T, int M, typename TI>
__device__ __forceinline__ void solve_gcf_direct_body(long n_sys, int N, T h, const TI* __restrict__ g, const TI* __restrict__ c,
                                                      const TI* __restrict__ f, long ld, TI* lam_out, TI* gam_out, TI* X_out,
                                                      TI* dX_out, int* info_out, int flags, int* fix_count, long* fix_sys,
                                                      double* fix_center) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long sys = (long)blockIdx.x * wpb + wave;
  const bool valid = sys < n_sys;
  const long sysc = valid ? sys : (n_sys - 1);
  T* Xs = (X_out || dX_out) ? reinterpret_cast<T*>(smem_raw) + (size_t)wave * lds_pitch(N) : nullptr;
  SrcDirect<T, TI> src{g + sysc * ld, c + sysc * ld, f + sysc * ld};
  WaveSolver<T, M> ws;
  SolveInfo inf{0, 0};
  const bool want_vec = gam_out || X_out || dX_out;     // (kernel-uniform)
  const bool bad = ws.template setup<SrcDirect<T, TI>, true>(src, N, h);
  T lam = T(0);
  if (!bad) {
    T g_, w_;
    ws.trial_guess(g_, w_);
    lam = ws.template solve<true>(inf, true, g_, w_);
    // a suspect solve (WaveSolver::solve<true>) is LISTED: k_fix_gcf closes it again in division form and repeats its eigenvector
    // stage after this launch, so that the hot kernel carries none of that code (flags bit 1: only marked, diagnostics)
    if (flags & 1) {                                       // (kernel-uniform)
      if (ws.suspect) list_suspect<T>(fix_count, fix_sys, fix_center, valid, lane == 0, sysc, ws.rho_last, lam);      // (wave-uniform, rare)
    } else if (flags != 0) {
      const int w = ws.why();
      if (w != 0) inf.status |= 8 | (w << 5);
    }
  } else { inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); }
  if (!want_vec) {                         // (kernel-uniform) eigenvalues only
    if (lane == 0 && valid) {
      if (lam_out) lam_out[sysc] = (TI)lam;
      if (info_out) info_out[sysc] = inf.iters | (inf.status << 16);
    }
    return;
  }
  finish_chunk<T, M, SrcDirect<T, TI>, false, 1, TI>(ws, src, N, h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                                     valid ? gam_out : nullptr, valid ? X_out : nullptr,
                                                     valid ? dX_out : nullptr, nullptr, valid ? info_out : nullptr);
}
template <typename T, int M, typename TI>
__global__ void __launch_bounds__(256) k_solve_gcf_direct(long n_sys, int N, T h, const TI* __restrict__ g,
                                                          const TI* __restrict__ c, const TI* __restrict__ f, long ld,
                                                          TI* lam_out, TI* gam_out, TI* X_out, TI* dX_out, int* info_out, int flags,
                                                          int* fix_count, long* fix_sys, double* fix_center) {
  solve_gcf_direct_body<T, M, TI>(n_sys, N, h, g, c, f, ld, lam_out, gam_out, X_out, dX_out, info_out, flags, fix_count, fix_sys, fix_center);
}
// The same kernel held to two waves per SIMD (256 registers).  The allocator left to itself takes AGPRs from M = 21 on (one wave
// per SIMD, `valu_issue` 0.39-0.53); capped, M = 21 .. 28 spill 12-232 B per lane and still gain: FP64 rows 1.17-1.5x (N_zeta =
// 1344 .. 1792), FP32 rows 1.5x up to M = 30.  Beyond that the spills cost more than the second wave brings (N_zeta = 2048: 384 B
// of scratch, 0.77x; with the backward solution moved to LDS to relieve the tail: 0.66x -- the spills are the shift iteration's,
// not the tail's), so M = 29+ (FP32 rows: 31+) keep one wave per SIMD.   tools/bench_direct.py, docs/EXPERIMENTS.md R5.6
constexpr bool direct_two_waves(int M, bool f32_rows) { return M >= 21 && M <= (f32_rows ? 30 : 28); }
template <typename T, int M, typename TI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_solve_gcf_direct_w2(long n_sys, int N, T h, const TI* __restrict__ g, const TI* __restrict__ c, const TI* __restrict__ f, long ld,
                      TI* lam_out, TI* gam_out, TI* X_out, TI* dX_out, int* info_out, int flags, int* fix_count, long* fix_sys,
                      double* fix_center) {
  solve_gcf_direct_body<T, M, TI>(n_sys, N, h, g, c, f, ld, lam_out, gam_out, X_out, dX_out, info_out, flags, fix_count, fix_sys, fix_center);
}

// The listed suspects of a big-batch launch (k_solve_gcf_direct*, k_solve_gcf_g): one wave (= one block) per system.
//   1. the rows of the division form -- d_r, e_r^2, f_r (utils.py:1584-1592) -- are formed once, in parallel, into LDS;
//   2. ONE multisection pass over a bracket of half width 2^11 eps ||A|| about the polish the solver kernel recorded places lam_max
//      in an interval 65 eps ||A|| wide by division-form counts (a polish further off first costs the passes that move the bracket);
//   3. set-up and a sweep pair at the interval's middle: the twisted factorisation's Rayleigh polish, clamped into the interval, is
//      the eigenvalue returned (within 65 eps ||A|| of lam_max a priori, ~1 eps typically), and, when the growth rate or the
//      eigenfunction was asked for, a second sweep pair AT it feeds the growth-rate stage (utils.py:1601-1621) that overwrites the
//      system's provisional outputs.
// The info word gains the informational status bit 3 and the passes.  The chain of n dependent divisions per pass is what this
// kernel's time is (~50 us at N_zeta = 2048); it runs after the solver kernel, on at most a few dozen waves.  A persistent grid:
// the list length is only known on the device.
template <int M, typename TI>
__global__ void __launch_bounds__(64) k_fix_gcf(const int* __restrict__ n_fix, const long* __restrict__ fix_sys,
                                                const double* __restrict__ fix_center, int N, double h, const TI* __restrict__ g,
                                                const TI* __restrict__ c, const TI* __restrict__ f, long ld, TI* lam_out, TI* gam_out,
                                                TI* X_out, TI* dX_out, int* info_out) {
  using T = double;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int nf = *n_fix;
  const int n = N - 2;
  const bool want_vec = gam_out || X_out || dX_out;     // (kernel-uniform)
  T* rd = reinterpret_cast<T*>(smem_raw); T* re2 = rd + n; T* rf = re2 + n;
  T* Xs = rf + n;                                        // (one padded row, used when X / dX are written or M < 3)
  const T ih2 = T(1) / (h * h);
  for (int i = blockIdx.x; i < nf; i += gridDim.x) {
    const long sys = fix_sys[i];
    SrcDirect<T, TI> src{g + sys * ld, c + sys * ld, f + sys * ld};
    for (int r = lane; r < n; r += kWave) {              // the diagonal exactly as WaveSolver::setup forms it
      const T e_lo = T(0.5) * (src.g(r) + src.g(r + 1)) * ih2, e_hi = T(0.5) * (src.g(r + 1) + src.g(r + 2)) * ih2;
      rd[r] = src.c(r + 1) - (e_lo + e_hi); re2[r] = e_lo * e_lo; rf[r] = src.f(r + 1);
    }
    wave_lds_sync();
    WaveSolver<T, M> ws;
    (void)ws.template setup<SrcDirect<T, TI>, false>(src, N, h);
    const T epsA = Eps<T>::v * ws.normA, center = (T)fix_center[i];
    T lo = center - T(2048) * epsA, hi = center + T(2048) * epsA, lam;
    int passes;
    const bool ok = multisect<T>([&](T sig) { return count_above_rows<T>(rd, re2, rf, n, sig); }, lo, hi, ws.normA, T(128), lane, lam, passes);
    const int old = info_out ? info_out[sys] : 0;
    SolveInfo inf{(old & 0xffff) + passes, ((old >> 16) & 0x7fff) | 8 | (ok ? 0 : 1)};
    if (ok) {
      // the interval [lam - w, lam + w] is certified by division-form counts; the polish of a sweep pair at its middle refines within it
      const T w = T(33) * epsA;
      ws.sweep(lam);
      const T rho = ws.twisted(lam);
      const T lam2 = WaveSolver<T, M>::U(finite_of(rho)) ? xmin(xmax(rho, lam - w), lam + w) : lam;
      if (want_vec) {
        ws.sweep(lam2); ws.twisted(lam2);
        if constexpr (M >= 3)
          finish_chunk<T, M, SrcDirect<T, TI>, false, 1, TI>(ws, src, N, h, Xs, lam2, inf, sys, lam_out, gam_out, X_out, dX_out, nullptr, info_out);
        else
          finish<T, M, SrcDirect<T, TI>, false, NoTangent, TI>(ws, src, N, h, Xs, lam2, inf, sys, lam_out, gam_out, X_out, dX_out, nullptr, info_out);
      } else if (lane == 0) {
        if (lam_out) lam_out[sys] = (TI)lam2;
        if (info_out) info_out[sys] = inf.iters | (inf.status << 16);
      }
    } else if (lane == 0 && info_out) info_out[sys] = inf.iters | (inf.status << 16);
    wave_lds_sync();                                     // (the rows are rebuilt for this wave's next system)
  }
}

// FP32 eigenvalues only, rows straight from global memory: the all-FP32 shift iteration of k_solve_gcf<float, M> and its FP64
// certificate (count 0 at lam32 + n eps32 ||A||, >= 1 at lam32 - n eps32 ||A||; a system that fails it is solved in FP64,
// informational status bit 2), every lane reading its own chunk of the three FP32 rows -- once for the FP32 set-up, once more
// (widened, exactly) for the certificate's.  No LDS: k_solve_gcf<float, M> stages 27.7 KB per wave at N_zeta = 2048.
template <int M>
__device__ __forceinline__ void solve_gcf_f32lam_direct_body(long n_sys, int N, float h, const float* __restrict__ g,
                                                             const float* __restrict__ c, const float* __restrict__ f, long ld,
                                                             float* lam_out, int* info_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long sys = (long)blockIdx.x * wpb + wave;
  const bool valid = sys < n_sys;
  const long sysc = valid ? sys : (n_sys - 1);
  SolveInfo inf{0, 0};
  float lam = 0.0f;
  bool bad;
  {
    SrcDirect<float, float> src{g + sysc * ld, c + sysc * ld, f + sysc * ld};
    WaveSolver<float, M> ws;
    bad = ws.template setup<SrcDirect<float, float>, false>(src, N, h);
    if (!bad) lam = ws.solve(inf);
    else inf.status = 2;
  }
  if (!bad) {
    SrcDirect<double, float> srcw{g + sysc * ld, c + sysc * ld, f + sysc * ld};
    WaveSolver<double, M> wd;
    const bool bad2 = wd.template setup<SrcDirect<double, float>, true>(srcw, N, (double)h);
    const double tolc = (double)(N - 2) * (double)Eps<float>::v * wd.normA;
    bool ok = !bad2 && wd.sweep_fwd((double)lam + tolc) == 0;
    if (ok) ok = wd.sweep_fwd((double)lam - tolc) >= 1;
    if (!ok && !bad2) {                                  // (wave-uniform)
      SolveInfo inf2{0, 0};
      double g_, w_;
      wd.trial_guess(g_, w_);
      lam = (float)wd.solve(inf2, true, g_, w_);
      inf.iters += inf2.iters + 2; inf.status |= inf2.status | 4;
    } else inf.iters += 2;
  }
  if (lane == 0 && valid) {
    if (lam_out) lam_out[sysc] = lam;
    if (info_out) info_out[sysc] = inf.iters | (inf.status << 16);
  }
}
template <int M>
__global__ void __launch_bounds__(256) k_solve_gcf_f32lam_direct(long n_sys, int N, float h, const float* __restrict__ g,
                                                                 const float* __restrict__ c, const float* __restrict__ f, long ld,
                                                                 float* lam_out, int* info_out) {
  solve_gcf_f32lam_direct_body<M>(n_sys, N, h, g, c, f, ld, lam_out, info_out);
}
template <int M>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_solve_gcf_f32lam_direct_w2(long n_sys, int N, float h, const float* __restrict__ g, const float* __restrict__ c,
                             const float* __restrict__ f, long ld, float* lam_out, int* info_out) {
  solve_gcf_f32lam_direct_body<M>(n_sys, N, h, g, c, f, ld, lam_out, info_out);
}

// FP32 systems whose growth rate (or eigenfunction) is wanted: FP32 in HBM, FP64 in the solver.  The FD4 / Simpson growth rate
// subtracts two sums of size ||A|| ~ 4 / h^2, so an FP32 eigenvector's noise is multiplied by ~N^2 (usable at N_zeta <= 512,
// noise above: round 2); the rows are therefore widened while they are staged -- exactly: every float is a double -- and the
// system the FP32 arrays DEFINE is solved by the FP64 path (certified by counts, gam to ~1e-12 of that system), results
// rounded to FP32 on the way out.  Half the HBM bytes of the FP64 entry point, the same arithmetic.  (lam alone, the
// throughput / stress form of config 5, stays with the all-FP32 kernel k_solve_gcf<float, M>.)
// growth-rate stage of k_solve_gcf_wide at M < 3: g, c from the widened LDS rows, f (whose row now holds X) re-read from global memory
struct SrcGCFW {
  static constexpr bool kHasGh = false;
  const double* gs; const double* cs; const float* fg;
  __device__ __forceinline__ double g(int j) const { return gs[lpos(j)]; }
  __device__ __forceinline__ void gcf(int j, double& g_, double& c_, double& f_) const { const int q = lpos(j); g_ = gs[q]; c_ = cs[q]; f_ = (double)fg[j]; }
};
template <int M>
__global__ void __launch_bounds__(256) k_solve_gcf_wide(long n_sys, int N, float h, const float* __restrict__ g,
                                                        const float* __restrict__ c, const float* __restrict__ f, long ld,
                                                        float* lam_out, float* gam_out, float* X_out, float* dX_out,
                                                        int* info_out, int flags) {
  using T = double;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long sys = (long)blockIdx.x * wpb + wave;
  const bool valid = sys < n_sys;
  const long sysc = valid ? sys : (n_sys - 1);
  const int P = lds_pitch(N);
  T* gs = smem + (size_t)wave * 3 * P;
  T* cs = gs + P; T* fs = cs + P; T* Xs = fs;
  const float* gg = g + sysc * ld; const float* cg = c + sysc * ld; const float* fg = f + sysc * ld;
  for (int j = lane; j < N; j += kWave) { const int q = lpos(j); gs[q] = (T)gg[j]; cs[q] = (T)cg[j]; fs[q] = (T)fg[j]; }
  wave_lds_sync();
  SrcGCF<T> src{gs, cs, fs};
  WaveSolver<T, M> ws;
  SolveInfo inf{0, 0};
  const bool want_vec = gam_out || X_out || dX_out;     // (kernel-uniform)
  const bool bad = ws.template setup<SrcGCF<T>, true>(src, N, (T)h);
  wave_lds_sync();
  T lam = T(0);
  bool redo = false;   // the solve was re-closed in division form: the eigenvector stage is repeated at the end
  if (!bad) {
    T g_, w_;
    ws.trial_guess(g_, w_);
    lam = ws.template solve<true>(inf, true, g_, w_);
    const SrcDirect<T, float> sd{gg, cg, fg};
    redo = reclose_suspect<T, M>(ws, inf, sd, N, (T)h, flags, lam);
  } else { inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); }
  if (!want_vec) {                         // (kernel-uniform) eigenvalues only
    if (lane == 0 && valid) {
      if (lam_out) lam_out[sysc] = (float)lam;
      if (info_out) info_out[sysc] = inf.iters | (inf.status << 16);
    }
    return;
  }
  if constexpr (M >= 3) {
    finish_chunk<T, M, SrcGCF<T>, false, 1, float>(ws, src, N, (T)h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                                    valid ? gam_out : nullptr, valid ? X_out : nullptr,
                                                    valid ? dX_out : nullptr, nullptr, valid ? info_out : nullptr);
  } else {
    // (the LDS-round-trip growth-rate stage needs f after X has taken its row: re-read, widened, from global memory)
    const SrcGCFW srcf{gs, cs, fg};
    finish<T, M, SrcGCFW, false, NoTangent, float>(ws, srcf, N, (T)h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                                valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                                nullptr, valid ? info_out : nullptr);
  }
  if (redo) {                              // (wave-uniform, rare) eigenvector stage once more, at the re-closed eigenvalue
    wave_lds_sync();
    for (int j = lane; j < N; j += kWave) fs[lpos(j)] = (T)fg[j];     // (X / dX went through f's slot)
    wave_lds_sync();
    WaveSolver<T, M> w2;
    (void)w2.template setup<SrcGCF<T>, false>(src, N, (T)h);
    wave_lds_sync();
    w2.sweep(lam); w2.twisted(lam);
    if constexpr (M >= 3) {
      finish_chunk<T, M, SrcGCF<T>, false, 1, float>(w2, src, N, (T)h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                                      valid ? gam_out : nullptr, valid ? X_out : nullptr,
                                                      valid ? dX_out : nullptr, nullptr, valid ? info_out : nullptr);
    } else {
      const SrcGCFW srcf{gs, cs, fg};
      finish<T, M, SrcGCFW, false, NoTangent, float>(w2, srcf, N, (T)h, Xs, lam, inf, sysc, valid ? lam_out : nullptr,
                                                  valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                                  nullptr, valid ? info_out : nullptr);
    }
  }
}

// ---------------------------------------------------------------- raw (g, c, f) systems on long grids: one LDS row per wave
// k_solve_gcf stages g, c and f of its system side by side (3 N doubles per wave: 55 KB at N_zeta = 2048 -> two waves per
// CU, 28 KB at 1024 -> five).  Here the three rows pass through ONE row, one after the other, the way k_sturm_count moves
// them: set-up is split into a g pass (half-grid e, diagonal scaling), an f pass and a c pass; after the solve the growth
// rate re-stages g (scaling of the eigenvector + the g dX^2 sum), then c, then f.  LDS per wave: N doubles.
template <typename T, int M, typename TI = T>
__device__ __forceinline__ void stage_row(T* row, const TI* __restrict__ src, int N, int lane) {
  // batches of up to 17 coalesced loads in flight (34 VGPRs for FP64 rows), then their LDS writes.  (The whole row in one batch --
  // 33 loads at N_zeta = 2048, one exposed memory latency per row instead of two -- spills more than it hides for FP64 rows, 1.72e7
  // against 1.83e7 solves/s, and changes nothing for FP32 rows: 12.6 against 12.4 us per three staged rows.)
  constexpr int B = 17;
#pragma unroll
  for (int k0 = 0; k0 <= M; k0 += B) {
    TI v[B];
    if constexpr (sizeof(TI) == 8) {
#pragma unroll
      for (int k = 0; k < B; ++k) { const int j = lane + (k0 + k) * kWave; v[k] = (k0 + k <= M && j < N) ? src[j] : TI(0); }
    } else {
      // (FP32 rows: a load under a lane condition is compiled as a branch with a wait behind EVERY such load -- 1.8 us longer per
      //  staged row.  N > 64 (M - 1) + 2 for this instantiation, so the loads k <= M - 2 are in bounds for every lane and
      //  unconditional, at immediate offsets from one base; only the last two take a clamped index: three staged rows 12.4 instead
      //  of 21.6 us at N_zeta = 2048.  The FP64 rows keep the conditional form: the other one spills 216 instead of 20 bytes at
      //  M = 32 and takes 26.7 instead of 17.0 us.)
#pragma unroll
      for (int k = 0; k < B; ++k) {
        if (k0 + k <= M) {
          const int j = lane + (k0 + k) * kWave;
          if (k0 + k <= M - 2) v[k] = src[j];
          else v[k] = src[j < N ? j : N - 1];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      if (k0 + k <= M) { const int j = lane + (k0 + k) * kWave; if (j < N) row[lpos(j)] = (T)v[k]; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <typename T, int M, typename TI = T>
struct SrcRows {
  static constexpr bool kHasGh = false;
  T* row; const TI* gg; const TI* cg; const TI* fg; int N, lane, which;     // which: the array the row holds now (0 g, 1 c, 2 f)
  __device__ __forceinline__ T g(int j) const { return row[lpos(j)]; }
  __device__ __forceinline__ void gcf(int j, T& g_, T& c_, T& f_) const {
    const T v = row[lpos(j)];
    g_ = which == 0 ? v : T(0); c_ = which == 1 ? v : T(0); f_ = which == 2 ? v : T(0);
  }
  __device__ __forceinline__ void hold(int w) {
    wave_lds_sync();
    stage_row<T, M, TI>(row, w == 0 ? gg : (w == 1 ? cg : fg), N, lane);
    wave_lds_sync();
    which = w;
  }
  __device__ __forceinline__ void begin_pass(int p) { if (which != p) hold(p); }
};
// set-up of WaveSolver from the three rows streamed through `s.row` (same quantities as WaveSolver::setup; the bounds are
// formed from the scaled rows: c/f = (c s^2)/(f s^2), ...)
// Also forms the Rayleigh quotient of the trial vector x_j = sin(pi j / (N - 1)) (WaveSolver::setup<Src, true>) -- from sums that
// are SEPARABLE in (g, c, f), so they fit the three passes:  x'Tx = sum_j c_j x_j^2 - sum_edges e_{j+1/2} (x_{j+1} - x_j)^2
// (summation by parts, x_0 = x_{N-1} = 0),  x'Fx = sum_j f_j x_j^2.  (The residual bound del is not separable: the kernel
// estimates it from a sample of rows, see k_solve_gcf_rows.)
template <typename T, int M, typename TI = T>
__device__ __forceinline__ bool setup_rows(WaveSolver<T, M>& ws, SrcRows<T, M, TI>& s, int N, T h) {
  const int lane = s.lane;
  ws.lane = lane;
  const int n = N - 2;
  const int rem = n - kWave * (M - 1);
  ws.has_last = lane < rem;
  const bool hl = ws.has_last;
  int a = WaveSolver<T, M>::rows_start(lane, n);
  const T ih2 = T(1) / (h * h);
  T esum[M], s2[M];
  bool bad = false;
  // trial vector at grid points a and a + 1 (a = the left neighbour of this lane's first row); every pass replays the recurrence
  T x_a, x_a1, two_cd;
  {
    const T dl = T(3.14159265358979323846) / T(N - 1);
    T s0, c0, sd, cd;
    trial_sincos(T(a) * dl, s0, c0);
    trial_sincos(dl, sd, cd);
    x_a = s0; x_a1 = xfma(s0, cd, c0 * sd); two_cd = T(2) * cd;
  }
  s.hold(0);                                             // ---- g: half-grid e (utils.py:1574-1576), scaling e s_i s_{i+1} = 1
  T e_first, e_last;
  T tE;                                                  // sum over this lane's edges of e (x_{j+1} - x_j)^2
  {
    const T g0 = s.row[lpos(a)];
    T gcur = s.row[lpos(a + 1)];
    T e_lo = T(0.5) * (g0 + gcur) * ih2;
    e_first = e_lo;
    bad = !(g0 > T(0)) || !(gcur > T(0));
    T sc = T(1);
    T xp = x_a, xc = x_a1;
    tE = lane == 0 ? e_first * (xc - xp) * (xc - xp) : T(0);   // the edge below row 1 (x_0 = 0); every other edge is its lower row's
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if ((i < M - 1) || hl) {
        const T gnext = s.row[lpos(a + i + 2)];
        const T e_hi = T(0.5) * (gcur + gnext) * ih2;
        bad = bad || !(gnext > T(0)) || !(e_hi > T(0));
        esum[i] = e_lo + e_hi; s2[i] = sc * sc;
        sc = fast_rcp(e_hi * sc);
        gcur = gnext; e_lo = e_hi;
        const T xn = xfma(two_cd, xc, -xp), dx = xn - xc;
        tE = xfma(e_hi * dx, dx, tE);
        xp = xc; xc = xn;
      } else { esum[i] = T(0); s2[i] = T(0); }
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    ws.kap = sc; ws.ikap = fast_rcp(sc);
    e_last = e_lo;
    bad = bad || !(e_first > T(0));
  }
  T sum_c = T(0), sum_f = T(0), tB = T(0), tCx = T(0);
  s.hold(2);                                             // ---- f
  {
    T xp = x_a, xc = x_a1;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if ((i < M - 1) || hl) {
        const T fj = s.row[lpos(a + i + 1)];
        ws.Ph[i] = fj * s2[i];
        sum_f += fj;
        tB = xfma(fj * xc, xc, tB);
        bad = bad || !(fj > T(0));
        const T xn = xfma(two_cd, xc, -xp);
        xp = xc; xc = xn;
      } else ws.Ph[i] = T(0);
    }
  }
  s.hold(1);                                             // ---- c: d = c - (e_lo + e_hi)  (utils.py:1584-1592)
  T vhi = -T(1e300), vlo = -T(1e300), vna = T(0);
  {
    T xp = x_a, xc = x_a1;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if ((i < M - 1) || hl) {
        const T cj = s.row[lpos(a + i + 1)];
        const T d = cj - esum[i];
        ws.D[i] = d * s2[i];
        const T rPh = fast_rcp(ws.Ph[i]);                // bounds only (margins added below)
        vhi = xmax(vhi, (cj * s2[i]) * rPh);
        vlo = xmax(vlo, ws.D[i] * rPh);
        vna = xmax(vna, ((xabs(d) + esum[i]) * s2[i]) * rPh);
        sum_c += cj;
        tCx = xfma(cj * xc, xc, tCx);
        bad = bad || !finite_of(cj);
        const T xn = xfma(two_cd, xc, -xp);
        xp = xc; xc = xn;
      } else ws.D[i] = T(0);
    }
  }
  const T e0 = readlane_t(e_first, 0), en = readlane_t(e_last, kWave - 1);
  const T sc_all = wave_sum(sum_c), sf_all = wave_sum(sum_f);
  ws.normA = uniform(wave_max(vna));
  ws.hi = uniform(wave_max(vhi));
  ws.lo = uniform(xmax(wave_max(vlo), (sc_all - e0 - en) / sf_all));
  ws.hi += T(8) * Eps<T>::v * ws.normA;
  ws.lo -= T(8) * Eps<T>::v * ws.normA;
  ws.trial_rho = wave_sum(tCx - tE) / wave_sum(tB);
  ws.trial_del = T(-1);                                  // (set by the caller)
  ws.trial_mrg = T(8 + N / 2) * Eps<T>::v * ws.normA;
  ws.chk_slack = T(N < 256 ? 2 * N : 512) * Eps<T>::v * ws.normA;
  return __any(bad) != 0;
}

// Estimate of the trial vector's residual bound del (WaveSolver::setup<Src, true>) from every fourth row, straight from global
// memory before anything else is live: (T x)_j needs g, c and f of a row at once.  8 rows per lane at N_zeta = 2048, all 40 loads in
// flight together.  del only sets the width of the first bracket (rho + del / 4 and up): an estimate will do.
template <typename T, typename TI = T>
__device__ __forceinline__ T trial_del_sampled(const TI* gq, const TI* cq, const TI* fq, int N, T h, int lane) {
  constexpr int kS = 8, kStep = 4;
  const T ih2 = T(1) / (h * h);
  const T dl = T(3.14159265358979323846) / T(N - 1);
  T gm[kS], g0[kS], gp[kS], cc[kS], ff[kS];
#pragma unroll
  for (int k = 0; k < kS; ++k) {
    const int j = 2 + kStep * (lane + kWave * k);
    const int jc = j <= N - 2 ? j : 2;
    gm[k] = (T)gq[jc - 1]; g0[k] = (T)gq[jc]; gp[k] = (T)gq[jc + 1]; cc[k] = (T)cq[jc]; ff[k] = (T)fq[jc];
  }
  T sj, cj, s1, c1, sW, cW;
  trial_sincos(T(2 + kStep * lane) * dl, sj, cj);
  trial_sincos(dl, s1, c1);
  trial_sincos(T(kStep * kWave) * dl, sW, cW);
  T tA = T(0), tB = T(0), tC = T(0);
#pragma unroll
  for (int k = 0; k < kS; ++k) {
    const int j = 2 + kStep * (lane + kWave * k);
    if (j <= N - 2) {
      const T e_lo = T(0.5) * (gm[k] + g0[k]) * ih2, e_hi = T(0.5) * (g0[k] + gp[k]) * ih2;
      const T d = cc[k] - (e_lo + e_hi);
      const T xm = xfma(sj, c1, -cj * s1), xp = xfma(sj, c1, cj * s1);            // sin((j -+ 1) dl)
      const T Tx = xfma(e_lo, xm, xfma(d, sj, e_hi * xp));
      tA = xfma(sj, Tx, tA); tB = xfma(ff[k] * sj, sj, tB); tC = xfma(Tx * fast_rcp(ff[k]), Tx, tC);
    }
    const T sn = xfma(sj, cW, cj * sW);
    cj = xfma(cj, cW, -sj * sW); sj = sn;
  }
  const T A = wave_sum(tA), B = wave_sum(tB), C = wave_sum(tC);
  const T rho_s = A / B;
  return approx_sqrt(xmax(xfma(-rho_s, A, C), T(0)) / B);
}

// TI = the element type in memory (float: FP32 systems widened as they are staged and solved in FP64, like k_solve_gcf_wide)
template <typename T, int M, typename TI = T>
__global__ void __launch_bounds__(256) k_solve_gcf_rows(long n_sys, int N, T h, const TI* __restrict__ g,
                                                        const TI* __restrict__ c, const TI* __restrict__ f, long ld,
                                                        TI* lam_out, TI* gam_out, TI* X_out, TI* dX_out, int* info_out, int flags) {
  static_assert(M >= 3, "long grids only");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long sys = (long)blockIdx.x * wpb + wave;
  const bool valid = sys < n_sys;
  const long sysc = valid ? sys : (n_sys - 1);
  T* row = smem + (size_t)wave * lds_pitch(N);
  SrcRows<T, M, TI> src{row, g + sysc * ld, c + sysc * ld, f + sysc * ld, N, lane, -1};
  WaveSolver<T, M> ws;
  SolveInfo inf{0, 0};
  // trial-vector bracket (WaveSolver::trial_guess): rho from set-up's separable sums, del from a sample of rows (a full pass over
  // global memory for del cost what the bracket saved: 10.0 instead of 16.2 sweeps, 1.73e7 solves/s either way)
  IBS_PROBE_AT(0);                                       // (phase stamps: tools/rows_probe.py, debug builds only)
  const T t_del = trial_del_sampled<T, TI>(src.gg, src.cg, src.fg, N, h, lane);
  IBS_PROBE_AT(1);
  const bool bad = setup_rows<T, M, TI>(ws, src, N, h);
  ws.trial_del = t_del;
  IBS_PROBE_AT(2);
  T lam = T(0);
  bool redo = false;   // the solve was re-closed in division form: the eigenvector stage is repeated at the end
  if (!bad) {
    T g_, w_;
    ws.trial_guess(g_, w_);
    lam = ws.template solve<true>(inf, true, g_, w_);
    const SrcDirect<T, TI> sd{src.gg, src.cg, src.fg};
    redo = reclose_suspect<T, M>(ws, inf, sd, N, h, flags, lam);
  } else { inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); }
  IBS_PROBE_AT(3);
  // (the first batch of the next row requested while the current row is worked on -- to hide one of the two exposed memory
  //  latencies per staged row, 33 of this kernel's 63 us -- was built and measured: the 34 registers it keeps live cost more in
  //  AGPR traffic than the latency it hides, 1.71e7 against 1.83e7 solves/s)
  if (!gam_out && !X_out && !dX_out) {                   // (kernel-uniform) eigenvalues only
    if (lane == 0 && valid) {
      if (lam_out) lam_out[sysc] = (TI)lam;
      if (info_out) info_out[sysc] = inf.iters | (inf.status << 16);
    }
    return;
  }
  src.hold(0);                                           // g: the eigenvector's scaling is rebuilt from it, then the g dX^2 sum
  finish_chunk<T, M, SrcRows<T, M, TI>, false, 3, TI>(ws, src, N, h, row, lam, inf, sysc, valid ? lam_out : nullptr,
                                                      valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                                      nullptr, valid ? info_out : nullptr);
  IBS_PROBE_AT(4);
  if (redo) {                              // (wave-uniform, rare) eigenvector stage once more, at the re-closed eigenvalue
    WaveSolver<T, M> w2;
    (void)setup_rows<T, M, TI>(w2, src, N, h);
    w2.sweep(lam); w2.twisted(lam);
    src.hold(0);
    finish_chunk<T, M, SrcRows<T, M, TI>, false, 3, TI>(w2, src, N, h, row, lam, inf, sysc, valid ? lam_out : nullptr,
                                                        valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                                        nullptr, valid ? info_out : nullptr);
  }
}

// ---------------------------------------------------------------- geometry-fed theta0 scan
// grid = n_lines * ceil(n_theta0 / wpb) blocks; block = wpb waves; wave w solves theta0 index part*wpb + w.
// dynamic LDS = (7 + wpb) * lds_pitch(N) * sizeof(T).  Geometry arrays are [n_lines][ld].
template <typename T, int M>
__global__ void __launch_bounds__(scan_max_threads(M)) k_gamma_scan(int n_lines, int n_theta0, int N, T h,
                                                     const T* __restrict__ bmag, const T* __restrict__ gradpar,
                                                     const T* __restrict__ cvdrift, const T* __restrict__ cvdrift0,
                                                     const T* __restrict__ gds2, const T* __restrict__ gds21,
                                                     const T* __restrict__ gds22, long ld,
                                                     const T* __restrict__ dPdrho, const T* __restrict__ theta0,
                                                     T* gam_out, T* lam_out, T* X_out, T* dX_out, T* dth0_out,
                                                     int* info_out, const T* __restrict__ lam_guess, T guess_width,
                                                     int lines_per_surf, int* surf_counter, T* pack, int pack_mode,
                                                     int t0_stride) {
  // t0_stride: 0 = the theta0 grid is shared by all lines; 1 (with n_theta0 = 1) = one theta0 per line (ibs_gamma_points_f64)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  IBS_PROBE_AT(0);
  // XCD-aware block -> (line, part) map.  Workgroups are dealt round-robin over the 8 XCDs, so blocks b
  // and b+8 share an L2: the `nparts` blocks that scan different theta0 of ONE line are placed 8 apart
  // and the line's geometry is fetched from HBM once per XCD instead of once per block (speed only).
  const int nparts = (n_theta0 + wpb - 1) / wpb;
  int line, part;
  {
    const int b = blockIdx.x;
    const int chunk = b / (8 * nparts), r = b - chunk * 8 * nparts;
    const int lines_here = min(8, n_lines - chunk * 8);
    line = chunk * 8 + r % lines_here;
    part = r / lines_here;
    if (part >= nparts) return;   // cannot happen (r < lines_here*nparts is guaranteed by the grid size)
  }
  const int P = lds_pitch(N);
  T* A1 = smem; T* A3 = A1 + P; T* C0 = A3 + P; T* C1 = C0 + P; T* G0 = C1 + P; T* G1 = G0 + P; T* G2 = G1 + P;
  T* Xs = G2 + P + (size_t)wave * P;
  // this wave's theta0 (and warm-start guess): requested before the staging loads so that their latency is
  // covered by the geometry's instead of following the block barrier
  const int it0 = part * wpb + wave;
  const bool valid = it0 < n_theta0;
  const int it0c = valid ? it0 : (n_theta0 - 1);
  const T th0 = theta0[(long)line * t0_stride + it0c];
  const long sys = (long)line * n_theta0 + it0c;
  const T guess = lam_guess ? lam_guess[sys] : T(0);
  {
    const long off = (long)line * ld;
    const T mdP = -dPdrho[line];
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const T B = bmag[off + j], gp = xabs(gradpar[off + j]);
      const T inv = T(1) / (gp * B);                    // 1/(|gradpar| B)   (IEEE divisions: the coefficients are
      const int q = lpos(j);                            //  bit-identical to the oracle's, which near-degenerate
      A1[q] = gp / B;                                   //  eigenvectors amplify by ~1e9)   g: utils.py:1560
      A3[q] = inv / (B * B);                            // f = gds2/B^2 /(|gradpar| B)   (utils.py:1562)
      C0[q] = mdP * cvdrift[off + j] * inv;             // c = -dPdrho cvdrift/(|gradpar| B) (utils.py:1561)
      C1[q] = mdP * cvdrift0[off + j] * inv;
      G0[q] = gds2[off + j]; G1[q] = gds21[off + j]; G2[q] = gds22[off + j];
    }
  }
  __syncthreads();
  IBS_PROBE_AT(1);
  SrcGeo<T> src{A1, A3, C0, C1, G0, G1, G2, th0, T(2) * th0, th0 * th0};
  WaveSolver<T, M> ws;
  SolveInfo inf{0, 0};
#ifdef IBS_PROBE_TWICE
  // experiment (round 5, tools/scan_probe.py): the set-up executed twice, the first result discarded -- is its time the code's
  // first execution (instruction fetch) or its dependent chains?  stamp 15 = end of the first execution.  (Answer: the chains.)
  { int N2 = N; asm volatile("" : "+s"(N2)); WaveSolver<T, M> w0; (void)w0.template setup<SrcGeo<T>, true>(src, N2, h); asm volatile("" :: "v"(w0.kap), "v"(w0.D[0]), "v"(w0.Ph[M - 1]), "v"(w0.trial_rho)); }
  IBS_PROBE_AT(15);
#endif
  const bool bad = ws.template setup<SrcGeo<T>, true>(src, N, h);
  IBS_PROBE_AT(2);
  T lam = T(0);
  if (!bad) {
    // one inlined copy of the solver: a caller's guess, or the bracket of the trial vector (WaveSolver::trial_guess)
    T g_ = guess, w_ = guess_width;
    if (!lam_guess) ws.trial_guess(g_, w_);
    lam = ws.solve(inf, true, g_, w_);
  } else { inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); }
  IBS_PROBE_AT(3);
  if constexpr (M >= 3) {
    finish_chunk<T, M, SrcGeo<T>, true>(ws, src, N, h, Xs, lam, inf, sys, valid ? lam_out : nullptr,
                                        valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                        valid ? dth0_out : nullptr, valid ? info_out : nullptr);
  } else {
    finish<T, M, SrcGeo<T>, true>(ws, src, N, h, Xs, lam, inf, sys, valid ? lam_out : nullptr,
                                  valid ? gam_out : nullptr, valid ? X_out : nullptr, valid ? dX_out : nullptr,
                                  valid ? dth0_out : nullptr, valid ? info_out : nullptr);
  }
  IBS_PROBE_AT(4);
  // ---- fused per-surface argmax (ball_scan.py:279-295: first maximum of the surface's (alpha, theta0) table).
  // Last-block-done: every block publishes its growth rates (plain stores -> each wave drains its stores -> block
  // barrier -> ONE lane: agent-scope release, drain, relaxed agent-scope add on the surface's counter); the block whose
  // add completes the surface acquires (agent scope), then reduces the table.  Placement-independent (no assumption
  // on dispatch order or XCD co-location: cdna_hip_programming.md G16); the counter word is reset by the last arriver.
  if (pack) {                                   // (kernel argument: uniform over the grid)
    int* flag = reinterpret_cast<int*>(smem + (size_t)(7 + wpb) * P);   // 128 B behind the staging rows (launcher adds them)
    double* sv = reinterpret_cast<double*>(flag + 4);
    int* si = flag + 4 + 2 * 8;
    const int surf = line / lines_per_surf;
    const int nblk_surf = lines_per_surf * nparts;
    // pack_mode 1 (one block per CU: the regime MI355X_MICROARCH.md measured this hand-off in): the growth rates were
    // stored write-through (sc1) by finish(), every storing wave drains its stores, and the add follows the block barrier;
    // the last arriver reads them back with sc1 loads -- no fences.  pack_mode 2 (any placement): agent-scope release
    // before the add, agent-scope acquire in the block that completes the surface.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (pack_mode != 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const int old = __hip_atomic_fetch_add(&surf_counter[surf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (old == nblk_surf - 1) ? 1 : 0;
      if (last) {
        __hip_atomic_store(&surf_counter[surf], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pack_mode != 1) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      flag[0] = last;
    }
    __syncthreads();
    if (flag[0]) {                              // block-uniform
      const int n_per = lines_per_surf * n_theta0;
      const T* gsurf = gam_out + (size_t)surf * n_per;
      T best = -T(1.7976931348623157e308);
      int bi = 0x7fffffff;
      for (int i = threadIdx.x; i < n_per; i += blockDim.x) {
        // (agent-scope loads: served by L2, never by this CU's L1, which other CUs' stores do not refresh)
        const T v = __hip_atomic_load(gsurf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const T v2 = __shfl_xor(best, d);
        const int i2 = __shfl_xor(bi, d);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
      }
      if (lane == 0) { sv[wave] = best; si[wave] = bi; }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int k = 1; k < wpb; ++k)
          if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        pack[2 * surf] = best; pack[2 * surf + 1] = (T)bi;
      }
    }
  }
}


// Chained form of the scan for batches much larger than the chip: every wave solves `chain` consecutive theta0
// values of its line one after the other and starts each solve from the eigenvalue of the previous one (first
// neighbour: lam_prev +- w1 |lam_prev|; from the second on: linear extrapolation +- w2 |last difference|) -- the
// reference's own theta0 loop warm-starts its ARPACK vector the same way (ball_scan.py:265-274).  A wrong guess
// costs sweeps, never correctness: the bracket still moves on counts only.  The staged line serves chain x wpb
// solves instead of wpb, and without X / dX output no per-wave LDS row is allocated at all.
template <typename T, int M>
__global__ void __launch_bounds__(scan_max_threads(M)) k_gamma_scan_chain(
    int n_lines, int n_theta0, int N, T h, const T* __restrict__ bmag, const T* __restrict__ gradpar,
    const T* __restrict__ cvdrift, const T* __restrict__ cvdrift0, const T* __restrict__ gds2,
    const T* __restrict__ gds21, const T* __restrict__ gds22, long ld, const T* __restrict__ dPdrho,
    const T* __restrict__ theta0, T* gam_out, T* lam_out, T* X_out, T* dX_out, T* dth0_out, int* info_out,
    int chain, T w1, T w2) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int waves_per_line = (n_theta0 + chain - 1) / chain;
  const int nparts = (waves_per_line + wpb - 1) / wpb;
  int line, part;
  {   // XCD-aware block -> (line, part) map of k_gamma_scan
    const int b = blockIdx.x;
    const int chunk = b / (8 * nparts), r = b - chunk * 8 * nparts;
    const int lines_here = min(8, n_lines - chunk * 8);
    line = chunk * 8 + r % lines_here;
    part = r / lines_here;
    if (part >= nparts) return;
  }
  const int P = lds_pitch(N);
  T* A1 = smem; T* A3 = A1 + P; T* C0 = A3 + P; T* C1 = C0 + P; T* G0 = C1 + P; T* G1 = G0 + P; T* G2 = G1 + P;
  T* Xs = (M < 3 || X_out || dX_out) ? G2 + P + (size_t)wave * P : nullptr;
  {
    const long off = (long)line * ld;
    const T mdP = -dPdrho[line];
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const T B = bmag[off + j], gp = xabs(gradpar[off + j]);
      const T inv = T(1) / (gp * B);
      const int q = lpos(j);
      A1[q] = gp / B; A3[q] = inv / (B * B);
      C0[q] = mdP * cvdrift[off + j] * inv; C1[q] = mdP * cvdrift0[off + j] * inv;
      G0[q] = gds2[off + j]; G1[q] = gds21[off + j]; G2[q] = gds22[off + j];
    }
  }
  __syncthreads();
  const int first = (part * wpb + wave) * chain;
  T lam_p1 = T(0), lam_p2 = T(0);
  int have = 0;                       // eigenvalues of this wave's previous solves in hand (0, 1, 2)
  for (int q = 0; q < chain; ++q) {
    const int it0 = first + q;
    if (it0 >= n_theta0) break;       // wave-uniform
    // N is made opaque once per solve: otherwise every LDS address of setup() and of the growth-rate stage is
    // loop-invariant, gets hoisted out of this loop and is kept alive (spilled) across all the solves
    int Nq = N;
    asm volatile("" : "+s"(Nq));
    const T th0 = theta0[it0];
    SrcGeo<T> src{A1, A3, C0, C1, G0, G1, G2, th0, T(2) * th0, th0 * th0};
    WaveSolver<T, M> ws;
    SolveInfo inf{0, 0};
    const bool bad = ws.template setup<SrcGeo<T>, true>(src, Nq, h);
    const long sys = (long)line * n_theta0 + it0;
    T lam = T(0);
    if (!bad) {
      const T floor_w = T(4096) * T(64) * Eps<T>::v * ws.normA;
      T guess = have == 2 ? T(2) * lam_p1 - lam_p2 : lam_p1;
      T width = have == 2 ? xmax(w2 * xabs(lam_p1 - lam_p2), floor_w) : xmax(w1 * xabs(lam_p1), floor_w);
      if (have == 0) ws.trial_guess(guess, width);          // the first solve of a chain starts from the trial vector's bracket
      lam = ws.solve(inf, true, guess, width);
      lam_p2 = lam_p1; lam_p1 = lam; have = have < 2 ? have + 1 : 2;
    } else {
      inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); have = 0;
    }
    if constexpr (M >= 3)
      finish_chunk<T, M, SrcGeo<T>, true>(ws, src, Nq, h, Xs, lam, inf, sys, lam_out, gam_out, X_out, dX_out, dth0_out,
                                          info_out);
    else
      finish<T, M, SrcGeo<T>, true>(ws, src, Nq, h, Xs, lam, inf, sys, lam_out, gam_out, X_out, dX_out, dth0_out, info_out);
    if (Xs) wave_lds_sync();          // the row is reused by the next solve of this wave
  }
}

// ---------------------------------------------------------------- objective + Hellmann-Feynman gradient
// One wave per (alpha, theta0) evaluation point (utils.py:1632-1728 obj_w_grad given the three field
// lines alpha-d/2, alpha, alpha+d/2).  geo: [n_pts][3][8][ld] in the order bmag, gradpar, cvdrift,
// cvdrift0, gds2, gds21, gds22, gbdrift.  out: val[n_pts] = -gam, jac[n_pts][2] = (-dgam/dalpha, -dgam/dtheta0).
template <typename T>
__device__ __forceinline__ T line_dPdrho(const T* p, long ld, int N, int lane) {
  T s = T(0);
  for (int j = lane; j < N; j += kWave) { const T B = p[j]; s += (p[2 * ld + j] - p[7 * ld + j]) * B * B; }
  return T(-0.5) * wave_sum(s) / T(N);                                // utils.py:1657 / 1691 / 1703
}
template <typename T, int M>
__global__ void __launch_bounds__(256) k_obj_w_grad(int n_pts, int N, T h, const T* __restrict__ geo, long ld,
                                                    long line_stride,
                                                    const T* __restrict__ theta0, T del_alpha, T* val_out,
                                                    T* jac_out, T* gam_out, T* dalpha_out, T* dth0_out, int* info_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int pt = blockIdx.x * wpb + wave;
  const bool valid = pt < n_pts;
  const int ptc = valid ? pt : (n_pts - 1);
  const int P = lds_pitch(N);
  T* A1 = smem + (size_t)wave * 8 * P;
  T* A3 = A1 + P; T* C0 = A3 + P; T* C1 = C0 + P; T* G0 = C1 + P; T* G1 = G0 + P; T* G2 = G1 + P; T* Xs = G2 + P;
  // `ld` = distance between the 8 arrays of one line, `line_stride` = distance between consecutive lines:
  // (ld, 8 ld) for the [n_pts][3][8][ld] layout of the C ABI, (n_lines ld, ld) for the [8][n_lines][ld] planes the
  // geometry kernel writes
  const T* pl = geo + ((long)ptc * 3 + 0) * line_stride;
  const T* pc = geo + ((long)ptc * 3 + 1) * line_stride;
  const T* pr = geo + ((long)ptc * 3 + 2) * line_stride;
  const T dP_l = line_dPdrho(pl, ld, N, lane), dP_c = line_dPdrho(pc, ld, N, lane), dP_r = line_dPdrho(pr, ld, N, lane);
  for (int j = lane; j < N; j += kWave) {
    const T B = pc[j], gp = xabs(pc[ld + j]);
    const T inv = T(1) / (gp * B);
    const int q = lpos(j);
    A1[q] = gp / B; A3[q] = inv / (B * B);
    C0[q] = -dP_c * pc[2 * ld + j] * inv; C1[q] = -dP_c * pc[3 * ld + j] * inv;
    G0[q] = pc[4 * ld + j]; G1[q] = pc[5 * ld + j]; G2[q] = pc[6 * ld + j];
  }
  wave_lds_sync();   // the staging rows are private to this wave
  const T th0 = theta0[ptc];
  SrcGeo<T> src{A1, A3, C0, C1, G0, G1, G2, th0, T(2) * th0, th0 * th0};
  WaveSolver<T, M> ws;
  SolveInfo inf{0, 0};
  const bool bad = ws.template setup<SrcGeo<T>, true>(src, N, h);
  T lam = T(0);
  if (!bad) { T g_, w_; ws.trial_guess(g_, w_); lam = ws.solve(inf, true, g_, w_); }
  else { inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); }
  AlphaTangent<T> tan{pl, pr, ld, -dP_l, -dP_r, th0, T(2) * th0, th0 * th0, T(1) / del_alpha};
  // results land in per-point scratch slots of the output arrays, then are folded into (val, jac)
  finish<T, M, SrcGeo<T>, true, AlphaTangent<T>>(ws, src, N, h, Xs, lam, inf, ptc, nullptr, valid ? gam_out : nullptr,
                                                 nullptr, nullptr, valid ? dth0_out : nullptr,
                                                 valid ? info_out : nullptr, &tan, valid ? dalpha_out : nullptr);
  if (valid && lane == 0) {      // lane 0 wrote the three scratch values itself
    val_out[pt] = -gam_out[pt];                                      // utils.py:1728
    jac_out[2 * pt] = -dalpha_out[pt];
    jac_out[2 * pt + 1] = -dth0_out[pt];
  }
}

// ---------------------------------------------------------------- one round of the device-resident maximiser (row F2)
// One block (4 waves) per point of the batch; see ibs_refine.hpp.  The objective is utils.py:1632-1728 (as k_obj_w_grad),
// the optimizer step ibs_lbfgsb2.hpp (= scipy's L-BFGS-B for ball_scan.py:307-314) on the point's state, and the block
// that finishes last re-packs the batch (finished points leave it) for the next round.
//   all waves : ONE pass over the three lines -- centre line staged as the 7 derived arrays of the scan kernels (c without
//               its factor -dPdrho), alpha-tangent of g and f (utils.py:1707-1718) and the two halves of c's, partial sums
//               of dPdrho of the three lines (utils.py:1657 / 1691 / 1703) -- 24 coalesced loads per grid point in flight
//   wave 0    : set-up, eigen-solve (warm-started), eigenvector -> X in LDS
//   all waves : Simpson sums of the growth rate and the two Hellmann-Feynman derivatives (utils.py:1618-1621, 1676-1680,
//               1721-1725), a quarter of the grid each
//   wave 0    : gam, jac, optimizer step, state write-back, last-block-done compaction
template <typename T, int M>
__global__ void __launch_bounds__(256) k_refine_eval(RefineEvalArgs<T> a) {
  static_assert(sizeof(T) == 8, "the refinement is an FP64 path");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int n_c = a.ctrl->n_c;
  const int slot = blockIdx.x;
  if (slot >= n_c) return;                                 // (block-uniform: the grid was sized for an earlier, larger batch)
  const T h = a.ctrl->h;
  IBS_PROBE_AT(0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = a.N;
  const int P = lds_pitch(N);
  T* A1 = smem; T* A3 = A1 + P; T* C0 = A3 + P; T* C1 = C0 + P; T* G0 = C1 + P; T* G1 = G0 + P; T* G2 = G1 + P; T* Xs = G2 + P;
  T* Tg = Xs + P; T* Tf = Tg + P; T* Tcr = Tf + P; T* Tcl = Tcr + P;
  T* s_red = smem + (size_t)(a.lds_tangent ? 12 : 8) * P;       // [4 waves][8] partial sums
  double* s_state = s_red + 32;
  constexpr int kStateWords = (int)(sizeof(RefineState) / 8);
  lbfgsb2::Work* s_work = reinterpret_cast<lbfgsb2::Work*>(s_state + kStateWords);   // the optimizer step's run-time-indexed arrays
  const int k = a.idx[slot];
  const T th0 = a.th0[slot], two_th0 = T(2) * th0, th0sq = th0 * th0;
  const T inv_da = T(1) / a.prm.del_alpha;
  const long ld = a.ld, as = (long)a.plane;                // line pitch / distance between the 8 arrays of a line
  const T* pl = a.geo + (long)(3 * slot) * ld; const T* pc = pl + ld; const T* pr = pc + ld;
  {
    T sl = T(0), sc_ = T(0), sr = T(0);
#pragma unroll 2
    for (int j = threadIdx.x; j < N; j += 256) {
      const int q = lpos(j);
      const T B = pc[j], gp = xabs(pc[as + j]), cv = pc[2 * as + j], cv0 = pc[3 * as + j];
      const T g0 = pc[4 * as + j], g1 = pc[5 * as + j], g2 = pc[6 * as + j], gb = pc[7 * as + j];
      const T Bl = pl[j], gpl = xabs(pl[as + j]), cvl = pl[2 * as + j], cv0l = pl[3 * as + j];
      const T g0l = pl[4 * as + j], g1l = pl[5 * as + j], g2l = pl[6 * as + j], gbl = pl[7 * as + j];
      const T Br = pr[j], gpr = xabs(pr[as + j]), cvr = pr[2 * as + j], cv0r = pr[3 * as + j];
      const T g0r = pr[4 * as + j], g1r = pr[5 * as + j], g2r = pr[6 * as + j], gbr = pr[7 * as + j];
      sc_ += (cv - gb) * B * B; sl += (cvl - gbl) * Bl * Bl; sr += (cvr - gbr) * Br * Br;
      const T inv = T(1) / (gp * B);
      A1[q] = gp / B; A3[q] = inv / (B * B);               // g: utils.py:1560, f: utils.py:1562
      C0[q] = cv * inv; C1[q] = cv0 * inv;                 // c / (-dPdrho): utils.py:1561
      G0[q] = g0; G1[q] = g1; G2[q] = g2;
      if (a.lds_tangent) {
        // (g, c / (-dPdrho), f) of the lines at alpha -+ del_alpha/2 with theta0 folded in (utils.py:1692-1713)
        const T gdl = g0l + two_th0 * g1l + th0sq * g2l, gdr = g0r + two_th0 * g1r + th0sq * g2r;
        // (reciprocals good to ~1 ulp instead of IEEE divisions: the tangent is a difference quotient over del_alpha, its
        //  last bits carry no information; the centre line above keeps the divisions, whose rounding the eigenvector sees)
        const T rBl = fast_rcp(Bl), rBr = fast_rcp(Br), rgl = fast_rcp(gpl * Bl), rgr = fast_rcp(gpr * Br);
        const T gl_ = gpl * gdl * rBl, gr_ = gpr * gdr * rBr;
        const T fl_ = gdl * (rBl * rBl) * rgl, fr_ = gdr * (rBr * rBr) * rgr;
        Tg[q] = (gr_ - gl_) * inv_da; Tf[q] = (fr_ - fl_) * inv_da;          // utils.py:1716, 1718
        Tcr[q] = (cvr + th0 * cv0r) * rgr; Tcl[q] = (cvl + th0 * cv0l) * rgl;
      }
    }
    sl = wave_sum(sl); sc_ = wave_sum(sc_); sr = wave_sum(sr);
    if (lane == 0) { s_red[8 * wave] = sl; s_red[8 * wave + 1] = sc_; s_red[8 * wave + 2] = sr; }
    if (wave == 3) {                                       // the optimizer state of the point comes into LDS
      const double* src = reinterpret_cast<const double*>(a.st + k);
      for (int i = lane; i < kStateWords; i += kWave) s_state[i] = src[i];
    }
  }
  __syncthreads();
  const T dP_l = T(-0.5) * (((s_red[0] + s_red[8]) + s_red[16]) + s_red[24]) / T(N);
  const T dP_c = T(-0.5) * (((s_red[1] + s_red[9]) + s_red[17]) + s_red[25]) / T(N);
  const T dP_r = T(-0.5) * (((s_red[2] + s_red[10]) + s_red[18]) + s_red[26]) / T(N);
  IBS_PROBE_AT(2);
  SrcGeo<T, true> src{A1, A3, C0, C1, G0, G1, G2, th0, two_th0, th0sq, -dP_c};
  RefineState& S = *reinterpret_cast<RefineState*>(s_state);
  WaveSolver<T, M> ws;
  SolveInfo inf{0, 0};
  T lam = T(0), guess = T(0);
  bool bad = false, warm = false;
  T xe0 = T(0), xe1 = T(0);
  if (wave == 0) {
    bad = ws.template setup<SrcGeo<T, true>, true>(src, N, h);
    IBS_PROBE_AT(3);
    // warm start: lam of the point's previous evaluation moved along the Hellmann-Feynman gradient found there; the bracket
    // still moves on counts only, so a poor guess costs sweeps, never correctness
    xe0 = S.q.x[0]; xe1 = S.q.x[1];
    warm = !bad && S.have != 0;
    T g_ = T(0), w_ = T(0);
    if (warm) {
      const T lin = -(S.g_prev[0] * (xe0 - S.x_prev[0]) + S.g_prev[1] * (xe1 - S.x_prev[1]));
      guess = S.lam_prev + lin;
      const T floor_w = T(4096) * T(64) * Eps<T>::v * ws.normA;
      g_ = guess; w_ = xmax(xmax(T(0.5) * xabs(lin), T(4) * S.err_prev), floor_w);
    } else if (!bad) {
      ws.trial_guess(g_, w_);                              // first evaluation of a point: the trial vector's bracket
    }
    if (!bad) lam = ws.solve(inf, true, g_, w_);
    else { inf.status = 2; ws.sweep(ws.hi); ws.twisted(ws.hi); }
    IBS_PROBE_AT(4);
    // eigenvector -> X = v / max|v| with zero end points (utils.py:1602-1608) in the block's LDS row
    T x[M];
    ws.assemble(src, nullptr, N, h, x);
    T m = T(0);
#pragma unroll
    for (int i = 0; i < M; ++i) m = xmax(m, xabs(x[i]));
    m = uniform(wave_max(m));
    const int a0 = WaveSolver<T, M>::rows_start(lane, N - 2);
    const T rm = T(1) / m;
#pragma unroll
    for (int i = 0; i < M; ++i)
      if ((i < M - 1) || ws.has_last) Xs[lpos(a0 + i + 1)] = x[i] * rm;
    if (lane == 0) { Xs[lpos(0)] = T(0); Xs[lpos(N - 1)] = T(0); }
  }
  __syncthreads();
  {
    // dX (utils.py:1610-1616) in the branch-free form of finish(); Simpson weights with the common 1/3 dropped
    const T ih = T(1) / h;
    const T A_in = (T(2) / T(3)) * ih, B_in = -ih / T(12), A_e1 = T(0.5) * ih, A_e0 = T(2) * ih, B_e0 = T(-0.5) * ih;
    const AlphaTangent<T> tang{pl, pr, as, -dP_l, -dP_r, th0, two_th0, th0sq, inv_da};
    T y0 = T(0), y1 = T(0), hc = T(0), hg = T(0), hf = T(0), ac = T(0), ag = T(0), af = T(0);
    for (int j = threadIdx.x; j < N; j += 256) {
      const int jm1 = j > 0 ? j - 1 : 0, jm2 = j > 1 ? j - 2 : 0;
      const int jp1 = j < N - 1 ? j + 1 : N - 1, jp2 = j < N - 2 ? j + 2 : N - 1;
      const T X = Xs[lpos(j)];
      const T d1 = Xs[lpos(jp1)] - Xs[lpos(jm1)], d2 = Xs[lpos(jp2)] - Xs[lpos(jm2)];
      const bool end0 = (j == 0) || (j == N - 1), end1 = (j == 1) || (j == N - 2);
      const T A = end0 ? A_e0 : (end1 ? A_e1 : A_in), Bc = end0 ? B_e0 : (end1 ? T(0) : B_in);
      const T dX = xfma(A, d1, Bc * d2);
      const T w = end0 ? T(1) : ((j & 1) ? T(4) : T(2));
      const T X2 = w * (X * X), dX2 = w * (dX * dX);
      T g_, c_, f_, gt, ct, ft, ga, ca, fa;
      src.gcf_with_tangent(j, g_, c_, f_, gt, ct, ft);
      if (a.lds_tangent) {
        const int q = lpos(j);
        ga = Tg[q]; fa = Tf[q];
        ca = (-dP_r * Tcr[q] - (-dP_l) * Tcl[q]) * inv_da;                 // utils.py:1717 (each line with its own dPdrho)
      } else {
        tang.at(j, ga, ca, fa);
      }
      y0 += c_ * X2 - g_ * dX2;                                            // utils.py:1618
      y1 = xfma(f_, X2, y1);                                               // utils.py:1619
      hc = xfma(ct, X2, hc); hg = xfma(gt, dX2, hg); hf = xfma(ft, X2, hf);        // utils.py:1676-1680
      ac = xfma(ca, X2, ac); ag = xfma(ga, dX2, ag); af = xfma(fa, X2, af);        // utils.py:1721-1725
    }
    y0 = wave_sum(y0); y1 = wave_sum(y1); hc = wave_sum(hc); hg = wave_sum(hg); hf = wave_sum(hf);
    ac = wave_sum(ac); ag = wave_sum(ag); af = wave_sum(af);
    // (every wave read the dPdrho partial sums in s_red before the barrier above)
    if (lane == 0) {
      T* r = s_red + 8 * wave;
      r[0] = y0; r[1] = y1; r[2] = hc; r[3] = hg; r[4] = hf; r[5] = ac; r[6] = ag; r[7] = af;
    }
  }
  __syncthreads();
  if (wave != 0) return;                                   // (no block barrier below)
  IBS_PROBE_AT(5);
  T t[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) t[q] = ((s_red[q] + s_red[8 + q]) + s_red[16 + q]) + s_red[24 + q];
  const T gam = t[0] / t[1];                                                // utils.py:1621
  const T dth0 = t[2] / t[1] - t[3] / t[1] - gam * t[4] / t[1];             // utils.py:1676-1680
  const T dalpha = t[5] / t[1] - t[6] / t[1] - gam * t[7] / t[1];           // utils.py:1721-1725
  if (lane == 0) {
    if (a.info) a.info[slot] = inf.iters | (inf.status << 16);
    const double val = -gam;                                                // utils.py:1728
    const double g[2] = {-dalpha, -dth0};
    S.err_prev = warm ? xabs(lam - guess) : T(0);
    S.lam_prev = lam; S.x_prev[0] = xe0; S.x_prev[1] = xe1; S.g_prev[0] = g[0]; S.g_prev[1] = g[1];
    S.have = bad ? 0 : 1;
    S.sweeps += inf.iters;
    S.nev++;
    S.active = lbfgsb2::step(S.q, val, g, *s_work) ? 1 : 0;
  }
  IBS_PROBE_AT(6);
  wave_lds_sync();
  {
    double* dst = reinterpret_cast<double*>(a.st + k);
    for (int i = lane; i < kStateWords; i += kWave) dst[i] = s_state[i];
  }
  IBS_PROBE_AT(7);
  // ---- last block done: compact the batch for the next round (ibs_refine.hpp)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  int last = 0;
  if (lane == 0) last = (__hip_atomic_fetch_add(&a.ctrl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_c - 1) ? 1 : 0;
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  int count = 0;
  for (int base = 0; base < n_c; base += kWave) {
    const int j = base + lane;
    const int kk = j < n_c ? a.idx[j] : -1;
    const bool act = kk >= 0 && a.st[kk].active != 0;
    const unsigned long long mask = __ballot(act);
    const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
    if (act) {
      const double x[2] = {a.st[kk].q.x[0], a.st[kk].q.x[1]};
      a.idx[pos] = kk;
      refine_emit(x, a.prm.del_alpha, pos, min(max(a.pt_surf[kk], 0), a.prm.n_surf - 1), a.line_surf, a.line_alpha, a.th0);
    }
    count += __popcll(mask);
  }
  if (lane == 0) {
    const int r = a.ctrl->round + 1;
    a.ctrl->round = r; a.ctrl->n_c = count; a.ctrl->n_lines = 3 * count;
    __hip_atomic_store(&a.ctrl->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (r < a.hist_len) __hip_atomic_store(&a.hist[r], count + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------- Sturm count at given shifts
// The bandwidth kernel of the path ("batched Sturm/tridiag sweep"): one forward sweep per system, no
// scaling set-up, no divisions.  Works on the un-normalised three-term recurrence
//     u[r+1] = -(d[r] - sig f[r]) u[r] - e[r]^2 u[r-1]
// (signs only matter, so the 2x2 scan carries no exponent).  Each array is streamed through a per-wave
// LDS row with 16-byte coalesced loads and handed to the lanes as contiguous chunks.
// A wave's row of N values held in registers as KP 16-byte pieces per lane (coalesced: lane l owns pieces
// l, l+64, ...), loaded from the first 16-byte aligned element on; the (at most one) head and tail elements
// are loaded singly.  All three coefficient rows are put in flight before any of them is consumed.
template <typename T, int KP>
struct RowRegs {
  double2 v[KP];
  T head_v, tail_v;
  int head;
  __device__ __forceinline__ void load(const T* __restrict__ src, int N, int lane) {
    static_assert(sizeof(T) == 8, "f64 rows");
    head = (reinterpret_cast<uintptr_t>(src) & 15) ? 1 : 0;
    const int npair = (N - head) / 2;
    const double2* s2 = reinterpret_cast<const double2*>(src + head);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int p = lane + k * kWave;
      v[k] = (p < npair) ? s2[p] : double2{0.0, 0.0};
    }
    head_v = src[0];
    tail_v = src[N - 1];
  }
  __device__ __forceinline__ void store(T* dst, int N, int lane) const {
    const int npair = (N - head) / 2;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int p = lane + k * kWave;
      if (p < npair) { dst[head + 2 * p] = v[k].x; dst[head + 2 * p + 1] = v[k].y; }
    }
    if (lane == 0) { dst[0] = head_v; dst[N - 1] = tail_v; }
  }
};

template <typename T>
struct M2s { T a, b, c, d; };
template <typename T>
__device__ __forceinline__ M2s<T> muls(const M2s<T>& A, const M2s<T>& B, bool norm) {
  M2s<T> R;
  R.a = xfma(A.a, B.a, A.b * B.c); R.b = xfma(A.a, B.b, A.b * B.d);
  R.c = xfma(A.c, B.a, A.d * B.c); R.d = xfma(A.c, B.b, A.d * B.d);
  if (norm) {
    const int ex = fexp(xmax(xmax(xabs(R.a), xabs(R.b)), xmax(xabs(R.c), xabs(R.d))));
    R.a = xldexp(R.a, -ex); R.b = xldexp(R.b, -ex); R.c = xldexp(R.c, -ex); R.d = xldexp(R.d, -ex);
  }
  return R;
}
template <typename T, int CTRL, int ROWMASK>
__device__ __forceinline__ M2s<T> dpp_fetch_s(const M2s<T>& s) {
  M2s<T> r;
  r.a = dppz_t<CTRL, ROWMASK>(s.a); r.b = dppz_t<CTRL, ROWMASK>(s.b);
  r.c = dppz_t<CTRL, ROWMASK>(s.c); r.d = dppz_t<CTRL, ROWMASK>(s.d);
  return r;
}

template <typename T, int M>
__global__ void __launch_bounds__(256) k_sturm_count(long n_sys, int N, T h, const T* __restrict__ g,
                                                     const T* __restrict__ c, const T* __restrict__ f, long ld,
                                                     const T* __restrict__ shift, int* count_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long sys = (long)blockIdx.x * wpb + wave;
  const bool valid = sys < n_sys;
  const long sysc = valid ? sys : (n_sys - 1);
  T* row = smem + (size_t)wave * N;
  const int n = N - 2;
  const int rem = n - kWave * (M - 1);
  const bool has_last = lane < rem;
  const int a = WaveSolver<T, M>::rows_start(lane, n);
  const T ih2 = T(1) / (h * h);
  const T sig = shift[sysc];
  T t[M], e2[M];
  constexpr int KP = M / 2 + 1;      // 16-byte pieces per lane: ceil(ceil(N/2)/64) <= M/2 + 1
  RowRegs<T, KP> rg, rc, rf;
  rg.load(g + sysc * ld, N, lane);
  rc.load(c + sysc * ld, N, lane);
  rf.load(f + sysc * ld, N, lane);
  // g -> e = half-grid g / h^2 (utils.py:1574-1576)
  rg.store(row, N, lane);
  wave_lds_sync();
  {
    T gprev = row[a], gcur = row[a + 1];
    T e_lo = T(0.5) * (gprev + gcur) * ih2;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if ((i < M - 1) || has_last) {
        const T gnext = row[a + i + 2];
        const T e_hi = T(0.5) * (gcur + gnext) * ih2;
        e2[i] = e_lo * e_lo;
        t[i] = -(e_lo + e_hi);          // d = c - (e_lo + e_hi), c added below
        gcur = gnext; e_lo = e_hi;
      } else { e2[i] = T(0); t[i] = T(0); }
    }
  }
  wave_lds_sync();
  rc.store(row, N, lane);
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < M; ++i)
    if ((i < M - 1) || has_last) t[i] += row[a + i + 1];
  wave_lds_sync();
  rf.store(row, N, lane);
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < M; ++i)
    if ((i < M - 1) || has_last) t[i] = xfma(-sig, row[a + i + 1], t[i]);   // t = d - sig f
  // chunk transfer matrix: (u_a, u_{a-1}) -> (u_{a+cnt}, u_{a+cnt-1})
  T fA = T(1), fAp = T(0), fB = T(0), fBp = T(1);
#pragma unroll
  for (int i = 0; i < M; ++i) {
    if ((i < M - 1) || has_last) {
      const T nA = xfma(-t[i], fA, -(e2[i] * fAp)), nB = xfma(-t[i], fB, -(e2[i] * fBp));
      fAp = fA; fA = nA; fBp = fB; fB = nB;
    }
  }
  M2s<T> P{fA, fB, fAp, fBp};
  P = muls(P, M2s<T>{T(1), T(0), T(0), T(1)}, true);
  {
    const int l16 = lane & 15, rw = lane >> 4;
    { const M2s<T> F = dpp_fetch_s<T, 0x111, 0xF>(P); if (l16 >= 1) P = muls(P, F, false); }
    { const M2s<T> F = dpp_fetch_s<T, 0x112, 0xF>(P); if (l16 >= 2) P = muls(P, F, true); }
    { const M2s<T> F = dpp_fetch_s<T, 0x114, 0xF>(P); if (l16 >= 4) P = muls(P, F, false); }
    { const M2s<T> F = dpp_fetch_s<T, 0x118, 0xF>(P); if (l16 >= 8) P = muls(P, F, true); }
    { const M2s<T> F = dpp_fetch_s<T, 0x142, 0xA>(P); if (rw & 1) P = muls(P, F, false); }
    { const M2s<T> F = dpp_fetch_s<T, 0x143, 0xC>(P); if (rw >= 2) P = muls(P, F, true); }
  }
  T zc = dpp_t<0x138, 0xF>(T(1), P.a), zp = dpp_t<0x138, 0xF>(T(0), P.c);   // incoming (u_a, u_{a-1})
  // (sign changes counted from ONE incoming pair per lane; the step into the next lane's first row belongs to that lane:
  //  see WaveSolver::sweep_fwd)
  const int ncount = (has_last ? M : M - 1) - (lane == kWave - 1 ? 0 : 1);
  int cnt = __popcll(__ballot(sign_differs(zc, zp)));
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const bool act = (i < M - 1) || has_last;
    const T zn = xfma(-t[i], zc, -(e2[i] * zp));
    const bool flip = (i < M - 2 || i < ncount) && sign_differs(zn, zc);
    cnt += __popcll(__ballot(flip));
    if (act) { zp = zc; zc = zn; }
  }
  if (valid && lane == 0) count_out[sys] = cnt;
}

// ---------------------------------------------------------------- launchers
template <typename T>
static hipError_t launch_gcf(const GcfArgs<T>& a, hipStream_t st) {
  const int wpb = a.wpb;
  const size_t lds = (size_t)wpb * 3 * lds_pitch(a.N) * sizeof(T);
  const long nblk = (a.n_sys + wpb - 1) / wpb;
  auto kern = k_solve_gcf<T, IBS_M>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld,
                     a.lam, a.gam, a.X, a.dX, a.info, a.gh, a.flags);
  note_launch(nblk, wpb * 64, "ibs::k_solve_gcf<%s, %d>", type_name<T>(), IBS_M);
  return hipGetLastError();
}
static hipError_t launch_gcf_wide(const GcfArgs<float>& a, hipStream_t st) {
  const int wpb = a.wpb;
  const size_t lds = (size_t)wpb * 3 * lds_pitch(a.N) * sizeof(double);
  const long nblk = (a.n_sys + wpb - 1) / wpb;
  auto kern = k_solve_gcf_wide<IBS_M>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld,
                     a.lam, a.gam, a.X, a.dX, a.info, a.flags);
  note_launch(nblk, wpb * 64, "ibs::k_solve_gcf_wide<%d>", IBS_M);
  return hipGetLastError();
}
template <typename TI>
static hipError_t launch_gcf_direct(const GcfArgs<TI>& a, hipStream_t st) {
  if constexpr (IBS_M >= IBS_DIRECT_MIN_M) {
    const int wpb = a.wpb > 0 ? a.wpb : 4;
    const size_t lds = (a.X || a.dX) ? (size_t)wpb * lds_pitch(a.N) * sizeof(double) : 0;
    const long nblk = (a.n_sys + wpb - 1) / wpb;
    constexpr bool w2 = direct_two_waves(IBS_M, sizeof(TI) == 4);
    void (*kern)(long, int, double, const TI*, const TI*, const TI*, long, TI*, TI*, TI*, TI*, int*, int, int*, long*, double*);
    if constexpr (w2) kern = k_solve_gcf_direct_w2<double, IBS_M, TI>;      // (only the form that runs is instantiated)
    else kern = k_solve_gcf_direct<double, IBS_M, TI>;
    if (lds) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, (double)a.h, a.g, a.c, a.f, a.ld,
                       a.lam, a.gam, a.X, a.dX, a.info, a.flags, a.fix_count, a.fix_sys, a.fix_center);
    note_launch(nblk, wpb * 64, w2 ? "ibs::k_solve_gcf_direct_w2<double, %d, %s>" : "ibs::k_solve_gcf_direct<double, %d, %s>", IBS_M, type_name<TI>());
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
// second launch of the big-batch forms: the suspects they listed (k_fix_gcf)
template <typename TI>
static hipError_t launch_gcf_fix(const GcfArgs<TI>& a, hipStream_t st) {
  const size_t lds = ((size_t)3 * (a.N - 2) + lds_pitch(a.N)) * sizeof(double);
  long nblk = a.n_sys < 128 ? a.n_sys : 128;          // (a few dozen suspects per million systems; an empty pass of 1,024 blocks cost 25 us)
  auto kern = k_fix_gcf<IBS_M, TI>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64), lds, st, a.fix_count, a.fix_sys, a.fix_center, a.N, (double)a.h, a.g, a.c, a.f,
                     a.ld, a.lam, a.gam, a.X, a.dX, a.info);
  return hipGetLastError();
}
#ifdef IBS_WITH_F32
// the occupancy cap pays where the allocator would otherwise take AGPRs (one wave per SIMD): checked per M in the resource table
constexpr bool f32lam_two_waves(int M) { return M >= 21 && M <= 30; }   // (M = 32: 4.4 against 5.7e7 solves/s capped against free, rough family; tools/bench_f32lam.py)
static hipError_t launch_gcf_f32lam_direct(const GcfArgs<float>& a, hipStream_t st) {
  if constexpr (IBS_M >= IBS_F32LAM_DIRECT_MIN_M) {
    const int wpb = a.wpb > 0 ? a.wpb : 4;
    const long nblk = (a.n_sys + wpb - 1) / wpb;
    constexpr bool w2 = f32lam_two_waves(IBS_M);
    void (*kern)(long, int, float, const float*, const float*, const float*, long, float*, int*);
    if constexpr (w2) kern = k_solve_gcf_f32lam_direct_w2<IBS_M>;
    else kern = k_solve_gcf_f32lam_direct<IBS_M>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), 0, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld, a.lam, a.info);
    note_launch(nblk, wpb * 64, w2 ? "ibs::k_solve_gcf_f32lam_direct_w2<%d>" : "ibs::k_solve_gcf_f32lam_direct<%d>", IBS_M);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
#endif
template <typename T>
static hipError_t launch_gcf_rows(const GcfArgs<T>& a, hipStream_t st) {
  if constexpr (IBS_M >= 3) {
    const int wpb = a.wpb;
    const size_t lds = (size_t)wpb * lds_pitch(a.N) * sizeof(T);
    const long nblk = (a.n_sys + wpb - 1) / wpb;
    auto kern = k_solve_gcf_rows<T, IBS_M>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld,
                       a.lam, a.gam, a.X, a.dX, a.info, a.flags);
    note_launch(nblk, wpb * 64, "ibs::k_solve_gcf_rows<%s, %d, %s>", type_name<T>(), IBS_M, type_name<T>());
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
#ifdef IBS_WITH_F32
static hipError_t launch_gcf_rows_wide(const GcfArgs<float>& a, hipStream_t st) {
  if constexpr (IBS_M >= 3) {
    const int wpb = a.wpb;
    const size_t lds = (size_t)wpb * lds_pitch(a.N) * sizeof(double);
    const long nblk = (a.n_sys + wpb - 1) / wpb;
    auto kern = k_solve_gcf_rows<double, IBS_M, float>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, (double)a.h, a.g, a.c, a.f, a.ld,
                       a.lam, a.gam, a.X, a.dX, a.info, a.flags);
    note_launch(nblk, wpb * 64, "ibs::k_solve_gcf_rows<double, %d, float>", IBS_M);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
#endif
template <typename T>
static hipError_t launch_scan(const ScanArgs<T>& a, hipStream_t st) {
  const int wpb = a.wpb;
  const size_t lds = (size_t)(7 + wpb) * lds_pitch(a.N) * sizeof(T) + (a.pack ? 128 : 0);
  auto kern = k_gamma_scan<T, IBS_M>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  dim3 grid((unsigned)(((a.n_theta0 + wpb - 1) / wpb) * a.n_lines));
  hipLaunchKernelGGL(kern, grid, dim3(wpb * 64), lds, st, a.n_lines, a.n_theta0, a.N, a.h, a.bmag, a.gradpar,
                     a.cvdrift, a.cvdrift0, a.gds2, a.gds21, a.gds22, a.ld, a.dPdrho, a.theta0, a.gam, a.lam, a.X,
                     a.dX, a.dth0, a.info, a.lam_guess, a.guess_width, a.lines_per_surf, a.surf_counter, a.pack, a.pack_mode,
                     a.t0_stride);
  note_launch(grid.x, wpb * 64, "ibs::k_gamma_scan<%s, %d>", type_name<T>(), IBS_M);
  return hipGetLastError();
}
template <typename T>
static hipError_t launch_scan_chain(const ScanArgs<T>& a, hipStream_t st) {
  const int wpb = a.wpb, chain = a.chain;
  const bool need_x = (IBS_M < 3) || a.X || a.dX;          // (the LDS-round-trip finish of M < 3 always uses the row)
  const size_t lds = (size_t)(7 + (need_x ? wpb : 0)) * lds_pitch(a.N) * sizeof(T);
  auto kern = k_gamma_scan_chain<T, IBS_M>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int waves_per_line = (a.n_theta0 + chain - 1) / chain;
  dim3 grid((unsigned)(((waves_per_line + wpb - 1) / wpb) * a.n_lines));
  hipLaunchKernelGGL(kern, grid, dim3(wpb * 64), lds, st, a.n_lines, a.n_theta0, a.N, a.h, a.bmag, a.gradpar,
                     a.cvdrift, a.cvdrift0, a.gds2, a.gds21, a.gds22, a.ld, a.dPdrho, a.theta0, a.gam, a.lam, a.X,
                     a.dX, a.dth0, a.info, chain, a.chain_w1, a.chain_w2);
  note_launch(grid.x, wpb * 64, "ibs::k_gamma_scan_chain<%s, %d>", type_name<T>(), IBS_M);
  return hipGetLastError();
}
template <typename T>
static hipError_t launch_sturm(const SturmArgs<T>& a, hipStream_t st) {
  const int wpb = a.wpb;
  const size_t lds = (size_t)wpb * a.N * sizeof(T);
  const long nblk = (a.n_sys + wpb - 1) / wpb;
  auto kern = k_sturm_count<T, IBS_M>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld,
                     a.shift, a.count);
  note_launch(nblk, wpb * 64, "ibs::k_sturm_count<%s, %d>", type_name<T>(), IBS_M);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_grad(const GradArgs<T>& a, hipStream_t st) {
  const int wpb = a.wpb;
  const size_t lds = (size_t)wpb * 8 * lds_pitch(a.N) * sizeof(T);
  auto kern = k_obj_w_grad<T, IBS_M>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((a.n_pts + wpb - 1) / wpb), dim3(wpb * 64), lds, st, a.n_pts, a.N, a.h, a.geo,
                     a.arr_stride ? a.arr_stride : a.ld, a.line_stride ? a.line_stride : 8 * a.ld, a.theta0, a.del_alpha, a.val, a.jac, a.gam, a.dalpha, a.dth0, a.info);
  note_launch((a.n_pts + wpb - 1) / wpb, wpb * 64, "ibs::k_obj_w_grad<%s, %d>", type_name<T>(), IBS_M);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_refine_eval(const RefineEvalArgs<T>& a, hipStream_t st) {
  const size_t lds = (size_t)(a.lds_tangent ? 12 : 8) * lds_pitch(a.N) * sizeof(T) + 32 * sizeof(T) + sizeof(RefineState) + sizeof(lbfgsb2::Work);
  auto kern = k_refine_eval<T, IBS_M>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(a.n_c_max), dim3(256), lds, st, a);
  note_launch(a.n_c_max, 256, "ibs::k_refine_eval<%s, %d>", type_name<T>(), IBS_M);
  return hipGetLastError();
}

#define IBS_CAT2(a, b) a##b
#define IBS_CAT(a, b) IBS_CAT2(a, b)
struct IBS_CAT(Registrar, IBS_M) {
  IBS_CAT(Registrar, IBS_M)() {
    LaunchTable& t = launch_table();
    t.gcf_f64[IBS_M] = &launch_gcf<double>;
#if IBS_M >= 24
    // long grids (N >= 1475, e.g. N_zeta = 2048): one LDS row per wave -> four waves per CU (registers: one per SIMD) where
    // the three-row staging admits three (M = 24..28) or two (M = 29..32).  Below, the three-row kernel holds as many or
    // more waves and is faster (N_zeta = 1024, M = 16: 198 VGPRs, two waves per SIMD, 3.7e7 against 3.0e7 solves/s)
    t.gcf_rows_f64[IBS_M] = &launch_gcf_rows<double>;
#ifdef IBS_WITH_F32
    t.gcf_f32w_rows[IBS_M] = &launch_gcf_rows_wide;
#endif
#endif
#if IBS_M >= IBS_DIRECT_MIN_M
    // N > 578: big batches read their rows straight from global memory (no LDS staging: occupancy is the registers')
    t.gcf_direct_f64[IBS_M] = &launch_gcf_direct<double>;
#ifdef IBS_WITH_F32
    t.gcf_direct_f32w[IBS_M] = &launch_gcf_direct<float>;
#endif
#endif
#if defined(IBS_WITH_F32) && IBS_M >= IBS_F32LAM_DIRECT_MIN_M
    t.gcf_direct_f32lam[IBS_M] = &launch_gcf_f32lam_direct;
#endif
    t.gcf_fix_f64[IBS_M] = &launch_gcf_fix<double>;
#ifdef IBS_WITH_F32
    t.gcf_fix_f32w[IBS_M] = &launch_gcf_fix<float>;
#endif
    t.scan_f64[IBS_M] = &launch_scan<double>;
    t.scan_chain_f64[IBS_M] = &launch_scan_chain<double>;
    t.sturm_f64[IBS_M] = &launch_sturm<double>;
    t.grad_f64[IBS_M] = &launch_grad<double>;
    t.refine_f64[IBS_M] = &launch_refine_eval<double>;
#ifdef IBS_WITH_F32
    t.gcf_f32[IBS_M] = &launch_gcf<float>;
    t.gcf_f32_wide[IBS_M] = &launch_gcf_wide;
#endif
  }
};
static IBS_CAT(Registrar, IBS_M) IBS_CAT(registrar_instance_, IBS_M);

}  // namespace ibs

#ifdef IBS_PROBE
extern "C" int ibs_probe_read(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ibs::ibs_probe_buf), sizeof(long long) * n);
}
#endif
