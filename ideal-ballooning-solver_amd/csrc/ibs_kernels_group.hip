// Sub-wave kernels: G = 64/P systems per wavefront (ibs_group.hpp).  One translation unit per (P, M):
// compiled with -DIBS_P=<32|16> -DIBS_M=<rows per lane>.  Used by the C-ABI layer for large batches of
// short grids; results are the same quantities as the P = 64 kernels of ibs_kernels.hip.
#include <type_traits>
#include "ibs_group.hpp"
#include "ibs_launch.hpp"

#if !defined(IBS_M) || !defined(IBS_P)
#error "compile with -DIBS_P=<lanes per system> -DIBS_M=<rows per lane>"
#endif

namespace ibs {

// TI = the element type in memory: FP32 systems whose growth rate is wanted are widened (exactly) as they are read and solved
// by the FP64 solver (T = double, TI = float: half the bytes, the same arithmetic; see k_solve_gcf_wide in ibs_kernels.hip)
template <typename T, typename TI = T>
struct SrcGlobal {   // raw (g, c, f) straight from global memory: a lane's chunk is M contiguous values
  static constexpr bool kHasGh = false;
  const TI* gs; const TI* cs; const TI* fs;
  __device__ __forceinline__ T g(int j) const { return (T)gs[j]; }
  __device__ __forceinline__ T c(int j) const { return (T)cs[j]; }
  __device__ __forceinline__ T f(int j) const { return (T)fs[j]; }
};

template <typename T>
struct SrcGeoG {     // as SrcGeo (ibs_kernels.hip): 7 derived arrays of the line in LDS, theta0 per lane
  static constexpr bool kHasGh = false;
  const T* A1; const T* A3; const T* C0; const T* C1; const T* G0; const T* G1; const T* G2;
  T th0, two_th0, th0sq;
  // rows are padded: element j at lpos(j)
  __device__ __forceinline__ T gd(int j) const { const int q = lpos(j); return G0[q] + two_th0 * G1[q] + th0sq * G2[q]; }
  __device__ __forceinline__ T g(int j) const { return A1[lpos(j)] * gd(j); }
  __device__ __forceinline__ T c(int j) const { const int q = lpos(j); return C0[q] + th0 * C1[q]; }
  __device__ __forceinline__ T f(int j) const { return A3[lpos(j)] * gd(j); }
  __device__ __forceinline__ T gdp(int j) const { const int q = lpos(j); return T(2) * G1[q] + two_th0 * G2[q]; }
  __device__ __forceinline__ T g_t(int j) const { return A1[lpos(j)] * gdp(j); }
  __device__ __forceinline__ T c_t(int j) const { return C1[lpos(j)]; }
  __device__ __forceinline__ T f_t(int j) const { return A3[lpos(j)] * gdp(j); }
};

// one grid point of the Simpson sums (utils.py:1618-1619; with WT also the theta0-tangent sums, utils.py:1676-1680)
template <typename T, class Src, bool WT>
__device__ __forceinline__ void simpson_point_g(const Src& src, int j, T w, T X, T dX, T& y0, T& y1, T& hc, T& hg, T& hf) {
  const T X2 = w * (X * X), dX2 = w * (dX * dX);
  y0 += src.c(j) * X2 - src.g(j) * dX2;
  y1 = xfma(src.f(j), X2, y1);
  if constexpr (WT) { hc = xfma(src.c_t(j), X2, hc); hg = xfma(src.g_t(j), dX2, hg); hf = xfma(src.f_t(j), X2, hf); }
}

// eigenvector -> growth rate (utils.py:1601-1621, +1666-1680 when HF) with the eigenfunction kept in the lanes' row
// chunks (see finish_chunk in ibs_kernels.hip): the four
// stencil neighbours beyond a chunk come from the adjacent lanes of the group, every lane sums its own rows, the
// first and last lane of the group add the end points j = 0, N-1.  X / dX, when requested, go through the group's
// LDS row afterwards.
template <typename T, int M, int P, class Src, bool HF, typename TO = T>
__device__ __forceinline__ void finish_chunk_g(GroupSolver<T, M, P>& ws, const Src& src, int N, T h, T* Xs, T lam,
                                               int iters, int status, long sys, bool valid, TO* lam_out, TO* gam_out,
                                               TO* X_out, TO* dX_out, TO* dth0_out, int* info_out) {
  static_assert(M >= 3, "the halo exchange takes two rows from each neighbour lane");
  using GP = Grp<P>;
  const int lane = ws.lane, lg = ws.lg;
  const int n = N - 2;
  const bool hl = ws.has_last, first = lg == 0, last = lg == P - 1;

  T x[M];
  ws.assemble(src, N, h, x);
  T m = T(0);
#pragma unroll
  for (int i = 0; i < M; ++i) m = xmax(m, xabs(x[i]));
  m = GP::max(m, lane);
  const T rm = T(1) / m;
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] *= rm;                                   // utils.py:1605
  int a = GroupSolver<T, M, P>::rows_start(lg, n);
  asm volatile("" : "+v"(a));   // keep setup()'s addresses from being carried across the iteration (see finish_chunk)
  const T lastv = hl ? x[M - 1] : x[M - 2], last2 = hl ? x[M - 2] : x[M - 3];
  // neighbours across the wave; at the ends of a group the zero end points X[0], X[N-1] take their place
  const T sp1 = dpp_t<0x130, 0xF>(T(0), x[0]), sp2 = dpp_t<0x130, 0xF>(T(0), x[1]);      // wave_shl:1
  const T sm2 = dpp_t<0x138, 0xF>(T(0), last2), sm1 = dpp_t<0x138, 0xF>(T(0), lastv);    // wave_shr:1
  const T xp1 = last ? T(0) : sp1, xp2 = last ? T(0) : sp2;
  T xe[M + 4];
  xe[0] = first ? T(0) : sm2; xe[1] = first ? T(0) : sm1;
#pragma unroll
  for (int i = 0; i < M - 1; ++i) xe[i + 2] = x[i];
  xe[M + 1] = hl ? x[M - 1] : xp1; xe[M + 2] = hl ? xp1 : xp2; xe[M + 3] = xp2;
  const T ih = T(1) / h;
  const T A_in = (T(2) / T(3)) * ih, B_in = -ih / T(12), A_e1 = T(0.5) * ih, A_e0 = T(2) * ih, B_e0 = T(-0.5) * ih;
  bool do_hf = false;
  if constexpr (HF) do_hf = dth0_out != nullptr;
  T y0 = T(0), y1 = T(0), hc = T(0), hg = T(0), hf = T(0);
  const T w_even = ((a + 1) & 1) ? T(4) : T(2), w_odd = ((a + 1) & 1) ? T(2) : T(4);
  const int i_end = last ? (hl ? M - 1 : M - 2) : -1;
  // one-sided dX of the two end points, formed now so that the rows it needs are not kept alive across the row loop
  const T dX_end = xfma(A_e0, first ? x[0] : -lastv, B_e0 * (first ? x[1] : -last2));
  auto all_points = [&](auto with_tangent) {
    constexpr bool WT = HF && decltype(with_tangent)::value;
    if (first || last)                                                 // j = 0, N-1: utils.py:1610, 1614
      simpson_point_g<T, Src, WT>(src, first ? 0 : N - 1, T(1), T(0), dX_end, y0, y1, hc, hg, hf);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const bool act = (i < M - 1) || hl;
      const bool e1 = (i == 0 && first) || (i == i_end);               // j = 1, N-2: utils.py:1611, 1613
      const T A = e1 ? A_e1 : A_in, B = e1 ? T(0) : B_in;
      const T dX = xfma(A, xe[i + 3] - xe[i + 1], B * (xe[i + 4] - xe[i]));   // utils.py:1616
      const T w = act ? ((i & 1) ? w_odd : w_even) : T(0);
      simpson_point_g<T, Src, WT>(src, a + i + 1, w, xe[i + 2], dX, y0, y1, hc, hg, hf);
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (HF) {
    if (do_hf) all_points(std::true_type{});
    else all_points(std::false_type{});
  } else {
    all_points(std::false_type{});
  }
  y0 = GP::sum(y0, lane); y1 = GP::sum(y1, lane);
  const T gam = y0 / y1;                                               // utils.py:1621
  if constexpr (HF) {
    if (do_hf) {
      hc = GP::sum(hc, lane); hg = GP::sum(hg, lane); hf = GP::sum(hf, lane);
      const T jac = hc / y1 - hg / y1 - gam * hf / y1;                 // utils.py:1676-1680
      if (first && valid) dth0_out[sys] = (TO)jac;
    }
  }
  if (first && valid) {
    if (lam_out) lam_out[sys] = (TO)lam;
    if (gam_out) gam_out[sys] = (TO)gam;
    if (info_out) info_out[sys] = iters | (status << 16);
  }
  if (X_out || dX_out) {                   // kernel-uniform
    for (int pass = 0; pass < 2; ++pass) {
      TO* out = pass ? dX_out : X_out;
      if (!out) continue;
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const bool e1 = (i == 0 && first) || (i == i_end);
        const T A = e1 ? A_e1 : A_in, B = e1 ? T(0) : B_in;
        const T v = pass ? xfma(A, xe[i + 3] - xe[i + 1], B * (xe[i + 4] - xe[i])) : xe[i + 2];
        if ((i < M - 1) || hl) Xs[lpos(a + i + 1)] = v;
      }
      if (first) Xs[lpos(0)] = pass ? dX_end : T(0);
      if (last) Xs[lpos(N - 1)] = pass ? dX_end : T(0);
      wave_lds_sync();
      if (valid) for (int j = lg; j < N; j += P) out[sys * N + j] = (TO)Xs[lpos(j)];
    }
  }
}

// raw (g, c, f): wave w of block b solves systems (b*wpb + w)*G .. +G-1; dynamic LDS = wpb * G * lds_pitch(N) * sizeof(T) (X only)
// (2 waves per SIMD: the shift iteration keeps ~210 VGPRs live.  Forcing 3 waves (168 VGPRs) spills 45 of them:
//  6 % faster (9.6e7 vs 9.0e7 solves/s) but the scratch traffic doubles the HBM bytes per launch (8.1 vs 4.3 GB
//  PMC), so it is not used; 4 waves per SIMD halve the rate.)
template <typename T, int M, int P, typename TI = T>
__global__ void __launch_bounds__(256, 2) k_solve_gcf_g(long n_sys, int N, T h, const TI* __restrict__ g,
                                                     const TI* __restrict__ c, const TI* __restrict__ f, long ld,
                                                     TI* lam_out, TI* gam_out, TI* X_out, TI* dX_out, int* info_out, int flags,
                                                     int* fix_count, long* fix_sys, double* fix_center) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr int G = 64 / P;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int gid = lane / P;
  const long sys = ((long)blockIdx.x * wpb + wave) * G + gid;
  const bool valid = sys < n_sys;
  const long sysc = valid ? sys : (n_sys - 1);
  T* Xs = smem + ((size_t)wave * G + gid) * lds_pitch(N);
  SrcGlobal<T, TI> src{g + sysc * ld, c + sysc * ld, f + sysc * ld};
  GroupSolver<T, M, P> ws;
  const bool bad = ws.template setup<SrcGlobal<T, TI>, true>(src, N, h);
  int iters = 0, status = 0;
  T g_ = T(0), w_ = T(0);
  bool warm_ = false;
  ws.trial_guess(true, g_, w_, warm_);                 // cold solves start from the trial vector's bracket
  const T lam = ws.template solve<true>(bad, iters, status, warm_, g_, w_);
  if constexpr (sizeof(T) == 8) {
    // a group whose closing bracket failed the consistency check of solve<true> is LISTED (flags bit 0): k_fix_gcf (ibs_kernels.hip)
    // closes it again in division form and repeats its eigenvector stage after this launch; flags bit 1: only marked (diagnostics)
    if (flags & 1) {                       // (kernel-uniform)
      if (ws.suspect && valid && ws.lg == 0 && fix_count) {           // (rare)
        const int i = atomicAdd(fix_count, 1);
        fix_sys[i] = sysc;
        fix_center[i] = finite_of(ws.rho_last) ? (double)ws.rho_last : (double)lam;
      }
    } else if (flags != 0) {
      const int w = ws.why();
      status |= w != 0 ? (8 | (w << 5)) : 0;
    }
  }
  if (!gam_out && !X_out && !dX_out) {     // (kernel-uniform) eigenvalues only: no eigenvector, no Simpson sums
    if (ws.lg == 0 && valid) {
      if (lam_out) lam_out[sysc] = (TI)lam;
      if (info_out) info_out[sysc] = iters | (status << 16);
    }
    return;
  }
  finish_chunk_g<T, M, P, SrcGlobal<T, TI>, false, TI>(ws, src, N, h, Xs, lam, iters, status, sysc, valid, lam_out, gam_out, X_out,
                                                       dX_out, static_cast<TI*>(nullptr), info_out);
}

// geometry-fed scan: block = wpb waves of one line part; wave w solves theta0 indices (part*wpb + w)*G .. +G-1
template <typename T, int M, int P>
__global__ void __launch_bounds__(scan_max_threads_g(M), 2) k_gamma_scan_g(
    int n_lines, int n_theta0, int N, T h, const T* __restrict__ bmag, const T* __restrict__ gradpar,
    const T* __restrict__ cvdrift, const T* __restrict__ cvdrift0, const T* __restrict__ gds2,
    const T* __restrict__ gds21, const T* __restrict__ gds22, long ld, const T* __restrict__ dPdrho,
    const T* __restrict__ theta0, T* gam_out, T* lam_out, T* X_out, T* dX_out, T* dth0_out, int* info_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr int G = 64 / P;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int gid = lane / P;
  const int per_blk = wpb * G;
  const int nparts = (n_theta0 + per_blk - 1) / per_blk;
  int line, part;
  {   // XCD-aware block -> (line, part) map, see k_gamma_scan
    const int b = blockIdx.x;
    const int chunk = b / (8 * nparts), r = b - chunk * 8 * nparts;
    const int lines_here = min(8, n_lines - chunk * 8);
    line = chunk * 8 + r % lines_here;
    part = r / lines_here;
  }
  const int PT = lds_pitch(N);
  T* A1 = smem; T* A3 = A1 + PT; T* C0 = A3 + PT; T* C1 = C0 + PT; T* G0 = C1 + PT; T* G1 = G0 + PT; T* G2 = G1 + PT;
  T* Xs = G2 + PT + ((size_t)wave * G + gid) * PT;
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
  const int it0 = (part * wpb + wave) * G + gid;
  const bool valid = it0 < n_theta0;
  const int it0c = valid ? it0 : (n_theta0 - 1);
  const T th0 = theta0[it0c];
  SrcGeoG<T> src{A1, A3, C0, C1, G0, G1, G2, th0, T(2) * th0, th0 * th0};
  GroupSolver<T, M, P> ws;
  const bool bad = ws.template setup<SrcGeoG<T>, true>(src, N, h);
  int iters = 0, status = 0;
  T g_ = T(0), w_ = T(0);
  bool warm_ = false;
  ws.trial_guess(true, g_, w_, warm_);                 // cold solves start from the trial vector's bracket
  const T lam = ws.solve(bad, iters, status, warm_, g_, w_);
  const long sys = (long)line * n_theta0 + it0c;
  finish_chunk_g<T, M, P, SrcGeoG<T>, true>(ws, src, N, h, Xs, lam, iters, status, sys, valid, lam_out, gam_out, X_out,
                                      dX_out, dth0_out, info_out);
}

// Chained / warm-started form (see k_gamma_scan_chain in ibs_kernels.hip): group gid of wave w solves the theta0
// indices ((part*wpb + w)*G + gid)*chain + q, q = 0 .. chain-1, one after the other, each solve warm-started from the
// eigenvalue of the previous one; with lam_guess (chain = 1) every solve starts from the caller's eigenvalue of a
// nearby problem instead (ibs_gamma_scan_warm_f64).  No X / dX output => no per-group LDS row.
template <typename T, int M, int P>
__global__ void __launch_bounds__(scan_max_threads_g(M), 2) k_gamma_scan_g_chain(
    int n_lines, int n_theta0, int N, T h, const T* __restrict__ bmag, const T* __restrict__ gradpar,
    const T* __restrict__ cvdrift, const T* __restrict__ cvdrift0, const T* __restrict__ gds2,
    const T* __restrict__ gds21, const T* __restrict__ gds22, long ld, const T* __restrict__ dPdrho,
    const T* __restrict__ theta0, T* gam_out, T* lam_out, T* X_out, T* dX_out, T* dth0_out, int* info_out,
    int chain, T w1, T w2, const T* __restrict__ lam_guess, T guess_width) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr int G = 64 / P;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int gid = lane / P;
  const int per_blk = wpb * G * chain;
  const int nparts = (n_theta0 + per_blk - 1) / per_blk;
  int line, part;
  {   // XCD-aware block -> (line, part) map, see k_gamma_scan
    const int b = blockIdx.x;
    const int chunk = b / (8 * nparts), r = b - chunk * 8 * nparts;
    const int lines_here = min(8, n_lines - chunk * 8);
    line = chunk * 8 + r % lines_here;
    part = r / lines_here;
  }
  const int PT = lds_pitch(N);
  T* A1 = smem; T* A3 = A1 + PT; T* C0 = A3 + PT; T* C1 = C0 + PT; T* G0 = C1 + PT; T* G1 = G0 + PT; T* G2 = G1 + PT;
  T* Xs = (X_out || dX_out) ? G2 + PT + ((size_t)wave * G + gid) * PT : nullptr;
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
  const int first = ((part * wpb + wave) * G + gid) * chain;
  T lam_p1 = T(0), lam_p2 = T(0);
  int have = 0;
  for (int q = 0; q < chain; ++q) {
    const int it0 = first + q;
    const bool valid = it0 < n_theta0;
    if (!__any(valid)) break;
    int Nq = N;                          // opaque once per solve: keeps the LDS addresses from being hoisted out of the loop
    asm volatile("" : "+s"(Nq));
    const int it0c = valid ? it0 : (n_theta0 - 1);
    const T th0 = theta0[it0c];
    SrcGeoG<T> src{A1, A3, C0, C1, G0, G1, G2, th0, T(2) * th0, th0 * th0};
    GroupSolver<T, M, P> ws;
    const bool bad = ws.template setup<SrcGeoG<T>, true>(src, Nq, h);
    const long sys = (long)line * n_theta0 + it0c;
    int iters = 0, status = 0;
    const T floor_w = T(4096) * T(64) * Eps<T>::v * ws.normA;
    T guess = have == 2 ? T(2) * lam_p1 - lam_p2 : lam_p1;
    T width = have == 2 ? xmax(w2 * xabs(lam_p1 - lam_p2), floor_w) : xmax(w1 * xabs(lam_p1), floor_w);
    bool warm = have > 0;
    if (lam_guess) { guess = lam_guess[sys]; width = guess_width; warm = true; }
    ws.trial_guess(!warm, guess, width, warm);           // the first solve of a chain: the trial vector's bracket
    const T lam = ws.solve(bad, iters, status, warm, guess, width);
    lam_p2 = lam_p1; lam_p1 = lam; have = bad ? 0 : (have < 2 ? have + 1 : 2);
    finish_chunk_g<T, M, P, SrcGeoG<T>, true>(ws, src, Nq, h, Xs, lam, iters, status, sys, valid, lam_out, gam_out, X_out,
                                              dX_out, dth0_out, info_out);
    if (Xs) wave_lds_sync();
  }
}

template <typename T>
static hipError_t launch_gcf_g(const GcfArgs<T>& a, hipStream_t st) {
  constexpr int G = 64 / IBS_P;
  const int wpb = a.wpb;
  const size_t lds = (size_t)wpb * G * lds_pitch(a.N) * sizeof(T);
  const long nwaves = (a.n_sys + G - 1) / G;
  const long nblk = (nwaves + wpb - 1) / wpb;
  auto kern = k_solve_gcf_g<T, IBS_M, IBS_P>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld,
                     a.lam, a.gam, a.X, a.dX, a.info, a.flags, a.fix_count, a.fix_sys, a.fix_center);
  note_launch(nblk, wpb * 64, "ibs::k_solve_gcf_g<%s, %d, %d, %s>", type_name<T>(), IBS_M, IBS_P, type_name<T>());
  return hipGetLastError();
}
#ifdef IBS_WITH_F32
// FP32 in HBM, FP64 in the solver (the sub-wave sibling of k_solve_gcf_wide)
static hipError_t launch_gcf_g_wide(const GcfArgs<float>& a, hipStream_t st) {
  constexpr int G = 64 / IBS_P;
  const int wpb = a.wpb;
  const size_t lds = (size_t)wpb * G * lds_pitch(a.N) * sizeof(double);
  const long nwaves = (a.n_sys + G - 1) / G;
  const long nblk = (nwaves + wpb - 1) / wpb;
  auto kern = k_solve_gcf_g<double, IBS_M, IBS_P, float>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(wpb * 64), lds, st, a.n_sys, a.N, (double)a.h, a.g, a.c, a.f, a.ld,
                     a.lam, a.gam, a.X, a.dX, a.info, a.flags, a.fix_count, a.fix_sys, a.fix_center);
  note_launch(nblk, wpb * 64, "ibs::k_solve_gcf_g<double, %d, %d, float>", IBS_M, IBS_P);
  return hipGetLastError();
}
#endif
template <typename T>
static hipError_t launch_scan_g(const ScanArgs<T>& a, hipStream_t st) {
  constexpr int G = 64 / IBS_P;
  const int wpb = a.wpb;
  const size_t lds = (size_t)(7 + wpb * G) * lds_pitch(a.N) * sizeof(T);
  auto kern = k_gamma_scan_g<T, IBS_M, IBS_P>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int per_blk = wpb * G;
  dim3 grid((unsigned)(((a.n_theta0 + per_blk - 1) / per_blk) * a.n_lines));
  hipLaunchKernelGGL(kern, grid, dim3(wpb * 64), lds, st, a.n_lines, a.n_theta0, a.N, a.h, a.bmag, a.gradpar,
                     a.cvdrift, a.cvdrift0, a.gds2, a.gds21, a.gds22, a.ld, a.dPdrho, a.theta0, a.gam, a.lam, a.X,
                     a.dX, a.dth0, a.info);
  note_launch(grid.x, wpb * 64, "ibs::k_gamma_scan_g<%s, %d, %d>", type_name<T>(), IBS_M, IBS_P);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_scan_g_chain(const ScanArgs<T>& a, hipStream_t st) {
  constexpr int G = 64 / IBS_P;
  const int wpb = a.wpb, chain = a.chain < 1 ? 1 : a.chain;
  const bool need_x = a.X || a.dX;
  const size_t lds = (size_t)(7 + (need_x ? wpb * G : 0)) * lds_pitch(a.N) * sizeof(T);
  auto kern = k_gamma_scan_g_chain<T, IBS_M, IBS_P>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int per_blk = wpb * G * chain;
  dim3 grid((unsigned)(((a.n_theta0 + per_blk - 1) / per_blk) * a.n_lines));
  hipLaunchKernelGGL(kern, grid, dim3(wpb * 64), lds, st, a.n_lines, a.n_theta0, a.N, a.h, a.bmag, a.gradpar,
                     a.cvdrift, a.cvdrift0, a.gds2, a.gds21, a.gds22, a.ld, a.dPdrho, a.theta0, a.gam, a.lam, a.X,
                     a.dX, a.dth0, a.info, chain, a.chain_w1, a.chain_w2, a.lam_guess, a.guess_width);
  note_launch(grid.x, wpb * 64, "ibs::k_gamma_scan_g_chain<%s, %d, %d>", type_name<T>(), IBS_M, IBS_P);
  return hipGetLastError();
}

#define IBS_CAT3_(a, b, c) a##b##_##c
#define IBS_CAT3(a, b, c) IBS_CAT3_(a, b, c)
struct IBS_CAT3(RegistrarG, IBS_P, IBS_M) {
  IBS_CAT3(RegistrarG, IBS_P, IBS_M)() {
    LaunchTable& t = launch_table();
    constexpr int pi = (IBS_P == 32) ? 0 : 1;
    t.gcf_f64_g[pi][IBS_M] = &launch_gcf_g<double>;
#ifdef IBS_WITH_F32
    t.gcf_f32w_g[pi][IBS_M] = &launch_gcf_g_wide;
#endif
    t.scan_f64_g[pi][IBS_M] = &launch_scan_g<double>;
    t.scan_chain_f64_g[pi][IBS_M] = &launch_scan_g_chain<double>;
  }
};
static IBS_CAT3(RegistrarG, IBS_P, IBS_M) IBS_CAT3(registrar_g_instance_, IBS_P, IBS_M);

}  // namespace ibs
