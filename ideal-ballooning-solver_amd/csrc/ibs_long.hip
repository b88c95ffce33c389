// Grids beyond the register-resident kernels (N > 64 * kMaxM + 2 = 2050 points, up to kMaxLongN): the generic path that keeps the
// drop-in from refusing what the reference computes -- utils.py:1556-1624 accepts any length, and the reference's own grid rule
// N = 2 mpol ntor 4 + 1 (ball_scan.py:201-208) passes 2050 from mpol ntor > 256 on.
//
// One wavefront per system, everything in DIVISION form on the original rows (no scaled rows, nothing register-resident):
//   1. bounds        lam_max <= max c/f (Gershgorin), lam_max >= max d/f (unit vectors) and >= the Rayleigh
//                    quotients of four trial vectors sin^p(pi j / (N - 1)) taken in the same pass, ||A||; data checks
//   2. eigenvalue    64-way multisection on division-form Sturm counts (ibs_wave.hpp: multisect; the recurrence of SURVEY Appendix A /
//                    LAPACK dstebz, the rows passed through LDS in chunks: count_above_chunked) to a bracket of 2 eps ||A||: first on
//                    every 16th and every 8th grid point (two-grid start), then 3-5 passes on all rows (9 from the Gershgorin bracket alone)
//   3. eigenvector   twisted factorisation N_k D_k N_k^T of T - lam F (the getvec step of MRRR): forward pivots D+ (lane 0) and
//                    backward pivots D- (lane 1) in one serial pass, gamma_r = D+_r + D-_r - (d_r - lam f_r) in parallel, twist row
//                    k = argmin |gamma_r|, then z_k = 1, z_{r-1} = -e_r z_r / D+_{r-1} downwards (lane 0) and
//                    z_{r+1} = -e_{r+1} z_r / D-_{r+1} upwards (lane 1)
//   4. growth rate   X = z / max |z| with zero end points, FD2/FD4 derivative, composite Simpson quotient: the arithmetic of
//                    finish() in ibs_kernels.hip, i.e. utils.py:1601-1621
// Per-wave workspace in global memory: 3 N doubles (D+, D-, d - lam f; the last becomes z).  The grid is persistent: a fixed
// number of waves, each taking systems blockIdx.x, blockIdx.x + gridDim.x, ...
#include "ibs_wave.hpp"
#include "ibs_launch.hpp"

namespace ibs {

template <typename TI, bool HAS_GH>
struct SrcLong {
  static constexpr bool kHasGh = HAS_GH;
  const TI* gg; const TI* cg; const TI* fg; const TI* ghg;
  __device__ __forceinline__ double g(int j) const { return (double)gg[j]; }
  __device__ __forceinline__ double c(int j) const { return (double)cg[j]; }
  __device__ __forceinline__ double f(int j) const { return (double)fg[j]; }
  __device__ __forceinline__ double gh(int k) const { return (double)ghg[k]; }
  // e_k = half-grid g between grid points k and k + 1, over h^2 (utils.py:1574-1576, 1584-1592)
  __device__ __forceinline__ double e(int k, double ih2) const {
    if constexpr (HAS_GH) return (double)ghg[k] * ih2;
    else return 0.5 * ((double)gg[k] + (double)gg[k + 1]) * ih2;
  }
};

// Every S-th grid point of a system: the same operator on a coarser grid (two-grid start of the multisection, solve_long_one)
template <class Src>
struct SrcCoarse {
  static constexpr bool kHasGh = false;
  const Src& s; int S;
  __device__ __forceinline__ double g(int j) const { return s.g(S * j); }
  __device__ __forceinline__ double c(int j) const { return s.c(S * j); }
  __device__ __forceinline__ double f(int j) const { return s.f(S * j); }
  __device__ __forceinline__ double e(int k, double ih2c) const { return 0.5 * (s.g(S * k) + s.g(S * (k + 1))) * ih2c; }
};

// Division-form count of one system at 64 shifts (lane L = shift L) with the rows passed through LDS in chunks of kLongChunk: the 64
// lanes form d_r, e_r^2, f_r of a chunk together (coalesced loads), then every lane runs the recurrence over the chunk from LDS (all
// lanes read the same address: broadcast) -- the serial chain never waits for global memory (read per row it cost one exposed memory
// latency per four rows: 27 ms per system at N = 8,193; so: 3).  Same arithmetic as count_above_rows (ibs_wave.hpp).
constexpr int kLongChunk = 768;      // 18 KB per block: eight blocks per CU (the persistent grid of long_waves(), ibs_api.hip) stay resident -- the
                                     // recurrence is a chain of dependent divisions, two waves per SIMD overlap almost freely.  Chunks of 1,024: six
                                     // blocks per CU, 2,048 systems ran as 1,536 + 512 (24 ms at N = 16,385 against 15); chunks of 384, sixteen
                                     // blocks: one system 25 % slower (twice the chunk boundaries), 8,192 systems no faster
static_assert(8 * 3 * kLongChunk * sizeof(double) <= 160 * 1024, "eight blocks per CU");
constexpr int kVecChunk = kLongChunk / 2;      // pivots / eigenvector: two directions x (two operands + one result) in the same LDS
template <class Src>
__device__ __forceinline__ int count_above_chunked(const Src& src, int n, double ih2, double sig, double* lds, int lane) {
  constexpr double pivmin = 2.2250738585072014e-292;
  double* rd = lds; double* re2 = lds + kLongChunk; double* rf = lds + 2 * kLongChunk;
  int cnt = 0;
  double q = 1.0;
  for (int r0 = 0; r0 < n; r0 += kLongChunk) {
    const int m = n - r0 < kLongChunk ? n - r0 : kLongChunk;
    for (int i = lane; i < m; i += kWave) {
      const int r = r0 + i;
      const double e_lo = src.e(r, ih2), e_hi = src.e(r + 1, ih2);
      rd[i] = src.c(r + 1) - (e_lo + e_hi); re2[i] = e_lo * e_lo; rf[i] = src.f(r + 1);
    }
    wave_lds_sync();
    int i = 0;
    if (r0 == 0) {
      q = xfma(-sig, rf[0], rd[0]);
      q = xabs(q) < pivmin ? -pivmin : q;
      cnt += q > 0.0 ? 1 : 0;
      i = 1;
    }
#pragma unroll 8
    for (; i < m; ++i) {
      const double a = xfma(-sig, rf[i], rd[i]);
      q = xfma(-re2[i], fast_rcp(q), a);
      q = xabs(q) < pivmin ? -pivmin : q;
      cnt += q > 0.0 ? 1 : 0;
    }
    wave_lds_sync();
  }
  return cnt;
}

__device__ __forceinline__ void long_fence() {      // stores of two lanes, read by all lanes of the same wave afterwards
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <typename TI, bool HAS_GH>
__device__ __forceinline__ void solve_long_one(const SrcLong<TI, HAS_GH>& src, int N, double h, long sys, double* work, TI* lam_out,
                                               TI* gam_out, TI* X_out, TI* dX_out, int* info_out, double* lds) {
  constexpr double pivmin = 2.2250738585072014e-292;
  const int lane = threadIdx.x & 63;
  const int n = N - 2;
  const double ih2 = 1.0 / (h * h);
  // ---- 1. bounds and data checks (lanes strided over the rows)
  // The Gershgorin-type lower end max_r d_r / f_r is ~ -||A|| (d_r = c_r - 2 g / h^2) while lam_max is O(1): every factor 64 of
  // bracket is a pass over the rows.  Rayleigh quotients of four trial vectors x_j = sin^p(pi j / (N - 1)), p = 1, 4, 16, 64 (zero at
  // both ends like the eigenfunctions, ever more localised about the middle of the grid where ballooning modes sit) are rigorous
  // lower bounds of lam_max (Courant-Fischer) and cost this one parallel pass: on s-alpha and geometry lines the bracket starts
  // O(1) wide instead of O(||A||), three passes of nine less; on rough coefficients the bounds are poor and nothing changes.
  double vhi = -1e300, vlo = -1e300, vna = 0.0;
  double tn[4] = {0.0, 0.0, 0.0, 0.0}, td[4] = {0.0, 0.0, 0.0, 0.0};
  bool bad = false;
  const double dth = 3.141592653589793 / (double)(N - 1);
  double sdl, cdl;
  sincos(dth, &sdl, &cdl);
  for (int r = lane; r < n; r += kWave) {
    const int j = r + 1;
    const double e_lo = src.e(r, ih2), e_hi = src.e(j, ih2);
    const double cj = src.c(j), fj = src.f(j), gj = src.g(j);
    const double d = cj - (e_lo + e_hi);
    const double rf = 1.0 / fj;
    vhi = xmax(vhi, cj * rf); vlo = xmax(vlo, d * rf); vna = xmax(vna, (xabs(d) + e_lo + e_hi) * rf);
    bad = bad || !(fj > 0.0) || !(gj > 0.0) || !(e_lo > 0.0) || !(e_hi > 0.0) || !finite_of(cj) || !finite_of(fj) || !finite_of(e_lo + e_hi);
    double sj, cjs;
    sincos(dth * (double)j, &sj, &cjs);
    double xm = sj * cdl - cjs * sdl, x0 = sj, xp = sj * cdl + cjs * sdl;      // sin at j - 1, j, j + 1
    xm = j == 1 ? 0.0 : xm; xp = j == N - 2 ? 0.0 : xp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tn[k] = xfma(x0, xfma(e_lo, xm, xfma(d, x0, e_hi * xp)), tn[k]);
      td[k] = xfma(fj * x0, x0, td[k]);
      xm *= xm; xm *= xm; x0 *= x0; x0 *= x0; xp *= xp; xp *= xp;               // p -> 4 p
    }
  }
  if (lane == 0) bad = bad || !(src.g(0) > 0.0) || !(src.g(N - 1) > 0.0);
  const double normA = uniform(wave_max(vna));
  double hi = uniform(wave_max(vhi)) + 8.0 * Eps<double>::v * normA;
  double lo = uniform(wave_max(vlo)) - 8.0 * Eps<double>::v * normA;
  {
    double rho = -1e300;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double a = wave_sum(tn[k]), b = wave_sum(td[k]);
      const double q = a / b;
      rho = (b > 0.0 && finite_of(q) && q > rho) ? q : rho;
    }
    rho = uniform(rho) - (8.0 + 0.5 * (double)N) * Eps<double>::v * normA;       // (rounding of the sums: N terms)
    if (rho > lo && rho < hi) lo = rho;
  }
  int status = 0, passes = 0;
  double lam = 0.0;
  const bool want_vec = gam_out || X_out || dX_out;        // (kernel-uniform)
  if (__any(bad) || !finite_of(normA)) {
    status = 2;
    lam = __builtin_nan("");
  } else {
    // ---- 2. eigenvalue.  Two-grid start: where the trial vectors have brought the bracket down from O(||A||) to O(1) -- smooth
    // coefficients -- lam_max of the same operator on every 16th and every 8th grid point (a sixteenth / an eighth of a pass per
    // multisection pass, closed to 1e-7 of the bracket) differ from the fine grid's by C (16 h)^2 and C (8 h)^2: their Richardson
    // extrapolate, +- a quarter of their difference, is where the fine multisection starts (a bracket that misses is moved and
    // widened by multisect itself, at a pass per miss).  6 fine passes become 3-4 + 0.6.
    int coarse_passes = 0;
    if ((N - 1) % 16 == 0 && N >= 1025 && (hi - lo) < 1e-3 * normA) {
      double l16 = 0.0, l8 = 0.0;
      int p16 = 0, p8 = 0;
      const double stop_c = 1e-7 * (hi - lo) / (Eps<double>::v * normA);      // multisect ends at width <= stop eps ||A||
      const SrcCoarse<SrcLong<TI, HAS_GH>> c16{src, 16}, c8{src, 8};
      const int n16 = (N - 1) / 16 - 1, n8 = (N - 1) / 8 - 1;
      const bool ok16 = multisect<double>([&](double sig) { return count_above_chunked(c16, n16, ih2 * (1.0 / 256.0), sig, lds, lane); },
                                          lo, hi, normA, stop_c, lane, l16, p16);
      bool ok8 = false;
      if (ok16) {
        const double w8 = 1e-2 * (hi - lo);                                  // (the two coarse grids differ by far less on smooth data)
        ok8 = multisect<double>([&](double sig) { return count_above_chunked(c8, n8, ih2 * (1.0 / 64.0), sig, lds, lane); },
                                xmax(lo, l16 - w8), xmin(hi, l16 + w8), normA, stop_c, lane, l8, p8);
      }
      coarse_passes = (p16 + 15) / 16 + (p8 + 7) / 8;                         // (in units of a fine pass, rounded up)
      if (ok16 && ok8) {
        const double dl = l16 - l8;
        const double est = l8 - dl * (1.0 / 3.0);
        const double w = xmax(0.25 * xabs(dl), 8.0 * 1e-7 * (hi - lo));
        const double lo2 = xmax(lo, est - w), hi2 = xmin(hi, est + w);
        if (lo2 < hi2) { lo = lo2; hi = hi2; }
      }
    }
    if (!multisect<double>([&](double sig) { return count_above_chunked(src, n, ih2, sig, lds, lane); }, lo, hi, normA, 2.0, lane, lam, passes))
      status = 1;
    passes += coarse_passes;
  }
  double gam = __builtin_nan("");
  if (want_vec && status == 0) {
    double* Dp = work; double* Dm = work + N; double* A = work + 2 * (size_t)N;     // A[r] = d_r - lam f_r, later z_r
    // ---- 3a. pivots: lane 0 walks the rows upwards (D+), lane 1 downwards (D-).  The rows pass through LDS in chunks of kVecChunk
    // per direction: all 64 lanes form a_r and e_r^2 (coalesced loads), lanes 0 / 1 run the two recurrences from LDS eight rows at a
    // time (operands in registers before the dependent chain starts), all lanes write the pivots out (coalesced).  Read and written
    // row by row from the two lanes the chain waited for global memory at every step: 400 cycles per row, as long as the whole
    // multisection.
    {
      const int dsel = lane & 1;
      const double* xa = lds + dsel * 3 * kVecChunk; const double* xe = xa + kVecChunk; double* xq = lds + dsel * 3 * kVecChunk + 2 * kVecChunk;
      double q = 1.0;
      for (int c0 = 0; c0 < n; c0 += kVecChunk) {
        const int m = n - c0 < kVecChunk ? n - c0 : kVecChunk;
        for (int i = lane; i < m; i += kWave) {
          {
            const int r = c0 + i, j = r + 1;
            const double e_lo = src.e(r, ih2), e_hi = src.e(j, ih2);
            lds[i] = xfma(-lam, src.f(j), src.c(j) - (e_lo + e_hi)); lds[kVecChunk + i] = e_lo * e_lo;
          }
          {
            const int r = n - 1 - (c0 + i), j = r + 1;
            const double e_lo = src.e(r, ih2), e_hi = src.e(j, ih2);
            lds[3 * kVecChunk + i] = xfma(-lam, src.f(j), src.c(j) - (e_lo + e_hi)); lds[4 * kVecChunk + i] = e_hi * e_hi;
          }
        }
        wave_lds_sync();
        if (lane < 2) {
          for (int i0 = 0; i0 < m; i0 += 8) {
            double av[8], ev[8], qv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u < m ? i0 + u : m - 1; av[u] = xa[i]; ev[u] = xe[i]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (i0 + u < m) {
                q = (c0 + i0 + u == 0) ? av[u] : xfma(-ev[u], fast_rcp(q), av[u]);      // (the arithmetic of the counts)
                q = xabs(q) < pivmin ? -pivmin : q;
              }
              qv[u] = q;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u < m) xq[i0 + u] = qv[u];
          }
        }
        wave_lds_sync();
        for (int i = lane; i < m; i += kWave) {
          Dp[c0 + i] = lds[2 * kVecChunk + i]; A[c0 + i] = lds[i]; Dm[n - 1 - (c0 + i)] = lds[5 * kVecChunk + i];
        }
        wave_lds_sync();
      }
    }
    long_fence();
    // ---- 3b. twist row: the smallest |gamma_r|
    double best = 1e300;
    int bi = 0;
    for (int r = lane; r < n; r += kWave) {
      const double gm = xabs(Dp[r] + Dm[r] - A[r]);
      if (gm < best) { best = gm; bi = r; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const double b2 = __shfl_xor(best, d);
      const int i2 = __shfl_xor(bi, d);
      if (b2 < best || (b2 == best && i2 < bi)) { best = b2; bi = i2; }
    }
    const int k = __builtin_amdgcn_readfirstlane(bi);
    long_fence();                                           // (A is overwritten by z below: every lane has read it)
    // ---- 3c. eigenvector from the twist row outwards (staged like 3a: lane 0 downwards from row k, lane 1 upwards)
    //   downwards: z_t = -e_{t+1} z_{t+1} / D+_t;  upwards: z_t = -e_t z_{t-1} / D-_t
    {
      const int dsel = lane & 1;
      const double* xe = lds + dsel * 3 * kVecChunk; double* xz = lds + dsel * 3 * kVecChunk + 2 * kVecChunk;
      const int steps_dn = k, steps_up = n - 1 - k;
      const int smax = steps_dn > steps_up ? steps_dn : steps_up;
      double z = 1.0;
      if (lane == 0) A[k] = 1.0;
      for (int c0 = 0; c0 < smax; c0 += kVecChunk) {
        const int md = steps_dn - c0 < 0 ? 0 : (steps_dn - c0 < kVecChunk ? steps_dn - c0 : kVecChunk);
        const int mu = steps_up - c0 < 0 ? 0 : (steps_up - c0 < kVecChunk ? steps_up - c0 : kVecChunk);
        for (int i = lane; i < kVecChunk; i += kWave) {
          if (i < md) { const int t = k - 1 - (c0 + i); lds[i] = -src.e(t + 1, ih2) / Dp[t]; }        // z_t / z_{t+1}: formed by all lanes,
          if (i < mu) { const int t = k + 1 + (c0 + i); lds[3 * kVecChunk + i] = -src.e(t, ih2) / Dm[t]; } // the chain is one product per row
        }
        wave_lds_sync();
        if (lane < 2) {
          const int m = dsel ? mu : md;
          for (int i0 = 0; i0 < m; i0 += 8) {
            double rv[8], zv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) rv[u] = xe[i0 + u < m ? i0 + u : m - 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (i0 + u < m) z *= rv[u];
              zv[u] = z;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u < m) xz[i0 + u] = zv[u];
          }
        }
        wave_lds_sync();
        for (int i = lane; i < kVecChunk; i += kWave) {
          if (i < md) A[k - 1 - (c0 + i)] = lds[2 * kVecChunk + i];
          if (i < mu) A[k + 1 + (c0 + i)] = lds[5 * kVecChunk + i];
        }
        wave_lds_sync();
      }
    }
    long_fence();
    // ---- 4. growth rate (utils.py:1601-1621; the arithmetic of finish() in ibs_kernels.hip)
    double m = 0.0;
    for (int r = lane; r < n; r += kWave) m = xmax(m, xabs(A[r]));
    m = uniform(wave_max(m));
    const double rm = 1.0 / m;
    auto Xat = [&](int j) { return (j <= 0 || j >= N - 1) ? 0.0 : A[j - 1] * rm; };     // utils.py:1605, 1607-1608
    const double ih = 1.0 / h;
    const double A_in = (2.0 / 3.0) * ih, B_in = -ih / 12.0, A_e1 = 0.5 * ih, A_e0 = 2.0 * ih, B_e0 = -0.5 * ih;
    double y0 = 0.0, y1 = 0.0;
    for (int j0 = 0; j0 < N; j0 += kWave) {
      const int j = j0 + lane;
      const bool in = j < N;
      const int jc = in ? j : N - 1;
      const int jm1 = jc > 0 ? jc - 1 : 0, jm2 = jc > 1 ? jc - 2 : 0;
      const int jp1 = jc < N - 1 ? jc + 1 : N - 1, jp2 = jc < N - 2 ? jc + 2 : N - 1;
      const double X = Xat(jc);
      const double d1 = Xat(jp1) - Xat(jm1), d2 = Xat(jp2) - Xat(jm2);
      const bool end0 = (jc == 0) || (jc == N - 1), end1 = (jc == 1) || (jc == N - 2);
      const double Ac = end0 ? A_e0 : (end1 ? A_e1 : A_in), Bc = end0 ? B_e0 : (end1 ? 0.0 : B_in);
      const double dX = xfma(Ac, d1, Bc * d2);                                       // utils.py:1610-1616
      const double w = in ? (end0 ? 1.0 : ((jc & 1) ? 4.0 : 2.0)) : 0.0;             // Simpson weights (the 1/3 cancels)
      const double X2 = w * (X * X), dX2 = w * (dX * dX);
      y0 += src.c(jc) * X2 - src.g(jc) * dX2;                                        // utils.py:1618
      y1 = xfma(src.f(jc), X2, y1);                                                  // utils.py:1619
      if (X_out && in) X_out[sys * N + j] = (TI)X;
      if (dX_out && in) dX_out[sys * N + j] = (TI)dX;
    }
    y0 = wave_sum(y0); y1 = wave_sum(y1);
    gam = y0 / y1;                                                                   // utils.py:1621
  }
  if (lane == 0) {
    if (lam_out) lam_out[sys] = (TI)lam;
    if (gam_out) gam_out[sys] = (TI)gam;
    if (info_out) info_out[sys] = passes | (status << 16);
  }
}

template <typename TI>
__global__ void __launch_bounds__(64) k_solve_gcf_long(long n_sys, int N, double h, const TI* __restrict__ g, const TI* __restrict__ c,
                                                       const TI* __restrict__ f, const TI* __restrict__ gh, long ld, TI* lam_out,
                                                       TI* gam_out, TI* X_out, TI* dX_out, int* info_out, double* work) {
  __shared__ double lds[3 * kLongChunk];
  double* my = work + (size_t)blockIdx.x * 3 * (size_t)N;
  for (long sys = blockIdx.x; sys < n_sys; sys += gridDim.x) {
    if (gh) {
      const SrcLong<TI, true> src{g + sys * ld, c + sys * ld, f + sys * ld, gh + sys * ld};
      solve_long_one<TI, true>(src, N, h, sys, my, lam_out, gam_out, X_out, dX_out, info_out, lds);
    } else {
      const SrcLong<TI, false> src{g + sys * ld, c + sys * ld, f + sys * ld, nullptr};
      solve_long_one<TI, false>(src, N, h, sys, my, lam_out, gam_out, X_out, dX_out, info_out, lds);
    }
    long_fence();                                           // (the workspace is reused by this wave's next system)
  }
}

// eigenvalues of (T, F) above shift[sys]: one wave per system, every lane the same shift (division form: exact for a pencil a few
// ulp away, whatever N).  Replaces check_ball's verdict (bishop_ball_s-alpha.py:20-115) on grids beyond 2050 points.
__global__ void __launch_bounds__(64) k_sturm_count_long(long n_sys, int N, double h, const double* __restrict__ g,
                                                         const double* __restrict__ c, const double* __restrict__ f, long ld,
                                                         const double* __restrict__ shift, int* count_out) {
  __shared__ double lds[3 * kLongChunk];
  const double ih2 = 1.0 / (h * h);
  for (long sys = blockIdx.x; sys < n_sys; sys += gridDim.x) {
    const SrcLong<double, false> src{g + sys * ld, c + sys * ld, f + sys * ld, nullptr};
    const int cnt = count_above_chunked(src, N - 2, ih2, shift[sys], lds, (int)(threadIdx.x & 63));
    if ((threadIdx.x & 63) == 0) count_out[sys] = cnt;
  }
}

// Division-form Sturm count with LANES AS SYSTEMS (round 6): every lane walks its own system's rows serially -- q_r = (d_r - sig f_r) -
// e_r^2 / q_{r-1}, IEEE division, pivmin guard: exact for a pencil a few ulp away, any N (the prefix-product sweep k_sturm_count is
// exact only for ~N eps ||A||, ~N^2 eps ||A|| on iid-random coefficients) -- while the rows reach it through an LDS transpose: per chunk
// the wave loads, for 64 systems, ONE 128-byte line of g, c and f per system (four systems per load instruction), writes them to LDS as
// [system][point] and reads them back one system per lane.  The next chunk's 48 loads are in flight while the current one is worked on.
// Rows of N doubles start at any multiple of 8 bytes: a chunk of 16 grid points at the same j in every system straddled two lines per
// system and array, and the second half was gone from the L2 by the time the next chunk asked for it (PMC: 1.82 x the algorithmic bytes).
// So every system is walked from ITS OWN line boundary: lane s runs a_s = (its row's first element) mod 16 points behind the chunk index
// -- the systems are independent, nothing needs them at the same grid point at the same time -- and a row is eliminated one step late,
// when g of the next point has arrived (c and f of the row wait one step in registers).  Steps before point 1 / after point N - 2 of a
// lane are masked.
constexpr int kDivChunk = 16, kDivPitch = kDivChunk + 1, kDivWaves = 2;      // (2 waves per block: 52 KB of LDS, three blocks per CU)
__global__ void __launch_bounds__(64 * kDivWaves) k_sturm_count_div(long n_sys, int N, double h, const double* __restrict__ g,
                                                         const double* __restrict__ c, const double* __restrict__ f, long ld,
                                                         const double* __restrict__ shift, int* count_out) {
  constexpr double pivmin = 2.2250738585072014e-292;
  __shared__ double tile[kDivWaves][3][kWave * kDivPitch];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long base = ((long)blockIdx.x * kDivWaves + wave) * kWave;    // first system of this wave
  if (base >= n_sys) return;                                           // (wave-uniform; no block-level barrier below)
  const int sub = lane >> 4, rr = lane & 15;                           // loader role: system 4 k + sub, element rr of its line
  const long mine = base + lane < n_sys ? base + lane : n_sys - 1;     // worker role: this lane's system
  const double sig = shift[mine];
  const double ih2 = 1.0 / (h * h);
  auto phase = [&](long sy) { return (int)((reinterpret_cast<unsigned long long>(g + sy * ld) >> 3) & 15ull); };   // in doubles, of g's row
  double* tg = tile[wave][0]; double* tc = tile[wave][1]; double* tf = tile[wave][2];
  double vg[16], vc[16], vf[16];
  auto load_chunk = [&](int B) {                                       // system sy: grid points 16 B - a_sy + (0 .. 15), clamped into the row
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      long sy = base + 4 * k + sub; sy = sy < n_sys ? sy : n_sys - 1;
      int pnt = 16 * B - phase(sy) + rr;
      pnt = pnt < 0 ? 0 : (pnt > N - 1 ? N - 1 : pnt);
      const long o = sy * ld + pnt;
      vg[k] = g[o]; vc[k] = c[o]; vf[k] = f[o];
    }
  };
  const int a_m = phase(mine);
  double gm2 = 0.0, gm1 = 0.0, cprev = 0.0, fprev = 1.0, q = 1.0;
  int cnt = 0;
  const int nB = (N - 1 + 15) / 16 + 1;                                // chunks until every lane has seen its point N - 1
  load_chunk(0);
  for (int B = 0; B < nB; ++B) {
    wave_lds_sync();                                                   // (the previous chunk has been consumed)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int o = (4 * k + sub) * kDivPitch + rr;
      tg[o] = vg[k]; tc[o] = vc[k]; tf[o] = vf[k];
    }
    wave_lds_sync();
    if (B + 1 < nB) load_chunk(B + 1);                                 // in flight during the recurrence below
    const int p0 = 16 * B - a_m;                                       // this lane's grid point at i = 0
#pragma unroll 4
    for (int i = 0; i < kDivChunk; ++i) {
      const double gp = tg[lane * kDivPitch + i], cp = tc[lane * kDivPitch + i], fp = tf[lane * kDivPitch + i];
      const int j = p0 + i - 1;                                        // the row that can be eliminated now: g[j + 1] has arrived
      const double e_lo = 0.5 * (gm2 + gm1) * ih2, e_hi = 0.5 * (gm1 + gp) * ih2;      // utils.py:1574-1576
      const double a = xfma(-sig, fprev, cprev - (e_lo + e_hi));                       // utils.py:1584-1592
      double qn = (j == 1) ? a : a - (e_lo * e_lo) / q;
      qn = xabs(qn) < pivmin ? -pivmin : qn;
      const bool row = j >= 1 && j <= N - 2;
      q = row ? qn : q;
      cnt += (row && qn > 0.0) ? 1 : 0;
      gm2 = gm1; gm1 = gp; cprev = cp; fprev = fp;
    }
  }
  if (base + lane < n_sys) count_out[base + lane] = cnt;
}

// (g, c, f) of every (line, theta0) system of a geometry-fed scan, and their theta0 tangents, written out as rows: the long-grid
// form of the staging the scan kernels do in LDS -- the same expressions in the same order (k_gamma_scan, SrcGeo), i.e.
// ball_scan.py:267-268 + utils.py:1560-1562 and utils.py:1669-1673.
__global__ void __launch_bounds__(256) k_assemble_gcf_long(int n_lines, int n_theta0, int N, const double* __restrict__ bmag,
                                                           const double* __restrict__ gradpar, const double* __restrict__ cvdrift,
                                                           const double* __restrict__ cvdrift0, const double* __restrict__ gds2,
                                                           const double* __restrict__ gds21, const double* __restrict__ gds22, long ld,
                                                           const double* __restrict__ dPdrho, const double* __restrict__ theta0,
                                                           int t0_stride, double* g, double* c, double* f, double* gt, double* ct,
                                                           double* ft) {
  const long n_sys = (long)n_lines * n_theta0;
  for (long sys = blockIdx.y; sys < n_sys; sys += gridDim.y) {
    const int line = (int)(sys / n_theta0), it0 = (int)(sys - (long)line * n_theta0);
    const double th0 = theta0[(long)line * t0_stride + it0], two_th0 = 2.0 * th0, th0sq = th0 * th0;
    const double mdP = -dPdrho[line];
    const long off = (long)line * ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
      const double B = bmag[off + j], gp = xabs(gradpar[off + j]);
      const double inv = 1.0 / (gp * B);
      const double A1 = gp / B, A3 = inv / (B * B);
      const double C0 = mdP * cvdrift[off + j] * inv, C1 = mdP * cvdrift0[off + j] * inv;
      const double G0 = gds2[off + j], G1 = gds21[off + j], G2 = gds22[off + j];
      const double d = G0 + two_th0 * G1 + th0sq * G2;
      const long o = sys * N + j;
      g[o] = A1 * d; c[o] = C0 + th0 * C1; f[o] = A3 * d;
      if (gt) { const double dp = 2.0 * G1 + two_th0 * G2; gt[o] = A1 * dp; ct[o] = C1; ft[o] = A3 * dp; }
    }
  }
}

hipError_t launch_gcf_long(const LongGcfArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)(a.n_sys < a.n_waves ? a.n_sys : a.n_waves);
  if (a.f32) {
    hipLaunchKernelGGL(k_solve_gcf_long<float>, dim3(grid), dim3(64), 0, st, a.n_sys, a.N, a.h, (const float*)a.g, (const float*)a.c,
                       (const float*)a.f, (const float*)a.gh, a.ld, (float*)a.lam, (float*)a.gam, (float*)a.X, (float*)a.dX, a.info, a.work);
    note_launch(grid, 64, "ibs::k_solve_gcf_long<float>");
  } else {
    hipLaunchKernelGGL(k_solve_gcf_long<double>, dim3(grid), dim3(64), 0, st, a.n_sys, a.N, a.h, (const double*)a.g, (const double*)a.c,
                       (const double*)a.f, (const double*)a.gh, a.ld, (double*)a.lam, (double*)a.gam, (double*)a.X, (double*)a.dX, a.info,
                       a.work);
    note_launch(grid, 64, "ibs::k_solve_gcf_long<double>");
  }
  return hipGetLastError();
}

hipError_t launch_sturm_long(const SturmArgs<double>& a, hipStream_t st) {
  const long cap = 16384;
  const unsigned grid = (unsigned)(a.n_sys < cap ? a.n_sys : cap);
  hipLaunchKernelGGL(k_sturm_count_long, dim3(grid), dim3(64), 0, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld, a.shift, a.count);
  note_launch(grid, 64, "ibs::k_sturm_count_long");
  return hipGetLastError();
}

hipError_t launch_sturm_div(const SturmArgs<double>& a, hipStream_t st) {
  const long per = 64 * kDivWaves, nblk = (a.n_sys + per - 1) / per;
  hipLaunchKernelGGL(k_sturm_count_div, dim3((unsigned)nblk), dim3((unsigned)per), 0, st, a.n_sys, a.N, a.h, a.g, a.c, a.f, a.ld, a.shift, a.count);
  note_launch(nblk, (int)per, "ibs::k_sturm_count_div");
  return hipGetLastError();
}

hipError_t launch_assemble_long(const ScanArgs<double>& a, double* g, double* c, double* f, double* gt, double* ct, double* ft,
                                hipStream_t st) {
  const long n_sys = (long)a.n_lines * a.n_theta0;
  int bx = (a.N + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_assemble_gcf_long, dim3((unsigned)bx, (unsigned)(n_sys < 32768 ? n_sys : 32768)), dim3(256), 0, st, a.n_lines, a.n_theta0, a.N, a.bmag,
                     a.gradpar, a.cvdrift, a.cvdrift0, a.gds2, a.gds21, a.gds22, a.ld, a.dPdrho, a.theta0, a.t0_stride, g, c, f, gt, ct, ft);
  return hipGetLastError();
}

}  // namespace ibs
