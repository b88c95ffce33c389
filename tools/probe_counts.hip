// Debug (round 6): for ONE raw system (file: g, c, f = 3 N doubles) the three Sturm counts side by side at a ladder of shifts --
// forward and backward prefix-product sweeps of WaveSolver (what moves the solver's bracket) and the division-form count
// (count_above_div: the arbiter) -- then the solver's own trace from its trial-vector start.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DIBS_TRACE -I ideal-ballooning-solver_amd/csrc tools/probe_counts.hip -o /tmp/pc && /tmp/pc 1025 sys.bin lam_lo lam_hi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "ibs_wave.hpp"
using namespace ibs;
struct Src { static constexpr bool kHasGh = false; const double* gs; const double* cs; const double* fs;
  __device__ double g(int j) const { return gs[j]; } __device__ double c(int j) const { return cs[j]; } __device__ double f(int j) const { return fs[j]; } };
template <int M>
__global__ void __launch_bounds__(64) k(int N, double h, const double* g, const double* c, const double* f, int ns, const double* shifts, int* counts,
                                        double* trace, double* out) {
  Src src{g, c, f};
  WaveSolver<double, M> ws;
  ws.template setup<Src, true>(src, N, h);
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < ns; ++i) {
    const int cf = ws.sweep_fwd(shifts[i]);
    const int cb = ws.sweep_bwd(shifts[i]);
    const double rho = ws.twisted(shifts[i]);
    const int cd = count_above_div<double, Src>(src, N, 1.0 / (h * h), shifts[i]);
    if (lane == 0) { counts[3 * i] = cf; counts[3 * i + 1] = cb; counts[3 * i + 2] = cd; out[8 + i] = rho; }
  }
  ws.trace = trace;
  SolveInfo inf{0, 0};
  double g_, w_;
  ws.trial_guess(g_, w_);
  double lam = ws.template solve<true>(inf, true, g_, w_);
  if (lane == 0) { out[0] = lam; out[1] = inf.iters; out[2] = ws.normA; out[3] = ws.rho_last; out[4] = ws.suspect; out[5] = ws.why; }
}
int main(int argc, char** argv) {
  int N = atoi(argv[1]);
  double h = 8 * M_PI / (N - 1);
  std::vector<double> g(N), c(N), f(N);
  FILE* fp = fopen(argv[2], "rb"); fread(g.data(), 8, N, fp); fread(c.data(), 8, N, fp); fread(f.data(), 8, N, fp); fclose(fp);
  double l2 = atof(argv[3]), l1 = atof(argv[4]);     // the eigenvalue returned and the true lam_max
  std::vector<double> sh;
  const double gap = l1 - l2;
  for (int k = -6; k <= 26; ++k) sh.push_back(l2 + gap * k / 20.0);
  for (int k = -12; k <= -1; ++k) { sh.push_back(l2 + gap * pow(2.0, k)); sh.push_back(l2 - gap * pow(2.0, k)); sh.push_back(l1 - gap * pow(2.0, k)); sh.push_back(l1 + gap * pow(2.0, k)); }
  int ns = sh.size();
  double *dg, *dc, *df, *dt, *dout, *dsh; int* dcnt;
  hipMalloc(&dg, N * 8); hipMalloc(&dc, N * 8); hipMalloc(&df, N * 8); hipMalloc(&dt, 6 * 65 * 8); hipMalloc(&dout, (8 + ns) * 8); hipMalloc(&dsh, ns * 8); hipMalloc(&dcnt, ns * 12);
  hipMemset(dt, 0, 6 * 65 * 8);
  hipMemcpy(dg, g.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), N * 8, hipMemcpyHostToDevice);
  hipMemcpy(dsh, sh.data(), ns * 8, hipMemcpyHostToDevice);
  int M = (N - 2 + 63) / 64;
  if (M == 16) hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, N, h, dg, dc, df, ns, dsh, dcnt, dt, dout);
  else if (M == 8) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, N, h, dg, dc, df, ns, dsh, dcnt, dt, dout);
  else if (M == 32) hipLaunchKernelGGL(k<32>, dim3(1), dim3(64), 0, 0, N, h, dg, dc, df, ns, dsh, dcnt, dt, dout);
  else { printf("M=%d not built\n", M); return 1; }
  std::vector<double> tr(6 * 65), out(8 + ns); std::vector<int> cnt(3 * ns);
  hipMemcpy(tr.data(), dt, 6 * 65 * 8, hipMemcpyDeviceToHost); hipMemcpy(out.data(), dout, (8 + ns) * 8, hipMemcpyDeviceToHost); hipMemcpy(cnt.data(), dcnt, ns * 12, hipMemcpyDeviceToHost);
  printf("returned lam %.15e (lam - l2 %+.3e, lam - l1 %+.3e) iters %g normA %g tol %.3e  polish - lam %+.3e suspect %g bucket %g\n", out[0], out[0] - l2, out[0] - l1, out[1], out[2],
         64 * 2.22e-16 * out[2], out[3] - out[0], out[4], out[5]);
  printf("%26s %6s %6s %6s %14s\n", "(shift - l2) / gap", "C_fwd", "C_bwd", "C_div", "polish - l2 [gap]");
  for (int i = 0; i < ns; ++i) printf("%+26.6e %6d %6d %6d %+14.4e\n", (sh[i] - l2) / gap, cnt[3 * i], cnt[3 * i + 1], cnt[3 * i + 2], (out[8 + i] - l2) / gap);
  for (int i = 0; i <= (int)out[1] && i < 64; ++i)
    printf("%2d (sig - l2)/gap %+.4e C %g rho-lam %+.3e (lo-l2)/gap %+.3e (hi-l2)/gap %+.3e\n", i, (tr[6*i]-l2)/gap, tr[6*i+1], tr[6*i+2]-out[0], (tr[6*i+3]-l2)/gap, (tr[6*i+4]-l2)/gap);
  return 0;
}
