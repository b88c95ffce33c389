// Debug: per-iteration trace of the wave solver for one s-alpha system (run on the GPU box).
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DIBS_TRACE -I ideal-ballooning-solver_amd/csrc tools/trace_solve.hip -o /tmp/ts && /tmp/ts 1025 0.3 0.6 0.0
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
__global__ void __launch_bounds__(64) k(int N, double h, const double* g, const double* c, const double* f, double* trace, double* out) {
  Src src{g, c, f};
  WaveSolver<double, M> ws;
  ws.setup(src, N, h);
  ws.trace = trace;
  SolveInfo inf{0, 0};
  double lam = ws.solve(inf);
  if ((threadIdx.x & 63) == 0) { out[0] = lam; out[1] = inf.iters; out[2] = ws.normA; }
}
int main(int argc, char** argv) {
  int N = atoi(argv[1]);
  double h = 8 * M_PI / (N - 1);
  std::vector<double> g(N), c(N), f(N);
  if (argc == 3) {   // raw file: g, c, f (3N doubles)
    FILE* fp = fopen(argv[2], "rb"); fread(g.data(), 8, N, fp); fread(c.data(), 8, N, fp); fread(f.data(), 8, N, fp); fclose(fp);
  } else {
    double sh = atof(argv[2]), al = atof(argv[3]), t0 = atof(argv[4]);
    for (int j = 0; j < N; ++j) { double th = -4 * M_PI + j * h; double lam = sh * (th - t0) - al * (sin(th) - sin(t0)); g[j] = 1 + lam * lam; c[j] = al * (cos(th) + sin(th) * lam); f[j] = g[j]; }
  }
  double *dg, *dc, *df, *dt, *dout;
  hipMalloc(&dg, N * 8); hipMalloc(&dc, N * 8); hipMalloc(&df, N * 8); hipMalloc(&dt, 6 * 65 * 8); hipMalloc(&dout, 64);
  hipMemset(dt, 0, 6 * 65 * 8);
  hipMemcpy(dg, g.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), N * 8, hipMemcpyHostToDevice);
  int M = (N - 2 + 63) / 64;
  if (M == 16) hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, N, h, dg, dc, df, dt, dout);
  else if (M == 8) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, N, h, dg, dc, df, dt, dout);
  else { printf("M=%d not built\n", M); return 1; }
  std::vector<double> tr(6 * 65); double out[3];
  hipMemcpy(tr.data(), dt, 6 * 65 * 8, hipMemcpyDeviceToHost); hipMemcpy(out, dout, 24, hipMemcpyDeviceToHost);
  printf("lam %.15e iters %g normA %g tol %.3e\n", out[0], out[1], out[2], 64 * 2.22e-16 * out[2]);
  for (int i = 0; i <= (int)out[1] && i < 64; ++i)
    printf("%2d sig-lam %+.3e C %g rho-lam %+.3e lo-lam %+.3e hi-lam %+.3e  dt %.2f us\n", i, tr[6*i]-out[0], tr[6*i+1], tr[6*i+2]-out[0], tr[6*i+3]-out[0], tr[6*i+4]-out[0], i ? (tr[6*i+5]-tr[6*i-1])*0.01 : 0.0);
  return 0;
}
