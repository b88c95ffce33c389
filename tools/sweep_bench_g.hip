// Micro-benchmark of the sub-wave solver phases (P lanes per system), companion of sweep_bench.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "ibs_group.hpp"
using namespace ibs;
struct Src { static constexpr bool kHasGh = false; const double* gs; const double* cs; const double* fs;
  __device__ double g(int j) const { return gs[j]; } __device__ double c(int j) const { return cs[j]; } __device__ double f(int j) const { return fs[j]; } };

template <int MODE, int M, int P>
__global__ void __launch_bounds__(256) k(int N, double h, const double* g, const double* c, const double* f, int reps, double* out) {
  Src src{g, c, f};
  GroupSolver<double, M, P> ws;
  ws.setup(src, N, h);
  double acc = 0, sig = ws.hi;
  for (int r = 0; r < reps; ++r) {
    int C = ws.sweep_fwd(sig);
    if (MODE >= 1) { ws.sweep_bwd(sig); double rho = ws.twisted(sig); acc += rho; }
    acc += C;
    sig = sig * 0.999 + 1e-9 * acc * 1e-9;
  }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}
template <int MODE, int M, int P> void run(const char* name, int nblk, int N, double h, double* dg, double* dc, double* df, double* dout) {
  int reps = 200;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, M, P>), dim3(nblk), dim3(256), 0, 0, N, h, dg, dc, df, 10, dout);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, M, P>), dim3(nblk), dim3(256), 0, 0, N, h, dg, dc, df, reps, dout);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double per = ms * 1e-3 / reps;
  printf("%-26s P=%2d M=%2d blocks=%5d  %.3f us per round, %.2f ns per SYSTEM-iteration\n", name, P, M, nblk, per * 1e6,
         per * 1e9 / (nblk * 4.0 * (64 / P)));
}
int main() {
  int N = 513; double h = 8 * M_PI / (N - 1);
  std::vector<double> g(N), c(N), f(N);
  for (int j = 0; j < N; ++j) { double th = -4 * M_PI + j * h; double lam = 1.0 * th - 0.8 * sin(th); g[j] = 1 + lam * lam; c[j] = 0.8 * (cos(th) + sin(th) * lam); f[j] = g[j]; }
  double *dg, *dc, *df, *dout;
  hipMalloc(&dg, N * 8); hipMalloc(&dc, N * 8); hipMalloc(&df, N * 8); hipMalloc(&dout, 1 << 20);
  hipMemcpy(dg, g.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), N * 8, hipMemcpyHostToDevice);
  for (int nblk : {256, 768}) {
    run<0, 8, 64>("fwd only", nblk, N, h, dg, dc, df, dout);
    run<1, 8, 64>("fwd+bwd+twisted", nblk, N, h, dg, dc, df, dout);
    run<0, 16, 32>("fwd only", nblk, N, h, dg, dc, df, dout);
    run<1, 16, 32>("fwd+bwd+twisted", nblk, N, h, dg, dc, df, dout);
  }
  return 0;
}
