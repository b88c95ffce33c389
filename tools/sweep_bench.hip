// Micro-benchmark of the wave solver phases (run on the GPU box):
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -I ideal-ballooning-solver_amd/csrc tools/sweep_bench.hip -o /tmp/sb && /tmp/sb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "ibs_wave.hpp"
using namespace ibs;
constexpr int M = 8;
struct Src { static constexpr bool kHasGh = false; const double* gs; const double* cs; const double* fs;
  __device__ double g(int j) const { return gs[j]; } __device__ double c(int j) const { return cs[j]; } __device__ double f(int j) const { return fs[j]; } };

template <int MODE>
__global__ void __launch_bounds__(256) k(int N, double h, const double* g, const double* c, const double* f, int reps, double* out) {
  extern __shared__ double smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* gs = smem + wave * 3 * N; double* cs = gs + N; double* fs = cs + N;
  for (int j = lane; j < N; j += 64) { gs[j] = g[j]; cs[j] = c[j]; fs[j] = f[j]; }
  __syncthreads();
  Src src{gs, cs, fs};
  WaveSolver<double, M> ws;
  ws.setup(src, N, h);
  double acc = 0, sig = ws.hi;
  if (MODE == 3) {          // bisection loop on the count (minimal control logic)
    double lo = ws.lo, hi = ws.hi; const double lo0 = lo, hi0 = hi;
    sig = 0.5 * (lo + hi);
    for (int r = 0; r < reps; ++r) {
      const int C = ws.sweep_fwd(sig);
      if (C == 0) hi = sig; else lo = sig;
      if (hi - lo < 1e-13) { lo = lo0; hi = hi0; }
      sig = 0.5 * (lo + hi);
      acc += ws.u0_in;
    }
    if (lane == 0) out[blockIdx.x * 4 + wave] = acc + sig;
    return;
  }
  if (MODE == 4) {          // the full solve, repeated; reports time per solve
    for (int r = 0; r < reps; r += 16) {
      SolveInfo inf{0, 0};
      const double lo0 = ws.lo, hi0 = ws.hi;
      acc += ws.solve(inf) + inf.iters;
      ws.lo = lo0; ws.hi = hi0;
    }
    if (lane == 0) out[blockIdx.x * 4 + wave] = acc;
    return;
  }
  for (int r = 0; r < reps; ++r) {
    int C = (MODE == 2) ? ws.sweep_fwd(sig) : ws.sweep(sig);
    if (MODE == 1) { double rho = ws.twisted(sig); acc += rho; }
    acc += C;
    sig = sig * 0.999 + 1e-9 * acc * 1e-9;
  }
  if (lane == 0) out[blockIdx.x * 4 + wave] = acc;
}
template <int MODE> void run(const char* name, int nblk, int N, double h, double* dg, double* dc, double* df, double* dout) {
  int reps = 200;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 3 * N * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 4 * 3 * N * 8, 0, N, h, dg, dc, df, 10, dout);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 4 * 3 * N * 8, 0, N, h, dg, dc, df, reps, dout);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double per = ms * 1e-3 / reps;  // per iteration (all waves in parallel / rounds)
  printf("%-28s blocks=%5d  %.3f us per iteration-round, %.1f ns per wave-iteration (throughput)\n", name, nblk, per * 1e6, per * 1e9 / (nblk * 4.0));
}
int main() {
  int N = 513; double h = 8 * M_PI / (N - 1);
  std::vector<double> g(N), c(N), f(N);
  for (int j = 0; j < N; ++j) { double th = -4 * M_PI + j * h; double lam = 1.0 * th - 0.8 * sin(th); g[j] = 1 + lam * lam; c[j] = 0.8 * (cos(th) + sin(th) * lam); f[j] = g[j]; }
  double *dg, *dc, *df, *dout;
  hipMalloc(&dg, N * 8); hipMalloc(&dc, N * 8); hipMalloc(&df, N * 8); hipMalloc(&dout, 1 << 20);
  hipMemcpy(dg, g.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), N * 8, hipMemcpyHostToDevice);
  for (int nblk : {256, 768}) {
    run<0>("sweep only", nblk, N, h, dg, dc, df, dout);
    run<1>("sweep + twisted", nblk, N, h, dg, dc, df, dout);
    run<2>("fwd sweep only", nblk, N, h, dg, dc, df, dout);
    run<3>("bisection loop", nblk, N, h, dg, dc, df, dout);
    run<4>("full solve / 16", nblk, N, h, dg, dc, df, dout);
  }
  return 0;
}
