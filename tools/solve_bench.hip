// Micro-benchmark: one whole shift iteration (solve(): locate + interpolate + certificates + backward sweep + twisted)
// of the wave solver (scalar branches on wave-uniform conditions) against the sub-wave solver's select-based state
// machine run with P = 64 (one system per wave), one wave per SIMD.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -I ideal-ballooning-solver_amd/csrc tools/solve_bench.hip -o /tmp/sb && /tmp/sb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "ibs_group.hpp"
using namespace ibs;
struct Src { static constexpr bool kHasGh = false; const double* gs; const double* cs; const double* fs; double cscale;
  __device__ double g(int j) const { return gs[j]; } __device__ double c(int j) const { return cs[j] * cscale; } __device__ double f(int j) const { return fs[j]; } };

template <int KIND, int M>
__global__ void __launch_bounds__(256) k(int N, double h, const double* g, const double* c, const double* f, int reps, double* out, int* its) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  Src src{g, c, f, 1.0 + 1e-3 * (w % 97)};
  double acc = 0; int itsum = 0;
  if constexpr (KIND == 0) {
    WaveSolver<double, M> ws;
    ws.setup(src, N, h);
    const double lo0 = ws.lo, hi0 = ws.hi;
    for (int r = 0; r < reps; ++r) {
      ws.lo = lo0; ws.hi = hi0 + 1e-12 * acc * 1e-6;
      SolveInfo inf{0, 0};
      acc += ws.solve(inf);
      itsum += inf.iters;
    }
  } else {
    GroupSolver<double, M, 64> ws;
    const bool bad = ws.setup(src, N, h);
    const double lo0 = ws.lo, hi0 = ws.hi;
    for (int r = 0; r < reps; ++r) {
      ws.lo = lo0; ws.hi = hi0 + 1e-12 * acc * 1e-6;
      int it = 0, st = 0;
      acc += ws.solve(bad, it, st);
      itsum += it;
    }
  }
  if ((threadIdx.x & 63) == 0) { out[w] = acc; its[w] = itsum; }
}
template <int KIND, int M> void run(const char* name, int nblk, int N, double h, double* dg, double* dc, double* df, double* dout, int* dits) {
  const int reps = 50;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<KIND, M>), dim3(nblk), dim3(256), 0, 0, N, h, dg, dc, df, 3, dout, dits);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<KIND, M>), dim3(nblk), dim3(256), 0, 0, N, h, dg, dc, df, reps, dout, dits);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<int> its(nblk * 4); std::vector<double> o(nblk * 4);
  hipMemcpy(its.data(), dits, its.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; int mx = 0; for (int v : its) { s += v; mx = v > mx ? v : mx; }
  const double mean_it = s / its.size() / reps;
  printf("%-34s M=%2d blocks=%4d  %.2f us per solve (slowest wave sets it), mean %.2f sweeps/solve, max %.2f; lam[0] %.12e\n", name, M, nblk,
         ms * 1e3 / reps, mean_it, (double)mx / reps, o[0] / reps);
}
int main() {
  for (int N : {513, 1025}) {
    double h = 8 * M_PI / (N - 1);
    std::vector<double> g(N), c(N), f(N);
    for (int j = 0; j < N; ++j) { double th = -4 * M_PI + j * h; double lam = 1.0 * (th - 0.3) - 0.8 * (sin(th) - sin(0.3)); g[j] = 1 + lam * lam; c[j] = 0.8 * (cos(th) + sin(th) * lam); f[j] = g[j]; }
    double *dg, *dc, *df, *dout; int* dits;
    hipMalloc(&dg, N * 8); hipMalloc(&dc, N * 8); hipMalloc(&df, N * 8); hipMalloc(&dout, 1 << 20); hipMalloc(&dits, 1 << 20);
    hipMemcpy(dg, g.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), N * 8, hipMemcpyHostToDevice);
    for (int nblk : {256, 512}) {
      if (N == 513) {
        run<0, 8>("WaveSolver::solve (branches)", nblk, N, h, dg, dc, df, dout, dits);
        run<1, 8>("GroupSolver<P=64>::solve (selects)", nblk, N, h, dg, dc, df, dout, dits);
      } else {
        run<0, 16>("WaveSolver::solve (branches)", nblk, N, h, dg, dc, df, dout, dits);
        run<1, 16>("GroupSolver<P=64>::solve (selects)", nblk, N, h, dg, dc, df, dout, dits);
      }
    }
  }
  return 0;
}
