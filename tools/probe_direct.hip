// Debug (round 6): the LIBRARY's own raw kernels (k_solve_gcf_direct<double, M, double> and the LDS-staged k_solve_gcf<double, M>,
// compiled from ibs_kernels.hip in this translation unit) on ONE system from a file (g, c, f = 3 N doubles), with every sweep of the
// shift iteration traced.
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DIBS_WITH_F32 -DIBS_M=16 -DIBS_TRACE_ALL -I ideal-ballooning-solver_amd/csrc tools/probe_direct.hip -o /tmp/pd && /tmp/pd 1025 sys.bin l2 l1
#include "ibs_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
namespace ibs { LaunchTable& launch_table() { static LaunchTable t; return t; } LaunchNote& last_launch() { static LaunchNote n; return n; } void note_launch(long, int, const char*, ...) {} }
int main(int argc, char** argv) {
  int N = atoi(argv[1]);
  double h = 8 * M_PI / (N - 1);
  std::vector<double> buf(3 * N);
  FILE* fp = fopen(argv[2], "rb"); fread(buf.data(), 8, 3 * N, fp); fclose(fp);
  double l2 = atof(argv[3]), l1 = atof(argv[4]), gap = l1 - l2;
  double *d, *dlam; int* dinfo;
  hipMalloc(&d, 3 * N * 8); hipMalloc(&dlam, 64); hipMalloc(&dinfo, 64);
  hipMemcpy(d, buf.data(), 3 * N * 8, hipMemcpyHostToDevice);
  for (int staged = 0; staged < 2; ++staged) {
    if (staged) {
      size_t lds = (size_t)4 * 3 * ibs::lds_pitch(N) * 8;
      hipFuncSetAttribute(reinterpret_cast<const void*>(ibs::k_solve_gcf<double, IBS_M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((ibs::k_solve_gcf<double, IBS_M>), dim3(1), dim3(256), lds, 0, 1L, N, h, d, d + N, d + 2 * N, (long)N, dlam, (double*)nullptr, (double*)nullptr, (double*)nullptr, dinfo, (const double*)nullptr, 0);
    } else {
      hipLaunchKernelGGL((ibs::k_solve_gcf_direct<double, IBS_M, double>), dim3(1), dim3(256), 0, 0, 1L, N, h, d, d + N, d + 2 * N, (long)N, dlam, (double*)nullptr, (double*)nullptr, (double*)nullptr, dinfo, 0, (int*)nullptr, (long*)nullptr, (double*)nullptr);
    }
    hipDeviceSynchronize();
    double lam; int info; std::vector<double> tr(404);
    hipMemcpy(&lam, dlam, 8, hipMemcpyDeviceToHost); hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(ibs::ibs_trace_all), 404 * 8);
    printf("%s: lam %.15e  (lam - l2)/gap %+.6f  info %d sweeps %d\n", staged ? "k_solve_gcf (staged)" : "k_solve_gcf_direct", lam, (lam - l2) / gap, info >> 16, info & 0xffff);
    for (int i = 0; i < (int)tr[400] && i < 100; ++i)
      printf("  %2d  (sig - l2)/gap %+.6e   C %g   bracket before: (lo - l2)/gap %+.4e  (hi - l2)/gap %+.4e\n", i + 1, (tr[4 * i] - l2) / gap, tr[4 * i + 1], (tr[4 * i + 2] - l2) / gap, (tr[4 * i + 3] - l2) / gap);
  }
  return 0;
}
