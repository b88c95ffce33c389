// Debug: where the time of one k_gamma_scan launch goes (run on the GPU box).
//   python tools/make_d3d_geo.py && hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DIBS_M=8 -DIBS_PROBE \
//     -I ideal-ballooning-solver_amd/csrc tools/phase_probe.hip -o /tmp/pp && /tmp/pp tools/d3d_geo.bin
// d3d_geo.bin: 7 planes [128][513] (bmag gradpar cvdrift cvdrift0 gds2 gds21 gds22), then dPdrho[128], theta0[8].
#include "ibs_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace ibs;
namespace ibs { LaunchTable& launch_table() { static LaunchTable t{}; return t; } }
int main(int argc, char** argv) {
  // usage: pp geo.bin [wpb [N nl nt]]   (defaults: the bench step, N = 513, 128 lines x 8 theta0)
  const int N = argc > 3 ? atoi(argv[3]) : 513, nl = argc > 4 ? atoi(argv[4]) : 128, nt = argc > 5 ? atoi(argv[5]) : 8;
  const double h = 8 * M_PI / (N - 1);
  std::vector<double> buf(7 * nl * N + nl + nt);
  FILE* fp = fopen(argv[1], "rb"); if (!fp || fread(buf.data(), 8, buf.size(), fp) != buf.size()) { printf("bad input\n"); return 1; }
  double* d; hipMalloc(&d, buf.size() * 8); hipMemcpy(d, buf.data(), buf.size() * 8, hipMemcpyHostToDevice);
  double *gam, *lam, *dth; int* info; hipMalloc(&gam, nl * nt * 8); hipMalloc(&lam, nl * nt * 8); hipMalloc(&dth, nl * nt * 8); hipMalloc(&info, nl * nt * 4);
  const int wpb = argc > 2 ? atoi(argv[2]) : 4;
  ScanArgs<double> a{};
  a.n_lines = nl; a.n_theta0 = nt; a.N = N; a.h = h; a.ld = N; a.wpb = wpb;
  const size_t pl = (size_t)nl * N;
  a.bmag = d; a.gradpar = d + pl; a.cvdrift = d + 2 * pl; a.cvdrift0 = d + 3 * pl; a.gds2 = d + 4 * pl; a.gds21 = d + 5 * pl; a.gds22 = d + 6 * pl;
  a.dPdrho = d + 7 * pl; a.theta0 = d + 7 * pl + nl;
  a.gam = gam; a.lam = lam; a.dth0 = dth; a.info = info;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); launch_scan<double>(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("launch %d: %.1f us (events)\n", rep, ms * 1e3);
  }
  const int nw = (nt + wpb - 1) / wpb * nl * wpb;
  std::vector<long long> pb(16 * 4096);
  hipMemcpyFromSymbol(pb.data(), HIP_SYMBOL(ibs_probe_buf), pb.size() * 8);
  std::vector<int> inf(nl * nt); hipMemcpy(inf.data(), info, nl * nt * 4, hipMemcpyDeviceToHost);
  double mit = 0; for (int v : inf) mit += v & 0xffff; printf("mean iters %.2f\n", mit / inf.size());
  { int hist[16] = {0}; int worst = 0; for (size_t i = 0; i < inf.size(); ++i) { int v = inf[i] & 0xffff; hist[std::min(v / 4, 15)]++; if (v > (inf[worst] & 0xffff)) worst = (int)i; }
    printf("iters histogram (bins of 4):"); for (int b = 0; b < 16; ++b) printf(" %d", hist[b]); printf("\nworst system %d (line %d, theta0 %d): %d iters\n", worst, worst / nt, worst % nt, inf[worst] & 0xffff);
    for (size_t i = 0; i < inf.size(); ++i) if ((inf[i] & 0xffff) >= 28) printf("  sys %zu (line %zu th0 %zu) iters %d\n", i, i / nt, i % nt, inf[i] & 0xffff); }
  long long tmin = 1LL << 62, tmax = 0;
  double ph[4] = {0, 0, 0, 0}, phmax[4] = {0, 0, 0, 0};
  int cnt = 0;
  for (int w = 0; w < nw && w < 4096; ++w) {
    long long* q = &pb[w * 16];
    if (q[0] == 0) continue;
    tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[4]);
    for (int k = 0; k < 4; ++k) { double dt = (q[k + 1] - q[k]) * 0.01; ph[k] += dt; phmax[k] = std::max(phmax[k], dt); }
    ++cnt;
  }
  printf("waves %d  first start -> last end: %.2f us\n", cnt, (tmax - tmin) * 0.01);
  const char* nm[4] = {"stage geometry -> LDS", "setup", "solve", "finish"};
  for (int k = 0; k < 4; ++k) printf("  %-24s mean %.2f us  max %.2f us\n", nm[k], ph[k] / cnt, phmax[k]);
  { double per = 0; int c3 = 0; for (int w = 0; w < nw && w < 4096; ++w) { long long* q = &pb[w * 16]; if (!q[0] || !q[10]) continue; int sysi = -1; per += (q[10] - q[2]) * 0.01; ++c3; }
    printf("  solve loop only: mean %.2f us per wave -> %.3f us per iteration\n", per / c3, per / c3 / (mit / inf.size())); }
  { double a = 0, b = 0, n = 0; int c4 = 0; for (int w = 0; w < nw && w < 4096; ++w) { long long* q = &pb[w * 16]; if (!q[0] || !q[10]) continue; a += q[5] * 0.01; b += q[6] * 0.01; n += q[7]; ++c4; }
    printf("  in-loop: fwd sweeps %.2f us per wave, full-path logic %.2f us per wave over %.2f full-path iterations (%.3f us each)\n", a / c4, b / c4, n / c4, b / n); }
  { double s1 = 0, s2 = 0; int c2 = 0; for (int w = 0; w < nw && w < 4096; ++w) { long long* q = &pb[w * 16]; if (!q[0] || !q[8]) continue; s1 += (q[8] - q[1]) * 0.01; s2 += (q[9] - q[8]) * 0.01; ++c2; }
    if (c2) printf("  setup split: rows+scaling %.2f us, bounds/reductions %.2f us\n", s1 / c2, s2 / c2); }
  { double d[5] = {0}; int c2 = 0; for (int w = 0; w < nw && w < 4096; ++w) { long long* q = &pb[w * 16]; if (!q[0] || !q[10]) continue;
      d[0] += (q[11] - q[10]) * 0.01; d[1] += (q[12] - q[11]) * 0.01; d[2] += (q[13] - q[3]) * 0.01; d[3] += (q[14] - q[13]) * 0.01; d[4] += (q[4] - q[14]) * 0.01; ++c2; }
    if (c2) printf("  final bwd sweep %.2f us, twisted %.2f us | finish: assemble+normalise+LDS %.2f us, FD/Simpson loop %.2f us, reductions+store %.2f us\n", d[0] / c2, d[1] / c2, d[2] / c2, d[3] / c2, d[4] / c2); }
  double latest_start = 0; for (int w = 0; w < nw && w < 4096; ++w) if (pb[w * 16]) latest_start = std::max(latest_start, (pb[w * 16] - tmin) * 0.01);
  printf("  latest wave start after first: %.2f us\n", latest_start);
  return 0;
}
