// 64-bit DPP (row_newbcast) on gfx950: does v_fmac_f64_dpp read src0 from lane K of each 16-lane row, and at what rate against the
// plain v_fmac_f64?   hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp64_probe tools/dpp64_probe.hip && /tmp/dpp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define FMAC_DPP(acc, tab, x, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(tab), "v"(x))

__global__ void k_check(const double* tab, const double* x, double* out) {
  const int lane = threadIdx.x & 63;
  double t = tab[lane];
  double xv = x[lane];
  double a = 1.0, b = 2.0;
  asm volatile("s_nop 1");
  FMAC_DPP(a, t, xv, 3);
  FMAC_DPP(b, t, xv, 15);
  out[lane] = a; out[64 + lane] = b;
}

template <bool DPP>
__global__ void __launch_bounds__(1024) k_rate(const double* tab, double* out, int iters) {
  const int lane = threadIdx.x & 63;
  double t0 = tab[lane & 15], t1 = tab[16 + (lane & 15)];
  double x = 1.0 + 1e-9 * lane, y = 1.0 - 1e-9 * lane;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = i;
  asm volatile("s_nop 1");
  for (int it = 0; it < iters; ++it) {
    if constexpr (DPP) {
      FMAC_DPP(a[0], t0, x, 0); FMAC_DPP(a[1], t0, y, 1); FMAC_DPP(a[2], t0, x, 2); FMAC_DPP(a[3], t0, y, 3);
      FMAC_DPP(a[4], t0, x, 4); FMAC_DPP(a[5], t0, y, 5); FMAC_DPP(a[6], t0, x, 6); FMAC_DPP(a[7], t0, y, 7);
      FMAC_DPP(a[8], t1, x, 8); FMAC_DPP(a[9], t1, y, 9); FMAC_DPP(a[10], t1, x, 10); FMAC_DPP(a[11], t1, y, 11);
      FMAC_DPP(a[12], t1, x, 12); FMAC_DPP(a[13], t1, y, 13); FMAC_DPP(a[14], t1, x, 14); FMAC_DPP(a[15], t1, y, 15);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(i < 8 ? t0 : t1), "v"((i & 1) ? y : x));
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double *d_tab, *d_x, *d_out;
  std::vector<double> tab(64), x(64), out(128);
  for (int i = 0; i < 64; ++i) { tab[i] = 100.0 + i; x[i] = 0.5 + i; }
  hipMalloc(&d_tab, 64 * 8); hipMalloc(&d_x, 64 * 8); hipMalloc(&d_out, 256 * 1024 * 8);
  hipMemcpy(d_tab, tab.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(d_x, x.data(), 64 * 8, hipMemcpyHostToDevice);
  k_check<<<1, 64>>>(d_tab, d_x, d_out);
  hipMemcpy(out.data(), d_out, 128 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const double ea = 1.0 + tab[(l & ~15) + 3] * x[l], eb = 2.0 + tab[(l & ~15) + 15] * x[l];
    if (out[l] != ea || out[64 + l] != eb) { if (bad < 4) printf("lane %d: got %g %g expected %g %g\n", l, out[l], out[64 + l], ea, eb); ++bad; }
  }
  printf("row_newbcast semantics (src0 of lane K of the lane's 16-lane row): %s\n", bad ? "MISMATCH" : "as expected on all 64 lanes");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; ++wps) {                       // waves per SIMD (one block per CU)
    for (int dpp = 0; dpp < 2; ++dpp) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (dpp) k_rate<true><<<256, 256 * wps>>>(d_tab, d_out, iters); else k_rate<false><<<256, 256 * wps>>>(d_tab, d_out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%d wave(s) per SIMD, %s: %.3f ms for %d x 16 fmac per wave -> %.2f ns per wave-instruction per SIMD (4 clk at 2.3 GHz = 1.74 ns)\n",
             wps, dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64                 ", best, iters, best * 1e6 / ((double)wps * iters * 16));
    }
  }
  return bad != 0;
}
