// Is the matrix pipe a SECOND FP64 engine on gfx950?  (the geometry kernel's pair sums are a
// [points x pairs] x [pairs x columns] FP64 contraction; v_mfma_f64_16x16x4_f64 issues to the matrix pipe.)
// Measures, with every CU busy and 1 / 2 / 4 waves per SIMD:
//   (a) v_fma_f64 alone                      16 independent accumulators per wave
//   (b) v_mfma_f64_16x16x4_f64 alone         4 independent 16x16 accumulators per wave (1,024 multiply-adds each = 16 wave-FMAs)
//   (c) both in ONE wave's stream            1 MFMA per 16 FMAs (equal multiply-add counts) and 1 per 8
//   (d) two KINDS of waves on each SIMD      half the waves of a block run (a), the other half (b)
// and reports multiply-adds per clock per SIMD for each (the FP64 vector peak is 16: one wave64 v_fma_f64 per 4 clocks), with
// the shader clock taken from s_memtime against the wall clock inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma64_probe tools/mfma64_probe.hip && /tmp/mfma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// mode: 0 = FMA only, 1 = MFMA only, 2 = interleaved 1 MFMA : 16 FMA, 3 = interleaved 1 MFMA : 8 FMA,
//       4 = waves with (wave index within the SIMD's set) even run FMA, odd run MFMA (needs >= 2 waves per SIMD)
template <int MODE>
__global__ void __launch_bounds__(1024) k_probe(double* out, long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;                 // waves 0..3 sit on SIMDs 0..3, 4..7 again on 0..3, ...
  const bool second = ((wave >> 2) & 1) != 0;        // the second wave of each SIMD (mode 4)
  double x = 1.0 + 1e-9 * lane, y = 1.0 - 1e-9 * lane;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = i;
  double4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  auto fma16 = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"((i & 1) ? y : x), "v"((i & 2) ? x : y));
  };
  auto fma8 = [&](int h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[8 * h + i]) : "v"((i & 1) ? y : x), "v"((i & 2) ? x : y));
  };
  auto mfma = [&](int k) { acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[k], 0, 0, 0); };
  for (int it = 0; it < iters; it += 4) {            // (four iterations per trip: accumulator indices are compile-time)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (MODE == 0) { fma16(); }
      else if constexpr (MODE == 1) { mfma(k); }
      else if constexpr (MODE == 2) { mfma(k); fma16(); }
      else if constexpr (MODE == 3) { mfma(k); fma8(k & 1); }
      else { if (second) mfma(k); else fma16(); }
    }
  }
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  if (threadIdx.x == 256 && blockIdx.x == 0) { clk[2] = c1 - c0; clk[3] = w1 - w0; }      // first wave of the second set (mode 4: MFMA)
}

int main() {
  double* d_out; long long* d_clk;
  hipMalloc(&d_out, 256 * 1024 * 8); hipMalloc(&d_clk, 32);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[5] = {"v_fma_f64 alone", "v_mfma_f64_16x16x4 alone", "one stream, 1 MFMA : 16 FMA", "one stream, 1 MFMA : 8 FMA",
                          "two kinds of waves per SIMD (FMA | MFMA)"};
  printf("multiply-adds per clock per SIMD (FP64 vector peak = 16); clocks from s_memtime, GHz = cycles / wall (100 MHz)\n");
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int mode = 0; mode < 5; ++mode) {
      if (mode == 4 && wps < 2) continue;
      float best = 1e30f; long long clk[4] = {0, 0, 0, 0};
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        const dim3 g(256), b(256 * wps);
        switch (mode) {
          case 0: k_probe<0><<<g, b>>>(d_out, d_clk, iters); break;
          case 1: k_probe<1><<<g, b>>>(d_out, d_clk, iters); break;
          case 2: k_probe<2><<<g, b>>>(d_out, d_clk, iters); break;
          case 3: k_probe<3><<<g, b>>>(d_out, d_clk, iters); break;
          default: k_probe<4><<<g, b>>>(d_out, d_clk, iters); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) { best = ms; hipMemcpy(clk, d_clk, 32, hipMemcpyDeviceToHost); }
      }
      // multiply-adds per wave and iteration: FMA 16 x 64 lanes, MFMA 16 x 16 x 4
      double fma_w = 0, mfma_w = 0;                 // per SIMD, summed over its waves
      if (mode == 0) fma_w = wps * 16.0 * 64; else if (mode == 1) mfma_w = wps * 1024.0;
      else if (mode == 2) { fma_w = wps * 16.0 * 64; mfma_w = wps * 1024.0; }
      else if (mode == 3) { fma_w = wps * 8.0 * 64; mfma_w = wps * 1024.0; }
      else { fma_w = (wps / 2) * 16.0 * 64; mfma_w = (wps / 2) * 1024.0; }
      const double cyc = (double)clk[0];            // shader cycles of wave 0 of block 0 over the loop
      const double ghz = clk[1] ? cyc / (clk[1] * 10.0) : 0.0;
      // (mode 4: each kind over ITS OWN cycles -- the kind that finishes first ran beside the other all the time, the other one partly alone)
      const double cyc_m = (mode == 4 && clk[2]) ? (double)clk[2] : cyc;
      printf("%d wave(s)/SIMD  %-42s %8.3f ms  %6.2f GHz  FMA %6.2f + MFMA %6.2f = %6.2f multiply-adds/clk/SIMD%s\n", wps, names[mode], best, ghz,
             fma_w * iters / cyc, mfma_w * iters / cyc_m, fma_w * iters / cyc + mfma_w * iters / cyc_m,
             mode == 4 ? (cyc_m < cyc ? "  (MFMA waves done first)" : "  (FMA waves done first)") : "");
    }
  }
  return 0;
}
