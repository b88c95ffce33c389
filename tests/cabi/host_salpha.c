/* A host written against include/ibs.h alone (C99, no Python, no torch): the s-alpha systems of the reference's
 * bishop_ball_s-alpha.py:30-45 (g = 1 + L^2, c = alpha (cos t + L sin t), f = g, L = shat (t - t0) - alpha (sin t - sin t0)) on
 * theta = linspace(-4 pi, 4 pi, N), solved through ibs_solve_gcf_f64 with HOST pointers.  Prints one line per system:
 *   shat alpha theta0 lam gam info
 * tests/test_cabi_host.py compiles it (gcc host_salpha.c -I include -L .../lib -libs_hip -lamdhip64), runs it on the GPU
 * and compares the growth rates with the golden vectors captured from the reference (tests/golden/G1_salpha.npz).
 * usage: host_salpha N shat alpha theta0 [shat alpha theta0 ...] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ibs.h"

int main(int argc, char** argv) {
  if (argc < 5 || (argc - 2) % 3 != 0) { fprintf(stderr, "usage: %s N shat alpha theta0 [...]\n", argv[0]); return 2; }
  const int N = atoi(argv[1]);
  const int n_sys = (argc - 2) / 3;
  const double pi = 3.14159265358979323846;
  const double h = 8.0 * pi / (N - 1);
  double* g = malloc(sizeof(double) * (size_t)n_sys * N);
  double* c = malloc(sizeof(double) * (size_t)n_sys * N);
  double* lam = malloc(sizeof(double) * n_sys);
  double* gam = malloc(sizeof(double) * n_sys);
  int32_t* info = malloc(sizeof(int32_t) * n_sys);
  if (!g || !c || !lam || !gam || !info) return 2;
  for (int k = 0; k < n_sys; ++k) {
    const double shat = atof(argv[2 + 3 * k]), alpha = atof(argv[3 + 3 * k]), t0 = atof(argv[4 + 3 * k]);
    for (int j = 0; j < N; ++j) {
      const double t = (j == N - 1) ? 4.0 * pi : -4.0 * pi + j * h;                 /* numpy.linspace ends exactly */
      const double L = shat * (t - t0) - alpha * (sin(t) - sin(t0));
      g[(size_t)k * N + j] = 1.0 + L * L;
      c[(size_t)k * N + j] = alpha * (cos(t) + sin(t) * L);
    }
  }
  ibs_ctx* ctx = NULL;
  int rc = ibs_create(&ctx, 0);
  if (rc < 0) { fprintf(stderr, "ibs_create: %s\n", ibs_last_error()); return 1; }
  /* f = g: the same host array is passed twice; no eigenfunction wanted */
  rc = ibs_solve_gcf_f64(ctx, n_sys, N, h, g, c, g, N, lam, gam, NULL, NULL, info, IBS_MEM_HOST);
  if (rc < 0) { fprintf(stderr, "ibs_solve_gcf_f64: %s\n", ibs_last_error()); ibs_destroy(ctx); return 1; }
  for (int k = 0; k < n_sys; ++k)
    printf("%s %s %s %.17g %.17g %d\n", argv[2 + 3 * k], argv[3 + 3 * k], argv[4 + 3 * k], lam[k], gam[k], (int)info[k]);
  /* an argument error comes back as a negative code with a message, never as an exception or an abort */
  if (ibs_solve_gcf_f64(ctx, n_sys, 64, h, g, c, g, N, lam, gam, NULL, NULL, info, IBS_MEM_HOST) >= 0) { fprintf(stderr, "even N accepted\n"); return 1; }
  fprintf(stderr, "expected error: %s\n", ibs_last_error());
  ibs_destroy(ctx);
  free(g); free(c); free(lam); free(gam); free(info);
  return rc > 0 ? 3 : 0;                                                            /* rc > 0: non-converged systems */
}
