// Device-resident maximiser of gam over (alpha, theta0) (reference: ball_scan.py:305-339).
// Shared between ibs_api.hip (host loop, init / output kernels) and ibs_kernels.hip (the fused evaluation kernel).
//
// One L-BFGS-B state machine per start point (ibs_lbfgsb2.hpp: what scipy.optimize.minimize runs for
// ball_scan.py:307-314).  A ROUND evaluates every point still running:
//   geometry kernel  : the three field lines alpha - d/2, alpha, alpha + d/2 of each point (utils.py:1641-1646)
//   k_refine_eval    : one block per point -- dPdrho of the three lines, centre line and alpha-tangent staged in LDS by
//                      four waves, then ONE wave: eigen-solve (warm-started from the point's previous evaluation), growth
//                      rate and the Hellmann-Feynman gradient (utils.py:1632-1728), then the optimizer step on the state;
//                      the block that finishes last compacts the batch and publishes the count of points still running.
// Nothing returns to the host between rounds: the kernels of a round read the batch size from RefineCtrl, the host only
// reads the per-round counts the device posts into pinned memory, a few rounds behind the rounds it enqueues.
#pragma once
#include "ibs_lbfgsb2.hpp"

namespace ibs {

struct RefineState {
  lbfgsb2::State q;
  int active, nev;
  // warm start of the objective's eigen-solve: eigenvalue, point and gradient (of -gam) of the previous evaluation
  int have, sweeps;
  double lam_prev, x_prev[2], g_prev[2], err_prev;
};
static_assert(sizeof(RefineState) % 8 == 0, "copied as doubles");

struct RefineParams { double lo[2], hi[2], del_alpha, ftol, gtol; int maxiter, n_surf; };

struct RefineCtrl {
  int n_c;        // points in the batch of the next round (slots 0 .. n_c-1)
  int n_lines;    // = 3 n_c: what the geometry kernel reads
  int done;       // blocks of the current round that have finished (reset by the last one)
  int round;      // rounds completed
  double h;       // grid spacing of the (uniform) theta grid, formed on the device by k_refine_init
};

template <typename T>
struct RefineEvalArgs {
  int n_c_max, N;
  const T* geo; long ld; size_t plane;     // [8][lines][ld] planes of the geometry kernel, `plane` elements apart
  RefineState* st; RefineParams prm; RefineCtrl* ctrl;
  int* idx; const int* pt_surf;            // slot -> point, point -> surface
  int* line_surf; T* line_alpha; T* th0;   // the batch's evaluation requests (3 lines per slot, theta0 per slot)
  T* gam; T* dalpha; T* dth0; int* info;   // per-slot scratch of the growth-rate stage
  int* hist;                               // pinned host memory: hist[r] = 1 + points still running after round r
  int hist_len;
  int lds_tangent;                         // 1: alpha-tangent staged in LDS by the block (N small enough), 0: streamed by the solving wave
};

// evaluation request of a point into slot j of the batch: the three field lines of utils.py:1641-1646 and theta0
__device__ inline void refine_emit(const double (&x)[2], double del_alpha, int j, int surf, int* line_surf,
                                   double* line_alpha, double* th0) {
  for (int l = 0; l < 3; ++l) { line_surf[3 * j + l] = surf; line_alpha[3 * j + l] = x[0] + (l - 1) * 0.5 * del_alpha; }
  th0[j] = x[1];
}

}  // namespace ibs
