// Bounded quasi-Newton for the two unknowns (alpha, theta0) of the per-surface maximisation, written as a
// reverse-communication state machine that runs unchanged on the host (C ABI: ibs_lbfgsb2_*) and inside the
// refinement kernel (one thread per surface, ibs_refine_f64).
//
// What it restates (reference: /root/reference/ball_scan.py:307-314): the reference hands obj_w_grad to
//   scipy.optimize.minimize(..., jac=True, bounds=((0, pi), (0, pi/2)), options={ftol 5e-11, gtol 2e-8, maxiter 30})
// i.e. scipy's L-BFGS-B (un-vendored dependency; scipy 1.15.3 in the build container, which wraps L-BFGS-B 3.0 of
// Byrd, Lu, Nocedal, Zhu & Morales with m = 10 correction pairs, maxls = 20).  Because the Hellmann-Feynman jac is
// not the derivative of val (SURVEY 8 A6 note) the point where that algorithm stops is a property of the ALGORITHM
// (line-search collapses, memory restarts), not of the function: to land where the reference lands the same algorithm
// has to run.  Every decision of L-BFGS-B 3.0 is kept:
//   projected-gradient test (projgr) - generalized Cauchy point along the projected steepest-descent path (cauchy) -
//   subspace minimisation over the free variables with the projection / backtracking step of v3.0 (subsm) - More-Thuente
//   line search (dcsrch / dcstep: ftol 1e-3, gtol 0.9, xtol 0.1, first step 1 for a boxed problem, stpmax = 1 in the
//   first iteration, else the distance to the box along d) - <= 20 line-search steps, then restore and either restart
//   with an empty memory or stop ("ABNORMAL") - the f-reduction and projected-gradient stopping tests - the
//   curvature test s'y > eps * (-g's) that skips an update - m = 10 pairs with theta = y'y / s'y.
// What differs: with n = 2 the limited-memory matrix B = theta I - W M W' is FORMED (the m BFGS updates of theta I
// applied in order: identical to the compact representation in exact arithmetic) instead of carried as W and the
// Cholesky factors of its 2m x 2m middle matrix, so every product with B is a 2 x 2 product.  Rounding differs at
// the 1e-16 level; tests/test_lbfgsb2.py compares whole trajectories with scipy's on bounded 2-D test functions.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define IBS_HD __host__ __device__ inline
#else
#define IBS_HD inline
#endif

namespace ibs {
namespace lbfgsb2 {

constexpr int kM = 10;          // correction pairs kept (scipy default maxcor)
constexpr double kEps = 2.220446049250313e-16;

enum Task : int {
  T_START = 0,
  T_FG_START = 1,      // evaluate (f, g) at x: first evaluation
  T_FG_LNSRCH = 2,     // evaluate (f, g) at x: line-search trial point
  T_NEW_X = 3,         // (internal) an iteration has been completed
  T_CONV_PG = 10,      // CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL
  T_CONV_F = 11,       // CONVERGENCE: RELATIVE REDUCTION OF F <= FACTR*EPSMCH
  T_ABNORMAL = 12,     // ABNORMAL termination in the line search (no memory left to drop)
  T_STOP_MAXITER = 13, // STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT
  T_ERROR = 14
};

struct State {
  // problem (both variables bounded on both sides: nbd = 2 in L-BFGS-B's numbering)
  double l[2], u[2];
  double ftol;        // = factr * epsmch
  double pgtol;
  int maxiter, maxls;
  // iterate
  double x[2], f, g[2];
  // limited memory: pairs in a ring (oldest at head), B formed from them
  double ws[kM][2], wy[kM][2];
  double B[2][2];
  int col, head, itail, iupdat;
  double theta;
  // work vectors: z = Cauchy / subspace point, d = search direction, t = x at the start of the line search, r = g there
  double z[2], d[2], t[2], r[2];
  double fold, gd, gdold, stp, dnorm, dtd, stpmx, sbgnrm;
  int iwhere[2];
  int iter, ifun, iback, nfgv, nskip, n_iterations, n_restarts;
  int task;
  // More-Thuente line-search state (dcsrch)
  int ls_brackt, ls_stage, ls_task;      // ls_task: 0 START, 1 FG, 2 CONVERGENCE, 3 WARNING, 4 ERROR
  double ls_ginit, ls_gtest, ls_gx, ls_gy, ls_finit, ls_fx, ls_fy, ls_stx, ls_sty, ls_stmin, ls_stmax, ls_width, ls_width1;
};

// Work arrays of one step (indexed by run-time values: as locals of a GPU kernel they would live in scratch memory; the
// refinement kernel hands in a piece of LDS)
struct Work {
  double d[2], tb[2], ot[2], zc[2], dold[2], rr[2], xp[2];
  int iorder[2], oidx[2], ind[2];
};

IBS_HD double dmax(double a, double b) { return a > b ? a : b; }
IBS_HD double dmin(double a, double b) { return a < b ? a : b; }
IBS_HD double dabs(double a) { return a < 0 ? -a : a; }

// ------------------------------------------------------------------ B = theta I updated by the stored pairs
IBS_HD void form_B(State& s) {
  s.B[0][0] = s.theta; s.B[0][1] = 0.0; s.B[1][0] = 0.0; s.B[1][1] = s.theta;
  int p = s.head;
  for (int j = 0; j < s.col; ++j) {
    const double s0 = s.ws[p][0], s1 = s.ws[p][1], y0 = s.wy[p][0], y1 = s.wy[p][1];
    const double b0 = s.B[0][0] * s0 + s.B[0][1] * s1, b1 = s.B[1][0] * s0 + s.B[1][1] * s1;   // B s
    const double sBs = s0 * b0 + s1 * b1, sy = s0 * y0 + s1 * y1;
    s.B[0][0] += -b0 * b0 / sBs + y0 * y0 / sy;
    s.B[0][1] += -b0 * b1 / sBs + y0 * y1 / sy;
    s.B[1][1] += -b1 * b1 / sBs + y1 * y1 / sy;
    s.B[1][0] = s.B[0][1];
    p = (p + 1) % kM;
  }
}
IBS_HD void reset_memory(State& s) {      // "refresh the lbfgs memory and restart the iteration"
  s.col = 0; s.head = 0; s.theta = 1.0; s.iupdat = 0;
  form_B(s);
}

// ------------------------------------------------------------------ projgr: inf-norm of the projected gradient
IBS_HD double projgr(const State& s) {
  double nrm = 0.0;
  for (int i = 0; i < 2; ++i) {
    double gi = s.g[i];
    if (gi < 0.0) gi = dmax(s.x[i] - s.u[i], gi);
    else gi = dmin(s.x[i] - s.l[i], gi);
    nrm = dmax(nrm, dabs(gi));
  }
  return nrm;
}

// ------------------------------------------------------------------ cauchy: generalized Cauchy point -> s.z
// (piecewise-linear path x(t) = P(x - t g); the first local minimiser of the quadratic model along it)
IBS_HD void cauchy(State& s, Work& w) {
  if (s.sbgnrm <= 0.0) { s.z[0] = s.x[0]; s.z[1] = s.x[1]; return; }
  bool bnded = true;
  int nbreak = 0, ibkmin = 0, nfree_cnt = 0;
  double bkmin = 0.0, f1 = 0.0;
  double* d = w.d; double* tb = w.tb; int* iorder = w.iorder;
  tb[0] = tb[1] = 0.0; iorder[0] = iorder[1] = 0;
  for (int i = 0; i < 2; ++i) {
    const double neggi = -s.g[i];
    double tl = 0.0, tu = 0.0;
    if (s.iwhere[i] != 3 && s.iwhere[i] != -1) {
      tl = s.x[i] - s.l[i];
      tu = s.u[i] - s.x[i];
      const bool xlower = tl <= 0.0, xupper = tu <= 0.0;
      s.iwhere[i] = 0;
      if (xlower) { if (neggi <= 0.0) s.iwhere[i] = 1; }
      else if (xupper) { if (neggi >= 0.0) s.iwhere[i] = 2; }
      else { if (dabs(neggi) <= 0.0) s.iwhere[i] = -3; }
    }
    if (s.iwhere[i] != 0 && s.iwhere[i] != -1) {
      d[i] = 0.0;
    } else {
      d[i] = neggi;
      f1 -= neggi * neggi;
      if (neggi < 0.0) {                       // moves towards the lower bound
        iorder[nbreak] = i; tb[nbreak] = tl / (-neggi);
        if (nbreak == 0 || tb[nbreak] < bkmin) { bkmin = tb[nbreak]; ibkmin = nbreak; }
        ++nbreak;
      } else if (neggi > 0.0) {                // towards the upper bound
        iorder[nbreak] = i; tb[nbreak] = tu / neggi;
        if (nbreak == 0 || tb[nbreak] < bkmin) { bkmin = tb[nbreak]; ibkmin = nbreak; }
        ++nbreak;
      } else {                                 // (both-sided bounds: only neggi == 0 lands here)
        ++nfree_cnt;
        if (dabs(neggi) > 0.0) bnded = false;
      }
    }
  }
  s.z[0] = s.x[0]; s.z[1] = s.x[1];
  if (nbreak == 0 && nfree_cnt == 0) return;   // d is the zero vector
  // derivatives of the model along d at t = 0:  f1 = g'd = -d'd,  f2 = d'B d,  f2_org = theta d'd
  const double f2_org = -s.theta * f1;
  auto quad = [&](const double* dd) {
    const double b0 = s.B[0][0] * dd[0] + s.B[0][1] * dd[1], b1 = s.B[1][0] * dd[0] + s.B[1][1] * dd[1];
    return dd[0] * b0 + dd[1] * b1;
  };
  double f2 = s.col > 0 ? quad(d) : f2_org;
  double dtm = -f1 / f2, tsum = 0.0;
  if (nbreak > 0) {
    int nleft = nbreak, it = 1;
    double tj = 0.0;
    // breakpoints in ascending order (two at most)
    int* order_idx = w.oidx; double* order_t = w.ot;
    order_idx[0] = iorder[ibkmin]; order_t[0] = bkmin;
    if (nbreak == 2) { const int o = 1 - ibkmin; order_idx[1] = iorder[o]; order_t[1] = tb[o]; }
    bool all_fixed = false;
    double* zc = w.zc;                         // path displacement x(t_j) - x, accumulated segment by segment (the c of cauchy)
    zc[0] = zc[1] = 0.0;
    while (true) {
      const double tj0 = tj;
      tj = order_t[it - 1];
      const int ibp = order_idx[it - 1];
      const double dt = tj - tj0;
      if (dtm < dt) break;                     // the minimiser lies inside this segment
      tsum += dt; --nleft; ++it;
      const double dibp = d[ibp];
      double* d_old = w.dold;
      d_old[0] = d[0]; d_old[1] = d[1];
      d[ibp] = 0.0;
      double zibp;
      if (dibp > 0.0) { zibp = s.u[ibp] - s.x[ibp]; s.z[ibp] = s.u[ibp]; s.iwhere[ibp] = 2; }
      else { zibp = s.l[ibp] - s.x[ibp]; s.z[ibp] = s.l[ibp]; s.iwhere[ibp] = 1; }
      if (nleft == 0 && nbreak == 2) { dtm = dt; all_fixed = true; break; }   // every variable is fixed
      // derivatives of the model on the next segment, updated the way cauchy updates them (same cancellations):
      //   f1 += dt f2 + dibp^2 - theta dibp zibp + dibp w'Mc,   f2 += -theta dibp^2 + 2 dibp w'Mp - dibp^2 w'Mw
      // with W M W' = theta I - B:  w'Mc = theta zc_i - (B zc)_i,  w'Mp = theta d_i - (B d)_i,  w'Mw = theta - B_ii  (i = ibp)
      const double dibp2 = dibp * dibp;
      f1 = f1 + dt * f2 + dibp2 - s.theta * dibp * zibp;
      f2 = f2 - s.theta * dibp2;
      if (s.col > 0) {
        zc[0] += dt * d_old[0]; zc[1] += dt * d_old[1];
        const double wmc = s.theta * zc[ibp] - (s.B[ibp][0] * zc[0] + s.B[ibp][1] * zc[1]);
        const double wmp = s.theta * d_old[ibp] - (s.B[ibp][0] * d_old[0] + s.B[ibp][1] * d_old[1]);
        const double wmw = s.theta - s.B[ibp][ibp];
        f1 += dibp * wmc;
        f2 += 2.0 * dibp * wmp - dibp2 * wmw;
      }
      f2 = dmax(kEps * f2_org, f2);
      if (nleft > 0) { dtm = -f1 / f2; continue; }
      if (bnded) { f1 = 0.0; f2 = 0.0; dtm = 0.0; }
      else dtm = -f1 / f2;
      break;
    }
    if (all_fixed) return;
  }
  if (dtm <= 0.0) dtm = 0.0;
  tsum += dtm;
  for (int i = 0; i < 2; ++i) if (d[i] != 0.0) s.z[i] = s.x[i] + tsum * d[i];
}

// ------------------------------------------------------------------ subsm: subspace minimisation over the free variables
// at the Cauchy point, with the projection / backtracking step of L-BFGS-B 3.0.  s.z: in = xcp, out = the new point
IBS_HD void subsm(State& s, Work& w) {
  int* ind = w.ind; int nsub = 0;
  for (int i = 0; i < 2; ++i) if (s.iwhere[i] <= 0) ind[nsub++] = i;
  if (nsub == 0 || s.col == 0) return;
  // r = -Z'(g + B (xcp - x))
  const double dz0 = s.z[0] - s.x[0], dz1 = s.z[1] - s.x[1];
  double* rr = w.rr;
  rr[0] = -(s.g[0] + s.B[0][0] * dz0 + s.B[0][1] * dz1); rr[1] = -(s.g[1] + s.B[1][0] * dz0 + s.B[1][1] * dz1);
  double* d = w.d;                           // Newton direction in the free subspace (indexed like ind)
  d[0] = d[1] = 0.0;
  if (nsub == 2) {
    const double det = s.B[0][0] * s.B[1][1] - s.B[0][1] * s.B[1][0];
    d[0] = (s.B[1][1] * rr[0] - s.B[0][1] * rr[1]) / det;
    d[1] = (s.B[0][0] * rr[1] - s.B[1][0] * rr[0]) / det;
  } else {
    d[0] = rr[ind[0]] / s.B[ind[0]][ind[0]];
  }
  // try the projection of xcp + d onto the box
  double* xp = w.xp;
  xp[0] = s.z[0]; xp[1] = s.z[1];
  bool hit = false;
  for (int i = 0; i < nsub; ++i) {
    const int k = ind[i];
    double xk = dmax(s.l[k], s.z[k] + d[i]);
    xk = dmin(s.u[k], xk);
    s.z[k] = xk;
    if (xk == s.l[k] || xk == s.u[k]) hit = true;
  }
  if (!hit) return;
  // sign of the directional derivative of the projected step
  double dd_p = 0.0;
  for (int i = 0; i < 2; ++i) dd_p += (s.z[i] - s.x[i]) * s.g[i];
  if (dd_p > 0.0) {
    s.z[0] = xp[0]; s.z[1] = xp[1];
    double alpha = 1.0, temp1 = alpha;
    int ibd = -1;
    for (int i = 0; i < nsub; ++i) {
      const int k = ind[i];
      const double dk = d[i];
      if (dk < 0.0) {
        const double temp2 = s.l[k] - s.z[k];
        if (temp2 >= 0.0) temp1 = 0.0;
        else if (dk * alpha < temp2) temp1 = temp2 / dk;
      } else if (dk > 0.0) {
        const double temp2 = s.u[k] - s.z[k];
        if (temp2 <= 0.0) temp1 = 0.0;
        else if (dk * alpha > temp2) temp1 = temp2 / dk;
      }
      if (temp1 < alpha) { alpha = temp1; ibd = i; }
    }
    if (alpha < 1.0 && ibd >= 0) {
      const double dk = d[ibd];
      const int k = ind[ibd];
      if (dk > 0.0) { s.z[k] = s.u[k]; d[ibd] = 0.0; }
      else if (dk < 0.0) { s.z[k] = s.l[k]; d[ibd] = 0.0; }
    }
    for (int i = 0; i < nsub; ++i) s.z[ind[i]] += alpha * d[i];
  }
}

// ------------------------------------------------------------------ dcstep / dcsrch (More & Thuente, MINPACK-2)
IBS_HD void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp, double fp,
                   double dp, int& brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (dx / dabs(dx));
  double stpf;
  if (fp > fx) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double sc = dmax(dmax(dabs(theta), dabs(dx)), dabs(dp));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dx / sc) * (dp / sc));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (dabs(stpc - stx) < dabs(stpq - stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    brackt = 1;
  } else if (sgnd < 0.0) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double sc = dmax(dmax(dabs(theta), dabs(dx)), dabs(dp));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dx / sc) * (dp / sc));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (dabs(stpc - stp) > dabs(stpq - stp)) stpf = stpc;
    else stpf = stpq;
    brackt = 1;
  } else if (dabs(dp) < dabs(dx)) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double sc = dmax(dmax(dabs(theta), dabs(dx)), dabs(dp));
    double gamma = sc * sqrt(dmax(0.0, (theta / sc) * (theta / sc) - (dx / sc) * (dp / sc)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
    double stpc;
    if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (dabs(stpc - stp) < dabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      if (stp > stx) stpf = dmin(stp + 0.66 * (sty - stp), stpf);
      else stpf = dmax(stp + 0.66 * (sty - stp), stpf);
    } else {
      if (dabs(stpc - stp) > dabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      stpf = dmin(stpmax, stpf);
      stpf = dmax(stpmin, stpf);
    }
  } else {
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double sc = dmax(dmax(dabs(theta), dabs(dy)), dabs(dp));
      double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dy / sc) * (dp / sc));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) stpf = stpmax;
    else stpf = stpmin;
  }
  // the interval that contains a minimiser: (fp > fx) y <- trial;  else { (sgnd < 0) y <- x; }  x <- trial.  Written as
  // selects on the VALUES: conditional stores through the reference arguments become one store through a selected address,
  // which keeps the caller's copies in memory (a GPU kernel's scratch)
  const bool up = fp > fx, sw = !up && sgnd < 0.0;
  const double nsty = up ? stp : (sw ? stx : sty), nfy = up ? fp : (sw ? fx : fy), ndy = up ? dp : (sw ? dx : dy);
  const double nstx = up ? stx : stp, nfx = up ? fx : fp, ndx = up ? dx : dp;
  stx = nstx; fx = nfx; dx = ndx; sty = nsty; fy = nfy; dy = ndy;
  stp = stpf;
}

// one call of dcsrch with (f, g) at the current stp; ls_task on return: 1 = evaluate at the new stp, 2 = converged,
// 3 = warning (treated like convergence by lnsrlb), 4 = error
IBS_HD void dcsrch(State& s, double f, double g, double stpmin, double stpmax) {
  constexpr double ftol = 1.0e-3, gtol = 0.9, xtol = 0.1, xtrapl = 1.1, xtrapu = 4.0;
  if (s.ls_task == 0) {
    if (s.stp < stpmin || s.stp > stpmax || g >= 0.0) { s.ls_task = 4; return; }
    s.ls_brackt = 0; s.ls_stage = 1;
    s.ls_finit = f; s.ls_ginit = g; s.ls_gtest = ftol * g;
    s.ls_width = stpmax - stpmin; s.ls_width1 = s.ls_width / 0.5;
    s.ls_stx = 0.0; s.ls_fx = f; s.ls_gx = g;
    s.ls_sty = 0.0; s.ls_fy = f; s.ls_gy = g;
    s.ls_stmin = 0.0; s.ls_stmax = s.stp + xtrapu * s.stp;
    s.ls_task = 1;
    return;
  }
  const double ftest = s.ls_finit + s.stp * s.ls_gtest;
  if (s.ls_stage == 1 && f <= ftest && g >= 0.0) s.ls_stage = 2;
  int task = 1;
  if (s.ls_brackt && (s.stp <= s.ls_stmin || s.stp >= s.ls_stmax)) task = 3;     // rounding errors prevent progress
  if (s.ls_brackt && s.ls_stmax - s.ls_stmin <= xtol * s.ls_stmax) task = 3;      // xtol test satisfied
  if (s.stp == stpmax && f <= ftest && g <= s.ls_gtest) task = 3;                 // stp = stpmax
  if (s.stp == stpmin && (f > ftest || g >= s.ls_gtest)) task = 3;                // stp = stpmin
  if (f <= ftest && dabs(g) <= gtol * (-s.ls_ginit)) task = 2;                    // convergence
  if (task != 1) { s.ls_task = task; return; }
  // (ONE call of dcstep on local copies: two calls, one on locals and one on the state's fields, merge into one body that
  // reaches its operands through pointers, which puts the locals into a GPU kernel's scratch memory.  With shift = 0 the
  // arithmetic below returns its operands bit for bit.)
  const bool modified = s.ls_stage == 1 && f <= s.ls_fx && f > ftest;
  const double shift = modified ? s.ls_gtest : 0.0;                          // the modified function f - stp gtest
  double fxm = s.ls_fx - s.ls_stx * shift, fym = s.ls_fy - s.ls_sty * shift, gxm = s.ls_gx - shift, gym = s.ls_gy - shift;
  double stx = s.ls_stx, sty = s.ls_sty, stp = s.stp;
  int brackt = s.ls_brackt;
  dcstep(stx, fxm, gxm, sty, fym, gym, stp, f - s.stp * shift, g - shift, brackt, s.ls_stmin, s.ls_stmax);
  s.ls_fx = fxm + stx * shift; s.ls_fy = fym + sty * shift; s.ls_gx = gxm + shift; s.ls_gy = gym + shift;    // (shift = 0: the values themselves)
  s.ls_stx = stx; s.ls_sty = sty; s.stp = stp; s.ls_brackt = brackt;
  if (s.ls_brackt) {
    if (dabs(s.ls_sty - s.ls_stx) >= 0.66 * s.ls_width1) s.stp = s.ls_stx + 0.5 * (s.ls_sty - s.ls_stx);
    s.ls_width1 = s.ls_width;
    s.ls_width = dabs(s.ls_sty - s.ls_stx);
  }
  if (s.ls_brackt) { s.ls_stmin = dmin(s.ls_stx, s.ls_sty); s.ls_stmax = dmax(s.ls_stx, s.ls_sty); }
  else { s.ls_stmin = s.stp + xtrapl * (s.stp - s.ls_stx); s.ls_stmax = s.stp + xtrapu * (s.stp - s.ls_stx); }
  s.stp = dmax(s.stp, stpmin);
  s.stp = dmin(s.stp, stpmax);
  if ((s.ls_brackt && (s.stp <= s.ls_stmin || s.stp >= s.ls_stmax)) ||
      (s.ls_brackt && s.ls_stmax - s.ls_stmin <= xtol * s.ls_stmax))
    s.stp = s.ls_stx;
  s.ls_task = 1;
}

// ------------------------------------------------------------------ lnsrlb: one step of the line search
// returns info (0 ok, -4 = d is not a descent direction); sets task to T_FG_LNSRCH (x = trial point) or T_NEW_X
IBS_HD int lnsrlb(State& s, bool first) {
  if (first) {
    s.dtd = s.d[0] * s.d[0] + s.d[1] * s.d[1];
    s.dnorm = sqrt(s.dtd);
    s.stpmx = 1.0e10;
    if (s.iter == 0) s.stpmx = 1.0;
    else {
      for (int i = 0; i < 2; ++i) {
        const double a1 = s.d[i];
        if (a1 < 0.0) {
          const double a2 = s.l[i] - s.x[i];
          if (a2 >= 0.0) s.stpmx = 0.0;
          else if (a1 * s.stpmx < a2) s.stpmx = a2 / a1;
        } else if (a1 > 0.0) {
          const double a2 = s.u[i] - s.x[i];
          if (a2 <= 0.0) s.stpmx = 0.0;
          else if (a1 * s.stpmx > a2) s.stpmx = a2 / a1;
        }
      }
    }
    s.stp = 1.0;                               // (boxed problem: also in the first iteration)
    s.t[0] = s.x[0]; s.t[1] = s.x[1];
    s.r[0] = s.g[0]; s.r[1] = s.g[1];
    s.fold = s.f;
    s.ifun = 0; s.iback = 0;
    s.ls_task = 0;
  }
  s.gd = s.g[0] * s.d[0] + s.g[1] * s.d[1];
  if (s.ifun == 0) {
    s.gdold = s.gd;
    if (s.gd >= 0.0) return -4;                // the directional derivative >= 0: line search is impossible
  }
  dcsrch(s, s.f, s.gd, 0.0, s.stpmx);
  if (s.ls_task == 4) return -4;
  if (s.ls_task == 1) {
    s.task = T_FG_LNSRCH;
    ++s.ifun; ++s.nfgv;
    s.iback = s.ifun - 1;
    if (s.stp == 1.0) { s.x[0] = s.z[0]; s.x[1] = s.z[1]; }
    else { s.x[0] = s.stp * s.d[0] + s.t[0]; s.x[1] = s.stp * s.d[1] + s.t[1]; }
  } else {
    s.task = T_NEW_X;
  }
  return 0;
}

// ------------------------------------------------------------------ driver
IBS_HD void init(State& s, const double* x0, const double* lo, const double* hi, double ftol, double pgtol, int maxiter,
                 int maxls) {
  for (int i = 0; i < 2; ++i) {
    s.l[i] = lo[i]; s.u[i] = hi[i];
    s.x[i] = dmin(dmax(x0[i], lo[i]), hi[i]);                 // scipy clips x0 into the box; so does 'active'
    s.iwhere[i] = (hi[i] - lo[i] <= 0.0) ? 3 : 0;
    s.g[i] = 0.0; s.z[i] = s.x[i]; s.d[i] = 0.0; s.t[i] = s.x[i]; s.r[i] = 0.0;
  }
  s.ftol = ftol; s.pgtol = pgtol; s.maxiter = maxiter; s.maxls = maxls;
  s.f = 0.0; s.fold = 0.0; s.gd = 0.0; s.gdold = 0.0; s.stp = 0.0; s.dnorm = 0.0; s.dtd = 0.0; s.stpmx = 0.0; s.sbgnrm = 0.0;
  s.itail = 0;
  reset_memory(s);
  s.iter = 0; s.ifun = 0; s.iback = 0; s.nfgv = 0; s.nskip = 0; s.n_iterations = 0; s.n_restarts = 0;
  s.ls_brackt = 0; s.ls_stage = 0; s.ls_task = 0;
  s.ls_ginit = s.ls_gtest = s.ls_gx = s.ls_gy = s.ls_finit = s.ls_fx = s.ls_fy = s.ls_stx = s.ls_sty = 0.0;
  s.ls_stmin = s.ls_stmax = s.ls_width = s.ls_width1 = 0.0;
  for (int j = 0; j < kM; ++j) { s.ws[j][0] = s.ws[j][1] = 0.0; s.wy[j][0] = s.wy[j][1] = 0.0; }
  s.task = T_FG_START;                                        // first request: (f, g) at the projected start point
}

IBS_HD bool finished(const State& s) { return s.task >= T_CONV_PG; }

// Advance with (f, g) evaluated at s.x.  Returns true when another evaluation at the (new) s.x is wanted,
// false when the minimisation has ended (s.x, s.f = result; s.task says why).
IBS_HD bool step(State& s, double f, const double* g, Work& w) {
  if (finished(s)) return false;
  s.f = f; s.g[0] = g[0]; s.g[1] = g[1];
  bool new_search = false;
  if (s.task == T_FG_START) {
    s.nfgv = 1;
    s.sbgnrm = projgr(s);
    if (s.sbgnrm <= s.pgtol) { s.task = T_CONV_PG; return false; }
    new_search = true;
  }
  while (true) {
    if (new_search) {
      // ---- 222: generalized Cauchy point, subspace minimisation, direction
      cauchy(s, w);
      subsm(s, w);
      s.d[0] = s.z[0] - s.x[0]; s.d[1] = s.z[1] - s.x[1];
    }
    // ---- 666: line search
    const int info = lnsrlb(s, new_search);
    new_search = false;
    if (info != 0 || s.iback >= s.maxls) {
      s.x[0] = s.t[0]; s.x[1] = s.t[1]; s.g[0] = s.r[0]; s.g[1] = s.r[1]; s.f = s.fold;     // restore the previous iterate
      if (s.col == 0) {
        if (info == 0) { --s.nfgv; --s.ifun; --s.iback; }
        s.task = T_ABNORMAL; ++s.iter;
        return false;
      }
      if (info == 0) --s.nfgv;
      reset_memory(s);                          // refresh the memory and restart the iteration from the same point
      ++s.n_restarts;
      new_search = true;
      continue;
    }
    if (s.task == T_FG_LNSRCH) return true;     // evaluate at the trial point
    // ---- a new iterate
    ++s.iter;
    s.sbgnrm = projgr(s);
    // (scipy's driver: iteration counter and the maxiter stop come before the solver's own tests)
    ++s.n_iterations;
    if (s.n_iterations >= s.maxiter) { s.task = T_STOP_MAXITER; return false; }
    if (s.sbgnrm <= s.pgtol) { s.task = T_CONV_PG; return false; }
    const double ddum0 = dmax(dmax(dabs(s.fold), dabs(s.f)), 1.0);
    if ((s.fold - s.f) <= s.ftol * ddum0) { s.task = T_CONV_F; return false; }
    // ---- update of the limited-memory matrix with s = x_new - x_old, y = g_new - g_old
    const double y0 = s.g[0] - s.r[0], y1 = s.g[1] - s.r[1];
    const double rr = y0 * y0 + y1 * y1;
    double dr, ddum;
    double sd0 = s.d[0], sd1 = s.d[1];
    if (s.stp == 1.0) { dr = s.gd - s.gdold; ddum = -s.gdold; }
    else { dr = (s.gd - s.gdold) * s.stp; sd0 *= s.stp; sd1 *= s.stp; ddum = -s.gdold * s.stp; }
    if (dr <= kEps * ddum) {
      ++s.nskip;                                // skip the update: curvature condition fails
    } else {
      ++s.iupdat;
      if (s.iupdat <= kM) { s.col = s.iupdat; s.itail = (s.head + s.iupdat - 1) % kM; }
      else { s.itail = (s.itail + 1) % kM; s.head = (s.head + 1) % kM; }
      s.ws[s.itail][0] = sd0; s.ws[s.itail][1] = sd1;
      s.wy[s.itail][0] = y0; s.wy[s.itail][1] = y1;
      s.theta = rr / dr;
      form_B(s);
      // (formt's Cholesky of T = theta SS + L D^-1 L' fails only through rounding when every s'y > 0; its counterpart
      //  here: a B that is not positive definite -> refresh the memory)
      const double det = s.B[0][0] * s.B[1][1] - s.B[0][1] * s.B[1][0];
      if (!(s.B[0][0] > 0.0) || !(det > 0.0)) reset_memory(s);
    }
    new_search = true;
  }
}
IBS_HD bool step(State& s, double f, const double* g) {
  Work w;
  return step(s, f, g, w);
}

}  // namespace lbfgsb2
}  // namespace ibs
