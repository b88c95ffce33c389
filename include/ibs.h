/* libibs_hip.so -- C ABI of the MI355X-native ideal-ballooning hot path.
 *
 * Drop-in boundary for the reference's module-level Python operators (there is no FFI upstream:
 * the operators are reached by `from utils import *`, ball_scan.py:19).  Each entry point names
 * the reference interface it replaces (paths relative to the reference checkout).  Plain pointers
 * and sizes only; no exceptions cross the boundary.
 *
 * Conventions
 *   - return value: 0 = ok; < 0 = argument/runtime error (text via ibs_last_error());
 *     > 0 = number of systems whose info word reports a non-zero status (IBS_MEM_HOST calls only: a device-pointer
 *     call is asynchronous and returns 0 -- request the info words and look at their status bits).
 *   - `mem`: IBS_MEM_DEVICE = all data pointers are device (HBM) pointers, the call is
 *     asynchronous on the context's stream; IBS_MEM_HOST = host pointers, the library stages
 *     through its own device workspace and the call returns after the results are back.
 *   - grids are uniform in theta with N points (N odd, N >= 66) and spacing h; arrays of one system/line are contiguous,
 *     consecutive systems are `ld` elements apart.  Up to N = 2050 the register-resident kernels run (one wavefront or a part of
 *     one per system).  The reference takes any length (utils.py:1556-1624; its own rule N = 2 mpol ntor 4 + 1, ball_scan.py:201-208,
 *     passes 2050 from mpol ntor > 256 on): for 2050 < N <= 65537 ibs_solve_gcf_f64 / _f32, ibs_solve_gcfh_f64, ibs_gamma_scan_f64 /
 *     _warm_f64 (the guesses are then unused) / _argmax_f64, ibs_gamma_points_f64, ibs_obj_w_grad_f64 and ibs_sturm_count_f64 run a
 *     generic path that works in division form on the rows in memory (csrc/ibs_long.hip: correct to the same tolerances, slower per
 *     row); the on-device refinement ibs_refine_f64 stays limited to N <= 2050 (IBS_ERR_UNSUPPORTED): beyond, ibs_obj_w_grad_f64
 *     under the caller's optimizer -- upstream's own form (ball_scan.py:307-314); the Python host layer drives the library's
 *     L-BFGS-B state machines (ibs_lbfgsb2_*) with one batched ibs_obj_w_grad_f64 launch per round.
 *   - optional outputs may be NULL.
 *   - info word per system: bits 0..15 = sweeps used, bits 16.. = status
 *     (bit 0 = iteration cap hit, bit 1 = invalid data: non-finite, g <= 0 or f <= 0.  Bits 2-4 are INFORMATIONAL -- the returned
 *     values are good; return values > 0 and the host-side counts look at bits 0 and 1 only:
 *     bit 2, FP32 eigenvalue-only path: the all-FP32 result failed its FP64 certificate and the system was solved in FP64;
 *     bit 3, FP64 raw-system entry points: the closing bracket of the shift iteration disagreed with the twisted factorisation's
 *            Rayleigh polish by more than min(512, 2 N) eps ||A|| and lam_max was closed again by division-form multisection on the
 *            rows (the recurrence of LAPACK dstebz): 0 .. 60 systems per million on iid-random coefficients, none on field-line
 *            data.  With it |lam - lam_max| <= 4 N eps ||A|| holds a priori (tests/test_gpu_configs.py), ||A|| = max_r (|d_r| +
 *            e_r + e_{r+1}) / f_r of utils.py:1584-1592's rows;
 *     bit 4, only with option "sigma0" set and both lam and info requested: lam_max >= sigma0.  The reference's ARPACK call returns
 *            the eigenpair NEAREST sigma0 (utils.py:1597); this library always returns lam_max's.  They are the same eigenpair
 *            whenever lam_max < sigma0 -- and only then: this bit marks the one case in which upstream may have returned another).
 */
#ifndef IBS_H
#define IBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IBS_MEM_DEVICE 0
#define IBS_MEM_HOST 1

#define IBS_ERR_ARG (-1)
#define IBS_ERR_HIP (-2)
#define IBS_ERR_UNSUPPORTED (-3)

typedef struct ibs_ctx ibs_ctx;

int ibs_version(void);
const char* ibs_last_error(void);

/* context = device + stream + staging workspace.  One per process/GPU (ball_scan.py runs one
 * process per MPI rank; here one process per GPU). */
int ibs_create(ibs_ctx** ctx, int device_id);
int ibs_destroy(ibs_ctx* ctx);
/* run on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream */
int ibs_set_stream(ibs_ctx* ctx, void* hip_stream);
int ibs_synchronize(ibs_ctx* ctx);
int ibs_device_count(void);
/* Bounded quasi-Newton on the two unknowns (alpha, theta0), reverse communication, HOST side (no GPU involved).
 * Replaces: the optimizer behind ball_scan.py:307-314, scipy.optimize.minimize(obj_w_grad, jac=True,
 * bounds=((0, pi), (0, pi/2)), options={ftol, gtol, maxiter}) = scipy's L-BFGS-B (un-vendored dependency: scipy, pinned
 * to 1.15.3 by the build container; algorithm L-BFGS-B 3.0, m = 10, maxls = 20), restated for n = 2 in
 * csrc/ibs_lbfgsb2.hpp.  The same code runs per surface inside ibs_refine_f64; these entry points let a host drive
 * it and let the CPU tests compare its trajectories with scipy's.
 *   state: caller-owned buffer of ibs_lbfgsb2_state_bytes() bytes.
 *   init:  x0 is clipped into [lo, hi]; the first evaluation is wanted at the clipped point (returned by step's
 *          x_next convention: call ibs_lbfgsb2_result() for it, or clip yourself).
 *   step:  hand over (f, g[2]) evaluated at the point last requested; returns 1 = evaluate at x_next[2],
 *          0 = finished (x_next = result).
 *   result: x[2], f, counters[5] = {iterations, evaluations, task code (10 = projected gradient <= gtol,
 *          11 = f reduction <= ftol, 12 = abnormal line-search termination, 13 = maxiter), memory restarts, skipped updates}. */
int ibs_lbfgsb2_state_bytes(void);
int ibs_lbfgsb2_init(void* state, const double* x0, const double* lo, const double* hi, double ftol, double gtol,
                     int32_t maxiter, int32_t maxls);
int ibs_lbfgsb2_step(void* state, double f, const double* g, double* x_next);
int ibs_lbfgsb2_result(const void* state, double* x, double* f, int32_t* counters);

/* Native collective (optional): ONE all-gather of per-surface rows over RCCL on the context's stream, with no host
 * framework in between.  Replaces: the three comm_lead.Gather([x, MPI.DOUBLE], ..., root=0) of ball_scan.py:345-347
 * (theta0*, alpha*, gam per surface) as one all-gather of [n_surf_local][k] doubles per rank.
 *   ibs_comm_load(path)      bind librccl at run time (NULL: the copy the process already holds, else librccl.so.1);
 *   ibs_comm_unique_id(id)   rank 0: 128 bytes to hand to every rank by any means (MPI, a torch.distributed broadcast, a file);
 *   ibs_comm_init(ctx, id, rank, nranks)   collective over the ranks (ncclCommInitRank on the context's device);
 *   ibs_comm_allgather_f64(ctx, send, recv, count)   recv[nranks][count] <- every rank's send[count], device pointers,
 *                            asynchronous on the context's stream (ordered after the kernels that produced `send`);
 *   ibs_comm_allgather_start_f64(ctx, send, recv, count, slot, then_wait_slot)   the same gather, ordered after everything enqueued so
 *                            far on the context's stream but run on the communicator's OWN stream: the next scan does not
 *                            wait for the ranks to meet (the reference has nothing to overlap: its Gather is blocking).
 *                            slot in [0, 16) names the gather; then_wait_slot >= 0 (another slot) additionally does
 *                            ibs_comm_wait(ctx, then_wait_slot) in the same call, -1 = nothing, -2 - s = wait for slot s
 *                            on the HOST instead (event query; blocks only if that gather is still running) so that the
 *                            context's stream carries no wait at all -- for callers running several slots ahead;
 *   ibs_comm_wait(ctx, slot) orders the context's stream after the gather of `slot` (slot < 0: after every pending one)
 *                            without blocking the host: call it before `send` / `recv` of that slot are reused or read;
 *   ibs_comm_destroy(ctx)    (also done by ibs_destroy).
 * The Python layer (BallooningScan, bench.py) uses torch.distributed by default; Context.comm_init(dist) switches it. */
int ibs_comm_load(const char* librccl_path);
int ibs_comm_unique_id(void* id128);
int ibs_comm_init(ibs_ctx* ctx, const void* id128, int32_t rank, int32_t nranks);
int ibs_comm_allgather_f64(ibs_ctx* ctx, const double* send, double* recv, int64_t count_per_rank);
int ibs_comm_allgather_start_f64(ibs_ctx* ctx, const double* send, double* recv, int64_t count_per_rank, int32_t slot,
                                 int32_t then_wait_slot);
int ibs_comm_wait(ibs_ctx* ctx, int32_t slot);
int ibs_comm_destroy(ibs_ctx* ctx);

/* Diagnostic override of a dispatch heuristic of THIS context (tests, experiments; nothing upstream corresponds).
 * name: "force_p" (lanes per system 64|32|16), "scan_chain" (theta0 values chained through one wave),
 * "chain_w1" / "chain_w2" (relative widths of the chain's warm starts), "geo_lpp" (lanes per grid point of the
 * geometry kernel 1|2|4|8, or -2 = two grid points per lane), "gcf_rows" (0: three-row staging instead of the row-streamed raw
 * kernel on long grids), "gcf_direct" (raw systems, one wave per system: rows read straight from global memory instead of staged in LDS; -1 = by
 * batch size, 0 = never, 1 = wherever a one-wave-per-system form would run: not where the sub-wave forms are picked, with a half-grid g, or
 * below the rows-per-lane the direct kernels are built for), "f32_lam" (FP32 eigenvalue-only requests: 1 = all-FP32 iteration + FP64 certificate, 2 = FP64 solver on
 * the FP32 arrays; 0 = form 2 where the 16-lane sub-wave kernels run, else form 1), "reclose" (FP64 raw systems: 1 = a solve whose closing bracket
 * disagrees with its Rayleigh polish is closed again in division form, the default; 2 = only marked with status bit 3 and the distance
 * in bits 5..10; 0 = off, i.e. the round-5 results), "sigma0" (any finite value: solves that return lam and info flag lam_max >= sigma0
 * with the informational status bit 4 -- the nearest-sigma report of the drop-in, utils.py:1597; NaN = off, the default),
 * "sturm_form" (ibs_sturm_count_f64: 1 = prefix-product sweep, 2 = division form with lanes as systems, 3 = division form with one wave
 * per system; 0 = by grid length and batch size, see ibs_sturm_count_f64), "forget_rows" (an action: drop the remembered verdicts on device-resident mode rows, see ibs_fieldline_geometry_f64),
 * "pack_mode" (1|2: hand-off of the fused argmax), "refine_tangent" (refinement: the alpha-tangent of a point
 * staged in LDS, 1, or read from global memory by the sums, 0: two blocks per CU instead of one at N = 969; -1 = by batch size);
 * value 0 = automatic (refine_tangent, gcf_direct: -1); value NaN = back to what ibs_create() read from the environment
 * (IBS_FORCE_P, IBS_SCAN_CHAIN, IBS_CHAIN_W1, IBS_CHAIN_W2, IBS_GEO_LPP are read once, there); name "all" with NaN
 * resets every option.  Results never depend on these switches beyond rounding; only the kernel variant does. */
int ibs_set_option(ibs_ctx* ctx, const char* name, double value);

/* Raw (g, c, f) systems -> largest eigenvalue of the tridiagonal pencil and the reference growth rate.
 * Replaces: utils.py:1574-1624 (the part of gamma_ball_full after the coefficient assembly), batched.
 *   lam[n_sys]  matrix eigenvalue lam_max(T, F)            (optional)
 *   gam[n_sys]  Simpson/FD4 Rayleigh quotient (utils.py:1618-1621)  (optional)
 *   X, dX [n_sys][N]  normalised eigenfunction and its derivative (utils.py:1602-1616) (optional)
 *   info[n_sys] (optional) */
int ibs_solve_gcf_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, double h, const double* g, const double* c,
                      const double* f, int64_t ld, double* lam, double* gam, double* X, double* dX,
                      int32_t* info, int32_t mem);
/* Same with the half-grid values gh[n_sys][ld] (gh[k] between grid points k and k+1, N-1 used) supplied by the
 * caller instead of the mean of neighbouring g: what utils.py:1567-1576 produces for a NON-uniform theta_PEST
 * (coefficients regridded by np.interp onto the uniform grid, g interpolated at the uniform half points). */
int ibs_solve_gcfh_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, double h, const double* g, const double* gh,
                       const double* c, const double* f, int64_t ld, double* lam, double* gam, double* X, double* dX,
                       int32_t* info, int32_t mem);
/* FP32 form (BASELINE configs[4], the stress leg).  With lam alone (gam, X, dX all null) the shift iteration runs in FP32 --
 * the throughput form -- and EVERY result is then certified in FP64 on the staged rows: one Sturm-count pair at
 * lam32 +- n eps32 ||A|| (n = N - 2) must read (0 above, >= 1 below); a system that fails (an FP32 count off by one between
 * two close eigenvalues would otherwise return lam_2) is solved in FP64 and carries status bit 2.  Hence
 * |lam - lam64| <= n eps32 ||A|| + the rounding of lam to FP32 for every system.  When gam, X or dX is asked for the arrays
 * are widened to FP64 as they are read and the FP64 solver runs on them (FP32 in HBM only; results rounded to FP32; the same
 * kernel forms as the FP64 entry point): an FP32 eigenvector's noise is multiplied by ~N^2 in the FD4 / Simpson quotient, so
 * an all-FP32 growth rate would be noise above N_zeta = 512.  |gam - gam64| <= 1e-6 at every N_zeta on smooth systems. */
int ibs_solve_gcf_f32(ibs_ctx* ctx, int64_t n_sys, int32_t N, float h, const float* g, const float* c,
                      const float* f, int64_t ld, float* lam, float* gam, float* X, float* dX,
                      int32_t* info, int32_t mem);

/* Field-line geometry x theta0 grid -> growth rates.
 * Replaces: the inner loops of ball_scan.py:248-275 (theta0 fold :267-268, gamma_ball_full call :269)
 * and utils.py:1556-1624; with dgam_dtheta0 also utils.py:1666-1680 (Hellmann-Feynman d/dtheta0).
 *   geometry arrays [n_lines][ld] (first N of each row used): bmag, gradpar (gradpar_theta_pest),
 *   cvdrift, cvdrift0, gds2, gds21, gds22;  dPdrho[n_lines] (ball_scan.py:262);  theta0[n_theta0].
 *   outputs are [n_lines][n_theta0] (X, dX: [n_lines][n_theta0][N]). */
int ibs_gamma_scan_f64(ibs_ctx* ctx, int32_t n_lines, int32_t n_theta0, int32_t N, double h,
                       const double* bmag, const double* gradpar, const double* cvdrift, const double* cvdrift0,
                       const double* gds2, const double* gds21, const double* gds22, int64_t ld,
                       const double* dPdrho, const double* theta0, double* gam, double* lam, double* X,
                       double* dX, double* dgam_dtheta0, int32_t* info, int32_t mem);

/* Same scan, warm-started: lam_guess[n_lines][n_theta0] holds the eigenvalues of a nearby problem (the
 * previous optimizer iteration, or the base equilibrium when one of the DOF-perturbed equilibria of
 * sims_runner_NCSX.py:151-276 is re-scanned) and guess_width their expected absolute change.  The result
 * is certified exactly as in the cold scan (a wrong guess costs sweeps, never correctness). */
int ibs_gamma_scan_warm_f64(ibs_ctx* ctx, int32_t n_lines, int32_t n_theta0, int32_t N, double h,
                            const double* bmag, const double* gradpar, const double* cvdrift, const double* cvdrift0,
                            const double* gds2, const double* gds21, const double* gds22, int64_t ld,
                            const double* dPdrho, const double* theta0, const double* lam_guess, double guess_width,
                            double* gam, double* lam, double* X, double* dX, double* dgam_dtheta0, int32_t* info,
                            int32_t mem);

/* Scan + per-surface first maximum in ONE call (device pointers only): the lines are ordered surface-major,
 * n_lines / n_surf consecutive lines form a surface, and pack[n_surf][2] = (max gam of the surface's (alpha, theta0)
 * table, its first row-major index as a double) -- the buffer the per-surface all-gather sends.
 * Replaces: ball_scan.py:248-295 (coarse scan loops + argmax with the first-maximum rule :283-288) for all surfaces of a
 * rank at once.  For small batches this is a single kernel launch: the thread block that completes a surface reduces
 * it (agent-scope release / acquire on a per-surface arrival counter); large batches run the chained / sub-wave scan
 * kernels followed by the reduction kernel.  * The arrival counters are per (context, stream): launches of one context on ONE stream are ordered and may be queued
 * back to back; the same context used under two streams gets two counter sets.  A launch that faults leaves its
 * counters undefined: destroy the context. */
int ibs_gamma_scan_argmax_f64(ibs_ctx* ctx, int32_t n_lines, int32_t n_theta0, int32_t N, double h,
                              const double* bmag, const double* gradpar, const double* cvdrift, const double* cvdrift0,
                              const double* gds2, const double* gds21, const double* gds22, int64_t ld,
                              const double* dPdrho, const double* theta0, int32_t n_surf, double* gam, double* lam,
                              double* pack, int32_t* info);

/* One (field line, theta0) PAIR per point: growth rates of n_pts lines, line i at its own theta0[i].
 * Replaces: the final solve of ball_scan.py:322-339 (one more vmec_fieldlines + gamma_ball_full at the refined
 * (alpha*, theta0*) of every surface) for all surfaces of a rank -- or of all equilibria of an optimizer step -- at once.
 * Arrays as in ibs_gamma_scan_f64 with n_lines = n_pts ([n_pts][ld]); theta0[n_pts]; outputs [n_pts] (X, dX: [n_pts][N]). */
int ibs_gamma_points_f64(ibs_ctx* ctx, int32_t n_pts, int32_t N, double h, const double* bmag, const double* gradpar,
                         const double* cvdrift, const double* cvdrift0, const double* gds2, const double* gds21,
                         const double* gds22, int64_t ld, const double* dPdrho, const double* theta0, double* gam,
                         double* lam, double* X, double* dX, double* dgam_dtheta0, int32_t* info, int32_t mem);

/* Start points of the refinement from the coarse scan's per-surface maxima, on the device (device pointers only).
 * Replaces: ball_scan.py:279-295 -- the first maximum of the surface's (alpha, theta0) table gives x0 = (alpha_scan[i],
 * theta0_scan[j]) and sigma0 = 1.3 |gam| + 0.05 (:283-295); an all-zero table starts from (0, 0) with sigma0 = 0.05 (:279-282).
 *   pack[n_surf][2] = (max, first row-major index as a double): the output of ibs_gamma_scan_argmax_f64 /
 *   ibs_surface_argmax_pack_f64;  alpha[n_alpha], theta0[n_theta0]: the scan grids (ball_scan.py:225-226);
 *   start[n_surf][2] = (alpha, theta0): the `start` input of ibs_refine_f64;  sigma0[n_surf] (optional; the deterministic
 *   solver ignores it);  *n_bad (device int, NOT cleared here) += the number of surfaces whose maximum is not finite (a
 *   flagged or NaN table: ball_scan.py would have crashed in eigs; their start is (0, 0)). */
int ibs_scan_starts_f64(ibs_ctx* ctx, int32_t n_surf, int32_t n_alpha, int32_t n_theta0, const double* alpha,
                        const double* theta0, const double* pack, double* start, double* sigma0, int32_t* n_bad);

/* Objective and Hellmann-Feynman ("adjoint") gradient at n_pts points (alpha, theta0).
 * Replaces: utils.py:1632-1728 obj_w_grad, given the geometry of the three field lines
 * (alpha - del_alpha/2, alpha, alpha + del_alpha/2) that utils.py:1641-1646 obtains from vmec_fieldlines.
 *   geo [n_pts][3][8][ld]: per line bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift
 *   (dPdrho of each line is formed on the device, utils.py:1657/1691/1703);  theta0[n_pts];
 *   val[n_pts] = -gam;  jac[n_pts][2] = (-dgam/dalpha, -dgam/dtheta0)   (utils.py:1728). */
int ibs_obj_w_grad_f64(ibs_ctx* ctx, int32_t n_pts, int32_t N, double h, const double* geo, int64_t ld,
                       const double* theta0, double del_alpha, double* val, double* jac, int32_t* info,
                       int32_t mem);

/* Hellmann-Feynman derivative of gam with respect to any parameter p, given the eigenfunction and the tangent
 * coefficient arrays (d g/dp, d c/dp, d f/dp) on the grid:
 *   jac = [ S(c_p X^2) - S(g_p dX^2) - gam S(f_p X^2) ] / S(f X^2),   S = composite Simpson, unit spacing.
 * Replaces: utils.py:1676-1680 (p = theta0) and utils.py:1721-1725 (p = alpha) for caller-built tangents; the
 * fused entry point above builds both tangents itself.  All arrays [n_sys][ld], gam and jac [n_sys]; N odd. */
int ibs_hf_grad_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, const double* X, const double* dX, const double* f,
                    const double* g_p, const double* c_p, const double* f_p, int64_t ld, const double* gam,
                    double* jac, int32_t mem);

/* Field-line geometry on the device: the eight arrays of ball_scan.py:251-261 for
 * n_lines field lines (surface index, alpha) on the theta_PEST grid theta[N].
 * Replaces: the per-line arithmetic of vmec_fieldlines, utils.py:359-720 (theta_pest -> theta_vmec secant
 * solve :391-416, Fourier synthesis :420-468, metric algebra :474-720); the radial splines of
 * utils.py:37-158 / :311-357 stay on the host and provide, per surface,
 *   tab_mn  [n_surf][6][mnmax]      rmnc zmns lmns d_rmnc_d_s d_zmns_d_s d_lmns_d_s
 *   tab_nyq [n_surf][7][mnmax_nyq]  gmnc bmnc d_bmnc_d_s bsupvmnc bsubsmns bsubumnc bsubvmnc
 *   scal    [n_surf][6]             s iota d_iota_d_s d_pressure_d_s phiedge Aminor_p
 * and the mode numbers xm, xn [mnmax], xm_nyq, xn_nyq [mnmax_nyq] (xn includes nfp).
 *   geo [8][n_lines][ld]: bmag gradpar_theta_pest cvdrift cvdrift0 gds2 gds21 gds22 gbdrift
 *   dPdrho[n_lines] (optional): -0.5 mean((cvdrift - gbdrift) bmag^2), ball_scan.py:262.
 *   rows_mn [nrows_mn][2], rows_nyq [nrows_nyq][2] (optional, nrows = 0 to omit): {first mode, count} of each
 *   run of modes with equal m and n advancing by the common step dn_mn / dn_nyq (= nfp in VMEC's own
 *   ordering); enables the rotation-recurrence kernels (no sincos per mode).  Every row must lie inside its mode list
 *   (else IBS_ERR_ARG).  The row kernels hold at most 64 pair indices about a row's centre -- the mode with n = 0 if the
 *   row has one, else its first mode: rows of up to 129 modes n = -64 dn .. 64 dn, or up to 65 modes otherwise; tables
 *   with a longer row in rows_mn are valid and run on the one-sincos-per-mode kernel (split such rows to stay on the fast
 *   path: ibs_amd.geometry.mode_rows does).  Rows of one list must not overlap (IBS_ERR_ARG).  Device-resident rows are copied back
 *   and checked once per table set, the verdict kept under the rows' ADDRESSES and sizes (four sets): a caller that reuses that device
 *   memory for other rows calls ibs_set_option(ctx, "forget_rows", 1) first.
 * The first seven planes of geo and dPdrho are exactly the inputs of ibs_gamma_scan_f64. */
int ibs_fieldline_geometry_f64(ibs_ctx* ctx, int32_t n_surf, int32_t mnmax, int32_t mnmax_nyq, const double* xm,
                               const double* xn, const double* xm_nyq, const double* xn_nyq, const double* tab_mn,
                               const double* tab_nyq, const double* scal, int32_t n_lines, const int32_t* line_surf,
                               const double* line_alpha, int32_t N, const double* theta, int64_t ld, double* geo,
                               double* dPdrho, int32_t nrows_mn, const int32_t* rows_mn, int32_t nrows_nyq,
                               const int32_t* rows_nyq, double dn_mn, double dn_nyq, int32_t mem);

/* Host part of the geometry producer: the per-surface Fourier coefficient vectors of n_eq equilibria at n_s surfaces.
 * Replaces: vmec_splines (utils.py:58-119: one InterpolatedUnivariateSpline per mode and array) + their evaluation at the
 * surface (utils.py:311-357) for all equilibria of an optimizer step (sims_runner_NCSX.py:151-276) at once.  Cubic
 * interpolating splines are linear in the data and all modes share the radial meshes, so the caller supplies four weight
 * matrices [n_s][ns] -- value / derivative on VMEC's full mesh, value / derivative on the half mesh with a zero first
 * column (half-mesh data live in columns 1..ns-1) -- built once per (ns, surfaces) by splining the identity.
 *   tabs[n_eq][9]: pointers to the (modes, ns) row-major wout arrays rmnc zmns lmns (mnmax rows) gmnc bmnc bsupvmnc
 *   bsubsmns bsubumnc bsubvmnc (mnmax_nyq rows);  tab_mn [n_eq * n_s][6][mnmax], tab_nyq [n_eq * n_s][7][mnmax_nyq]: the
 *   inputs of ibs_fieldline_geometry_f64 (surface index = i_eq * n_s + i_s).  Host pointers only; n_threads <= 0: up to 16.
 * No GPU is involved; results do not depend on the thread count. */
int ibs_surface_tables_f64(int32_t n_eq, int32_t ns, int32_t n_s, int32_t mnmax, int32_t mnmax_nyq,
                           const double* const* tabs, const double* w_full, const double* w_full_d,
                           const double* w_half, const double* w_half_d, double* tab_mn, double* tab_nyq,
                           int32_t n_threads);

/* Refinement of the per-surface maximum: maximise gam over (alpha, theta0) in
 * [0, pi] x [0, pi/2] from n_pts start points at once, entirely on the device.
 * Replaces: ball_scan.py:305-314 (scipy.optimize.minimize(obj_w_grad, x0, jac=True, bounds, ftol, gtol, maxiter), i.e.
 * scipy's L-BFGS-B: restated for the two unknowns in csrc/ibs_lbfgsb2.hpp, one state machine per point) together with
 * the evaluations it drives (utils.py:1632-1728 on the three field lines of utils.py:1641-1646, produced by the
 * geometry kernel).  A round = the geometry kernel (3 lines per point still running) + one fused kernel per point
 * (objective, Hellmann-Feynman gradient, optimizer step; the eigen-solve is warm-started from the point's previous
 * evaluation; the block that finishes last re-packs the batch).  Nothing returns to the host between rounds: the device
 * posts the count of unfinished points per round into pinned memory, and the host enqueues rounds two ahead of the last
 * count it has seen (csrc/ibs_refine.hpp).
 *   surface tables, mode tables, row tables: as for ibs_fieldline_geometry_f64;  pt_surf[n_pts] surface of each point;
 *   start[n_pts][2] = (alpha, theta0);  theta[N] uniform theta_PEST grid;  del_alpha (utils.py:1639: 0.004);
 *   maxiter, ftol, gtol: ball_scan.py:312-313 (30, 5e-11, 2e-8); like scipy's driver the limit is tested after each
 *   completed iteration, so maxiter = 0 performs one.
 *   x_opt[n_pts][2], f_opt[n_pts] = -gam at x_opt (utils.py:1728 sign), n_evals[n_pts] (optional).  x_opt lies in the box up
 *   to the rounding of x + stp d (L-BFGS-B projects directions, not points).
 *   n_pts = 0 (a rank without surfaces) is a no-op: the per-point pointers may be null, 0 rounds are returned.
 * `mem` applies to every pointer.  IBS_MEM_HOST: synchronous.  IBS_MEM_DEVICE: the host returns once every round's count
 * has been read, but the kernel that writes x_opt / f_opt / n_evals may still be running: the outputs are STREAM-ORDERED
 * on the context's stream like those of every other device-pointer call (read them from that stream, or after
 * ibs_synchronize()).  A round that does not report within 120 s fails with IBS_ERR_HIP.
 * Returns the number of rounds (>= 0) or an error (< 0). */
int ibs_refine_f64(ibs_ctx* ctx, int32_t n_surf, int32_t mnmax, int32_t mnmax_nyq, const double* xm, const double* xn,
                   const double* xm_nyq, const double* xn_nyq, const double* tab_mn, const double* tab_nyq,
                   const double* scal, int32_t nrows_mn, const int32_t* rows_mn, int32_t nrows_nyq,
                   const int32_t* rows_nyq, double dn_mn, double dn_nyq, int32_t n_pts, const int32_t* pt_surf,
                   const double* start, int32_t N, const double* theta, double del_alpha, int32_t maxiter,
                   double ftol, double gtol, double* x_opt, double* f_opt, int32_t* n_evals, int32_t mem);

/* Statistics of the last ibs_refine_f64 call of this context (diagnostic; no reference counterpart):
 * out4 = {objective evaluations, forward sweeps of their eigen-solves, rounds needed, rounds enqueued}. */
int ibs_refine_stats(ibs_ctx* ctx, int64_t* out4);

/* Diagnostic (no reference counterpart): the solver / geometry kernel most recently launched by the calling thread -- its
 * name as rocprofv3 prints it ("ibs::k_gamma_scan<double, 8>"; the library picks lanes per system, theta0 chaining and the
 * geometry form from the batch size) and the launch dimensions (blocks, threads per block; waves = blocks * threads / 64).
 * bench.py uses it to look a leg's kernel up in the committed PMC passes by its exact name and batch size. */
int ibs_last_launch(char* name, int32_t len, int64_t* blocks, int32_t* threads);

/* Number of eigenvalues of (T, F) strictly above shift[i] for each system (Sturm sequence).
 * Replaces: tests/shifted-circle-s-alpha/bishop_ball_s-alpha.py:20-115 check_ball (isunstable <=> count(0) > 0).
 * Up to N = 2050 this is ONE prefix-product sweep per system (the bandwidth kernel): exact for a pencil perturbed by ~N eps ||A||
 * on smooth coefficients, but by up to ~N^2 eps ||A|| on iid-random ones -- within that distance of an eigenvalue the count can be
 * off by one.  Option "sturm_form" = 2 selects the DIVISION-form kernel instead (lanes as systems, the rows through an LDS transpose,
 * every system read from its own 128-byte line boundary: exact for a pencil a few ulp away, any N, 4.8-5.5 TB/s in batches of 65,536
 * systems or more against the sweep's 6 TB/s at N_zeta <= 512; Python: sturm_count(..., exact=True)); it is what grids beyond 2050
 * points get (batches under 64 systems: one wave per system) and, where it is also the faster one, what big batches get by default
 * (16 rows per lane = N in 963 .. 1026 from 65,536 systems, 32 rows = 1987 .. 2050 from 32,768: the sweep runs at 4.2 / 2.8 TB/s
 * there).  Even N is accepted here. */
int ibs_sturm_count_f64(ibs_ctx* ctx, int64_t n_sys, int32_t N, double h, const double* g, const double* c,
                        const double* f, int64_t ld, const double* shift, int32_t* count, int32_t mem);

/* Per-surface reduction of a scan: first (row-major) index of the maximum and its value, the rule of
 * ball_scan.py:279-295.  gam is [n_surf][n_per_surf]. */
int ibs_surface_argmax_f64(ibs_ctx* ctx, int32_t n_surf, int32_t n_per_surf, const double* gam,
                           int32_t* idx, double* val, int32_t mem);

/* Same reduction on device pointers, written as pack[n_surf][2] = (value, index as double): the one buffer the
 * per-surface all-gather sends (replaces the three comm_lead.Gather of ball_scan.py:345-347). */
int ibs_surface_argmax_pack_f64(ibs_ctx* ctx, int32_t n_surf, int32_t n_per_surf, const double* gam, double* pack);

#ifdef __cplusplus
}
#endif
#endif /* IBS_H */
