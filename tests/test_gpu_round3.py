"""GPU: the parity loose ends of round 2's review -- the bench's own workload against the oracle, BASELINE configs[3] with the
REFINED maxima the reference differentiates (ball_scan.py:305-347 -> sims_runner_NCSX.py:254-261), and the COBRAVMEC profile
check of the north star (tests/comparn_w_COBRAVMEC/cobra_compare_op.py:27-33) as a computed comparison.  Everything goes
through the C ABI; the oracle side is numpy geometry + C / numpy eigen-solver + scipy's own L-BFGS-B."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ctx():
    import ibs_amd
    c = ibs_amd.Context(0)
    yield c
    c.close()


def test_bench_workload_against_oracle(ctx):
    """bench.py's headline workload (configs[1] shape: 16 surfaces x 8 alpha x 8 theta0, N = 513) through the call the bench
    times -- ScanPlan.scan_argmax(), one fused launch -- against the C oracle: every growth rate to 1e-8, every per-surface
    argmax exact.  (bench.py itself sets parity_ok from the same comparison and exits non-zero when it fails.)"""
    import torch
    import ibs_amd
    from oracle import c_oracle as co
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda:0")
    h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
    plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF, n_pack=3)
    for slot in (0, 1, 2, 0):
        plan.scan_argmax(slot)
    torch.cuda.synchronize()
    gam = plan.gam.cpu().numpy()
    gam_c, lam_c, _ = co.gamma_scan(h, *[base[:, k, :] for k in range(7)], dP, theta0)
    assert int(((plan.info.cpu().numpy() >> 16) != 0).sum()) == 0
    assert np.abs(gam - gam_c).max() < 1e-8, np.abs(gam - gam_c).max()
    tab_c = gam_c.reshape(bench.N_SURF, -1)
    for slot in range(3):
        pk = plan.packs[slot].cpu().numpy()
        assert np.array_equal(pk[:, 1].astype(np.int64), tab_c.argmax(axis=1))           # first maximum (ball_scan.py:283-288)
        assert np.abs(pk[:, 0] - tab_c.max(axis=1)).max() < 1e-8


def test_config4_refined_maxima_and_gradient(ctx):
    """BASELINE configs[3] with the quantity the reference differentiates: 73 equilibria x 5 surfaces, coarse 24 x 15 scan ->
    argmax -> L-BFGS-B on the device for all 365 starts in ONE ibs_refine_f64 call -> final solve (ball_scan.py:305-339), then
    objective and 72-gradient from the REFINED gam (sims_runner_NCSX.py:254-261).  A random sample of 16 (equilibrium,
    surface) pairs is re-done with the oracle on every link (tests/helpers.py: numpy geometry, oracle objective, scipy's
    L-BFGS-B): refined gam within 1e-8."""
    import torch
    import ibs_amd
    from tests.helpers import oracle_surface_pipeline
    from tests.test_gpu_configs import emulated_equilibria
    dev = torch.device("cuda:0")
    wout0 = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    wouts, steps, x0 = emulated_equilibria(wout0)
    n_eq, ns, na, nt0 = len(wouts), 5, 24, 15
    svals = np.linspace(0.5, 0.95, ns)                                   # ball_scan.py:197
    th = ibs_amd.theta_grid_for(11, 11)                                  # 969 points
    big = ibs_amd.SurfaceTables.concat([ibs_amd.SurfaceTables.from_wout(w, svals) for w in wouts])
    scan = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), nalpha=na, ntheta0=nt0, tables=big, device=dev)
    # (tables.s repeats per equilibrium: the scan's nearest-s lookup must not be used; surfaces are addressed by index)
    scan.own = list(range(n_eq * ns))
    surf_idx = np.arange(n_eq * ns)
    tabs = scan_coarse_by_index(ctx, big, surf_idx, scan, dev)
    starts = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in tabs])
    xo, fo, ne, rounds = ctx.refine(big, surf_idx, starts, th, scan.del_alpha, device=dev)
    assert ne.min() >= 1 and ne.max() <= 2 + 42 * 30 and rounds == ne.max()
    gam = final_solve_by_index(ctx, big, surf_idx, xo, scan, dev)
    coarse_max = tabs.reshape(n_eq * ns, -1).max(axis=1)
    assert np.all(gam >= coarse_max - 1e-9)                               # L-BFGS-B never ends below its start
    assert np.abs(gam + fo).max() < 1e-9                                  # final solve = the optimizer's last accepted value
    # --- 16 random (equilibrium, surface) pairs with the oracle on every link
    rng = np.random.default_rng(11)
    worst = 0.0
    for k in rng.choice(n_eq * ns, size=16, replace=False):
        q, js = divmod(int(k), ns)
        ref = oracle_surface_pipeline(wouts[q], float(svals[js]), th, na, nt0, scan.del_alpha)
        assert np.abs(ref["table"] - tabs[k]).max() < 1e-8
        assert np.allclose(ref["start"], starts[k], rtol=0, atol=0)
        worst = max(worst, abs(ref["gam"] - gam[k]))
        assert abs(ref["gam"] - gam[k]) < 1e-8, (k, ref["gam"], gam[k], ref["x_opt"], xo[k], ref["nfev"], ne[k])
    # --- objective and forward-difference gradient from the refined maxima, sims_runner_NCSX.py:249-261 written out
    gmax = gam.reshape(n_eq, ns)
    f_other = 0.8 + 0.01 * np.arange(n_eq)
    thresh, prefac = -2.0e-4, 50.0                                        # sims_runner_NCSX.py:56-57
    f0_arr = np.zeros(n_eq); df0 = np.zeros(n_eq - 1)
    for i in range(n_eq):
        f0_arr[i] = f_other[i] + prefac * np.sum(np.maximum(gmax[i] - thresh, 0.0))      # :254-256
        if i > 0:
            df0[i - 1] = (f0_arr[i] - f0_arr[0]) / steps[i] * 0.5 * 1 / np.sqrt(f0_arr[0])   # :258-261
    f_gpu = ibs_amd.ballooning_objective(f_other, gmax, gamma_thresh=thresh, prefac=prefac)
    d_gpu = ibs_amd.dof_fd_gradient(f_gpu, ibs_amd.dof_steps(x0, (np.abs(np.concatenate([[0.0], x0])) <= 1.0e-2).astype(int)))
    assert np.abs(f_gpu - f0_arr).max() < 1e-12 and np.abs(d_gpu - df0).max() < 1e-9 * max(1.0, np.abs(df0).max())
    print("config 3 refined: %d maxima, evaluations %d..%d, worst |gam - oracle| of the sample %.2e" % (len(gam), ne.min(), ne.max(), worst))


def scan_coarse_by_index(ctx, tables, surf_idx, scan, dev):
    """coarse (alpha, theta0) tables of the surfaces tables.s[surf_idx] (BallooningScan.coarse with explicit indices)"""
    import torch
    na = len(scan.alpha_scan)
    surf = np.repeat(surf_idx, na)
    r = ctx.fieldline_geometry(tables, surf, np.tile(scan.alpha_scan, len(surf_idx)), scan.theta, device=dev)
    t0 = torch.from_numpy(scan.theta0_scan).to(dev)
    out = ctx.gamma_scan(scan.h, *[r["geo"][k] for k in range(7)], r["dPdrho"], t0, want_info=True)
    assert int(((out["info"] >> 16) != 0).sum().item()) == 0
    return out["gam"].cpu().numpy().reshape(len(surf_idx), na, len(scan.theta0_scan))


def final_solve_by_index(ctx, tables, surf_idx, xo, scan, dev):
    """gam at the refined points (ball_scan.py:322-339: one more geometry + solve at the optimum)"""
    import torch
    r = ctx.fieldline_geometry(tables, surf_idx, np.ascontiguousarray(xo[:, 0]), scan.theta, device=dev)
    n, N = len(surf_idx), len(scan.theta)
    geo = r["geo"].reshape(8, n, 1, N).expand(8, n, 3, N).permute(1, 2, 0, 3).contiguous()
    val, _ = ctx.obj_w_grad(scan.h, geo, torch.from_numpy(np.ascontiguousarray(xo[:, 1])).to(dev), scan.del_alpha)
    return -val.cpu().numpy()


def test_J1_cobravmec_profile(ctx):
    """The COBRAVMEC comparison of the north star as a computed check (tests/comparn_w_COBRAVMEC/cobra_compare_op.py:27-33):
    gam_max(s) on s = linspace(0.01, 0.995, 48) of the shipped NCSX_op equilibrium at the reference's defaults (24 x 15 coarse
    grid, N = 969, L-BFGS-B refinement, final solve) against the 48 values the reference stores (gamma_max_op.npy).  SURVEY 4:
    the stored run's settings are not recorded, so the pin is sign + magnitude band (every gam_max < 0: the optimised
    configuration is stable everywhere; ratio to the stored value within [0.3, 3]); six of the 48 surfaces are re-done with
    the oracle on every link to 1e-8."""
    import torch
    import ibs_amd
    from tests.helpers import oracle_surface_pipeline
    dev = torch.device("cuda:0")
    g7 = np.load(os.path.join(G, "G7_cobra_pins.npz"))
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    svals = np.linspace(0.01, 0.995, 48)                                  # cobra_compare_op.py:28
    assert np.allclose(svals, g7["s_op"])
    th = ibs_amd.theta_grid_for(11, 11)
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    scan = ibs_amd.BallooningScan(ctx, None, th, svals, tables=tabs, device=dev)
    t0, al, gam = scan.run()
    stored = g7["gamma_max_op"]
    assert np.all(np.isfinite(gam)) and np.all(gam < 0), gam               # stable on every surface, like the stored profile
    ratio = gam / stored
    assert ratio.min() >= 0.3 and ratio.max() <= 3.0, (ratio.min(), ratio.max(), svals[np.argmin(ratio)], svals[np.argmax(ratio)])
    worst = 0.0
    for k in (0, 9, 20, 31, 40, 47):                                       # incl. the two end surfaces s = 0.01, 0.995
        ref = oracle_surface_pipeline(wout, float(svals[k]), th)
        worst = max(worst, abs(ref["gam"] - gam[k]))
        assert abs(ref["gam"] - gam[k]) < 1e-8, (k, svals[k], ref["gam"], gam[k], ref["x_opt"], (al[k], t0[k]))
    print("J1: gam_max/stored in [%.2f, %.2f]; worst |gam - oracle| on 6 surfaces %.2e" % (ratio.min(), ratio.max(), worst))


def test_geometry_row_kernel_variants(ctx):
    """k_geo_rows on row structures other than VMEC's own: rows split into pieces of <= 12 modes (21 + 29 rows: the MAXR = 24
    instantiation, rows that repeat an m, rows centred away from n = 0 in the (P, Q) pass) and pieces of <= 5 modes (more rows
    than the register-resident (P, Q) pass holds: the one-sincos-per-mode kernel); every form, against the default rows."""
    import torch
    import ibs_amd
    from ibs_amd.geometry import mode_rows
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    dev = torch.device("cuda:0")
    tabs = ibs_amd.SurfaceTables.from_wout(wout, [0.35, 0.8])
    surf = [0, 0, 1, 1, 1, 0]; al = [0.0, 1.3, 0.4, 2.0, np.pi, 2.9]
    for N in (969, 257):
        th = ibs_amd.theta_grid(N)
        ref = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)["geo"].cpu().numpy()
        scale = np.abs(ref).max(axis=2, keepdims=True)
        for max_len in (12, 5):
            t2 = ibs_amd.SurfaceTables.from_wout(wout, [0.35, 0.8])
            t2.rows_mn, t2.dn_mn = mode_rows(t2.xm, t2.xn, max_len=max_len)
            t2.rows_nyq, t2.dn_nyq = mode_rows(t2.xm_nyq, t2.xn_nyq, max_len=max_len)
            assert (len(t2.rows_mn) > 12) and (len(t2.rows_mn) <= 24) == (max_len == 12)
            for lpp in (0, 1, -2, 4):
                ctx.set_option("geo_lpp", lpp)
                r = ctx.fieldline_geometry(t2, surf, al, th, device=dev)
                assert (np.abs(r["geo"].cpu().numpy() - ref) / scale).max() < 1e-11, (N, max_len, lpp)
            ctx.set_option("geo_lpp", None)


def _subset_wout(wout, keep_mn, keep_nyq):
    """the same equilibrium file with a subset of its Fourier modes (a data manipulation for shape coverage: both sides
    evaluate the same formulas on it)"""
    w = dict(wout)
    for k in ("rmnc", "zmns", "lmns"):
        w[k] = wout[k][keep_mn]
    for k in ("gmnc", "bmnc", "bsupvmnc", "bsubsmns", "bsubumnc", "bsubvmnc"):
        w[k] = wout[k][keep_nyq]
    w["xm"], w["xn"] = wout["xm"][keep_mn], wout["xn"][keep_mn]
    w["xm_nyq"], w["xn_nyq"] = wout["xm_nyq"][keep_nyq], wout["xn_nyq"][keep_nyq]
    return w


@pytest.mark.parametrize("shape", ["axisymmetric", "few_modes", "one_sided_n"])
def test_geometry_on_other_mode_sets(ctx, shape):
    """F1 on mode tables that are NOT NCSX's 12 x 23 / 14 x 29 rectangles, against the numpy oracle (utils.py:359-720 restated):
    a tokamak-like table (n = 0 only: every row is one mode -- the D3D equilibria of configs[1] look like this), a table with
    a handful of modes (rows of different lengths, some m missing) and one whose rows are not symmetric in n (the (P, Q) pass's
    centred-row logic); one line and many lines, N = 67 .. 2049, every lanes-per-point form."""
    import torch
    import ibs_amd
    from oracle import geometry_oracle as go
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    xm, xn, xmq, xnq = wout["xm"], wout["xn"], wout["xm_nyq"], wout["xn_nyq"]
    nfp = int(np.min(np.abs(xn[xn != 0])))
    if shape == "axisymmetric":
        kmn, knq = xn == 0, xnq == 0
    elif shape == "few_modes":
        kmn = ((xm <= 3) & (np.abs(xn) <= 2 * nfp) & (xm != 2)) | ((xm == 7) & (xn == 0)) | ((xm == 0) & (xn == 0))
        knq = ((xmq <= 4) & (np.abs(xnq) <= nfp)) | ((xmq == 9) & (xnq == 3 * nfp))
    else:
        kmn = (xn >= 0) | ((xm == 1) & (xn >= -2 * nfp))
        knq = (xnq <= 0) | ((xmq == 2) & (xnq <= 4 * nfp))
    w = _subset_wout(wout, kmn, knq)
    svals = np.array([0.3, 0.77])
    tabs = ibs_amd.SurfaceTables.from_wout(w, svals)
    otab = go.surface_tables_from_wout(w, svals)
    dev = torch.device("cuda:0")
    for N, surf, al in ((67, [1], [0.7]), (969, [0, 1, 1], [0.0, 2.2, np.pi]), (2049, [0, 0, 1, 0], [0.3, 1.1, 2.9, 3.0])):
        th = ibs_amd.theta_grid(N)
        ref = np.stack([go.fieldline_geometry(otab, s, np.array([a]), th)[0] for s, a in zip(surf, al)])     # (lines, 8, N)
        scale = np.abs(ref).max(axis=2, keepdims=True)
        for lpp in (None, 1, -2, 2, 8):
            ctx.set_option("geo_lpp", lpp)
            r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
            got = r["geo"].cpu().numpy().transpose(1, 0, 2)             # planes [8][lines][N] -> (lines, 8, N)
            err = (np.abs(got - ref) / scale).max()
            assert err < 1e-10, (shape, N, lpp, err)
            dP = -0.5 * np.mean((ref[:, 2] - ref[:, 7]) * ref[:, 0] ** 2, axis=1)
            assert np.abs(r["dPdrho"].cpu().numpy() - dP).max() < 1e-12 * max(1.0, np.abs(dP).max())
        ctx.set_option("geo_lpp", None)


def test_refinement_edge_batches(ctx):
    """ibs_refine_f64 on the batch shapes the scan driver never produces: no point, one point, every point on ONE surface,
    starts on the corners and edges of the box (the reference's bounds, ball_scan.py:311), maxiter = 0 (one iteration, like
    scipy's driver), host-pointer and device-pointer calls.  Checks: same optimum whatever the batch a point is refined in (1e-9 in gam:
    the batch size picks the geometry form, and the optimizer's path reacts to its rounding), every optimum inside the box (to rounding),
    gam(x_opt) recomputed by the scan kernel on the oracle's geometry to 1e-8, never below the start's value."""
    import torch
    import ibs_amd
    from oracle import geometry_oracle as go
    from oracle import ballooning_oracle as bo
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    svals = np.array([0.4, 0.9])
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    otab = go.surface_tables_from_wout(wout, svals)
    N = 513; th = ibs_amd.theta_grid(N); dev = torch.device("cuda:0")
    # no point
    xo, fo, ne, rounds = ctx.refine(tabs, np.zeros(0, dtype=np.int32), np.zeros((0, 2)), th, device=dev)
    assert xo.shape == (0, 2) and fo.shape == (0,) and rounds == 0
    xo, fo, ne, rounds = ctx.refine(tabs, np.zeros(0, dtype=np.int32), np.zeros((0, 2)), th)
    assert xo.shape == (0, 2) and rounds == 0
    starts = np.array([[0.0, 0.0], [np.pi, 0.5 * np.pi], [0.0, 0.5 * np.pi], [np.pi, 0.0], [1.7, 0.0], [0.0, 0.6], [2.4, 0.9],
                       [0.9, 0.5 * np.pi], [2.0, 1.2]])
    ps = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1], dtype=np.int32)

    def gam_at(s_idx, x):
        line = go.fieldline_geometry(otab, int(s_idx), np.array([x[0]]), th)[0]
        dP = -0.5 * np.mean((line[2] - line[7]) * line[0] ** 2)
        cv, gd = bo.fold_theta0(x[1], line[2], line[3], line[4], line[5], line[6])
        return bo.gamma_ball_full(dP, th, line[0], line[1], cv, gd)[0]

    xb, fb, nb, rb = ctx.refine(tabs, ps, starts, th, device=dev)                 # all nine on one surface, device pointers
    xh, fh, nh, rh = ctx.refine(tabs, ps, starts, th)                             # host pointers
    assert np.abs(fb - fh).max() < 1e-9 and rb >= 1
    # (inside the box up to the rounding of x = x_k + stp d in the line search, lnsrlb: L-BFGS-B projects the search direction, not x)
    eb = 1e-15
    assert (xb[:, 0] >= -eb).all() and (xb[:, 0] <= np.pi + eb).all() and (xb[:, 1] >= -eb).all() and (xb[:, 1] <= 0.5 * np.pi + eb).all()
    for k in range(len(ps)):
        x1, f1, n1, r1 = ctx.refine(tabs, ps[k:k + 1], starts[k:k + 1], th, device=dev)       # the same point alone
        assert abs(f1[0] - fb[k]) < 1e-9, (k, f1[0], fb[k])
        assert abs(gam_at(ps[k], xb[k]) + fb[k]) < 1e-8                                           # f_opt = -gam(x_opt)
        assert -fb[k] >= gam_at(ps[k], starts[k]) - 1e-12                                         # a maximiser never ends below its start
    # maxiter = 0: scipy's driver tests the limit after each COMPLETED iteration, so it performs exactly one (as maxiter = 1;
    # tests/test_lbfgsb2.py::test_iteration_limit_like_scipy pins that on the CPU)
    x0, f0, n0, r0 = ctx.refine(tabs, ps[:3], starts[:3], th, maxiter=0, device=dev)
    x1_, f1_, n1_, r1_ = ctx.refine(tabs, ps[:3], starts[:3], th, maxiter=1, device=dev)
    assert np.array_equal(x0, x1_) and np.array_equal(f0, f1_) and np.array_equal(n0, n1_) and (n0 <= 22).all()
    for k in range(3):
        assert abs(gam_at(ps[k], x0[k]) + f0[k]) < 1e-8 and -f0[k] >= gam_at(ps[k], starts[k]) - 1e-12
    # the alpha-tangent read from global memory in the sums (the form batches larger than the chip take) against staged in LDS
    ctx.set_option("refine_tangent", 0)
    xt, ft, nt, rt = ctx.refine(tabs, ps, starts, th, device=dev)
    ctx.set_option("refine_tangent", None)
    assert np.abs(ft - fb).max() < 1e-9
    # mixed surfaces, unsorted
    psm = np.array([1, 0, 1, 0], dtype=np.int32); stm = starts[[6, 6, 8, 8]]
    xm_, fm, nm, rm = ctx.refine(tabs, psm, stm, th, device=dev)
    assert abs(fm[0] - fb[6]) < 1e-9 and abs(fm[2] - fb[8]) < 1e-9
    for k in (1, 3):
        assert abs(gam_at(0, xm_[k]) + fm[k]) < 1e-8


def test_refinement_is_bitwise_repeatable(ctx):
    """The rounds are plain launches enqueued ahead of the device, the batch is re-packed by whichever block finishes last and the
    geometry form follows the batch size: none of it may leak into the results.  Same call, same bits (x_opt, f_opt, evaluation
    counts), for a batch smaller and one larger than the number of CUs."""
    import torch
    import ibs_amd
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    dev = torch.device("cuda:0")
    N = 513; th = ibs_amd.theta_grid(N)
    svals = np.linspace(0.3, 0.9, 4)
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    rng = np.random.default_rng(5)
    for n in (7, 300):
        ps = rng.integers(0, len(svals), n).astype(np.int32)
        st = np.stack([rng.uniform(0.2, 3.0, n), rng.uniform(0.05, 1.5, n)], axis=1)
        runs = [ctx.refine(tabs, ps, st, th, device=dev) for _ in range(3)]
        for r in runs[1:]:
            assert np.array_equal(runs[0][0], r[0]) and np.array_equal(runs[0][1], r[1]) and np.array_equal(runs[0][2], r[2]) and runs[0][3] == r[3]
        assert np.isfinite(runs[0][1]).all()
