"""GPU, round 4: the product driver runs the fused path end to end on the device -- the new entry points
(one line per point, start points on the device, device-pointer refinement), BallooningScan.device_rows against the
host-driven steps it replaces, and BASELINE configs[3] as ONE AdjointStep.run() against the oracle on every link."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ctx():
    import ibs_amd
    c = ibs_amd.Context(0)
    yield c
    c.close()


def test_gamma_points_one_theta0_per_line(ctx):
    """ibs_gamma_points_f64 (the final solve of ball_scan.py:322-339 for many surfaces at once): line i at ITS OWN theta0[i]
    equals the scan of that line with a one-entry theta0 grid bit for bit, matches the C oracle to 1e-8, host and device
    pointers agree, eigenfunctions and the theta0 derivative come out as in the scan."""
    import torch
    from oracle import c_oracle as co
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = g3["geo_513"][:40]                                    # (40, 8, 513)
    n, N = geo.shape[0], geo.shape[2]
    h = 8 * np.pi / (N - 1)
    rng = np.random.default_rng(5)
    t0 = rng.uniform(0, np.pi / 2, n)
    dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)
    arrs = [np.ascontiguousarray(geo[:, k]) for k in range(7)]
    host = ctx.gamma_points(h, *arrs, dP, t0, want_X=True, want_dtheta0=True, want_info=True)
    assert host["nbad"] == 0
    dev = torch.device("cuda:0")
    d = ctx.gamma_points(h, *[torch.from_numpy(a).to(dev) for a in arrs], torch.from_numpy(dP).to(dev),
                         torch.from_numpy(t0).to(dev), want_X=True, want_dtheta0=True, want_info=True)
    for k in ("gam", "lam", "X", "dX", "dgam_dtheta0"):
        assert np.array_equal(d[k].cpu().numpy(), host[k]), k
    for i in range(n):
        one = ctx.gamma_scan(h, *[a[i:i + 1] for a in arrs], dP[i:i + 1], t0[i:i + 1], want_X=True, want_dtheta0=True)
        assert one["gam"][0, 0] == host["gam"][i] and one["lam"][0, 0] == host["lam"][i]
        assert np.array_equal(one["X"][0, 0], host["X"][i]) and one["dgam_dtheta0"][0, 0] == host["dgam_dtheta0"][i]
        ref, _, _ = co.gamma_scan(h, *[a[i:i + 1] for a in arrs], dP[i:i + 1], t0[i:i + 1])
        assert abs(ref[0, 0] - host["gam"][i]) < 1e-8
    # an invalid line is flagged and does not disturb its neighbours
    bad = [a.copy() for a in arrs]
    bad[4][3, 100] = np.nan
    r = ctx.gamma_points(h, *bad, dP, t0, want_info=True)
    assert r["nbad"] == 1 and (r["info"][3] >> 16) == 2
    keep = np.arange(n) != 3
    assert np.array_equal(r["gam"][keep], host["gam"][keep])
    assert ctx.gamma_points(h, *[a[:0] for a in arrs], dP[:0], t0[:0])["gam"].shape == (0,)


def test_scan_starts_rule_on_the_device(ctx):
    """ibs_scan_starts_f64 = pick_start (ball_scan.py:279-295) for every surface at once: first maximum -> (alpha_scan[i],
    theta0_scan[j]); an all-zero table -> (0, 0); a non-finite maximum is counted and starts from (0, 0)."""
    import torch
    import ibs_amd
    dev = torch.device("cuda:0")
    na, nt0 = 24, 15
    al = np.linspace(0, np.pi, na); t0 = np.linspace(0, np.pi / 2, nt0)
    rng = np.random.default_rng(2)
    tabs = rng.uniform(-1e-3, 1e-3, size=(9, na, nt0))
    tabs[2, 5, 7] = tabs[2, 11, 3] = tabs[2].max() + 1e-4          # a tie: the first (row-major) maximum wins
    tabs[4] = 0.0                                                   # nothing scanned: ball_scan.py:279-282
    tabs[6] = -np.abs(tabs[6])                                      # all stable: the maximum is negative
    d_tabs = torch.from_numpy(tabs.reshape(9, -1)).to(dev)
    idx, val = ctx.surface_argmax(d_tabs)
    pack = torch.stack([val, idx.double()], dim=1).contiguous()
    pack[8, 0] = float("nan")
    n_bad = torch.zeros(1, dtype=torch.int32, device=dev)
    start = ctx.scan_starts(torch.from_numpy(al).to(dev), torch.from_numpy(t0).to(dev), pack, n_bad).cpu().numpy()
    assert int(n_bad.item()) == 1 and np.array_equal(start[8], [0.0, 0.0])
    for k in range(8):
        a0, th0, sigma0, ij = ibs_amd.pick_start(tabs[k], al, t0)
        assert np.array_equal(start[k], [a0, th0]), k
    assert np.array_equal(start[4], [0.0, 0.0]) and np.array_equal(start[2], [al[5], t0[7]])


def test_last_launch_names_the_kernel_the_bench_times(ctx):
    """ibs_last_launch: the bench looks its counters up by the exact kernel name and waves per launch"""
    import torch
    import ibs_amd
    import bench
    dev = torch.device("cuda:0")
    h, geo7, dP_d, th0_d, *_ = bench.build_workload(0, dev)
    plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF)
    plan.scan_argmax()
    assert ctx.last_launch() == ("ibs::k_gamma_scan<double, 8>", 1024)
    g = torch.ones((4096, 513), dtype=torch.float64, device=dev)
    ctx.sturm_count(h, g, g, g, torch.zeros(4096, dtype=torch.float64, device=dev))
    name, waves = ctx.last_launch()
    assert name == "ibs::k_sturm_count<double, 8>" and waves == 4096
    torch.cuda.synchronize()


def test_device_rows_equal_the_host_driven_steps(ctx):
    """BallooningScan.run() on the device pipeline (geometry -> fused scan + argmax -> starts on the device -> device-pointer
    refinement -> one-line final solve, one copy back) against the steps it replaces, driven from the host: coarse() table ->
    pick_start -> refine_device(host starts) -> final_solve_device.  Same kernels on the same inputs: bit for bit."""
    import torch
    import ibs_amd
    dev = torch.device("cuda:0")
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    sv = np.linspace(0.5, 0.95, 5)
    th = ibs_amd.theta_grid(969)
    scan = ibs_amd.BallooningScan(ctx, None, th, sv, tables=ibs_amd.SurfaceTables.from_wout(wout, sv), device=dev)
    ph = {}
    rows, bad = scan.device_rows(True, ph)
    rows = rows.cpu().numpy()
    assert float(bad) == 0 and set(ph) == {"geometry_ms", "scan_argmax_ms", "refine_ms", "final_solve_ms"}
    tabs = scan.coarse()
    starts = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in tabs])
    xo, fo, ne = scan.refine_device(starts)
    gam = scan.final_solve_device(xo)
    assert np.array_equal(rows[:, 0], xo[:, 1]) and np.array_equal(rows[:, 1], xo[:, 0]) and np.array_equal(rows[:, 2], gam)
    assert np.array_equal(scan.last_refine["n_evals"].cpu().numpy(), ne)
    t0a, ala, gama = scan.run()
    assert np.array_equal(gama, gam)
    c0, ca, cg = scan.run(refine=False)                                  # coarse maxima: (theta0, alpha, max of the table)
    assert np.array_equal(cg, tabs.reshape(5, -1).max(axis=1)) and np.array_equal(np.stack([ca, c0], axis=1), starts)
    # explicit surface indices into a table set that holds the surfaces in another order
    sv2 = sv[::-1].copy()
    scan2 = ibs_amd.BallooningScan(ctx, None, th, sv, tables=ibs_amd.SurfaceTables.from_wout(wout, sv2), device=dev,
                                   surf_index=[4, 3, 2, 1, 0])
    assert np.abs(scan2.run()[2] - gam).max() < 1e-10


def test_adjoint_step_configs3_end_to_end(ctx):
    """BASELINE configs[3] as the product runs it: ONE AdjointStep.run() = radial tables of the 73 equilibria (native host
    routine) -> geometry -> coarse 24 x 15 scan + argmax -> L-BFGS-B for all 365 maxima -> final solve -> objective and the
    72-gradient (sims_runner_NCSX.py:249-261).  12 random (equilibrium, surface) pairs are re-done with the oracle on every
    link (tests/helpers.py: numpy geometry, C-oracle scan, scipy's L-BFGS-B on the oracle objective, oracle final solve):
    refined gam within 1e-8; objective and gradient against the reference's formulas written out."""
    import torch
    import ibs_amd
    import bench
    from tests.helpers import oracle_surface_pipeline
    dev = torch.device("cuda:0")
    wout0 = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    wouts, steps, x0 = bench.emulated_equilibria(wout0)
    n_eq, ns, na, nt0 = len(wouts), 5, 24, 15
    svals = np.linspace(0.5, 0.95, ns)                                   # ball_scan.py:197
    th = ibs_amd.theta_grid_for(11, 11)                                  # 969 points
    f_other = 0.8 + 0.01 * np.arange(n_eq)
    step = ibs_amd.AdjointStep(ctx, th, svals, dev, nalpha=na, ntheta0=nt0, gamma_thresh=-2.0e-4, prefac=50.0)
    ph = {}
    out = step.run(wouts, f_other, steps, phases=ph)
    again = step.run(wouts, f_other, steps)                              # resident inputs reused: the same bits
    assert np.array_equal(out["gam"], again["gam"]) and np.array_equal(out["dfobj"], again["dfobj"])
    gam = out["gam"]
    assert gam.shape == (n_eq, ns) and set(ph) >= {"host_tables_ms", "geometry_ms", "scan_argmax_ms", "refine_ms", "final_solve_ms"}
    rng = np.random.default_rng(12)
    worst = 0.0
    for k in rng.choice(n_eq * ns, size=12, replace=False):
        q, js = divmod(int(k), ns)
        ref = oracle_surface_pipeline(wouts[q], float(svals[js]), th, na, nt0, step.del_alpha)
        worst = max(worst, abs(ref["gam"] - gam[q, js]))
        assert abs(ref["gam"] - gam[q, js]) < 1e-8, (q, js, ref["gam"], gam[q, js])
        assert abs(ref["x_opt"][0] - out["alpha"][q, js]) < 1e-5 and abs(ref["x_opt"][1] - out["theta0"][q, js]) < 1e-5
    f0_arr = np.zeros(n_eq); df0 = np.zeros(n_eq - 1)
    for i in range(n_eq):                                                 # sims_runner_NCSX.py:254-261 written out
        f0_arr[i] = f_other[i] + 50.0 * np.sum(np.maximum(gam[i] - (-2.0e-4), 0.0))
        if i > 0:
            df0[i - 1] = (f0_arr[i] - f0_arr[0]) / steps[i] * 0.5 * 1 / np.sqrt(f0_arr[0])
    assert np.abs(out["f0"] - f0_arr).max() < 1e-12 and np.abs(out["dfobj"] - df0).max() < 1e-9 * max(1.0, np.abs(df0).max())
    assert abs(out["fobj"] - np.sqrt(f0_arr[0])) < 1e-15
    print("configs[3] as one AdjointStep: phases %s; worst |gam - oracle| of the sample %.2e" % (
        {k: round(v, 3) for k, v in ph.items()}, worst))


def test_geometry_on_a_mode_set_too_big_for_the_one_lane_per_point_image(ctx):
    """F1 on tables with twice NCSX's modes (11 rows of 47 + 15 rows of 51: the size of a W7-X-like run): the image of the
    one-lane-per-point forms is sized for one pair per mode and does not fit the LDS there, so every batch size must run on the
    lanes-per-point forms (NOT on the one-sincos-per-mode kernel) and agree with the numpy oracle (utils.py:359-720 restated).
    The extra modes carry small coefficients (a data manipulation for shape coverage: both sides evaluate the same formulas)."""
    import torch
    import ibs_amd
    from oracle import geometry_oracle as go
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    nfp = int(np.min(np.abs(wout["xn"][wout["xn"] != 0])))
    rng = np.random.default_rng(11)
    w = dict(wout)

    def widen(xm, xn, names, nmax):
        have = set(zip(xm.astype(int).tolist(), xn.astype(int).tolist()))
        extra = [(m, n * nfp) for m in range(int(xm.max()) + 1) for n in range(-nmax, nmax + 1)
                 if (m > 0 or n >= 0) and (m, n * nfp) not in have]
        em = np.array([e[0] for e in extra], dtype=float); en = np.array([e[1] for e in extra], dtype=float)
        order = np.lexsort((np.concatenate([xn, en]), np.concatenate([xm, em])))           # VMEC order: m, then n
        prof = np.linspace(0, 1, wout[names[0]].shape[1]) ** 2
        for k in names:
            scale = 1e-4 * np.abs(wout[k]).max()
            add = scale * rng.standard_normal((len(extra), 1)) * prof[None, :] / (1.0 + np.abs(en[:, None]) / nfp)
            w[k] = np.concatenate([wout[k], add])[order]
        return np.concatenate([xm, em])[order], np.concatenate([xn, en])[order]

    w["xm"], w["xn"] = widen(wout["xm"], wout["xn"], ("rmnc", "zmns", "lmns"), 23)
    w["xm_nyq"], w["xn_nyq"] = widen(wout["xm_nyq"], wout["xn_nyq"], ("gmnc", "bmnc", "bsupvmnc", "bsubsmns", "bsubumnc", "bsubvmnc"), 25)
    assert len(w["xm"]) > 480 and len(w["xm_nyq"]) > 650
    svals = np.array([0.45, 0.8])
    tabs = ibs_amd.SurfaceTables.from_wout(w, svals)
    otab = go.surface_tables_from_wout(w, svals)
    dev = torch.device("cuda:0")
    for N, n_lines in ((257, 3), (969, 40), (969, 400)):
        th = ibs_amd.theta_grid(N)
        surf = np.arange(n_lines) % 2; al = np.linspace(0.0, np.pi, n_lines)
        r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
        name = ctx.last_launch()[0]
        assert "k_geo_rows<1, " in name and "k_geo_rows<1, 1," not in name, name
        got = r["geo"].cpu().numpy().transpose(1, 0, 2)
        for i in range(0, n_lines, max(1, n_lines // 5)):
            ref = go.fieldline_geometry(otab, int(surf[i]), np.array([al[i]]), th)[0]
            err = (np.abs(got[i] - ref) / np.abs(ref).max(axis=1, keepdims=True)).max()
            assert err < 1e-10, (N, n_lines, i, name, err)
