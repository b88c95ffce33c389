"""GPU: parity of the HIP path (through the C ABI) with the oracle and the golden vectors.
FP64 tolerance: |gam_HIP - gam_ref| < 1e-8 (BASELINE.json north_star); measured errors are ~1e-12."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-8


@pytest.fixture(scope="module")
def ctx():
    import ibs_amd
    return ibs_amd.Context(0)


@pytest.fixture(autouse=True)
def _default_dispatch(ctx):
    """tests steer kernel variants through the context's option setter (ibs_set_option); every test starts and ends
    with the library's own dispatch"""
    ctx.reset_options()
    yield
    ctx.reset_options()


@pytest.fixture(scope="module")
def bo():
    from oracle import ballooning_oracle
    return ballooning_oracle


def eigvec_tol(bo, th, g, c, f):
    """agreement to expect between the recurrence-based (twisted) eigenvector and LAPACK's: the recurrences carry
    a relative error ~n eps, so the eigenvalue the vector belongs to is off by ~n eps ||A|| and the vector by that
    over the gap lam_1 - lam_2 (NCSX lines at N = 1025: 1e-7 measured; LAPACK itself is good to 3e-12 there).
    The growth rate is second order in this error (5e-13 measured)."""
    from scipy.linalg import eigh_tridiagonal
    d, e, fd = bo.assemble(th, g, c, f)[:3]
    n = len(d)
    a = d / fd
    b = e[1:n] / np.sqrt(fd[:-1] * fd[1:])
    w = eigh_tridiagonal(a, b, eigvals_only=True, select="i", select_range=(n - 2, n - 1))
    norm_a = np.max(np.abs(a)) + 2 * np.max(np.abs(b))
    return max(1e-8, 2 * n * np.finfo(float).eps * norm_a / (w[1] - w[0]))


def salpha_batch(bo, N, params):
    th = bo.theta_grid(N)
    g = np.empty((len(params), N)); c = np.empty_like(g)
    for k, (sh, al, t0) in enumerate(params):
        g[k], c[k] = bo.salpha_gc(th, sh, al, t0)
    return th, g, c


@pytest.mark.parametrize("N", [257, 513, 1025])
def test_G1_salpha_against_reference_goldens(ctx, bo, N):
    g1 = np.load(os.path.join(G, "G1_salpha.npz"))
    sel = g1["params"][:, 0] == N
    params = g1["params"][sel][:, 1:]
    th, g, c = salpha_batch(bo, N, params)
    r = ctx.solve_gcf(th[1] - th[0], g, c, g, want_X=True, want_info=True)
    assert r["nbad"] == 0
    assert np.abs(r["gam"] - g1["gam"][sel]).max() < TOL
    for k, (sh, al, t0) in enumerate(params):
        key = "X_%d_%g_%g_%g" % (N, sh, al, t0)
        if key in g1:
            Xref = g1[key] * np.sign(g1[key][np.argmax(np.abs(g1[key]))])
            dXref = g1["d" + key] * np.sign(g1[key][np.argmax(np.abs(g1[key]))])
            assert np.abs(r["X"][k] - Xref).max() < 1e-7
            assert np.abs(r["dX"][k] - dXref).max() < 1e-6
        go, lo, Xo, dXo = bo.solve_gcf(th, g[k], c[k], g[k])
        assert abs(r["lam"][k] - lo) < 1e-10
        assert abs(r["gam"][k] - go) < 1e-10
        assert np.abs(r["X"][k] - Xo).max() < 1e-8
        assert np.abs(r["dX"][k] - dXo).max() < 1e-7


@pytest.mark.parametrize("which,N,extent", [(3, 1601, 61), (4, 401, 20)])
def test_G2_sturm_count_reproduces_reference_stability_test(ctx, bo, which, N, extent):
    tab = np.load(os.path.join(G, "G2_salpha_stability.npz"))["table"]
    th = np.linspace(-extent * np.pi, extent * np.pi, N)
    g = np.empty((len(tab), N)); c = np.empty_like(g)
    for k, row in enumerate(tab):
        g[k], c[k] = bo.salpha_gc(th, row[0], row[1], row[2])
    cnt = ctx.sturm_count(th[1] - th[0], g, c, np.ones_like(g), np.zeros(len(tab)))
    assert ((cnt > 0).astype(int) == tab[:, which].astype(int)).all()


@pytest.mark.parametrize("N", [513, 969, 1025])
def test_G3_ncsx_scan(ctx, bo, N):
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = g3["geo_%d" % N]
    th = bo.theta_grid(N)
    a = [np.ascontiguousarray(geo[:, k, :]) for k in range(7)]
    r = ctx.gamma_scan(th[1] - th[0], *a, g3["dPdrho_%d" % N], g3["theta0"], want_X=True, want_dtheta0=True)
    assert r["nbad"] == 0
    assert np.abs(r["gam"] - g3["gam_tight_%d" % N]).max() < 1e-10     # reference formula, ARPACK converged
    assert np.abs(r["gam"] - g3["gam_%d" % N]).max() < TOL             # reference as shipped
    # eigenfunctions + Hellmann-Feynman d/dtheta0 against the oracle (utils.py:1666-1680)
    for i in (0, len(geo) - 1):
        bmag, gp, cv, cv0, gd2, gd21, gd22, gb = geo[i]
        for j, t0 in enumerate(g3["theta0"]):
            cvf, gdf = bo.fold_theta0(t0, cv, cv0, gd2, gd21, gd22)
            gam, X, dX, gg, cc, ff = bo.gamma_ball_full(g3["dPdrho_%d" % N][i], th, bmag, gp, cvf, gdf)
            assert np.abs(r["X"][i, j] - X).max() < eigvec_tol(bo, th, gg, cc, ff)
            g_t = np.abs(gp) * (2 * gd21 + 2 * t0 * gd22) / bmag
            c_t = -g3["dPdrho_%d" % N][i] * cv0 / (np.abs(gp) * bmag)
            f_t = (2 * gd21 + 2 * t0 * gd22) / bmag ** 2 / (np.abs(gp) * bmag)
            assert abs(r["dgam_dtheta0"][i, j] - bo.hf_derivative(gam, X, dX, ff, g_t, c_t, f_t)) < 1e-8


@pytest.mark.parametrize("N", [257, 513])
def test_G6_rough_random_eigenvalue(ctx, bo, N):
    g6 = np.load(os.path.join(G, "G6_random_rough.npz"))
    gcf = g6["gcf_%d" % N]
    th = bo.theta_grid(N)
    r = ctx.solve_gcf(th[1] - th[0], gcf[:, 0], gcf[:, 1], gcf[:, 2], want_info=True)
    assert r["nbad"] == 0
    for k in range(len(gcf)):
        go, lo, _, _ = bo.solve_gcf(th, *gcf[k])
        assert abs(r["lam"][k] - lo) < 1e-9
        assert abs(r["gam"][k] - g6["gam_tight_%d" % N][k]) < 1e-7


def test_gamma_ball_full_dropin_signature(ctx, bo):
    import ibs_amd
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    N = 513
    th = bo.theta_grid(N)
    bmag, gp, cv, cv0, gd2, gd21, gd22, gb = g3["geo_513"][5]
    dP = ibs_amd.dPdrho_of(cv, gb, bmag)
    cvf, gdf = bo.fold_theta0(0.5, cv, cv0, gd2, gd21, gd22)
    out = ibs_amd.gamma_ball_full(dP, th, bmag, gp, cvf, gdf, bo.vguess(th), 1.0, ctx=ctx)
    ref = bo.gamma_ball_full(dP, th, bmag, gp, cvf, gdf)
    assert len(out) == 6 and abs(out[0] - ref[0]) < 1e-10 and abs(out[0] - g3["gam_513"][5, 1]) < TOL
    for a, b in zip(out[1:], ref[1:]):
        assert a.shape == (N,) and np.abs(a - b).max() < 1e-7


def test_device_pointers_and_ragged_batch(ctx, bo):
    import torch
    N = 641                                  # D3D reference grid (ball_scan.py:203)
    rng = np.random.default_rng(7)
    params = np.stack([rng.uniform(0.1, 2, 37), rng.uniform(0, 1.2, 37), rng.uniform(0, 1.5, 37)], 1)
    th, g, c = salpha_batch(bo, N, params)
    dev = torch.device("cuda:0")
    r = ctx.solve_gcf(th[1] - th[0], torch.from_numpy(g).to(dev), torch.from_numpy(c).to(dev),
                      torch.from_numpy(g).to(dev), want_info=True)
    rh = ctx.solve_gcf(th[1] - th[0], g, c, g)
    torch.cuda.synchronize()
    assert np.array_equal(r["gam"].cpu().numpy(), rh["gam"])          # same kernel, same bits
    for k in range(0, 37, 6):
        assert abs(rh["gam"][k] - bo.solve_gcf(th, g[k], c[k], g[k])[0]) < 1e-10
    assert (((r["info"].cpu().numpy() >> 16) & 3) == 0).all()


def test_invalid_coefficients_are_flagged_not_propagated(ctx, bo):
    """non-finite / non-positive g or f: status 2 in the info word, the call returns the count of such systems,
    and the healthy systems of the same batch (same waves' neighbours) are untouched (include/ibs.h conventions)"""
    N = 513
    rng = np.random.default_rng(11)
    params = np.stack([rng.uniform(0.1, 2, 12), rng.uniform(0, 1.2, 12), rng.uniform(0, 1.5, 12)], 1)
    th, g, c = salpha_batch(bo, N, params)
    f = g.copy()
    clean = ctx.solve_gcf(th[1] - th[0], g, c, f, want_info=True)
    assert clean["nbad"] == 0
    g2, c2, f2 = g.copy(), c.copy(), f.copy()
    g2[1, 100] = np.nan
    f2[4, 7] = 0.0
    g2[7, 300] = -1.0
    c2[9, 17] = np.inf
    r = ctx.solve_gcf(th[1] - th[0], g2, c2, f2, want_info=True)
    bad = [1, 4, 7, 9]
    assert r["nbad"] == len(bad)
    st = r["info"] >> 16
    assert sorted(np.nonzero(st)[0].tolist()) == bad and (st[bad] == 2).all()
    good = [k for k in range(12) if k not in bad]
    assert np.array_equal(r["gam"][good], clean["gam"][good]) and np.array_equal(r["lam"][good], clean["lam"][good])


def test_empty_and_single_system_batches(ctx, bo):
    N = 257
    th, g, c = salpha_batch(bo, N, np.array([[0.8, 0.6, 0.0]]))
    r0 = ctx.solve_gcf(th[1] - th[0], g[:0], c[:0], g[:0])
    assert r0["gam"].shape == (0,) and r0["nbad"] == 0
    r1 = ctx.solve_gcf(th[1] - th[0], g, c, g)
    assert abs(r1["gam"][0] - bo.solve_gcf(th, g[0], c[0], g[0])[0]) < 1e-10
    cnt = ctx.sturm_count(th[1] - th[0], g[:0], c[:0], g[:0], np.zeros(0))
    assert cnt.shape == (0,)


@pytest.mark.parametrize("N", [67, 131, 195, 323, 451, 577, 705, 833, 961, 1089, 1217, 1473, 1729, 1985])
def test_every_rows_per_lane_instantiation(ctx, bo, N):
    """one system per kernel instantiation (rows per lane M = 2 ... 31) against the LAPACK oracle"""
    th, g, c = salpha_batch(bo, N, np.array([[0.9, 0.7, 0.3], [1.7, 0.4, 1.1]]))
    r = ctx.solve_gcf(th[1] - th[0], g, c, g, want_info=True)
    rx = ctx.solve_gcf(th[1] - th[0], g, c, g, want_X=True)          # X / dX leave through the wave's LDS row
    assert r["nbad"] == 0
    for k in range(2):
        gam, lam, X, dX = bo.solve_gcf(th, g[k], c[k], g[k])
        assert abs(r["gam"][k] - gam) < 1e-10 and abs(rx["gam"][k] - r["gam"][k]) < 1e-13
        assert np.abs(rx["X"][k] - X).max() < 1e-7 and np.abs(rx["dX"][k] - dX).max() < 1e-7 * np.abs(dX).max() + 1e-7


def test_hf_grad_generic_tangents(ctx, bo):
    """ibs_hf_grad_f64 (Hellmann-Feynman sums for caller-built tangents) against the oracle formula, host and
    device pointers; the theta0 tangents of utils.py:1669-1673 must reproduce the fused kernel's d(gam)/d(theta0)"""
    import torch
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    N = 513
    geo = g3["geo_%d" % N]
    th = bo.theta_grid(N)
    a = [np.ascontiguousarray(geo[:, k, :]) for k in range(7)]
    dP = g3["dPdrho_%d" % N]
    t0s = g3["theta0"]
    r = ctx.gamma_scan(th[1] - th[0], *a, dP, t0s, want_X=True, want_dtheta0=True)
    nl, nt = r["gam"].shape
    bmag, gp, cv, cv0, gd2, gd21, gd22 = [x[:, None, :] for x in a]
    T0 = t0s[None, :, None]
    gdp = 2 * gd21 + 2 * T0 * gd22
    g_t = np.abs(gp) * gdp / bmag + 0 * T0
    c_t = -dP[:, None, None] * cv0 / (np.abs(gp) * bmag) + 0 * T0
    f_t = gdp / bmag ** 2 / (np.abs(gp) * bmag)
    f = (gd2 + 2 * T0 * gd21 + T0 ** 2 * gd22) / bmag ** 2 / (np.abs(gp) * bmag)
    flat = lambda x: np.array(np.broadcast_to(x, (nl, nt, N)).reshape(nl * nt, N), copy=True)
    args = [flat(r["X"]), flat(r["dX"]), flat(f), flat(g_t), flat(c_t), flat(f_t)]
    jac = ctx.hf_grad(*args, r["gam"].reshape(-1))
    assert np.abs(jac.reshape(nl, nt) - r["dgam_dtheta0"]).max() < 1e-12
    k = 5
    assert abs(jac[k] - bo.hf_derivative(r["gam"].reshape(-1)[k], args[0][k], args[1][k], args[2][k], args[3][k], args[4][k], args[5][k])) < 1e-12
    dev = torch.device("cuda:0")
    jd = ctx.hf_grad(*[torch.from_numpy(x).to(dev) for x in args], torch.from_numpy(r["gam"].reshape(-1).copy()).to(dev))
    torch.cuda.synchronize()
    assert np.array_equal(jd.cpu().numpy(), jac)


@pytest.mark.parametrize("N", [513, 1025])
def test_nearly_degenerate_top_eigenvalues(ctx, bo, N):
    """two identical, well separated wells: lam_1 - lam_2 from 2e-2 down to 2e-9.  The count-certified bracket
    must still close on lam_max (no C == 1 window may ever be seen: pure bisection then), status 0."""
    from scipy.linalg import eigh_tridiagonal
    th = bo.theta_grid(N)
    cases = ((6.0, 3.0, 0.8), (8.0, 5.0, 0.5), (3.0, 2.0, 1.0), (9.0, 1.0, 1.5))
    g = np.ones((len(cases), N)); f = np.ones_like(g)
    c = np.stack([d * (np.exp(-((th - s) / w) ** 2) + np.exp(-((th + s) / w) ** 2)) - 0.5 for s, d, w in cases])
    r = ctx.solve_gcf(th[1] - th[0], g, c, f, want_info=True)
    assert r["nbad"] == 0 and (((r["info"] >> 16) & 3) == 0).all() and ((r["info"] & 0xffff) < 60).all()
    for k in range(len(cases)):
        d, e, fd = bo.assemble(th, g[k], c[k], f[k])[:3]
        n = len(d)
        w = eigh_tridiagonal(d / fd, e[1:n] / np.sqrt(fd[:-1] * fd[1:]), eigvals_only=True, select="i", select_range=(n - 1, n - 1))
        norm_a = np.max(np.abs(d / fd)) + 2 * np.max(e / fd.min())
        assert abs(r["lam"][k] - w[0]) <= 4 * 64 * np.finfo(float).eps * norm_a


def test_surface_argmax_first_tie(ctx):
    tab = np.array([[0.1, 0.5, 0.5, -1.0], [3.0, 3.0, 1.0, 3.0], [-2.0, -3.0, -2.5, -2.0]])
    idx, val = ctx.surface_argmax(tab)
    assert idx.tolist() == [1, 0, 0] and val.tolist() == [0.5, 3.0, -2.0]
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    idx, val = ctx.surface_argmax(g5["gam_table"].reshape(1, -1))
    assert divmod(int(idx[0]), 15) == tuple(int(v) for v in g5["argmax"])


def test_G5_coarse_scan_table(ctx, bo):
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    th = bo.theta_grid(513)
    a = [np.ascontiguousarray(g5["geo"][:, k, :]) for k in range(7)]
    r = ctx.gamma_scan(th[1] - th[0], *a, g5["dPdrho"], g5["theta0_scan"])
    assert r["gam"].shape == (24, 15)
    assert np.abs(r["gam"] - g5["gam_table"]).max() < TOL
    idx, val = ctx.surface_argmax(r["gam"].reshape(1, -1))
    assert divmod(int(idx[0]), 15) == tuple(int(v) for v in g5["argmax"])


def test_rejects_unsupported_grids(ctx):
    import ibs_amd
    for N in (512, 33, 65539, 4098):      # even, too short, beyond the long-grid path's 65,537 points (4,099 is served since round 6), even and long
        z = np.ones((2, N))
        with pytest.raises(ibs_amd.IbsError):
            ctx.solve_gcf(0.1, z, z, z)
    r = ctx.solve_gcf(0.1, -np.ones((1, 257)), np.ones((1, 257)), np.ones((1, 257)), want_info=True)
    assert r["nbad"] == 1 and (r["info"][0] >> 16) == 2


def test_G4_obj_w_grad_kernel(ctx, bo):
    """fused objective + Hellmann-Feynman gradient (utils.py:1632-1728) against the reference goldens"""
    g4 = np.load(os.path.join(G, "G4_obj_w_grad.npz"))
    th = bo.theta_grid(513)
    val, jac, info = ctx.obj_w_grad(th[1] - th[0], g4["geo"], g4["pts"][:, 2], float(g4["del_alpha"]), want_info=True)
    assert (((info >> 16) & 3) == 0).all()
    assert np.abs(val - g4["val_tight"]).max() < 1e-10 and np.abs(jac - g4["jac_tight"]).max() < 1e-9
    assert np.abs(val - g4["val"]).max() < TOL and np.abs(jac - g4["jac"]).max() < 1e-7   # shipped ARPACK tol
    # device-pointer path gives the same bits
    import torch
    dev = torch.device("cuda:0")
    v2, j2 = ctx.obj_w_grad(th[1] - th[0], torch.from_numpy(g4["geo"]).to(dev),
                            torch.from_numpy(np.ascontiguousarray(g4["pts"][:, 2])).to(dev), float(g4["del_alpha"]))
    assert np.array_equal(v2.cpu().numpy(), val) and np.array_equal(j2.cpu().numpy(), jac)


def test_driver_on_gpu_matches_oracle_backed_driver(ctx, bo):
    """ball_scan.py:248-339 counterpart end to end (coarse scan, argmax, L-BFGS-B refine, final solve)"""
    import ibs_amd
    from tests.helpers import OracleContext, synthetic_fieldlines
    N = 257
    th = bo.theta_grid(N)
    fl = synthetic_fieldlines(th)
    rho = np.linspace(0.5, 0.95, 3)
    gpu = ibs_amd.BallooningScan(ctx, fl, th, rho, nalpha=6, ntheta0=5)
    cpu = ibs_amd.BallooningScan(OracleContext(), fl, th, rho, nalpha=6, ntheta0=5)
    assert np.abs(gpu.coarse() - cpu.coarse()).max() < 1e-10
    tg, ag, gg = gpu.run(refine=True)
    tc, ac, gc = cpu.run(refine=True)
    assert np.abs(gg - gc).max() < TOL
    assert np.abs(tg - tc).max() < 1e-4 and np.abs(ag - ac).max() < 1e-4
    drop = ibs_amd.make_obj_w_grad(lambda vs, s, al, theta: fl(s, al), ctx=ctx)
    v, j = drop((1.0, 0.4), None, 0.7, th, None, 0.42)
    vo, jo = bo.obj_w_grad_lines(th, 0.4, *fl(0.7, np.array([1.0 - 0.002, 1.0, 1.0 + 0.002])))
    assert abs(v - vo) < 1e-10 and np.abs(j - jo).max() < 1e-8


@pytest.mark.parametrize("N", [129, 257, 513, 1025])
def test_fp32_variant_stated_tolerance(ctx, bo, N):
    """config 5 FP32 (host pointers): the all-FP32 kernel for eigenvalues alone -- error bounded relative to ||A|| (SURVEY H3:
    eps32 ||A||) -- and the widened form (FP32 arrays, FP64 solver) whenever the growth rate or the eigenfunction is asked for:
    gam within 1e-6 of the FP64 entry point's on the same systems, X within 1e-6"""
    rng = np.random.default_rng(11)
    params = np.stack([rng.uniform(0.3, 2, 24), rng.uniform(0.2, 1.2, 24), rng.uniform(0, 1.5, 24)], 1)
    th, g, c = salpha_batch(bo, N, params)
    h = th[1] - th[0]
    g32, c32 = g.astype(np.float32), c.astype(np.float32)
    r64 = ctx.solve_gcf(h, g, c, g, want_X=True)
    normA = 4.0 / h ** 2 + 4.0                                   # ~ max_j (2 g/h^2 + |c|)/f for f = g
    r32 = ctx.solve_gcf(h, g32, c32, g32, want_info=True, dtype=np.float32, want_gam=False)
    assert r32["lam"].dtype == np.float32 and r32["gam"] is None and (((r32["info"] >> 16) & 3) == 0).all()
    err = np.abs(r32["lam"].astype(np.float64) - r64["lam"])
    assert err.max() < 64 * 1.2e-7 * normA and np.median(err) < 8 * 1.2e-7 * normA    # stated FP32 tolerance
    rw = ctx.solve_gcf(h, g32, c32, g32, want_X=True, want_info=True, dtype=np.float32)
    assert rw["gam"].dtype == np.float32 and rw["X"].dtype == np.float32 and (((rw["info"] >> 16) & 3) == 0).all()
    assert np.abs(rw["lam"].astype(np.float64) - r64["lam"]).max() < 2 * 1.2e-7 * normA
    assert np.abs(rw["gam"].astype(np.float64) - r64["gam"]).max() < 1e-6
    assert np.abs(rw["X"].astype(np.float64) - r64["X"]).max() < 1e-5 and np.abs(rw["dX"].astype(np.float64) - r64["dX"]).max() < 1e-4 * np.abs(r64["dX"]).max()


def test_large_grid_2049(ctx, bo):
    """N_zeta = 2048 (rows-per-lane 32)"""
    N = 2049
    params = [(1.0, 0.8, 0.0), (0.5, 0.6, 0.3), (1.7, 1.1, 0.1)]
    th, g, c = salpha_batch(bo, N, params)
    r = ctx.solve_gcf(th[1] - th[0], g, c, g, want_info=True)
    assert r["nbad"] == 0
    for k in range(len(params)):
        go, lo, _, _ = bo.solve_gcf(th, g[k], c[k], g[k])
        assert abs(r["gam"][k] - go) < 1e-9 and abs(r["lam"][k] - lo) < 1e-9


def test_full_size_stress_properties(ctx):
    """BASELINE config 5 at its full size (10^6 systems, N_zeta = 512), checked through size-independent properties:
    the Sturm count is 0 just above the returned eigenvalue and exactly 1 just below it; lam is invariant
    under a common scaling of (g, c) and shifts by s under c -> c + s f."""
    import torch
    dev = torch.device("cuda:0")
    n, N = 1000000, 513
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)
    g = torch.exp(u(np.log(0.01), np.log(50.0), (n, N)))
    c = u(-2.5, 3.5, (n, N))
    f = torch.exp(u(np.log(0.2), np.log(3e3), (n, N)))
    h = 8 * np.pi / (N - 1)
    r = ctx.solve_gcf(h, g, c, f, want_info=True)
    lam = r["lam"]
    assert int((((r["info"] >> 16) & 3) != 0).sum()) == 0
    e = 0.5 * (g[:, :-1] + g[:, 1:]) / h ** 2
    d = c[:, 1:-1] - (e[:, :-1] + e[:, 1:])
    normA = ((d.abs() + e[:, :-1] + e[:, 1:]) / f[:, 1:-1]).amax(dim=1)        # the solver's ||A|| bound
    eps = 2e-13 * normA              # certified bracket is 256 ulp(||A||) = 5.7e-14 ||A|| wide
    above = ctx.sturm_count(h, g, c, f, lam + eps)
    below = ctx.sturm_count(h, g, c, f, lam - eps)
    # floating-point Sturm counts of rough systems are not perfectly monotone in the shift: allow a few
    # inconsistent counts, but every such system must still agree with the independent C oracle
    odd = torch.nonzero((above != 0) | (below < 1)).flatten().cpu().numpy()
    assert len(odd) <= n // 10000
    if len(odd):
        from oracle import c_oracle as co
        _, lam_c, _ = co.solve_gcf_batch(h, g[odd].cpu().numpy(), c[odd].cpu().numpy(), f[odd].cpu().numpy())
        assert np.abs(lam[odd].cpu().numpy() - lam_c).max() < 1e-11 * float(normA[odd].max())
    assert float((below == 1).double().mean()) > 0.999          # near-degenerate pairs are allowed but rare
    r2 = ctx.solve_gcf(h, 2 * g[:4096], 2 * c[:4096] + 0.25 * 2 * f[:4096], 2 * f[:4096])
    assert float(((r2["lam"] - (lam[:4096] + 0.25)).abs() / normA[:4096]).max()) < 3e-13   # inside the certified brackets


def test_config3_full_size_from_wout_tables(ctx, bo):
    """BASELINE config 3 at its full single-GPU size (64 surfaces x 32 alpha x 16 theta0 = 32,768 solves, N_zeta = 1024)
    from the shipped equilibrium's wout tables, geometry on the device: a random sample of lines against the C
    oracle, no system flagged, and the result of a line does not depend on where it sits in the batch (the same
    lines in reversed order give bitwise the same growth rates)."""
    import ibs_amd
    import torch
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    ns, na, nt0, N = 64, 32, 16, 1025
    tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.1, 0.95, ns))
    th = bo.theta_grid(N)
    surf = np.repeat(np.arange(ns), na); al = np.tile(np.linspace(0, np.pi, na), ns)
    r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    t0 = np.linspace(0, np.pi / 2, nt0)
    t0d = torch.from_numpy(t0).to(dev)
    h = th[1] - th[0]
    sc = ctx.gamma_scan(h, *[r["geo"][k] for k in range(7)], r["dPdrho"], t0d, want_info=True)
    assert int(((sc["info"] >> 16) != 0).sum()) == 0
    rev = torch.arange(ns * na - 1, -1, -1, device=dev)
    sc_r = ctx.gamma_scan(h, *[r["geo"][k][rev].contiguous() for k in range(7)], r["dPdrho"][rev].contiguous(), t0d)
    assert bool((sc_r["gam"][rev] == sc["gam"]).all()) and bool((sc_r["lam"][rev] == sc["lam"]).all())
    pick = np.random.default_rng(3).choice(ns * na, size=48, replace=False)
    geo_h = r["geo"][:, torch.from_numpy(pick).to(dev)].cpu().numpy()
    gam_c, lam_c, _ = co.gamma_scan(h, *[np.ascontiguousarray(geo_h[k]) for k in range(7)],
                                    r["dPdrho"].cpu().numpy()[pick], t0)
    assert np.abs(sc["gam"].cpu().numpy()[pick] - gam_c).max() < TOL
    assert np.abs(sc["lam"].cpu().numpy()[pick] - lam_c).max() < TOL
    idx, val = ctx.surface_argmax(sc["gam"].reshape(ns, -1))
    assert np.array_equal(idx.cpu().numpy(), sc["gam"].reshape(ns, -1).argmax(dim=1).cpu().numpy())


@pytest.mark.parametrize("chain,nt0", [(2, 16), (4, 15), (4, 16), (5, 7), (16, 16)])
def test_chained_scan_matches_unchained(ctx, bo, chain, nt0, monkeypatch):
    """k_gamma_scan_chain (several theta0 of a line solved one after the other by one wave, each warm-started from
    the previous eigenvalue): same certified results as one wave per theta0, including X / dX / d(gam)/d(theta0)
    outputs, ragged chains (theta0 count not a multiple of the chain) and deliberately bad warm-start widths."""
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = np.tile(g3["geo_1025"], (3, 1, 1))[:14]
    geo[:, 4:7] *= (1 + 0.02 * np.arange(len(geo)))[:, None, None]
    dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)
    th = bo.theta_grid(1025)
    t0 = np.linspace(0, np.pi / 2, nt0)
    a = [np.ascontiguousarray(geo[:, k]) for k in range(7)]
    ctx.set_option("scan_chain", "1")
    ref = ctx.gamma_scan(th[1] - th[0], *a, dP, t0, want_X=True, want_dtheta0=True, want_info=True)
    for w1, w2 in (("0.5", "1.0"), ("1e-6", "1e-6"), ("50", "50")):
        ctx.set_option("scan_chain", str(chain)); ctx.set_option("chain_w1", w1); ctx.set_option("chain_w2", w2)
        r = ctx.gamma_scan(th[1] - th[0], *a, dP, t0, want_X=True, want_dtheta0=True, want_info=True)
        assert r["nbad"] == 0
        # lam is certified to 256 ulp(||A||) (~3e-11 here); gam is second order in the eigenvector error
        assert np.abs(r["gam"] - ref["gam"]).max() < 1e-11 and np.abs(r["lam"] - ref["lam"]).max() < 1e-10
        assert np.abs(r["dgam_dtheta0"] - ref["dgam_dtheta0"]).max() < 1e-9
        assert np.abs(r["X"] - ref["X"]).max() < 1e-6 and np.abs(r["dX"] - ref["dX"]).max() < 1e-5
        rn = ctx.gamma_scan(th[1] - th[0], *a, dP, t0)                       # no X: no per-wave LDS row
        assert np.abs(rn["gam"] - ref["gam"]).max() < 1e-11
    sw_ref = (np.asarray(ref["info"]) & 0xffff).mean()
    ctx.set_option("chain_w1", "0.5"); ctx.set_option("chain_w2", "1.0")
    r = ctx.gamma_scan(th[1] - th[0], *a, dP, t0, want_info=True)
    assert (np.asarray(r["info"]) & 0xffff).mean() < sw_ref                 # the chain saves sweeps


@pytest.mark.parametrize("N,P,chain,nt0", [(513, 32, 4, 16), (513, 32, 2, 8), (257, 16, 2, 8), (257, 16, 4, 16), (641, 32, 3, 12)])
def test_subwave_chained_and_warm_scan(ctx, bo, N, P, chain, nt0, monkeypatch):
    """sub-wave kernels (32 / 16 lanes per system): the theta0 chain and caller-supplied warm starts give the same
    certified results as the cold one-wave-per-system scan, in fewer sweeps"""
    rng = np.random.default_rng(N + P + chain)
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    src_th = bo.theta_grid(513)
    th = bo.theta_grid(N)
    geo = np.stack([[np.interp(th, src_th, g3["geo_513"][l, k]) for k in range(8)] for l in range(10)])
    geo[:, 4:7] *= (1 + rng.uniform(-0.05, 0.05, len(geo)))[:, None, None]
    dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)
    t0 = np.linspace(0, np.pi / 2, nt0)
    a = [np.ascontiguousarray(geo[:, k]) for k in range(7)]
    h = th[1] - th[0]
    ctx.set_option("force_p", "64"); ctx.set_option("scan_chain", "1")
    ref = ctx.gamma_scan(h, *a, dP, t0, want_X=True, want_dtheta0=True, want_info=True)
    ctx.set_option("force_p", str(P)); ctx.set_option("scan_chain", str(chain))
    r = ctx.gamma_scan(h, *a, dP, t0, want_X=True, want_dtheta0=True, want_info=True)
    rn = ctx.gamma_scan(h, *a, dP, t0, want_info=True)
    for q in (r, rn):
        assert q["nbad"] == 0
        assert np.abs(q["gam"] - ref["gam"]).max() < 1e-11 and np.abs(q["lam"] - ref["lam"]).max() < 1e-10
    assert np.abs(r["dgam_dtheta0"] - ref["dgam_dtheta0"]).max() < 1e-9
    assert np.abs(r["X"] - ref["X"]).max() < 1e-6
    sw = lambda q: (np.asarray(q["info"]) & 0xffff).mean()
    assert sw(rn) < sw(ref)
    # caller-supplied guesses (ibs_gamma_scan_warm_f64) through the sub-wave kernel
    ctx.set_option("scan_chain", "1")
    pert = [x.copy() for x in a]
    for k in (4, 5, 6):
        pert[k] *= 1.002
    cold = ctx.gamma_scan(h, *pert, dP, t0, want_info=True)
    width = 3 * float(np.abs(cold["lam"] - ref["lam"]).max())
    warm = ctx.gamma_scan(h, *pert, dP, t0, want_info=True, lam_guess=ref["lam"], guess_width=width)
    bad = ctx.gamma_scan(h, *pert, dP, t0, want_info=True, lam_guess=ref["lam"] + 0.3, guess_width=1e-7)   # wrong guesses
    for q in (warm, bad):
        assert q["nbad"] == 0 and np.abs(q["gam"] - cold["gam"]).max() < 1e-11
    assert sw(warm) < sw(cold) - 3


@pytest.mark.parametrize("P", [64, 32])
def test_chained_scan_flags_invalid_lines_only(ctx, bo, P, monkeypatch):
    """a field line with non-finite / non-positive geometry inside a chained scan: its systems are flagged (status 2),
    every other line's results are untouched and the chain restarts cold after a flagged solve"""
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = g3["geo_513"][:6].copy()
    dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)
    th = bo.theta_grid(513)
    t0 = np.linspace(0, np.pi / 2, 16)
    h = th[1] - th[0]
    ctx.set_option("force_p", str(P)); ctx.set_option("scan_chain", "4")
    good = ctx.gamma_scan(h, *[np.ascontiguousarray(geo[:, k]) for k in range(7)], dP, t0, want_info=True)
    assert good["nbad"] == 0
    bad = geo.copy()
    bad[2, 4, 100] = np.nan            # gds2 of line 2
    bad[4, 0, 7] = -1.0                # bmag < 0 on line 4: f < 0
    r = ctx.gamma_scan(h, *[np.ascontiguousarray(bad[:, k]) for k in range(7)], dP, t0, want_info=True)
    st = np.asarray(r["info"]) >> 16
    assert r["nbad"] == 32 and (st[[2, 4]] == 2).all() and (st[[0, 1, 3, 5]] == 0).all()
    ok = [0, 1, 3, 5]
    assert np.array_equal(np.asarray(r["gam"])[ok], np.asarray(good["gam"])[ok])


def test_config3_shape_ncsx_1025_tiled(ctx, bo):
    """NCSX-shape config (N_zeta = 1024, 16 theta0 per line) on tiled golden lines vs the C oracle"""
    from oracle import c_oracle as co
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = np.tile(g3["geo_1025"], (8, 1, 1))                    # 48 lines
    geo[:, 4:7] *= (1 + 0.01 * np.arange(len(geo)))[:, None, None] / 1.2
    dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)
    th = bo.theta_grid(1025)
    t0 = np.linspace(0, np.pi / 2, 16)
    a = [np.ascontiguousarray(geo[:, k]) for k in range(7)]
    r = ctx.gamma_scan(th[1] - th[0], *a, dP, t0)
    gam_c, lam_c, _ = co.gamma_scan(th[1] - th[0], *a, dP, t0)
    assert r["nbad"] == 0 and np.abs(r["gam"] - gam_c).max() < TOL and np.abs(r["lam"] - lam_c).max() < TOL


@pytest.mark.parametrize("N,P", [(513, 32), (385, 32), (257, 16), (129, 16), (641, 32), (705, 32), (513, 16)])
def test_subwave_variants_match_full_wave(ctx, bo, N, P):
    """32 / 16 lanes per system (ibs_group.hpp) against the one-wave-per-system kernels and the oracle;
    (641, 32) is the D3D grid at 20 rows per lane; (705, 32) and (513, 16) are outside the sub-wave ranges and must
    silently use the full-wave kernel."""
    rng = np.random.default_rng(N + P)
    n_sys = 37                                                  # not a multiple of the systems per wave
    params = np.stack([rng.uniform(0.2, 2, n_sys), rng.uniform(0.1, 1.2, n_sys), rng.uniform(0, 1.5, n_sys)], 1)
    th, g, c = salpha_batch(bo, N, params)
    f = g * (1 + 0.3 * np.cos(th))[None]
    h = th[1] - th[0]
    ctx.set_option("force_p", "64")
    try:
        ref = ctx.solve_gcf(h, g, c, f, want_X=True, want_info=True)
        ctx.set_option("force_p", str(P))
        r = ctx.solve_gcf(h, g, c, f, want_X=True, want_info=True)
    finally:
        ctx.set_option("force_p", None)
    assert r["nbad"] == 0 and (((r["info"] >> 16) & 3) == 0).all()
    assert np.abs(r["lam"] - ref["lam"]).max() < 1e-10 and np.abs(r["gam"] - ref["gam"]).max() < 1e-10
    assert np.abs(r["X"] - ref["X"]).max() < 1e-7 and np.abs(r["dX"] - ref["dX"]).max() < 1e-6
    for k in (0, 17, 36):
        go, lo, Xo, dXo = bo.solve_gcf(th, g[k], c[k], f[k])
        assert abs(r["gam"][k] - go) < 1e-10 and abs(r["lam"][k] - lo) < 1e-10


def test_subwave_scan_matches_full_wave(ctx, bo):
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = g3["geo_513"]
    th = bo.theta_grid(513)
    a = [np.ascontiguousarray(geo[:, k, :]) for k in range(7)]
    t0 = np.linspace(0, np.pi / 2, 8)
    ctx.set_option("force_p", "64")
    try:
        ref = ctx.gamma_scan(th[1] - th[0], *a, g3["dPdrho_513"], t0, want_X=True, want_dtheta0=True)
        ctx.set_option("force_p", "32")
        r = ctx.gamma_scan(th[1] - th[0], *a, g3["dPdrho_513"], t0, want_X=True, want_dtheta0=True, want_info=True)
        r5 = ctx.gamma_scan(th[1] - th[0], *a, g3["dPdrho_513"], t0[:5])     # 5 theta0: falls back to full waves
    finally:
        ctx.set_option("force_p", None)
    assert r["nbad"] == 0
    assert np.abs(r["gam"] - ref["gam"]).max() < 1e-10 and np.abs(r["lam"] - ref["lam"]).max() < 1e-10
    assert np.abs(r["dgam_dtheta0"] - ref["dgam_dtheta0"]).max() < 1e-9
    assert np.abs(r["X"] - ref["X"]).max() < 1e-7
    assert np.abs(r5["gam"] - ref["gam"][:, :5]).max() < 1e-10


def test_F1_fieldline_geometry_kernel(ctx, bo):
    """device geometry (utils.py:359-720) against the arrays the reference produced (G3) and the oracle"""
    import ibs_amd
    import torch
    from oracle import geometry_oracle as go
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    ref = dict(np.load(os.path.join(G, "G8_surface_tables.npz")))
    tabs = ibs_amd.SurfaceTables.from_arrays(ref)
    for N in (513, 1025):
        th = bo.theta_grid(N)
        lines = g3["lines_%d" % N]
        surf = [int(np.argmin(np.abs(ref["s"] - s))) for s, a in lines]
        geo_ref = g3["geo_%d" % N]                                   # (n_lines, 8, N)
        for use_rows in (True, False):          # rotation-recurrence kernel and the one-sincos-per-mode kernel
            r = ctx.fieldline_geometry(tabs, surf, lines[:, 1], th, use_rows=use_rows)
            for q in range(8):
                scale = np.abs(geo_ref[:, q]).max(axis=1, keepdims=True)
                assert (np.abs(r["geo"][q] - geo_ref[:, q]) / scale).max() < 1e-10, (q, use_rows)
            assert np.abs(r["dPdrho"] - g3["dPdrho_%d" % N]).max() < 1e-12
    # tables built by the data-only vmec_splines counterpart from the raw wout arrays
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    tabs2 = ibs_amd.SurfaceTables.from_wout(wout, ref["s"])
    assert np.abs(tabs2.tab_mn - tabs.tab_mn).max() < 1e-12 and np.abs(tabs2.tab_nyq - tabs.tab_nyq).max() < 1e-11
    # device-resident chain: geometry kernel -> scan kernel without leaving HBM
    th = bo.theta_grid(513)
    lines = g3["lines_513"]
    surf = [int(np.argmin(np.abs(ref["s"] - s))) for s, a in lines]
    dev = torch.device("cuda:0")
    rd = ctx.fieldline_geometry(tabs, surf, lines[:, 1], th, device=dev)
    # the resident device copies belong to the tables object (an id()-keyed cache once served stale tables to a new
    # object that re-used the id); a second, different table set on the same context must give its own geometry
    assert "_device_copies" in tabs.__dict__ and not hasattr(ctx, "_tab_cache")
    one = ibs_amd.SurfaceTables.from_wout(wout, [float(ref["s"][3])])
    r1 = ctx.fieldline_geometry(one, [0], [0.7], th, device=dev)
    r1h = ctx.fieldline_geometry(one, [0], [0.7], th)
    assert np.abs(r1["geo"].cpu().numpy() - r1h["geo"]).max() < 1e-10
    t0 = torch.from_numpy(g3["theta0"]).to(dev)
    sc = ctx.gamma_scan(th[1] - th[0], *[rd["geo"][k] for k in range(7)], rd["dPdrho"], t0)
    assert np.abs(sc["gam"].cpu().numpy() - g3["gam_tight_513"]).max() < 1e-9
    assert np.abs(sc["gam"].cpu().numpy() - g3["gam_513"]).max() < TOL


def test_F1_geometry_lanes_per_point_variants_agree(ctx, bo, monkeypatch):
    """small batches split a grid point over 2 / 4 lanes (latency of the refinement rounds), large ones put two grid
    points on a lane and hand the points beyond a multiple of 512 (N = 1025: one per line) to the one-point-per-wave
    kernel: same arrays as the one-lane-per-point kernel up to the summation order"""
    import ibs_amd
    import torch
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    tabs = ibs_amd.SurfaceTables.from_wout(wout, [0.5, 0.8])
    surf = [0, 0, 1, 1, 1]; al = [0.0, 1.3, 0.4, 2.0, np.pi]
    for N in (969, 1025, 513):
        th = bo.theta_grid(N)
        out = {}
        for lpp in ("1", "2", "4", "8", "-2"):                  # -2: two grid points per lane (+ the tail kernel)
            ctx.set_option("geo_lpp", lpp)
            r = ctx.fieldline_geometry(tabs, surf, al, th, device=torch.device("cuda:0"))
            out[lpp] = (r["geo"].cpu().numpy(), r["dPdrho"].cpu().numpy())
        ctx.set_option("geo_lpp", None)
        r = ctx.fieldline_geometry(tabs, surf, al, th, device=torch.device("cuda:0"))     # automatic choice (4 here)
        auto = r["geo"].cpu().numpy()
        scale = np.abs(out["1"][0]).max(axis=2, keepdims=True)
        for lpp in ("2", "4", "8", "-2"):
            assert np.isfinite(out[lpp][0]).all()
            assert (np.abs(out[lpp][0] - out["1"][0]) / scale).max() < 1e-11, (N, lpp)
            assert np.abs(out[lpp][1] - out["1"][1]).max() < 1e-11 * np.abs(out["1"][1]).max()
        assert (np.abs(auto - out["1"][0]) / scale).max() < 1e-11


def test_driver_with_device_geometry_reproduces_reference_scan(ctx, bo):
    """ball_scan.py:248-295 on one NCSX_op surface with the geometry produced on the GPU: the coarse
    24 x 15 table and its argmax against the reference trace (G5), then a refinement step."""
    import ibs_amd
    import torch
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    s = float(g5["s"])
    tabs = ibs_amd.SurfaceTables.from_wout(wout, [s])
    th = bo.theta_grid(513)
    scan = ibs_amd.BallooningScan(ctx, None, th, [s], tables=tabs, device=torch.device("cuda:0"))
    tab = scan.coarse()[0]
    assert np.abs(tab - g5["gam_table"]).max() < TOL
    a0, t0, sig, ij = ibs_amd.pick_start(tab, scan.alpha_scan, scan.theta0_scan)
    assert ij == tuple(int(v) for v in g5["argmax"])
    # objective + gradient at the reference's first L-BFGS-B evaluation point
    x0 = g5["trace"][0]
    val, jac = scan.obj_w_grad((x0[0], x0[1]), s)
    assert abs(val - x0[2]) < TOL and np.abs(jac - x0[3:5]).max() < 1e-7
    t_opt, a_opt, gam_opt, res = scan.refine(s, a0, t0)
    assert gam_opt >= tab.max() - 1e-9
    assert abs(gam_opt - float(g5["gam_opt"])) < 2e-6          # same local maximum as the reference run


def test_F2_batched_refinement_matches_per_surface_lbfgsb(ctx, bo):
    """all surfaces refined in lockstep (one batched launch per evaluation) against the per-surface
    scipy L-BFGS-B path of ball_scan.py:307-314"""
    import ibs_amd
    import torch
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    svals = np.array([0.6, 0.8483, 0.9])
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = bo.theta_grid(513)
    scan = ibs_amd.BallooningScan(ctx, None, th, svals, nalpha=12, ntheta0=8, tables=tabs, device=torch.device("cuda:0"))
    tabs_c = scan.coarse()
    starts = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in tabs_c])
    xo, fo, nev = scan.refine_batched(starts)
    assert nev < 120
    for k, s in enumerate(svals):
        t_opt, a_opt, gam_opt, res = scan.refine(s, starts[k, 0], starts[k, 1])
        assert -fo[k] >= tabs_c[k].max() - 1e-9                        # never below the coarse maximum
        assert abs(-fo[k] - gam_opt) < 1e-8, (k, -fo[k], gam_opt)       # the same algorithm: same stopping point
    t0, al, gam = scan.run()
    # (device state machines against the host-driven lockstep loop: the two batch their geometry differently -- other lanes
    #  per grid point, i.e. another summation order -- and the end-game of L-BFGS-B reacts to rounding: a different number
    #  of line-search evaluations, stopping points up to ~1e-9 apart in gam; bar 1e-8)
    assert np.abs(gam + fo).max() < 2e-9


def test_F2_device_state_machine_matches_host_driven_refinement(ctx, bo):
    """ibs_refine_f64 (quasi-Newton state machine on the device, no host round trip per evaluation) reaches the same
    optimum as the host-driven refine_batched; host-pointer and device-pointer calls agree; the reference's refined
    maximum of G5 is reached."""
    import ibs_amd
    import torch
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    svals = np.array([0.6, float(g5["s"]), 0.9])
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = bo.theta_grid(513)
    scan = ibs_amd.BallooningScan(ctx, None, th, svals, tables=tabs, device=torch.device("cuda:0"))
    tabs_c = scan.coarse()
    starts = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in tabs_c])
    xh, fh, nev_h = scan.refine_batched(starts)
    xd, fd, ne = scan.refine_device(starts)
    # the same algorithm on both sides; on the G5 surface the search ends in a line search that runs on rounding noise
    # (SURVEY A6 note), where the last bit of an evaluation decides between two end points 5e-10 apart in gam
    assert np.abs(fd - fh).max() < 2e-9 and np.abs(xd - xh).max() < 2e-3
    assert ne.min() >= 1 and ne.max() <= 2 + 42 * 30       # trajectories may differ in the last bit (FMA contraction)
    for k in range(len(svals)):
        assert -fd[k] >= tabs_c[k].max() - 1e-9                          # never below the coarse maximum
    assert abs(-fd[1] - float(g5["gam_opt"])) < 2e-6                      # the reference run's refined maximum (G5)
    # objective at the returned optimum, recomputed by the fused kernel
    val, jac = scan.batched_obj_w_grad(np.arange(len(svals)), xd)
    assert np.abs(val - fd).max() < 1e-12
    # host-pointer entry (tables staged by the library) gives the same answer
    x2, f2, ne2, rounds = ctx.refine(tabs, np.arange(len(svals)), starts, th)
    assert np.abs(f2 - fd).max() < 1e-13 and np.array_equal(ne2, ne) and rounds >= ne.max()
    # start on the boundary with the gradient pointing outwards / empty batch
    x3, f3, ne3, _ = ctx.refine(tabs, [1], [[0.0, 0.0]], th)
    assert 0.0 <= x3[0, 0] <= np.pi and 0.0 <= x3[0, 1] <= 0.5 * np.pi and np.isfinite(f3[0])
    x4, f4, ne4, r4 = ctx.refine(tabs, [], np.zeros((0, 2)), th)
    assert x4.shape == (0, 2) and r4 == 0
    with pytest.raises(ibs_amd.IbsError):
        ctx.refine(tabs, [7], [[0.1, 0.1]], th)


def test_warm_started_rescan_is_certified_and_cheaper(ctx, bo):
    """re-scan of a DOF-perturbed equilibrium (sims_runner_NCSX.py:151-276 pattern) warm-started from the
    base scan: same certified result as a cold scan, fewer sweeps; a bad guess only costs sweeps"""
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = g3["geo_513"].copy()
    th = bo.theta_grid(513)
    t0 = np.linspace(0, np.pi / 2, 8)
    a = [np.ascontiguousarray(geo[:, k, :]) for k in range(7)]
    base = ctx.gamma_scan(th[1] - th[0], *a, g3["dPdrho_513"], t0, want_info=True)
    pert = [x.copy() for x in a]
    pert[4] *= 1.002; pert[5] *= 1.002; pert[6] *= 1.002; pert[2] *= 0.999          # rel 2e-3 (create_dict.py:70)
    cold = ctx.gamma_scan(th[1] - th[0], *pert, g3["dPdrho_513"], t0, want_info=True)
    warm = ctx.gamma_scan(th[1] - th[0], *pert, g3["dPdrho_513"], t0, want_info=True,
                          lam_guess=base["lam"], guess_width=2e-5)
    assert warm["nbad"] == 0 and np.abs(warm["gam"] - cold["gam"]).max() < 1e-10
    assert np.abs(warm["lam"] - cold["lam"]).max() < 1e-10
    it_c = (cold["info"] & 0xffff).mean(); it_w = (warm["info"] & 0xffff).mean()
    assert it_w < 0.75 * it_c, (it_w, it_c)
    bad = ctx.gamma_scan(th[1] - th[0], *pert, g3["dPdrho_513"], t0, want_info=True,
                         lam_guess=base["lam"] - 0.05, guess_width=1e-6)
    assert np.abs(bad["gam"] - cold["gam"]).max() < 1e-10


def test_scan_plan_packed_argmax(ctx, bo):
    """the pre-marshalled step (scan + packed per-surface argmax) used by bench.py"""
    import ibs_amd
    import torch
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    dev = torch.device("cuda:0")
    th = bo.theta_grid(513)
    geo7 = [torch.from_numpy(np.ascontiguousarray(g5["geo"][:, k, :])).to(dev) for k in range(7)]
    plan = ibs_amd.ScanPlan(ctx, th[1] - th[0], geo7, torch.from_numpy(g5["dPdrho"]).to(dev),
                            torch.from_numpy(g5["theta0_scan"]).to(dev), n_surf=1)
    plan()
    torch.cuda.synchronize()
    assert np.abs(plan.gam.cpu().numpy() - g5["gam_table"]).max() < TOL
    assert divmod(int(plan.best_idx[0].item()), 15) == tuple(int(v) for v in g5["argmax"])
    assert abs(plan.best_val[0].item() - g5["gam_table"].max()) < TOL


def test_config4_adjoint_step_emulated_dofs(ctx, bo):
    """BASELINE config 4 (FD gradient of the ballooning objective over boundary DOFs) with the DOF
    perturbations emulated on the NCSX_op tables (boundary rows of rmnc/zmns changed by
    abs 1e-3 / rel 2e-3, create_dict.py:67-70): GPU pipeline (geometry -> scan -> per-surface max ->
    objective -> forward differences, sims_runner_NCSX.py:249-261) against the same pipeline on the oracles."""
    import ibs_amd
    import torch
    from oracle import geometry_oracle as go
    wout0 = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    svals = np.array([0.6, 0.85])
    N, na, nt = 257, 4, 3
    th = bo.theta_grid(N)
    alphas = np.linspace(0, np.pi, na)
    t0s = np.linspace(0, np.pi / 2, nt)
    dofs = [("rmnc", 3), ("zmns", 5), ("rmnc", 14)]
    steps = [1.0]
    wouts = [wout0]
    for name, k in dofs:
        w = {kk: (v.copy() if isinstance(v, np.ndarray) else v) for kk, v in wout0.items()}
        x = w[name][k, -1]
        step = 1e-3 if abs(x) <= 1e-2 else 2e-3 * x
        w[name][k, :] = w[name][k, :] + step * np.linspace(0, 1, w[name].shape[1]) ** 2   # boundary change, smooth inward
        wouts.append(w)
        steps.append(step)
    dev = torch.device("cuda:0")
    gam_gpu = np.zeros((len(wouts), len(svals)))
    gam_cpu = np.zeros_like(gam_gpu)
    for i, w in enumerate(wouts):
        tabs = ibs_amd.SurfaceTables.from_wout(w, svals)
        surf = np.repeat(np.arange(len(svals)), na)
        r = ctx.fieldline_geometry(tabs, surf, np.tile(alphas, len(svals)), th, device=dev)
        sc = ctx.gamma_scan(th[1] - th[0], *[r["geo"][k] for k in range(7)], r["dPdrho"], torch.from_numpy(t0s).to(dev))
        idx, val = ctx.surface_argmax(sc["gam"].reshape(len(svals), -1))
        gam_gpu[i] = val.cpu().numpy()
        tab_o = go.surface_tables_from_wout(w, svals)
        for js in range(len(svals)):
            geo = go.fieldline_geometry(tab_o, js, alphas, th)
            dP = np.array([bo.dPdrho_of(g[2], g[7], g[0]) for g in geo])
            gam_cpu[i, js] = bo.coarse_scan(th, geo[:, :7], dP, t0s).max()
    assert np.abs(gam_gpu - gam_cpu).max() < TOL
    f_other = np.full(len(wouts), 0.8)
    f_gpu = ibs_amd.ballooning_objective(f_other, gam_gpu, gamma_thresh=-2e-3, prefac=50.0)
    f_cpu = ibs_amd.ballooning_objective(f_other, gam_cpu, gamma_thresh=-2e-3, prefac=50.0)
    d_gpu = ibs_amd.dof_fd_gradient(f_gpu, steps)
    d_cpu = ibs_amd.dof_fd_gradient(f_cpu, steps)
    assert np.abs(d_gpu).max() > 0 and np.abs(d_gpu - d_cpu).max() < 1e-4 * max(1.0, np.abs(d_cpu).max())


def test_gamma_ball_full_nonuniform_grid_regrids_like_reference(ctx, bo):
    """non-uniform theta_PEST: utils.py:1567-1576 regrid semantics (oracle = same np.interp restatement)"""
    import ibs_amd
    N = 257
    tu = bo.theta_grid(N)
    th = tu + 0.35 * (tu[1] - tu[0]) * np.sin(3 * tu)             # monotone, non-uniform, same end points
    th[0], th[-1] = tu[0], tu[-1]
    lam_ = 1.0 * th - 0.8 * np.sin(th)
    gds2 = 1 + lam_ ** 2
    cv = 0.8 * (np.cos(th) + np.sin(th) * lam_)
    B = 1 + 0.1 * np.cos(th)
    gp = np.ones(N)
    out = ibs_amd.gamma_ball_full(-1.0, th, B, gp, cv, gds2, ctx=ctx)
    ref = bo.gamma_ball_full(-1.0, th, B, gp, cv, gds2)
    assert abs(out[0] - ref[0]) < 1e-10
    for a, b in zip(out[1:], ref[1:]):
        assert np.abs(a - b).max() < 1e-7
    uni = ibs_amd.gamma_ball_full(-1.0, tu, B, gp, cv, gds2, ctx=ctx)
    assert abs(uni[0] - out[0]) > 1e-6                             # the regrid actually matters here


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("ns,na,nt0,N", [(16, 8, 8, 513), (5, 24, 15, 969), (3, 5, 7, 257), (32, 8, 16, 1025)])
def test_fused_scan_argmax_equals_two_launches(ctx, bo, ns, na, nt0, N, mode):
    """ibs_gamma_scan_argmax_f64 (scan whose epilogue reduces every completed surface: agent-scope release / acquire on a
    per-surface arrival counter) against scan + ibs_surface_argmax_pack_f64: bitwise, over many launches that alternate
    between two geometries (a stale read of the other launch's growth rates would show), with the surfaces' blocks
    finishing unevenly (ragged theta0 counts, different sweep counts per line)."""
    import ibs_amd
    import torch
    dev = torch.device("cuda:0")
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    src_th = bo.theta_grid(513)
    th = bo.theta_grid(N)
    rng = np.random.default_rng(ns * 1000 + N)
    plans = []
    for variant in range(2):
        geo = np.stack([[np.interp(th, src_th, g3["geo_513"][l % 16, k]) for k in range(8)] for l in range(ns * na)])
        geo[:, 4:7] *= (1 + rng.uniform(-0.08, 0.08, len(geo)))[:, None, None]
        geo[:, 2:4] *= (1 + rng.uniform(-0.08, 0.08, len(geo)))[:, None, None]
        dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        plans.append(ibs_amd.ScanPlan(ctx, th[1] - th[0], [t(geo[:, k]) for k in range(7)], t(dP),
                                      t(np.linspace(0, np.pi / 2, nt0)), ns))
    want = []
    for p in plans:
        p.scan(); p.argmax()
        torch.cuda.synchronize()
        want.append((p.pack.clone(), p.gam.clone()))
        p.pack.fill_(-7.0)
    assert not torch.equal(want[0][0], want[1][0])
    ctx.set_option("pack_mode", mode)              # 0 = the library's choice, 1 = write-through + sc1 loads, 2 = fences
    got = []
    reps = 60 if ns * na * nt0 <= 4096 else 6
    for r in range(reps):
        for k, p in enumerate(plans):
            p.scan_argmax()
            got.append((k, p.pack.clone()))        # (stream-ordered copy)
    torch.cuda.synchronize()
    for k, pk in got:
        assert torch.equal(pk, want[k][0])
    for k, p in enumerate(plans):
        assert torch.equal(p.gam, want[k][1])
        idx = p.pack[:, 1].long().cpu().numpy()
        tab = p.gam.reshape(ns, -1).cpu().numpy()
        assert np.array_equal(idx, tab.argmax(axis=1))                   # first maximum (ball_scan.py:283-288)


@pytest.mark.parametrize("N", [1475, 1537, 1985, 2049])
def test_row_streamed_raw_kernel_matches_three_row_staging(ctx, bo, N):
    """k_solve_gcf_rows (long grids: g, c, f streamed through ONE LDS row per wave, set-up and growth rate in passes)
    against k_solve_gcf (three rows staged side by side) and the oracle: lam, gam, X, dX, flags; ragged batch sizes"""
    rng = np.random.default_rng(N)
    n = 37
    params = np.stack([rng.uniform(0.3, 2, n), rng.uniform(0.2, 1.2, n), rng.uniform(0, 1.5, n)], 1)
    th, g, c = salpha_batch(bo, N, params)
    f = g * (1 + 0.3 * np.sin(th)[None] ** 2)
    h = th[1] - th[0]
    ctx.set_option("force_p", 64)
    ctx.set_option("gcf_rows", 0)
    ref = ctx.solve_gcf(h, g, c, f, want_X=True, want_info=True)
    ctx.set_option("gcf_rows", 1)
    new = ctx.solve_gcf(h, g, c, f, want_X=True, want_info=True)
    assert ref["nbad"] == 0 and new["nbad"] == 0
    normA = 4.0 / h ** 2 + 4.0                    # lam is certified to 256 ulp(||A||) = 5.7e-14 ||A||; gam is second order
    assert np.abs(new["lam"] - ref["lam"]).max() < 2e-13 * normA and np.abs(new["gam"] - ref["gam"]).max() < 1e-10
    assert np.abs(new["X"] - ref["X"]).max() < 1e-6 and np.abs(new["dX"] - ref["dX"]).max() < 1e-5
    for k in (0, 17, 36):
        go, lo, Xo, dXo = bo.solve_gcf(th, g[k], c[k], f[k])
        assert abs(new["gam"][k] - go) < 1e-9 and abs(new["lam"][k] - lo) < 1e-9
        assert np.abs(new["X"][k] - Xo).max() < max(1e-6, eigvec_tol(bo, th, g[k], c[k], f[k]))
    bad = g.copy(); bad[5, 100] = -1.0; bad[9, 7] = np.nan
    r = ctx.solve_gcf(h, bad, c, f, want_info=True)
    flagged = np.nonzero((r["info"] >> 16) != 0)[0]
    assert list(flagged) == [5, 9] and np.abs(np.delete(r["gam"], [5, 9]) - np.delete(new["gam"], [5, 9])).max() < 1e-12


def test_native_rccl_allgather_single_rank():
    """ibs_comm_* (ncclAllGather issued by the library on the context's stream): a one-rank communicator on this GPU --
    the gathered buffer equals what was sent, the call is ordered after the kernel that produced it, a second init and
    a gather without communicator are refused.  (More ranks need more GPUs: the N > 1 logic runs under gloo in the CPU suite.)"""
    import ibs_amd
    import torch
    c = ibs_amd.Context(0)
    dev = torch.device("cuda:0")
    send = torch.zeros(48, dtype=torch.float64, device=dev)
    recv = torch.full((48,), -1.0, dtype=torch.float64, device=dev)
    with pytest.raises(ibs_amd.IbsError):
        c.allgather(send, recv)
    c.comm_init(None, 0, 1)
    for k in range(5):
        send.copy_(torch.arange(48, dtype=torch.float64, device=dev) + k)       # stream-ordered producer
        c.allgather(send, recv)
        assert torch.equal(recv.cpu(), torch.arange(48, dtype=torch.float64) + k)
    with pytest.raises(ibs_amd.IbsError):
        c.comm_init(None, 0, 1)
    full = ibs_amd.gather_rows_tensor(send.reshape(16, 3), 16, 0, 1, None, c)
    assert torch.equal(full, send.reshape(16, 3))
    # overlapped form: gathers on the communicator's own stream, two slots in flight while the compute stream goes on;
    # comm_wait(slot) orders the compute stream after a gather before its buffers are reused / read
    sends = [torch.zeros(48, dtype=torch.float64, device=dev) for _ in range(2)]
    recvs = [torch.full((48,), -1.0, dtype=torch.float64, device=dev) for _ in range(2)]
    big = torch.zeros(1 << 22, dtype=torch.float64, device=dev)
    for k in range(12):
        slot = k & 1
        c.comm_wait(slot)
        if k >= 2:
            assert float(recvs[slot][5].item()) == 5.0 + (k - 2)                # the gather of step k-2 has landed
        big.add_(1.0)                                                            # (keeps the compute stream busy)
        sends[slot].copy_(torch.arange(48, dtype=torch.float64, device=dev) + k)
        c.allgather_start(sends[slot], recvs[slot], slot)
    c.comm_wait()
    assert torch.equal(recvs[0].cpu(), torch.arange(48, dtype=torch.float64) + 10)
    assert torch.equal(recvs[1].cpu(), torch.arange(48, dtype=torch.float64) + 11)
    with pytest.raises(ibs_amd.IbsError):
        c.allgather_start(sends[0], recvs[0], 16)                                # slot out of range (16 slots)
    with pytest.raises(ibs_amd.IbsError):
        c.allgather_start(sends[0], recvs[0], 1, then_wait=1)                    # a gather cannot wait for itself
    for k in range(6):                                                           # the one-call form bench.py uses
        slot = k & 1
        sends[slot].copy_(torch.arange(48, dtype=torch.float64, device=dev) + 100 + k)
        c.allgather_start(sends[slot], recvs[slot], slot, then_wait=1 - slot)
        if k >= 1:
            assert float(recvs[1 - slot][0].item()) == 100.0 + (k - 1)
    c.comm_wait()
    assert float(recvs[1][0].item()) == 105.0
    # three slots with the HOST-side wait (no wait on the compute stream): before slot s is reused its gather has finished
    sends.append(torch.zeros(48, dtype=torch.float64, device=dev)); recvs.append(torch.zeros(48, dtype=torch.float64, device=dev))
    for k in range(9):
        slot = k % 3
        big.add_(1.0)
        sends[slot].copy_(torch.arange(48, dtype=torch.float64, device=dev) + 200 + k)
        c.allgather_start(sends[slot], recvs[slot], slot, host_wait=(k + 1) % 3)
        if k >= 2:                                                               # (slot (k+1)%3 = the gather of step k-2)
            assert float(recvs[(k + 1) % 3][0].item()) == 200.0 + (k - 2)
    c.comm_wait()
    assert float(recvs[2][0].item()) == 208.0
    with pytest.raises(ibs_amd.IbsError):
        c.allgather_start(sends[0], recvs[0], 0, host_wait=0)                    # cannot wait for itself
    c.comm_wait(3)                                                               # nothing pending there: no-op
    c.comm_destroy()
    c.close()


def test_new_entry_points_reject_bad_arguments(ctx, bo):
    """error behaviour of the round-2 entry points: negative return codes / IbsError, nothing launched"""
    import ctypes as C
    import ibs_amd
    import torch
    dev = torch.device("cuda:0")
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = g3["geo_513"][:6]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)
    th = bo.theta_grid(513)
    with pytest.raises(ibs_amd.IbsError):                     # 6 lines do not split into 4 surfaces
        ibs_amd.ScanPlan(ctx, th[1] - th[0], [t(geo[:, k]) for k in range(7)], t(dP), t(np.linspace(0, 1, 4)), 4)
    plan = ibs_amd.ScanPlan(ctx, th[1] - th[0], [t(geo[:, k]) for k in range(7)], t(dP), t(np.linspace(0, 1, 4)), 3)
    lib = ctx._lib
    args = list(plan._fused_args[0])
    args[15] = 4                                              # n_surf = 4: n_lines % n_surf != 0
    assert lib.ibs_gamma_scan_argmax_f64(*args) == -1 and b"n_surf" in lib.ibs_last_error()
    args = list(plan._fused_args[0]); args[18] = C.c_void_p(None)   # pack = NULL
    assert lib.ibs_gamma_scan_argmax_f64(*args) == -1
    with pytest.raises(ibs_amd.IbsError):
        ctx.set_option("no_such_option", 1)
    assert lib.ibs_comm_allgather_f64(ctx._h, C.c_void_p(8), C.c_void_p(8), 1) < 0      # no communicator
    assert lib.ibs_lbfgsb2_init(None, None, None, None, 1e-9, 1e-9, 10, 20) < 0
    plan.scan_argmax()                                        # and the plan still works
    torch.cuda.synchronize()
    assert np.array_equal(plan.pack[:, 1].long().cpu().numpy(), plan.gam.reshape(3, -1).argmax(dim=1).cpu().numpy())
