"""TEST INFRASTRUCTURE (checker only; never imported by the product path): the reference's per-surface worker
(ball_scan.py:248-339) with the CPU oracle on every link.  Used by tests/ and by bench.py's parity check of the
configs[3] leg."""
import numpy as np

from oracle import ballooning_oracle as bo


def oracle_surface_pipeline(wout, s, theta, nalpha=24, ntheta0=15, del_alpha=0.004, start=None):
    """The reference's per-surface worker (ball_scan.py:248-339) with the oracle on every link: numpy geometry (utils.py:359-720
    restated in oracle/geometry_oracle.py) -> C-oracle coarse scan -> first maximum -> scipy L-BFGS-B (the reference's own
    optimizer call, ball_scan.py:307-314) on the oracle's obj_w_grad -> final solve at the optimum.
    Returns dict(table, start, x_opt, gam, nfev).  `start` = (alpha0, theta0_0) skips the coarse scan."""
    from scipy.optimize import minimize
    from oracle import c_oracle as co
    from oracle import geometry_oracle as go
    tab = go.surface_tables_from_wout(wout, np.array([s]))
    h = float(theta[1] - theta[0])
    alpha_scan = np.linspace(0, np.pi, nalpha); theta0_scan = np.linspace(0.0, 0.5 * np.pi, ntheta0)
    table = None
    if start is None:
        geo = go.fieldline_geometry(tab, 0, alpha_scan, theta)                         # (nalpha, 8, N)
        dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)          # ball_scan.py:262
        table, _, _ = co.gamma_scan(h, *[np.ascontiguousarray(geo[:, k]) for k in range(7)], dP, theta0_scan)
        i, j = np.unravel_index(int(np.argmax(table)), table.shape)                    # first maximum (ball_scan.py:283-288)
        start = (float(alpha_scan[i]), float(theta0_scan[j]))

    def obj(x):
        a = float(x[0])
        lines = go.fieldline_geometry(tab, 0, np.array([a - 0.5 * del_alpha, a, a + 0.5 * del_alpha]), theta)
        v, j = bo.obj_w_grad_lines(theta, float(x[1]), lines[0], lines[1], lines[2], del_alpha)
        return float(v), np.asarray(j, dtype=np.float64)

    res = minimize(obj, x0=start, jac=True, bounds=((0.0, np.pi), (0.0, 0.5 * np.pi)),
                   options={"ftol": 5.0e-11, "gtol": 2.0e-08, "maxiter": 30})            # ball_scan.py:307-314
    a, t = float(res.x[0]), float(res.x[1])
    line = go.fieldline_geometry(tab, 0, np.array([a]), theta)[0]
    dP = -0.5 * np.mean((line[2] - line[7]) * line[0] ** 2)
    cv, gd = bo.fold_theta0(t, line[2], line[3], line[4], line[5], line[6])
    gam = bo.gamma_ball_full(dP, theta, line[0], line[1], cv, gd)[0]                     # ball_scan.py:322-339
    return dict(table=table, start=start, x_opt=np.array([a, t]), gam=float(gam), nfev=int(res.nfev))
