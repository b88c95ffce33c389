"""Test-only helpers: an oracle-backed stand-in for ibs_amd.Context (CPU tests of the host driver)
and a synthetic field-line geometry family.  Never imported by the product path."""
import numpy as np

from oracle import ballooning_oracle as bo


class OracleContext:
    """same method surface as ibs_amd.Context, computed by the CPU oracle"""

    def gamma_scan(self, h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, dPdrho, theta0, **kw):
        nl, N = bmag.shape
        th = np.linspace(-h * (N - 1) / 2, h * (N - 1) / 2, N)
        gam = np.zeros((nl, len(theta0)))
        for i in range(nl):
            for j, t0 in enumerate(theta0):
                cv, gd = bo.fold_theta0(t0, cvdrift[i], cvdrift0[i], gds2[i], gds21[i], gds22[i])
                gam[i, j] = bo.gamma_ball_full(dPdrho[i], th, bmag[i], gradpar[i], cv, gd)[0]
        return dict(gam=gam, lam=gam.copy(), nbad=0)

    def obj_w_grad(self, h, geo, theta0, del_alpha=0.004):
        n, _, _, N = geo.shape
        th = np.linspace(-h * (N - 1) / 2, h * (N - 1) / 2, N)
        val = np.zeros(n); jac = np.zeros((n, 2))
        for k in range(n):
            val[k], jac[k] = bo.obj_w_grad_lines(th, theta0[k], geo[k, 0], geo[k, 1], geo[k, 2], del_alpha)
        return val, jac

    def surface_argmax(self, gam):
        idx = np.array([int(np.argmax(row)) for row in gam], dtype=np.int32)
        return idx, np.array([row.max() for row in gam])


def synthetic_fieldlines(theta):
    """s-alpha-like geometry with a field-line-label dependence, polynomial in theta0 like the real
    (gds2, gds21, gds22) / (cvdrift, cvdrift0) family.  Returns fieldlines(s, alphas) -> (nalpha, 8, N)."""
    theta = np.asarray(theta)

    def fieldlines(s, alphas):
        out = []
        for a in np.atleast_1d(alphas):
            shat = 0.4 + 1.2 * s
            am = (0.5 + 0.8 * s) * (1 + 0.35 * np.cos(a - 0.9) + 0.1 * np.cos(2 * a))
            lam0 = shat * theta - am * np.sin(theta)
            bmag = 1 + 0.1 * s * np.cos(theta)
            gradpar = np.ones_like(theta) * (1.0 + 0.05 * np.cos(a))
            gds2 = 1 + lam0 ** 2
            gds21 = -shat * lam0
            gds22 = np.full_like(theta, shat ** 2)
            cvdrift = am * (np.cos(theta) + np.sin(theta) * lam0)
            cvdrift0 = -am * shat * np.sin(theta)
            gbdrift = cvdrift - 2.0 / bmag ** 2          # => dPdrho = -0.5*mean(2) = -1
            out.append(np.stack([bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift]))
        return np.stack(out)

    return fieldlines


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
