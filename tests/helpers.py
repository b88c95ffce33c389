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


from oracle.pipeline import oracle_surface_pipeline  # noqa: E402,F401  (the reference's per-surface worker with the oracle on every link)
