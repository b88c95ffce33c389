"""Drop-in operators with the reference's names and signatures (utils.py:1550-1552, 1632).

    from ibs_amd import gamma_ball_full          # instead of `from utils import *` (ball_scan.py:19)

`vguess` only steers ARPACK upstream (utils.py:1597) and is accepted and ignored.  `sigma0` is ARPACK's shift: the reference
returns the eigenpair NEAREST sigma0, this library always the LARGEST eigenvalue's.  The two are the same eigenpair whenever
lam_max < sigma0 -- the case the reference is written for (sigma0 = 1.0 / 1.3 |gam| + 0.05 / 0.42 against growth rates of
1e-4 .. 1e-1: ball_scan.py:230, 289, 337) and the only one observed -- and ONLY then: with lam_max >= sigma0 upstream would
have returned another eigenpair (or lam_max itself, if it happens to be the nearest).  The drop-in reports that case instead of
hiding it: the solve carries the informational status bit 4 (include/ibs.h), `gamma_ball_full(..., info=d)` fills
d["above_sigma0"], and a NearestSigmaWarning is issued.  All eigen-work runs on the GPU through libibs_hip.so; the only host
arithmetic is the elementwise coefficient formulas returned to the caller as (g, c, f).
"""
import warnings

import numpy as np

from .solver import default_context
from ._lib import IbsError


def uniform_spacing(theta, rtol=1e-9):
    """h of a uniform grid (the batched geometry-fed entry points need one; gamma_ball_full also accepts
    non-uniform grids and regrids like the reference)."""
    theta = np.asarray(theta, dtype=np.float64)
    N = len(theta)
    h = (theta[-1] - theta[0]) / (N - 1)
    if np.max(np.abs(np.diff(theta) - h)) > rtol * abs(h) * N:
        raise IbsError("theta grid is not uniform: the batched entry points need a uniform grid (use gamma_ball_full)")
    return float(h)


def is_uniform(theta, rtol=1e-9):
    theta = np.asarray(theta, dtype=np.float64)
    h = (theta[-1] - theta[0]) / (len(theta) - 1)
    return bool(np.max(np.abs(np.diff(theta) - h)) <= rtol * abs(h) * len(theta))


class NearestSigmaWarning(UserWarning):
    """lam_max >= sigma0: utils.py:1597 (ARPACK shift-invert about sigma0) would have returned the eigenpair nearest sigma0"""


def _report_sigma(r, sigma0, info):
    """the library's nearest-sigma flag (status bit 4, option "sigma0") of a one-system solve -> info dict / warning"""
    word = int(np.asarray(r["info"]).ravel()[0])
    lam = float(np.asarray(r["lam"]).ravel()[0])
    above = bool((word >> 16) & 16)
    if info is not None:
        info.update(lam=lam, above_sigma0=above, status=(word >> 16) & 3, sweeps=word & 0xffff)
    if above:
        warnings.warn("lam_max = %.6g >= sigma0 = %.6g: the reference (eigs(..., sigma=sigma0), utils.py:1597) returns the eigenpair "
                      "nearest sigma0 here, this library the largest eigenvalue's" % (lam, sigma0), NearestSigmaWarning, stacklevel=3)


def gamma_ball_full(dPdrho, theta_PEST, B, gradpar, cvdrift, gds2, vguess=None, sigma0=0.42, ctx=None, info=None):
    """reference: utils.py:1550-1624.  Returns (gam, X, dX, g, c, f) with the same meaning.
    Uniform grids go through the fused geometry-fed kernel; a non-uniform theta_PEST is regridded exactly
    as upstream (np.interp of g, c, f onto the uniform grid, g interpolated at the uniform half points,
    utils.py:1567-1576 -- elementwise host glue) and solved by the raw (g, gh, c, f) kernel.
    The returned eigenpair is lam_max's; upstream's is the one nearest sigma0 -- the same whenever lam_max < sigma0.  Otherwise
    (module docstring) a NearestSigmaWarning is issued; info (optional dict) receives lam (the matrix eigenvalue), above_sigma0,
    status, sweeps."""
    ctx = ctx or default_context()
    ctx.set_option("sigma0", float(sigma0))
    try:
        return _gamma_ball_full(ctx, dPdrho, theta_PEST, B, gradpar, cvdrift, gds2, sigma0, info)
    finally:
        ctx.set_option("sigma0", None)


def _gamma_ball_full(ctx, dPdrho, theta_PEST, B, gradpar, cvdrift, gds2, sigma0, info):
    theta = np.asarray(theta_PEST, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    gradpar = np.asarray(gradpar, dtype=np.float64)
    cvdrift = np.asarray(cvdrift, dtype=np.float64)
    gds2 = np.asarray(gds2, dtype=np.float64)
    N = len(B)
    gp = np.abs(gradpar)
    g = gp * gds2 / B                         # utils.py:1560
    c = -1 * dPdrho * cvdrift * 1 / (gp * B)  # utils.py:1561
    f = gds2 / B ** 2 * 1 / (gp * B)          # utils.py:1562
    if is_uniform(theta):
        h = uniform_spacing(theta)
        z = np.zeros((1, N))
        r = ctx.gamma_scan(h, B[None], gradpar[None], cvdrift[None], z, gds2[None], z, z,
                           np.array([float(dPdrho)]), np.zeros(1), want_X=True, want_info=True)
        _report_sigma(r, sigma0, info)
        return float(r["gam"][0, 0]), r["X"][0, 0], r["dX"][0, 0], g, c, f
    tu = np.linspace(theta[0], theta[-1], N)                       # utils.py:1565
    g_u, c_u, f_u = np.interp(tu, theta, g), np.interp(tu, theta, c), np.interp(tu, theta, f)   # utils.py:1567-1571
    th_half = (tu[:-1] + tu[1:]) / 2                                # utils.py:1574
    h = np.diff(th_half)[2]                                         # utils.py:1575
    gh = np.zeros(N)
    gh[:-1] = np.interp(th_half, theta, g)                          # utils.py:1576
    r = ctx.solve_gcf(h, g_u[None], c_u[None], f_u[None], want_X=True, gh=gh[None], want_info=True)
    _report_sigma(r, sigma0, info)
    return float(r["gam"][0]), r["X"][0], r["dX"][0], g_u, c_u, f_u


def dPdrho_of(cvdrift, gbdrift, bmag):
    """ball_scan.py:262 / utils.py:1657"""
    return -1.0 * 0.5 * np.mean((cvdrift - gbdrift) * bmag ** 2)


def make_obj_w_grad(fieldlines, ctx=None, del_alpha=0.004):
    """Factory for a drop-in `obj_w_grad(x0, vs, rho_val, theta, vguess00, sigma00=0.42)` (utils.py:1632).

    `fieldlines(vs, rho_val, alphas, theta)` supplies the geometry exactly as the reference gets it from
    `vmec_fieldlines(vs, rho_val, alphas, theta1d=theta)` (utils.py:1641-1646) and must return an array
    (3, 8, N) in scan.GEO_ORDER.  With the reference available:
        fl = lambda vs, s, al, th: np.stack([[getattr(utils.vmec_fieldlines(vs, s, al, theta1d=th), k)[0][i]
                                             for k in GEO_ORDER] for i in range(3)])
    Returns (-gam, array([-dgam/dalpha, -dgam/dtheta0])) like utils.py:1728 (scipy jac=True convention)."""
    def obj_w_grad(x0, vs, rho_val, theta, vguess00=None, sigma00=0.42):
        c = ctx or default_context()
        alpha_val, theta0_val = float(x0[0]), float(x0[1])
        al = np.array([alpha_val - 0.5 * del_alpha, alpha_val, alpha_val + 0.5 * del_alpha])
        geo = np.asarray(fieldlines(vs, rho_val, al, theta), dtype=np.float64)
        val, jac = c.obj_w_grad(uniform_spacing(theta), geo[None], np.array([theta0_val]), del_alpha)
        return float(val[0]), np.asarray(jac[0], dtype=np.float64)
    return obj_w_grad


def theta_grid(N, theta_fac=4):
    """the theta_PEST grid of ball_scan.py:201-209: N points on [-theta_fac pi, theta_fac pi]"""
    return np.linspace(-theta_fac * np.pi, theta_fac * np.pi, int(N))
