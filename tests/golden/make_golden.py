#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the upstream reference *in place* from /root/reference (read-only) and
records input/output vectors of its hot path as small .npz fixtures next to this
script.  Nothing of the reference (source, bytecode) is written into the repo:
the fixtures hold numbers only.

Reference entry points exercised (file:line relative to /root/reference):
  utils.py:1550-1624  gamma_ball_full      -> G1, G3, G5, G6
  utils.py:1632-1728  obj_w_grad           -> G4, G5 (refine trace)
  utils.py:37-158     vmec_splines         -> producer of the NCSX geometry in G3-G5
  utils.py:161-864    vmec_fieldlines      -> producer of the NCSX geometry in G3-G5
  tests/shifted-circle-s-alpha/bishop_ball_s-alpha.py:20-209  check_ball,
      check_ball_long                       -> G2 (only the two defs are exec'd;
      importing that module would start its 60,000-case multiprocessing scan)
  tests/comparn_w_COBRAVMEC/gamma_max_{og,op}.npy -> G7 (data copied as numbers)

In-process shims needed to import utils.py here (no reference file is modified):
  * scipy.integrate.simps was removed in scipy >= 1.14 -> alias to simpson
  * simsopt is not installed -> empty stub modules providing a dummy Vmec class
  * the equilibrium comes from tests/comparn_w_COBRAVMEC/wout_NCSX_op.nc
    (NetCDF-3) through scipy.io.netcdf_file and a duck-typed Vmec object.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [G1 G2 ...]
"""
import os
import sys
import time
import types

import numpy as np
import scipy
import scipy.integrate

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

if not hasattr(scipy.integrate, "simps"):
    scipy.integrate.simps = scipy.integrate.simpson
for _name in ("simsopt", "simsopt.mhd", "simsopt.mhd.vmec"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules["simsopt.mhd.vmec"].Vmec = type("Vmec", (), {})
sys.path.insert(0, REF)
import utils as ref  # noqa: E402  (the reference, imported in place)

META = dict(numpy=np.__version__, scipy=scipy.__version__,
            reference="rahulgaur104/ideal-ballooning-solver @ 2025-07-18 (v0.2.0)")


class tight_arpack:
    """Context: re-run the reference with its ARPACK call converged (tol 1e-14 instead of the
    hard-coded 5e-7 of utils.py:1597) by wrapping the `eigs` name inside the imported module.
    The reference file is untouched; this separates ARPACK noise from formula parity."""

    def __enter__(self):
        self._orig = ref.eigs

        def eigs_tight(A, k, **kw):
            kw["tol"] = 1e-14
            kw["maxiter"] = 20000
            return self._orig(A, k, **kw)

        ref.eigs = eigs_tight

    def __exit__(self, *a):
        ref.eigs = self._orig


def save(name, **arrays):
    path = os.path.join(OUT, name)
    arrays["meta"] = np.array(repr(META))
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0), flush=True)


def theta_grid(N, fac=4):
    return np.linspace(-fac * np.pi, fac * np.pi, N)


def vguess_of(theta, fac=4):
    # ball_scan.py:209 / :233
    return (1 - np.tanh(theta[1:-1] / np.pi) ** 2) * np.cos(theta[1:-1] / (2 * fac))


def salpha_coeffs(theta, shat, alpha, theta0):
    # analytic s-alpha coefficients, bishop_ball_s-alpha.py:30-45
    lam = shat * (theta - theta0) - alpha * (np.sin(theta) - np.sin(theta0))
    g = 1 + lam ** 2
    c = alpha * (np.cos(theta) + np.sin(theta) * lam)
    return g, c


# ----------------------------------------------------------------------------------------------
def make_G1():
    """s-alpha solves through gamma_ball_full with B = gradpar = 1, dPdrho = -1,
    cvdrift = c, gds2 = g  (=> g_ref = g, c_ref = c, f_ref = g)."""
    shats = [0.3, 0.5, 1.0, 1.5]
    alphas = [0.3, 0.6, 0.8, 1.1]
    theta0s = [0.0, 0.1, 0.3]
    Ns = [257, 513, 1025]
    params, gams = [], []
    keepX = {}
    for N in Ns:
        th = theta_grid(N)
        vg = vguess_of(th)
        one = np.ones(N)
        for sh in shats:
            for al in alphas:
                for t0 in theta0s:
                    g, c = salpha_coeffs(th, sh, al, t0)
                    gam, X, dX, gg, cc, ff = ref.gamma_ball_full(-1.0, th, one, one, c, g, vg, 1.0)
                    params.append((N, sh, al, t0))
                    gams.append(gam)
                    if (sh, al) in ((1.0, 0.8), (0.5, 0.3)) and t0 in (0.0, 0.3):
                        keepX["X_%d_%g_%g_%g" % (N, sh, al, t0)] = X
                        keepX["dX_%d_%g_%g_%g" % (N, sh, al, t0)] = dX
    save("G1_salpha.npz", params=np.array(params), gam=np.array(gams), **keepX)


def make_G2():
    """Stability booleans of the reference's own s-alpha test functions."""
    src = open(os.path.join(REF, "tests/shifted-circle-s-alpha/bishop_ball_s-alpha.py")).read().split("\n")
    # function definitions only (the module body below line 211 launches a 40-process scan)
    defs = "\n".join(src[:211])
    ns = {}
    exec(compile(defs, "bishop_defs", "exec"), ns)
    shats = np.linspace(0.1, 1.9, 10)
    alphas = np.linspace(0.1, 1.5, 8)
    theta0s = [0.0, 0.1, 0.2]
    rows = []
    for t0 in theta0s:
        for sh in shats:
            for al in alphas:
                rows.append((sh, al, t0, ns["check_ball"](sh, al, t0), ns["check_ball_long"](sh, al, t0)))
    save("G2_salpha_stability.npz", table=np.array(rows))


# ----------------------------------------------------------------------------------------------
class _Wout:
    pass


class DuckVmec:
    """Duck-typed stand-in for simsopt.mhd.vmec.Vmec over a NetCDF-3 wout file
    (what vmec_splines touches: utils.py:46-135)."""

    def __init__(self, fname):
        from scipy.io import netcdf_file
        f = netcdf_file(fname, "r", mmap=False)
        w = _Wout()
        for k in ("rmnc", "zmns", "lmns", "gmnc", "bmnc", "bsupumnc", "bsupvmnc", "bsubsmns", "bsubumnc", "bsubvmnc"):
            setattr(w, k, np.array(f.variables[k][()], dtype=float).T.copy())  # (mn, ns)
        for k in ("pres", "chi", "iotas", "phi", "xm", "xn", "xm_nyq", "xn_nyq", "raxis_cc"):
            setattr(w, k, np.array(f.variables[k][()], dtype=float))
        for k in ("Aminor_p",):
            setattr(w, k, float(f.variables[k][()]))
        for k in ("mnmax", "mnmax_nyq", "nfp", "mpol", "ntor", "ns"):
            setattr(w, k, int(f.variables[k][()]))
        f.close()
        self.wout = w
        self.s_full_grid = np.linspace(0, 1, w.ns)
        ds = self.s_full_grid[1] - self.s_full_grid[0]
        self.s_half_grid = self.s_full_grid[1:] - 0.5 * ds

    def run(self):
        pass


_VS = None


def ncsx_splines():
    global _VS
    if _VS is None:
        v = DuckVmec(os.path.join(REF, "tests/comparn_w_COBRAVMEC/wout_NCSX_op.nc"))
        _VS = ref.vmec_splines(v)
    return _VS


GEO_KEYS = ("bmag", "gradpar_theta_pest", "cvdrift", "cvdrift0", "gds2", "gds21", "gds22", "gbdrift")


def line_geometry(vs, s, alphas, theta):
    """returns array (nalpha, 8, N) in GEO_KEYS order (ball_scan.py:251-261)."""
    fl = ref.vmec_fieldlines(vs, s, np.atleast_1d(alphas), theta1d=theta)
    return np.stack([np.stack([getattr(fl, k)[0][ia] for k in GEO_KEYS]) for ia in range(len(np.atleast_1d(alphas)))])


def solve_line(geo, theta, theta0, vguess, sigma):
    bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift = geo
    dPdrho = -1.0 * 0.5 * np.mean((cvdrift - gbdrift) * bmag ** 2)  # ball_scan.py:262
    cv = cvdrift + theta0 * cvdrift0                                  # ball_scan.py:267
    gd = gds2 + 2 * theta0 * gds21 + theta0 ** 2 * gds22              # ball_scan.py:268
    return (dPdrho,) + tuple(ref.gamma_ball_full(dPdrho, theta, bmag, gradpar, cv, gd, vguess, sigma))


def make_G3():
    vs = ncsx_splines()
    svals = [0.5, 0.7, 0.85, 0.95]
    avals = [0.0, 1.0, 2.0, np.pi]
    t0vals = [0.0, 0.5, 1.0, 0.5 * np.pi]
    out = {}
    for N, ss, aa in ((513, svals, avals), (969, svals[1:3], avals[1:3]), (1025, [0.5, 0.85, 0.95], [0.0, 2.0])):
        th = theta_grid(N)
        vg = vguess_of(th)
        geos, gams, dps, lines, Xs, dXs, gams_t = [], [], [], [], [], [], []
        for s in ss:
            G = line_geometry(vs, s, aa, th)
            for ia, a in enumerate(aa):
                geos.append(G[ia])
                lines.append((s, a))
                row = []
                for t0 in t0vals:
                    dP, gam, X, dX, g, c, f = solve_line(G[ia], th, t0, vg, 1.0)
                    row.append(gam)
                    if t0 == 0.5 and a in (0.0, 2.0):
                        Xs.append(X); dXs.append(dX)
                gams.append(row)
                dps.append(dP)
                with tight_arpack():
                    gams_t.append([solve_line(G[ia], th, t0, vg, 1.0)[1] for t0 in t0vals])
            print("G3 N=%d s=%g done" % (N, s), flush=True)
        out["geo_%d" % N] = np.array(geos)
        out["lines_%d" % N] = np.array(lines)
        out["gam_%d" % N] = np.array(gams)
        out["gam_tight_%d" % N] = np.array(gams_t)
        out["dPdrho_%d" % N] = np.array(dps)
        out["X_%d" % N] = np.array(Xs)
        out["dX_%d" % N] = np.array(dXs)
    save("G3_ncsx_lines.npz", theta0=np.array(t0vals), keys=np.array(GEO_KEYS), **out)


def make_G4():
    """obj_w_grad value + jacobian with the three-line geometry it saw (utils.py:1632-1728)."""
    vs = ncsx_splines()
    N = 513
    th = theta_grid(N)
    vg = vguess_of(th)
    pts = [(0.5, 0.3, 0.2), (0.5, 2.0, 1.0), (0.7, 1.0, 0.5), (0.85, 2.0, 1.0), (0.85, 0.7, 1.3), (0.95, 1.0, 0.5),
           (0.95, 2.9, 0.1)]
    geos, vals, jacs, vals_t, jacs_t = [], [], [], [], []
    for s, a, t0 in pts:
        val, jac = ref.obj_w_grad((a, t0), vs, s, th, vg, 1.0)
        with tight_arpack():
            vt, jt = ref.obj_w_grad((a, t0), vs, s, th, vg, 1.0)
        vals_t.append(vt)
        jacs_t.append(jt)
        G = line_geometry(vs, s, np.array([a - 0.002, a, a + 0.002]), th)  # del_alpha = 0.004, utils.py:1639-1646
        geos.append(G)
        vals.append(val)
        jacs.append(jac)
        print("G4", s, a, t0, val, jac, flush=True)
    save("G4_obj_w_grad.npz", pts=np.array(pts), geo=np.array(geos), val=np.array(vals), jac=np.array(jacs),
         val_tight=np.array(vals_t), jac_tight=np.array(jacs_t), keys=np.array(GEO_KEYS), del_alpha=np.array(0.004))


def make_G5():
    """One surface: the coarse 24 x 15 scan of ball_scan.py:223-275 (warm-start chain included),
    the argmax rule of ball_scan.py:279-295 and the L-BFGS-B refinement trace of ball_scan.py:305-339."""
    from scipy.optimize import minimize
    vs = ncsx_splines()
    N = 513
    s = 0.8483
    th = theta_grid(N)
    vguess = vguess_of(th)
    theta0_scan = np.linspace(0.0, 0.5 * np.pi, 15)
    alpha_scan = np.linspace(0, np.pi, 24)
    G = line_geometry(vs, s, alpha_scan, th)
    gam_tab = np.zeros((24, 15))
    vg_tab = np.zeros((24, 15, N - 2))
    dps = np.zeros(24)
    for i in range(24):
        for j in range(15):
            dP, gam, X, dX, g, c, f = solve_line(G[i], th, theta0_scan[j], vguess, 1.0)
            vguess = X[1:-1]
            vg_tab[i, j] = vguess
            gam_tab[i, j] = gam
        dps[i] = dP
        print("G5 alpha row", i, gam_tab[i].max(), flush=True)
    idx = np.where(gam_tab == np.max(gam_tab))
    i0, j0 = idx[0][0], idx[1][0]
    sigma0 = 1.3 * abs(gam_tab[i0, j0]) + 0.05
    trace = []

    def fun(x):
        val, jac = ref.obj_w_grad(x, vs, s, th, vg_tab[i0, j0], sigma0)
        trace.append((x[0], x[1], val, jac[0], jac[1]))
        return val, jac

    res = minimize(fun, x0=(alpha_scan[i0], theta0_scan[j0]), jac=True,
                   bounds=((0.0, np.pi), (0.0, 0.5 * np.pi)),
                   options={"ftol": 5.0e-11, "gtol": 2.0e-08, "maxiter": 30})
    Gf = line_geometry(vs, s, res.x[0], th)
    dP, gam_f, X, dX, g, c, f = solve_line(Gf[0], th, res.x[1], vg_tab[i0, j0], 0.42)
    print("G5 refine", res.x, res.fun, gam_f, len(trace), flush=True)
    save("G5_scan_trace.npz", s=np.array(s), geo=G[:, :7, :], dPdrho=dps, gam_table=gam_tab,
         alpha_scan=alpha_scan, theta0_scan=theta0_scan, argmax=np.array([i0, j0]), sigma0=np.array(sigma0),
         trace=np.array(trace), x_opt=res.x, gam_opt=np.array(gam_f), geo_opt=Gf[0], dPdrho_opt=np.array(dP),
         keys=np.array(GEO_KEYS))


def make_G6():
    """'rough' random tridiagonal systems (SURVEY.md §8d C5-ii envelopes) through gamma_ball_full
    with B = gradpar = 1, dPdrho = -1: g = gds2 ... we need independent g, c, f, so feed
    B, gradpar such that g, c, f come out as drawn:  gradpar = 1, B = sqrt(g/f) => g_ref = gds2/B,
    f_ref = gds2/B^3;  choose gds2 = g*B, then g_ref = g, f_ref = g/B^2 = f;  cvdrift = c*B."""
    out = {}
    for N in (257, 513):
        rng = np.random.default_rng(20240 + (N - 1))
        th = theta_grid(N)
        vg = vguess_of(th)
        nsys = 6
        g = np.exp(rng.uniform(np.log(0.01), np.log(50.0), size=(nsys, N)))
        c = rng.uniform(-2.5, 3.5, size=(nsys, N))
        f = np.exp(rng.uniform(np.log(0.2), np.log(3e3), size=(nsys, N)))
        gams, gams_t, gcf = [], [], []
        for k in range(nsys):
            B = np.sqrt(g[k] / f[k])
            gam, X, dX, gg, cc, ff = ref.gamma_ball_full(-1.0, th, B, np.ones(N), c[k] * B, g[k] * B, vg, 60.0)
            gams.append(gam)
            with tight_arpack():
                gams_t.append(ref.gamma_ball_full(-1.0, th, B, np.ones(N), c[k] * B, g[k] * B, vg, 60.0)[0])
            gcf.append(np.stack([gg, cc, ff]))
        out["gcf_%d" % N] = np.array(gcf)
        out["gam_%d" % N] = np.array(gams)
        out["gam_tight_%d" % N] = np.array(gams_t)
    save("G6_random_rough.npz", **out)


def make_G7():
    og = np.load(os.path.join(REF, "tests/comparn_w_COBRAVMEC/gamma_max_og.npy"))
    op = np.load(os.path.join(REF, "tests/comparn_w_COBRAVMEC/gamma_max_op.npy"))
    save("G7_cobra_pins.npz", gamma_max_og=og, gamma_max_op=op,
         s_og=np.linspace(0.05, 0.995, 48), s_op=np.linspace(0.01, 0.995, 48))


def make_G8():
    """Inputs of the field-line geometry (row F1): (a) the wout tables vmec_splines reads (utils.py:46-135),
    copied as numbers from tests/comparn_w_COBRAVMEC/wout_NCSX_op.nc; (b) the spline-evaluated Fourier
    coefficient vectors and profile scalars at four surfaces exactly as vmec_fieldlines forms them
    (utils.py:311-357).  Outputs to pin against: the geometry arrays already stored in G3."""
    v = DuckVmec(os.path.join(REF, "tests/comparn_w_COBRAVMEC/wout_NCSX_op.nc"))
    w = v.wout
    wout = {k: getattr(w, k) for k in ("rmnc", "zmns", "lmns", "gmnc", "bmnc", "bsupumnc", "bsupvmnc", "bsubsmns",
                                       "bsubumnc", "bsubvmnc", "pres", "chi", "iotas", "phi", "xm", "xn", "xm_nyq",
                                       "xn_nyq")}
    wout.update(Aminor_p=np.array(w.Aminor_p), nfp=np.array(w.nfp), ns=np.array(w.ns))
    save("G8_wout_ncsx_op.npz", **wout)
    vs = ncsx_splines()
    out = {}
    svals = np.array([0.5, 0.7, 0.85, 0.95])
    names_mn = ("rmnc", "zmns", "lmns", "d_rmnc_d_s", "d_zmns_d_s", "d_lmns_d_s")
    names_nyq = ("gmnc", "bmnc", "d_bmnc_d_s", "bsupumnc", "bsupvmnc", "bsubsmns", "bsubumnc", "bsubvmnc")
    for nm in names_mn + names_nyq:
        out[nm] = np.array([[spl(s) for spl in getattr(vs, nm)] for s in svals])
    out["iota"] = vs.iota(svals)
    out["d_iota_d_s"] = vs.d_iota_d_s(svals)
    out["d_pressure_d_s"] = vs.d_pressure_d_s(svals)
    out["pressure"] = vs.pressure(svals)
    save("G8_surface_tables.npz", s=svals, xm=vs.xm, xn=vs.xn, xm_nyq=vs.xm_nyq, xn_nyq=vs.xn_nyq,
         phiedge=np.array(vs.phiedge), Aminor_p=np.array(vs.Aminor_p), nfp=np.array(vs.nfp), **out)


def _scan_and_refine(vs, s, N, tight):
    """the per-surface body of ball_scan.py:223-339 (coarse 24 x 15 scan with the warm-start chain, argmax rule,
    L-BFGS-B refinement, final solve with the stale start vector and sigma = 0.42), run through the reference's own
    gamma_ball_full / obj_w_grad / vmec_fieldlines; every objective evaluation of the refinement is recorded."""
    from scipy.optimize import minimize
    import contextlib
    ctx = tight_arpack() if tight else contextlib.nullcontext()
    th = theta_grid(N)
    vguess = vguess_of(th)
    theta0_scan = np.linspace(0.0, 0.5 * np.pi, 15)
    alpha_scan = np.linspace(0, np.pi, 24)
    with ctx:
        G = line_geometry(vs, s, alpha_scan, th)
        gam_tab = np.zeros((24, 15))
        vg_tab = np.zeros((24, 15, N - 2))
        for i in range(24):
            for j in range(15):
                dP, gam, X, dX, g, c, f = solve_line(G[i], th, theta0_scan[j], vguess, 1.0)
                vguess = X[1:-1]                                   # ball_scan.py:272-274
                vg_tab[i, j] = vguess
                gam_tab[i, j] = gam
        idx = np.where(gam_tab == np.max(gam_tab))                 # ball_scan.py:283-295 (first maximum on ties)
        i0, j0 = idx[0][0], idx[1][0]
        vg0 = vg_tab[i0, j0]
        sigma0 = 1.3 * abs(gam_tab[i0, j0]) + 0.05
        trace = []

        def fun(x):
            val, jac = ref.obj_w_grad(x, vs, s, th, vg0, sigma0)
            trace.append((x[0], x[1], val, jac[0], jac[1]))
            return val, jac

        res = minimize(fun, x0=(alpha_scan[i0], theta0_scan[j0]), jac=True,
                       bounds=((0.0, np.pi), (0.0, 0.5 * np.pi)),
                       options={"ftol": 5.0e-11, "gtol": 2.0e-08, "maxiter": 30})
        Gf = line_geometry(vs, s, res.x[0], th)
        gam_f = solve_line(Gf[0], th, res.x[1], vg0, 0.42)[1]
    return dict(gam_table=gam_tab, argmax=np.array([i0, j0]), trace=np.array(trace), x_opt=np.array(res.x),
                gam_opt=float(gam_f), fun=float(res.fun), nit=int(res.nit), nfev=int(res.nfev), status=int(res.status),
                message=str(res.message))


def make_G5T():
    """G5 again with the reference's ARPACK call converged (tight_arpack): the L-BFGS-B trajectory is then free of
    ARPACK noise and can be compared point for point.  Same surface, grid and geometry as G5 (not stored again)."""
    vs = ncsx_splines()
    r = _scan_and_refine(vs, 0.8483, 513, tight=True)
    print("G5T", r["x_opt"], r["gam_opt"], len(r["trace"]), r["message"], flush=True)
    save("G5_scan_trace_tight.npz", s=np.array(0.8483), N=np.array(513), gam_table=r["gam_table"], argmax=r["argmax"],
         trace=r["trace"], x_opt=r["x_opt"], gam_opt=np.array(r["gam_opt"]), fun=np.array(r["fun"]),
         nit=np.array(r["nit"]), nfev=np.array(r["nfev"]), status=np.array(r["status"]), message=np.array(r["message"]))


def make_G9():
    """Refinement goldens on the reference's own NCSX grid (N = 969, ball_scan.py:203-208) for several surfaces:
    coarse table, argmax, the FULL evaluation trace of scipy's L-BFGS-B (ball_scan.py:305-314), x_opt and the final
    gam (ball_scan.py:322-339) -- once with ARPACK converged ('tight': the pin) and once as shipped (tol 5e-7: shows
    how far the reference's own ARPACK noise moves its optimum)."""
    vs = ncsx_splines()
    svals = [0.5, 0.6125, 0.7, 0.8483, 0.95]
    out = {}
    for k, s in enumerate(svals):
        for tag, tight in (("tight", True), ("shipped", False)):
            t = time.time()
            r = _scan_and_refine(vs, s, 969, tight=tight)
            print("G9 s=%g %s: x_opt=%s gam_opt=%.12e evals=%d nit=%d '%s' (%.0f s)" %
                  (s, tag, r["x_opt"], r["gam_opt"], len(r["trace"]), r["nit"], r["message"], time.time() - t), flush=True)
            for name in ("gam_table", "argmax", "trace", "x_opt"):
                out["%s_%s_%d" % (name, tag, k)] = r[name]
            for name in ("gam_opt", "fun", "nit", "nfev", "status", "message"):
                out["%s_%s_%d" % (name, tag, k)] = np.array(r[name])
    save("G9_refine_traces.npz", s=np.array(svals), N=np.array(969), **out)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["G1", "G2", "G3", "G4", "G5", "G6", "G7", "G8"]
    for name in todo:
        t = time.time()
        globals()["make_" + name]()
        print("%s done in %.1f s" % (name, time.time() - t), flush=True)
