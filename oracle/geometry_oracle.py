"""CPU oracle for the field-line geometry.  TEST INFRASTRUCTURE ONLY.

numpy restatement of the part of the reference's `vmec_fieldlines` (utils.py:302-720) that produces the
eight arrays the ballooning path consumes, and of `vmec_splines` (utils.py:37-158) on plain wout tables.
Parity status: PINNED -- checked in tests/test_geometry_oracle.py against the geometry arrays the
reference itself produced (tests/golden/G3_ncsx_lines.npz) from tests/golden/G8_*.npz inputs.
"""
import numpy as np

MU0 = 4 * np.pi * (1.0e-7)
NAMES_MN = ("rmnc", "zmns", "lmns", "d_rmnc_d_s", "d_zmns_d_s", "d_lmns_d_s")
NAMES_NYQ = ("gmnc", "bmnc", "d_bmnc_d_s", "bsupvmnc", "bsubsmns", "bsubumnc", "bsubvmnc")


def surface_tables_from_wout(wout, svals):
    """vmec_splines (utils.py:58-119) + the per-surface evaluation of utils.py:311-357 on wout tables
    (arrays stored (mn, ns) like simsopt's Vmec.wout).  Returns a dict of per-surface arrays."""
    from scipy.interpolate import InterpolatedUnivariateSpline as Spl
    ns = int(wout["ns"])
    s_full = np.linspace(0, 1, ns)
    ds = s_full[1] - s_full[0]
    s_half = s_full[1:] - 0.5 * ds
    svals = np.atleast_1d(np.asarray(svals, dtype=float))
    out = dict(s=svals, xm=wout["xm"], xn=wout["xn"], xm_nyq=wout["xm_nyq"], xn_nyq=wout["xn_nyq"],
               phiedge=float(wout["phi"][-1]), Aminor_p=float(wout["Aminor_p"]))

    def ev(tab, half, deriv=False):
        res = np.empty((len(svals), tab.shape[0]))
        for j in range(tab.shape[0]):
            sp = Spl(s_half, tab[j, 1:]) if half else Spl(s_full, tab[j, :])
            res[:, j] = (sp.derivative() if deriv else sp)(svals)
        return res

    out["rmnc"] = ev(wout["rmnc"], False); out["d_rmnc_d_s"] = ev(wout["rmnc"], False, True)     # utils.py:60, 68
    out["zmns"] = ev(wout["zmns"], False); out["d_zmns_d_s"] = ev(wout["zmns"], False, True)     # utils.py:63, 69
    out["lmns"] = ev(wout["lmns"], True); out["d_lmns_d_s"] = ev(wout["lmns"], True, True)        # utils.py:66, 70
    out["gmnc"] = ev(wout["gmnc"], True)                                                          # utils.py:83
    out["bmnc"] = ev(wout["bmnc"], True); out["d_bmnc_d_s"] = ev(wout["bmnc"], True, True)        # utils.py:86, 107
    out["bsupvmnc"] = ev(wout["bsupvmnc"], True)                                                  # utils.py:92
    out["bsubsmns"] = ev(wout["bsubsmns"], False)                                                 # utils.py:96 (full mesh)
    out["bsubumnc"] = ev(wout["bsubumnc"], True)                                                  # utils.py:99
    out["bsubvmnc"] = ev(wout["bsubvmnc"], True)                                                  # utils.py:102
    pres = Spl(s_half, wout["pres"][1:])                                                          # utils.py:112
    iota = Spl(s_half, wout["iotas"][1:])                                                         # utils.py:118
    out["d_pressure_d_s"] = pres.derivative()(svals)
    out["iota"] = iota(svals)
    out["d_iota_d_s"] = iota.derivative()(svals)
    return out


def theta_vmec_of(theta_p, phi, xm, xn, lmns, tol=1e-15, maxit=60):
    """theta_vmec with theta_vmec + sum lmns sin(m theta_vmec - n phi) = theta_pest (utils.py:391-416);
    secant from (theta_p, theta_p + 0.1) like scipy.optimize.newton without fprime."""
    def res(tv):
        return theta_p - (tv + np.sum(lmns[:, None] * np.sin(xm[:, None] * tv[None] - xn[:, None] * phi[None]), axis=0))
    p0 = theta_p.copy(); p1 = theta_p + 0.1
    q0 = res(p0); q1 = res(p1)
    for _ in range(maxit):
        den = q1 - q0
        step = np.where(den != 0, q1 * (p1 - p0) / np.where(den != 0, den, 1.0), 0.0)
        p = p1 - step
        p0, q0 = p1, q1
        p1 = p; q1 = res(p1)
        if np.max(np.abs(step)) < tol * max(1.0, np.max(np.abs(p1))):
            break
    return p1


def fieldline_geometry(tab, js, alphas, theta1d, phi_center=0.0):
    """The 8 arrays (bmag, gradpar_theta_pest, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift) for surface
    index js of the tables and each alpha: array (nalpha, 8, N).  Restates utils.py:359-720."""
    xm, xn, xmq, xnq = tab["xm"], tab["xn"], tab["xm_nyq"], tab["xn_nyq"]
    s = tab["s"][js]; iota = tab["iota"][js]; diota = tab["d_iota_d_s"][js]; dp = tab["d_pressure_d_s"][js]
    shat = (-2 * s / iota) * diota                                              # utils.py:316
    c = {k: tab[k][js] for k in NAMES_MN + NAMES_NYQ}
    etf = -tab["phiedge"] / (2 * np.pi)                                         # utils.py:474
    L = tab["Aminor_p"]; Bref = 2 * abs(etf) / (L * L); sgn = np.sign(etf); sq = np.sqrt(s)   # utils.py:654-665
    out = []
    for a in np.atleast_1d(alphas):
        tp = np.asarray(theta1d, dtype=float)
        phi = phi_center + (tp - a) / iota                                      # utils.py:373
        tv = theta_vmec_of(tp, phi, xm, xn, c["lmns"])
        ang = xm[:, None] * tv[None] - xn[:, None] * phi[None]
        ca, sa = np.cos(ang), np.sin(ang)
        R = c["rmnc"] @ ca; R_s = c["d_rmnc_d_s"] @ ca
        R_t = -(c["rmnc"] * xm) @ sa; R_p = (c["rmnc"] * xn) @ sa                 # utils.py:432-435
        Z_s = c["d_zmns_d_s"] @ sa; Z_t = (c["zmns"] * xm) @ ca; Z_p = -(c["zmns"] * xn) @ ca   # utils.py:437-440
        l_s = c["d_lmns_d_s"] @ sa; l_t = (c["lmns"] * xm) @ ca; l_p = -(c["lmns"] * xn) @ ca   # utils.py:442-444
        ang = xmq[:, None] * tv[None] - xnq[:, None] * phi[None]
        ca, sa = np.cos(ang), np.sin(ang)
        sqg = c["gmnc"] @ ca; modB = c["bmnc"] @ ca; B_s = c["d_bmnc_d_s"] @ ca
        B_t = -(c["bmnc"] * xmq) @ sa; B_p = (c["bmnc"] * xnq) @ sa              # utils.py:458-462
        Bsup_phi = c["bsupvmnc"] @ ca; Bsub_s = c["bsubsmns"] @ sa
        Bsub_t = c["bsubumnc"] @ ca; Bsub_p = c["bsubvmnc"] @ ca                 # utils.py:464-468
        sp, cp = np.sin(phi), np.cos(phi)
        X_t = R_t * cp; X_p = R_p * cp - R * sp; X_s = R_s * cp                  # utils.py:483-489
        Y_t = R_t * sp; Y_p = R_p * sp + R * cp; Y_s = R_s * sp
        gs = np.array([Y_t * Z_p - Z_t * Y_p, Z_t * X_p - X_t * Z_p, X_t * Y_p - Y_t * X_p]) / sqg      # utils.py:492-500
        gt = np.array([Y_p * Z_s - Z_p * Y_s, Z_p * X_s - X_p * Z_s, X_p * Y_s - Y_p * X_s]) / sqg      # utils.py:502-504
        gp = np.array([Y_s * Z_t - Z_s * Y_t, Z_s * X_t - X_s * Z_t, X_s * Y_t - Y_s * X_t]) / sqg      # utils.py:506-508
        gpsi = gs * etf                                                                                   # utils.py:515-517
        ls = l_s - (phi - phi_center) * diota
        galpha = ls * gs + (1 + l_t) * gt + (-iota + l_p) * gp                                           # utils.py:520-538
        BxgB_alpha = (Bsub_s * B_t * (l_p - iota) + Bsub_t * B_p * ls + Bsub_p * B_s * (1 + l_t)
                      - Bsub_p * B_t * ls - Bsub_t * B_s * (l_p - iota) - Bsub_s * B_p * (1 + l_t)) / sqg   # utils.py:603-618
        BxgB_psi = (Bsub_t * B_p - Bsub_p * B_t) / sqg * etf                                              # utils.py:646-650
        bmag = modB / Bref                                                                                # utils.py:678
        gradpar = L * (iota * Bsup_phi) / modB                                                            # utils.py:469, 679
        gds2 = np.sum(galpha * galpha, 0) * L * L * s                                                     # utils.py:682
        gds21 = np.sum(galpha * gpsi, 0) * shat / Bref                                                    # utils.py:683
        gds22 = np.sum(gpsi * gpsi, 0) * shat * shat / (L * L * Bref * Bref * s)                          # utils.py:684-689
        gbdrift = -1.0 * 2 * Bref * L * L * sq * BxgB_alpha / (modB * modB * modB) * sgn                  # utils.py:692-702
        gbdrift0 = -1.0 * BxgB_psi * 2 * shat / (modB * modB * modB * sq) * sgn                           # utils.py:704-711
        cvdrift = 1.0 * gbdrift - 2 * Bref * L * L * sq * MU0 * dp * sgn / (etf * modB * modB)            # utils.py:714-718
        out.append(np.stack([bmag, gradpar, cvdrift, gbdrift0, gds2, gds21, gds22, gbdrift]))            # cvdrift0 = gbdrift0 (:720)
    return np.stack(out)
