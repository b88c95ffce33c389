#!/usr/bin/env python3
"""Round 5: shift-iteration policies tried in the numpy model (tools/sim_solve.py) and NOT built -- on the bench batch (configs[1]
shape, 1,024 systems from the G3 lines) and on the rough family of configs[4] (iid coefficients, N = 513):
  * geometric instead of arithmetic halving towards the Rayleigh bound while every shift so far had count 0 (`sched`),
  * p(x) ~ (lam - x) exp(alpha + beta x) through three points instead of the parabola (`interp_exp`),
  * a bracket holding exactly two eigenvalues: the parabola's two roots, next shift between them (`pair`), and the count-2 end kept as
    the parabola's third point (`keep2`).
What it prints is quoted in docs/EXPERIMENTS.md R5.8.      python tools/sim_policies.py"""
import importlib.util
import math
import os
import sys

import numpy as np
from scipy.linalg import eigh_tridiagonal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ballooning_oracle as bo  # noqa: E402

_argv, sys.argv = sys.argv, ["sim_solve", "none"]
spec = importlib.util.spec_from_file_location("sim_solve", os.path.join(ROOT, "tools", "sim_solve.py"))
ss = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ss)
sys.argv = _argv
EPS = ss.EPS
Pt = ss.Pt
shoot = ss.shoot


def rough_systems(n, N, seed=1):
    rng = np.random.default_rng(seed)
    th = np.linspace(-4*np.pi, 4*np.pi, N); h = 8*np.pi/(N-1)
    out = []
    for k in range(n):
        g = np.exp(rng.uniform(np.log(0.01), np.log(50.0), N)); c = rng.uniform(-2.5, 3.5, N); f = np.exp(rng.uniform(np.log(0.2), np.log(3e3), N))
        d, efull, fd, hh, _, _, _ = bo.assemble(th, g, c, f)
        e = efull[1:-1]; s = 1/np.sqrt(fd)
        w = eigh_tridiagonal(d*s*s, e*s[:-1]*s[1:], eigvals_only=True)
        cj, fj = c[1:-1], f[1:-1]
        hi = np.max(cj/fj)
        lo = max(np.max(d/fd), (np.sum(cj) - efull[0] - efull[-1])/np.sum(fj))
        normA = np.max((np.abs(d) + efull[:-1] + efull[1:])/fd)
        out.append(dict(w=w, lo=lo - 8*EPS*normA, hi=hi + 8*EPS*normA, normA=normA, d=d, e=e, fd=fd))
    return out


def solve_v(sysd, div=2.0, trace=None, Nn=513, sched=None, pair=None, keep2=False, interp=None):
    interpolate = interp or ss.interpolate
    w = sysd["w"]; normA = sysd["normA"]
    count = lambda s: int(np.sum(w > s))
    tol = 64 * EPS * normA
    lo, hi = sysd["lo"], sysd["hi"]
    rho_t, dl = sysd["rho"], sysd["delta"]
    use = math.isfinite(rho_t) and dl > 0 and dl < 0.25 * (hi - lo)
    mrg = (8 + Nn / 2) * EPS * normA
    if use: lo = max(lo, rho_t - mrg)
    guess, width = (rho_t, 0.25 * dl) if use else (float("nan"), 0.0)
    sig = 0.5 * (lo + hi)
    expand = try_below = False; wstep = 0.0
    if use and lo < guess + width < hi:
        sig = guess + width; expand = True; wstep = 4 * width; try_below = True
    off_up = off_dn = tol; rho_trust = hi; aimed = 0
    lo1 = hi_f = old_ok = was_interp = conv = force_bis = False
    nz = 0; lo_cnt = 99; Pold2 = None
    lo_seen = False   # a shift with count >= 1 has been seen (lo is no longer the Rayleigh bound)
    lg_prev = 0.0
    Plo = Pt(lo); Phi = Pt(hi); Pold = Pt(hi)
    sig_prev = sig; it = 0; done = False
    while not done and it < 200:
        C = count(sig); sg, lg = shoot(w, sig); it += 1
        if trace is not None: trace.append((sig, C))
        if C == 0:
            nz += 1
            if hi_f: Pold = Phi; old_ok = True
            hi = sig; Phi = Pt(sig, sg, lg); hi_f = True
        else:
            if lo1 or (keep2 and lo_seen and lo_cnt == 2): Pold = Plo; old_ok = True
            if pair and not lo1 and lo_seen and lo_cnt == 2: Pold2 = Plo
            lo = sig; lo1 = (C == 1); Plo = Pt(sig, sg, lg); lo_seen = True; lo_cnt = C
        prevstep = abs(sig - sig_prev); sig_prev = sig
        if expand:
            if C != 0 and sig + wstep < hi: sig += wstep; wstep *= 4; continue
            expand = False
        if not lo1:
            if hi - lo <= 4 * tol: done = True; break
            moved = False
            if pair and lo_cnt == 2 and hi_f:
                cands = [P for P in ((Pold if old_ok else None), Pold2) if P is not None and P.x != Plo.x and P.x != Phi.x]
                if cands:
                    o = min(cands, key=lambda P: min(abs(P.x - lo), abs(P.x - hi)))
                    r = pair_roots(o, Plo, Phi)
                    if r is not None:
                        r2, r1 = r
                        t = pair(r2, r1)
                        wd = hi - lo
                        if lo + 0.02 * wd < t < hi - 0.02 * wd: sig = t; moved = True
            if moved: try_below = False; continue
            if try_below and lo < guess - width < hi: sig = guess - width
            elif use and hi_f and not lo_seen: sig = lo + (hi - lo) / (sched[min(nz, len(sched)) - 1] if sched else div)
            else: sig = 0.5 * (lo + hi)
            try_below = False
            continue
        if was_interp:
            red = lg_prev - lg
            conv = red >= 4; force_bis = red < 1
        elif aimed == 0: conv = False
        cert = aimed != 0
        if aimed > 0 and C != 0: off_up *= 2
        if aimed < 0 and C == 0: off_dn *= 2
        aimed = 0
        if hi - lo <= 4 * tol: done = True; break
        rho = sig; ok = False; near = False
        if cert: rho = rho_trust; ok = True; near = True
        elif hi_f and not force_bis:
            b_is_lo = Plo.lg <= Phi.lg
            lg_prev = Plo.lg if b_is_lo else Phi.lg
            b = Plo if b_is_lo else Phi; a = Phi if b_is_lo else Plo
            use_o = old_ok and Pold.x != a.x and Pold.x != b.x
            got, r = interpolate(Pold, use_o, a, b, lo, hi)
            q = 0.25 * (3 * a.x + b.x)
            inside = min(q, b.x) <= r <= max(q, b.x)
            stepb = abs(r - b.x)
            nr = was_interp and conv and stepb <= 4096 * tol
            acc = got and inside and (nr or (stepb < 0.5 * prevstep and stepb >= 9.5367431640625e-07 * prevstep))
            if acc: rho = r; ok = True; near = nr
        moved = False; interp_now = False
        if ok:
            if near:
                if not cert and abs(rho - rho_trust) > 4096 * tol: off_up = off_dn = tol
                rho_trust = rho
                up = max(rho, lo); dn = min(rho, hi); nxt = rho
                if hi > up + 2 * off_up: nxt = up + off_up; aimed = 1
                elif lo < dn - 2 * off_dn: nxt = dn - off_dn; aimed = -1
                if aimed != 0 and lo < nxt < hi: sig = nxt; moved = True
                else: aimed = 0
            else:
                sig = rho; moved = True; interp_now = True
        force_bis = False
        if not moved: sig = 0.5 * (lo + hi)
        was_interp = interp_now
    return it, 0.5 * (lo + hi)


def pair_roots(o, a, b):
    lgmax = max(a.lg, b.lg, o.lg)
    val = lambda p: p.sg * 2.0 ** max(p.lg - lgmax, -1000.0)
    xs = np.array([o.x, a.x, b.x]); ys = np.array([val(o), val(a), val(b)])
    x0 = b.x
    try:
        co = np.polyfit(xs - x0, ys, 2)
    except Exception:
        return None
    A, Bc, Cc = co
    disc = Bc * Bc - 4 * A * Cc
    if A == 0: return None
    if disc < 0:
        v = -Bc / (2 * A) + x0
        return (v, v)
    sq = math.sqrt(disc)
    q = -0.5 * (Bc + (sq if Bc >= 0 else -sq))
    rts = sorted([q / A + x0, (Cc / q + x0) if q != 0 else q / A + x0])
    return rts[0], rts[1]


def interp_exp(o, use_o, a, b, lo, hi):
    """p(x) ~ (lam - x) * exp(alpha + beta x) through three points (sign, log2 values); falls back to the built rule"""
    if not use_o: return ss.interpolate(o, use_o, a, b, lo, hi)
    lgmax = max(a.lg, b.lg, o.lg)
    val = lambda p: p.sg * 2.0 ** max(p.lg - lgmax, -1000.0)
    x1, x2, x3 = a.x, b.x, o.x
    u1, u2, u3 = val(a), val(b), val(o)
    if u1 == 0 or u2 == 0 or u3 == 0 or x1 == x2 or x2 == x3 or x1 == x3: return ss.interpolate(o, use_o, a, b, lo, hi)
    # g(beta) = (u1 E1 - u2) (x2 - x3) - (u2 - u3 E3) (x1 - x2),  Ei = exp(-beta (xi - x2))
    d1, d3 = x1 - x2, x3 - x2
    beta = 0.0
    okk = False
    for k in range(30):
        with np.errstate(all='ignore'):
            E1, E3 = math.exp(min(max(-beta * d1, -700), 700)), math.exp(min(max(-beta * d3, -700), 700))
            g = (u1 * E1 - u2) * (-d3) - (u2 - u3 * E3) * d1
            dg = (u1 * E1 * (-d1)) * (-d3) - (-u3 * E3 * (-d3)) * d1
        if dg == 0 or not math.isfinite(g) or not math.isfinite(dg): break
        step = g / dg
        beta -= step
        if abs(step) * max(abs(d1), abs(d3)) < 1e-12: okk = True; break
    if not okk: return ss.interpolate(o, use_o, a, b, lo, hi)
    E1 = math.exp(min(max(-beta * d1, -700), 700))
    den = u2 - u1 * E1
    if den == 0: return ss.interpolate(o, use_o, a, b, lo, hi)
    r = x2 - u2 * (x2 - x1) / den
    if not (math.isfinite(r) and lo < r < hi): return ss.interpolate(o, use_o, a, b, lo, hi)
    return True, r


if __name__ == "__main__":
    N = 513
    R = ss.add_trial(rough_systems(200, N), N)
    B = ss.add_trial(ss.systems())
    f34 = lambda r2, r1: r2 + 0.75 * (r1 - r2)
    rows = [("as built", {}),
            ("towards the Rayleigh bound by 2, 4, 4 ...", dict(sched=(2, 4))),
            ("... by 2, 2, 4, 4 ...", dict(sched=(2, 2, 4))),
            ("... by 4 from the first", dict(div=4.0)),
            ("exp-linear model instead of the parabola", dict(interp=interp_exp)),
            ("two-eigenvalue bracket: midpoint of the parabola's roots", dict(pair=lambda r2, r1: 0.5 * (r1 + r2))),
            ("... 3/4 of the way to the larger root", dict(pair=f34)),
            ("... the larger root itself", dict(pair=lambda r2, r1: r1)),
            ("count-2 end kept as the third point", dict(keep2=True)),
            ("3/4 rule + count-2 end kept", dict(pair=f34, keep2=True))]
    print("%-58s | rough family: mean max | bench batch: mean max  histogram from 8 sweeps" % "policy")
    for name, kw in rows:
        a = np.array([solve_v(s, **kw)[0] for s in R])
        b = np.array([solve_v(s, **kw)[0] for s in B])
        err = max(abs(solve_v(s, **kw)[1] - s["w"][-1]) / s["normA"] for s in R[:50] + B[:100])
        print("%-58s | %18.2f %3d | %17.3f %3d  %s   (max |lam - lam_1| / ||A|| %.1e)" % (name, a.mean(), a.max(), b.mean(), b.max(), np.bincount(b)[8:].tolist(), err))
