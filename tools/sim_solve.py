"""numpy model of WaveSolver::solve (cold start): counts from the exact spectrum, shooting value p(sig) = prod(lam_i - sig) in
(sign, log2) form.  Used to try changes of the shift iteration before building them."""
import numpy as np, sys, math
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ballooning_oracle as bo
from scipy.linalg import eigh_tridiagonal

EPS = 2.220446049250313e-16

def systems():
    g3 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'G3_ncsx_lines.npz'))
    geo = g3["geo_513"]; nl = 128
    rng = np.random.default_rng(1000)
    base = geo[np.arange(nl) % len(geo)].copy()
    eps = rng.uniform(-0.03, 0.03, size=(nl, 2))
    base[:, 4:7, :] *= (1 + eps[:, 0])[:, None, None]; base[:, 2:4, :] *= (1 + eps[:, 1])[:, None, None]; base[:, 7, :] *= (1 + eps[:, 1])[:, None]
    dP = -0.5 * np.mean((base[:, 2] - base[:, 7]) * base[:, 0] ** 2, axis=1)
    theta0 = np.linspace(0, np.pi / 2, 8)
    N = 513; th = bo.theta_grid(N); h = th[1] - th[0]
    out = []
    for ln in range(nl):
        for it0, t0 in enumerate(theta0):
            line = base[ln]
            cv, gd = bo.fold_theta0(t0, line[2], line[3], line[4], line[5], line[6])
            g, c, f = bo.gcf(dP[ln], line[0], line[1], cv, gd)
            d, efull, fd, hh, _, _, _ = bo.assemble(th, g, c, f)
            e = efull[1:-1]
            s = 1 / np.sqrt(fd)
            w = eigh_tridiagonal(d * s * s, e * s[:-1] * s[1:], eigvals_only=True)
            cj, fj = c[1:-1], f[1:-1]
            hi = np.max(cj / fj)
            lo = max(np.max(d / fd), (np.sum(cj) - efull[0] - efull[-1]) / np.sum(fj))
            normA = np.max((np.abs(d) + efull[:-1] + efull[1:]) / fd)
            out.append(dict(w=w, lo=lo - 8 * EPS * normA, hi=hi + 8 * EPS * normA, normA=normA, theta0_index=it0, line=ln, d=d, e=e, fd=fd))
    return out

class Pt:
    __slots__ = ("x", "sg", "lg")
    def __init__(self, x, sg=0.0, lg=-1e9): self.x = x; self.sg = sg; self.lg = lg

def shoot(w, sig):
    dlt = w - sig
    if np.any(dlt == 0): return 0.0, -1e9
    return float(np.prod(np.sign(dlt))), float(np.sum(np.log2(np.abs(dlt))))

def interpolate(o, use_o, a, b, lo, hi, larger_root=False):
    lgmax = max(a.lg, b.lg, o.lg if use_o else -1e18)
    val = lambda p: p.sg * 2.0 ** max(p.lg - lgmax, -1000.0)
    x2 = b.x; f1 = val(a); f2 = val(b); f0 = val(o)
    h2r = x2 - a.x
    if h2r == 0: return False, 0.0
    k = math.floor(math.log2(abs(h2r))); sc = 2.0 ** (-k); back = 2.0 ** k
    h2 = h2r * sc; h1 = (a.x - o.x) * sc
    df21 = f2 - f1
    with np.errstate(all='ignore'):
        z_s = -f2 * h2 / df21 if df21 != 0 else float('inf')
        r_s = z_s * back + x2
        in_s = math.isfinite(r_s) and lo < r_s < hi
        S = h1 + h2
        QA = df21 * h1 - (f1 - f0) * h2
        QB = QA * h2 + df21 * h1 * S
        H = h1 * h2 * S
        disc = QB * QB - 4 * QA * f2 * H
        sq = math.sqrt(max(disc, 0.0))
        den = QB + sq if QB >= 0 else QB - sq
        z1 = -2 * f2 * H / den if den != 0 else float('inf')
        z2 = -den / (2 * QA) if QA != 0 else float('inf')
    r1 = z1 * back + x2; r2 = z2 * back + x2
    par = use_o and disc >= 0 and H != 0
    in1 = par and math.isfinite(r1) and lo < r1 < hi
    in2 = par and math.isfinite(r2) and lo < r2 < hi
    if larger_root:
        if in1 and in2: return True, max(r1, r2)
        return False, 0.0
    pick1 = in1 and (not in2 or abs(z1) <= abs(z2))
    rho = r1 if pick1 else (r2 if in2 else r_s)
    return (in1 or in2 or in_s), rho

def solve(sysd, pair_rule=False, trace=False):
    w = sysd["w"]; lo = sysd["lo"]; hi = sysd["hi"]; normA = sysd["normA"]
    count = lambda s: int(np.sum(w > s))
    tol = 64 * EPS * normA
    sig = 0.5 * (lo + hi)
    off_up = off_dn = tol; rho_trust = hi; aimed = 0
    lo1 = False; hi_f = False; old_ok = False; was_interp = False; conv = False; force_bis = False
    lo_cnt = 99
    lg_prev = 0.0
    Plo = Pt(lo); Phi = Pt(hi); Pold = Pt(hi)
    sig_prev = sig; it = 0; done = False
    while not done and it < 200:
        C = count(sig); sg, lg = shoot(w, sig); it += 1
        if C == 0:
            if hi_f: Pold = Phi; old_ok = True
            hi = sig; Phi = Pt(sig, sg, lg); hi_f = True
        else:
            if lo1 or (pair_rule and lo_cnt <= 2): Pold = Plo; old_ok = True
            lo = sig; lo1 = (C == 1); lo_cnt = C; Plo = Pt(sig, sg, lg)
        prevstep = abs(sig - sig_prev); sig_prev = sig
        if not lo1:
            if hi - lo <= 4 * tol: done = True; break
            moved = False
            if pair_rule and lo_cnt == 2 and hi_f and old_ok:
                # two eigenvalues above lo, none above hi: the parabola through lo, hi and the last replaced end models both
                # roots; aim at the LARGER one
                o = Pold
                if o.x != Plo.x and o.x != Phi.x:
                    got, r = interpolate(o, True, Plo, Phi, lo, hi, larger_root=True)
                    if got and (r - lo) > 1e-3 * (hi - lo) and (hi - r) > 1e-3 * (hi - lo):
                        sig = r; moved = True
            if not moved: sig = 0.5 * (lo + hi)
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
    return it, (0.5 * (lo + hi))


def add_trial(S, N=513):
    """Rayleigh quotient and residual bound of the trial vector sin(pi j / (N - 1)) (WaveSolver::setup<Src, true>)"""
    th = bo.theta_grid(N); thi = th[1:-1]
    x = np.cos(thi / 8)
    for s in S:
        d, e, fd = s["d"], s["e"], s["fd"]
        Tx = d * x; Tx[:-1] += e * x[1:]; Tx[1:] += e * x[:-1]
        Fx = fd * x
        rho = x @ Tx / (x @ Fx); r = Tx - rho * Fx
        s["rho"] = rho; s["delta"] = np.sqrt(np.sum(r * r / fd)) / np.sqrt(x @ Fx)
    return S

def solve_trial(sysd, fracs=(1 / 16, 1 / 4, 1.0)):
    """locate phase from the trial vector's Rayleigh quotient: lo = rho (rigorous), shifts walk up rho + delta * frac; then the
    baseline iteration on the bracket found.  Returns total sweeps (the trial pass itself is counted by the caller)."""
    w = sysd["w"]; normA = sysd["normA"]
    count = lambda s: int(np.sum(w > s))
    lo = max(sysd["lo"], sysd["rho"] - 8 * EPS * normA); hi = sysd["hi"]
    n = 0; lo1 = False
    first = None
    for fr in fracs:
        s = sysd["rho"] + fr * sysd["delta"]
        if not (lo < s < hi): continue
        C = count(s); n += 1
        if C == 0: hi = s; break
        lo = s
        if C == 1: break
    # hand the bracket to the baseline solver (its bisection phase continues from the midpoint)
    sub = dict(sysd); sub["lo"] = lo; sub["hi"] = hi
    it, lam = solve(sub)
    return n + it, lam


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "dual"):
    S = add_trial(systems())
    base = np.array([solve(s)[0] for s in S])
    print("bench batch (configs[1] shape, 1,024 systems, N = 513)")
    print("  as built in rounds 1-2 (Gershgorin / diagonal-quotient bracket): forward sweeps mean %.2f max %d hist(10..) %s" % (base.mean(), base.max(), np.bincount(base)[10:].tolist()))
    for fr in ((0.25, 1.25, 5.25), (1.0,), (1 / 16, 1 / 4, 1.0)):
        its = np.array([solve_trial(s, fr)[0] for s in S])
        print("  trial-vector bracket, first shifts rho + delta * %s: mean %.2f max %d hist(8..) %s" % (str(fr), its.mean(), its.max(), np.bincount(its)[8:].tolist()))


# ------------------------------------------------------------------------------------------------------------------------------
# Round 4: the iteration AS BUILT in round 3 (trial-vector bracket handed to the warm path of WaveSolver::solve: first shift
# rho + del / 4, walk up x 4 while the count stays >= 1, then rho - del / 4, then the count / interpolation phases) and the
# TWO-SHIFTS-PER-ITERATION variant of VERDICT r3 item 5 (two waves per system, each sweeping its own shift; both see both results):
# solve_pair is the policy that was built (tools/experiments/pair_two_shifts.patch) -- measured and dropped, docs/EXPERIMENTS.md.
def solve_asbuilt(sysd, trace=None):
    w = sysd["w"]; normA = sysd["normA"]
    count = lambda s: int(np.sum(w > s))
    tol = 64 * EPS * normA
    lo, hi = sysd["lo"], sysd["hi"]
    rho_t, dl = sysd["rho"], sysd["delta"]
    use = math.isfinite(rho_t) and dl > 0 and dl < 0.25 * (hi - lo)
    mrg = (8 + 513 / 2) * EPS * normA
    if use: lo = max(lo, rho_t - mrg)
    guess, width = (rho_t, 0.25 * dl) if use else (float("nan"), 0.0)
    sig = 0.5 * (lo + hi)
    expand = try_below = False; wstep = 0.0
    if use and lo < guess + width < hi:
        sig = guess + width; expand = True; wstep = 4 * width; try_below = True
    off_up = off_dn = tol; rho_trust = hi; aimed = 0
    lo1 = hi_f = old_ok = was_interp = conv = force_bis = False
    lg_prev = 0.0
    Plo = Pt(lo); Phi = Pt(hi); Pold = Pt(hi)
    sig_prev = sig; it = 0; done = False
    while not done and it < 200:
        C = count(sig); sg, lg = shoot(w, sig); it += 1
        if trace is not None: trace.append((sig, C))
        if C == 0:
            if hi_f: Pold = Phi; old_ok = True
            hi = sig; Phi = Pt(sig, sg, lg); hi_f = True
        else:
            if lo1: Pold = Plo; old_ok = True
            lo = sig; lo1 = (C == 1); Plo = Pt(sig, sg, lg)
        prevstep = abs(sig - sig_prev); sig_prev = sig
        if expand:
            if C != 0 and sig + wstep < hi: sig += wstep; wstep *= 4; continue
            expand = False
        if not lo1:
            if hi - lo <= 4 * tol: done = True; break
            if try_below and lo < guess - width < hi: sig = guess - width
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


def solve_pair(sysd, first=(0.07, 0.21), mirror=0.5, trace=None):
    """bounded state, as it can be built: bracket ends Plo / Phi + TWO spare points (the most recently displaced or unused points
    with count <= 1); everything else as solve_dual4"""
    w = sysd["w"]; normA = sysd["normA"]
    count = lambda s: int(np.sum(w > s))
    tol = 64 * EPS * normA
    lo, hi = sysd["lo"], sysd["hi"]
    rho_t, dl = sysd["rho"], sysd["delta"]
    use = math.isfinite(rho_t) and dl > 0 and dl < 0.25 * (hi - lo)
    mrg = (8 + 513 / 2) * EPS * normA
    if use: lo = max(lo, rho_t - mrg)
    it = 0
    lo1 = False; hi_f = False
    Plo = Pt(lo); Phi = Pt(hi)
    spare = []                      # at most 2, newest first
    def push(P):
        spare.insert(0, P)
        del spare[2:]
    def take(x):
        C = count(x); sg, lg = shoot(w, x)
        return Pt(x, sg, lg), C
    def absorb(P, C):
        nonlocal lo, hi, lo1, hi_f, Plo, Phi
        if not (lo < P.x < hi):
            if C <= 1: push(P)
            return
        if C == 0:
            if hi_f: push(Phi)
            hi = P.x; Phi = P; hi_f = True
        else:
            if lo1: push(Plo)
            lo = P.x; Plo = P; lo1 = (C == 1)
    if use and lo < rho_t + first[0] * dl and rho_t + first[1] * dl < hi:
        s1, s2 = rho_t + first[0] * dl, rho_t + first[1] * dl
    else:
        s1, s2 = lo + (hi - lo) / 3, lo + 2 * (hi - lo) / 3
    lg_best_prev = None; off = tol
    while it < 200:
        (P1, C1), (P2, C2) = take(s1), take(s2); it += 1
        if trace is not None: trace.append((s1, C1, s2, C2, lo, hi))
        # the point farther from the root first, so that the nearer one ends up as the bracket end and the other as a spare
        if C1 >= 1: absorb(P1, C1); absorb(P2, C2)      # s1 below the root: s1 then s2 (s2 above: other end; s2 below: nearer)
        else: absorb(P2, C2); absorb(P1, C1)            # both above: s2 (farther) first
        width = hi - lo
        if width <= 4 * tol: break
        if not lo1 or not hi_f:
            s1, s2 = lo + width / 3, lo + 2 * width / 3
            continue
        b_is_lo = Plo.lg <= Phi.lg
        b = Plo if b_is_lo else Phi; a = Phi if b_is_lo else Plo
        cand = [P for P in spare if P.x != a.x and P.x != b.x]
        o = min(cand, key=lambda P: abs(P.x - b.x)) if cand else None
        got, r = interpolate(o if o is not None else a, o is not None, a, b, lo, hi)
        stalled = lg_best_prev is not None and (lg_best_prev - b.lg) < 1
        lg_best_prev = b.lg
        if not got or stalled:
            s1, s2 = lo + width / 3, lo + 2 * width / 3
            lg_best_prev = None
            continue
        d = abs(r - b.x)
        if d <= 4096 * tol:
            sA, sB = r - off, r + off
            off *= 2
        else:
            sgn = 1.0 if a.x > b.x else -1.0
            sA = r
            sB = r + sgn * mirror * d
            if not (lo < sB < hi): sB = 0.5 * (r + a.x)
        s1, s2 = (sA, sB) if sA < sB else (sB, sA)
        if s1 <= lo: s1 = 0.5 * (lo + min(s2, hi))
        if s2 >= hi: s2 = 0.5 * (max(s1, lo) + hi)
    return it, 0.5 * (lo + hi)


def report_round4():
    S = add_trial(systems())
    a = np.array([solve_asbuilt(s)[0] for s in S])
    print("as built (round 3), one shift per sweep: sweeps mean %.2f max %d hist(6..) %s   [GPU: 10.42 / 13, the same histogram]" % (
        a.mean(), a.max(), np.bincount(a)[6:].tolist()))
    for first in ((0.07, 0.21), (1 / 16, 1 / 4)):
        res = [solve_pair(s, first, 0.5) for s in S]
        d = np.array([r[0] for r in res])
        err = max(abs(r[1] - s["w"][-1]) / s["normA"] for r, s in zip(res, S))
        print("two shifts per iteration (two waves per system), first pair rho + %s del: iterations mean %.2f max %d hist(3..) %s ; max |lam - lam_1| / |A| %.1e"
              "   [GPU, first pair (0.07, 0.21): 7.41 / 11]" % (str(first), d.mean(), d.max(), np.bincount(d)[3:].tolist(), err))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "dual":
    report_round4()
