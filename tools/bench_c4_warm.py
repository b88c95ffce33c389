"""config-4 shape (73 DOF-perturbed equilibria x 5 surfaces x 24 alpha x 15 theta0, N = 969): cold scan of all
equilibria vs scan warm-started from the base equilibrium's eigenvalues (ibs_gamma_scan_warm_f64)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
n_eq, ns, na, nt0, N = 73, 5, 24, 15, 969
svals = np.linspace(0.5, 0.95, ns); th = ibs_amd.theta_grid(N); alphas = np.linspace(0, np.pi, na)
t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
tabs_all = []
for q in range(n_eq):
    w = dict(wout)
    if q:   # emulated DOF perturbation (create_dict.py:67-70: rel 2e-3) of one boundary-weighted Fourier row
        w["rmnc"] = wout["rmnc"].copy(); w["rmnc"][q % 200, :] *= (1 + 2e-3 * np.linspace(0, 1, wout["rmnc"].shape[1]) ** 2)
    tabs_all.append(ibs_amd.SurfaceTables.from_wout(w, svals))
big = ibs_amd.SurfaceTables(np.tile(svals, n_eq), tabs_all[0].xm, tabs_all[0].xn, tabs_all[0].xm_nyq, tabs_all[0].xn_nyq,
                            np.concatenate([t_.tab_mn for t_ in tabs_all]), np.concatenate([t_.tab_nyq for t_ in tabs_all]),
                            np.concatenate([t_.scal[:, 1] for t_ in tabs_all]), np.concatenate([t_.scal[:, 2] for t_ in tabs_all]),
                            np.concatenate([t_.scal[:, 3] for t_ in tabs_all]), tabs_all[0].scal[0, 4], tabs_all[0].scal[0, 5])
surf = np.repeat(np.arange(n_eq * ns), na); al = np.tile(alphas, n_eq * ns)
r = ctx.fieldline_geometry(big, surf, al, th, device=dev)
geo7 = [r["geo"][k] for k in range(7)]
h = th[1] - th[0]
nl0 = ns * na
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best, out
t_cold, cold = timed(lambda: ctx.gamma_scan(h, *geo7, r["dPdrho"], t0, want_info=True))
t_base, base = timed(lambda: ctx.gamma_scan(h, *[g[:nl0] for g in geo7], r["dPdrho"][:nl0], t0))
guess = base["lam"].repeat(n_eq, 1)
dl = (cold["lam"] - guess).abs().max().item()
for mult in (2.0, 4.0, 16.0):
    width = mult * dl
    t_warm, warm = timed(lambda: ctx.gamma_scan(h, *geo7, r["dPdrho"], t0, want_info=True, lam_guess=guess, guess_width=width))
    print("width %.1f x max|dlam| (%.2e): warm %.3f ms  sweeps %.2f  max|dgam| vs cold %.1e  flagged %d" % (
        mult, width, t_warm, float((warm["info"] & 0xffff).double().mean()), float((warm["gam"] - cold["gam"]).abs().max()),
        int(((warm["info"] >> 16) != 0).sum())))
print("cold scan of %d solves: %.3f ms (%.2f sweeps) | base equilibrium alone: %.3f ms" % (cold["gam"].numel(), t_cold,
      float((cold["info"] & 0xffff).double().mean()), t_base))
