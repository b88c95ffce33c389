"""End-to-end timing of the per-equilibrium pipeline on the GPU: wout tables -> geometry kernel -> scan."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
wout = dict(np.load(os.path.join(ROOT, 'tests/golden/G8_wout_ncsx_op.npz')))
for name, ns, na, nt0, N in (("reference batch (5 surfaces x 24 alpha x 15 theta0, N=969)", 5, 24, 15, 969),
                             ("config 2 (16 x 8 x 8, N=513)", 16, 8, 8, 513),
                             ("config 3 (64 x 32 x 16, N=1025)", 64, 32, 16, 1025)):
    svals = np.linspace(0.5, 0.95, ns) if ns <= 16 else np.linspace(0.1, 0.95, ns)
    t = time.time(); tabs = ibs_amd.SurfaceTables.from_wout(wout, svals); t_spl = time.time() - t
    th = ibs_amd.theta_grid(N); alphas = np.linspace(0, np.pi, na); t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
    surf = np.repeat(np.arange(ns), na); al = np.tile(alphas, ns)
    for rep in range(2):
        torch.cuda.synchronize(); e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
        e[1].record()
        sc = ctx.gamma_scan(th[1] - th[0], *[r['geo'][k] for k in range(7)], r['dPdrho'], t0, want_info=True)
        e[2].record(); torch.cuda.synchronize()
    nsolve = ns * na * nt0
    print('%s: host splines %.2f s | geometry %.3f ms (%d lines) | scan %.3f ms (%d solves, %.1f sweeps) | gam max per surface[:3] %s' % (
        name, t_spl, e[0].elapsed_time(e[1]), ns * na, e[1].elapsed_time(e[2]), nsolve,
        float((sc['info'] & 0xffff).double().mean()), sc['gam'].reshape(ns, -1).max(dim=1).values[:3].cpu().numpy()))

# config 4 shape: one FD-gradient step of the optimizer = 73 equilibria x 5 surfaces x 24 alpha x 15 theta0, N = 969
n_eq, ns, na, nt0, N = 73, 5, 24, 15, 969
svals = np.linspace(0.5, 0.95, ns); th = ibs_amd.theta_grid(N); alphas = np.linspace(0, np.pi, na)
t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
t = time.time()
tabs_all = []
for q in range(n_eq):
    w = dict(wout)
    if q:
        w["rmnc"] = wout["rmnc"].copy(); w["rmnc"][q % 200, :] *= (1 + 2e-3 * np.linspace(0, 1, wout["rmnc"].shape[1]) ** 2)
    tabs_all.append(ibs_amd.SurfaceTables.from_wout(w, svals))
t_spl = time.time() - t
# one table set holding all equilibria (n_eq * ns "surfaces")
big = ibs_amd.SurfaceTables(np.tile(svals, n_eq), tabs_all[0].xm, tabs_all[0].xn, tabs_all[0].xm_nyq, tabs_all[0].xn_nyq,
                            np.concatenate([t_.tab_mn for t_ in tabs_all]), np.concatenate([t_.tab_nyq for t_ in tabs_all]),
                            np.concatenate([t_.scal[:, 1] for t_ in tabs_all]), np.concatenate([t_.scal[:, 2] for t_ in tabs_all]),
                            np.concatenate([t_.scal[:, 3] for t_ in tabs_all]), tabs_all[0].scal[0, 4], tabs_all[0].scal[0, 5])
surf = np.repeat(np.arange(n_eq * ns), na); al = np.tile(alphas, n_eq * ns)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    r = ctx.fieldline_geometry(big, surf, al, th, device=dev)
    torch.cuda.synchronize(); t_geo = time.time() - t; t = time.time()
    sc = ctx.gamma_scan(th[1] - th[0], *[r['geo'][k] for k in range(7)], r['dPdrho'], t0)
    idx, val = ctx.surface_argmax(sc['gam'].reshape(n_eq * ns, -1))
    torch.cuda.synchronize(); t_scan = time.time() - t
f = ibs_amd.ballooning_objective(np.full(n_eq, 0.8), val.cpu().numpy().reshape(n_eq, ns), -2e-4, 50.0)
print('config 4 shape (%d equilibria, %d lines, %d solves): host splines %.2f s | geometry %.1f ms | scan+argmax %.1f ms | f0[:3] %s' % (
    n_eq, len(surf), n_eq * ns * na * nt0, t_spl, t_geo * 1e3, t_scan * 1e3, f[:3]))

# refinement of all 365 per-surface maxima of that step in one batch (ibs_refine_f64: L-BFGS-B state machines on the device)
scan = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), nalpha=na, ntheta0=nt0, tables=big, device=dev)
tab = sc['gam'].reshape(n_eq * ns, na, nt0).cpu().numpy()
starts = np.array([ibs_amd.pick_start(t_, scan.alpha_scan, scan.theta0_scan)[:2] for t_ in tab])
scan.refine_device(starts)
torch.cuda.synchronize(); t = time.time()
xo, fo, ne = scan.refine_device(starts)
t_ref = time.time() - t
print('config 4 shape: refinement of %d maxima in one batch %.1f ms (%d..%d evaluations per point, mean %.1f)' % (
    len(starts), t_ref * 1e3, int(np.min(ne)), int(np.max(ne)), float(np.mean(ne))))
