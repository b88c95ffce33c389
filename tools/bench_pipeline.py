"""End-to-end timing of the per-equilibrium pipeline on the GPU: wout tables -> geometry kernel -> scan."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
from oracle import ballooning_oracle as bo
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
wout = dict(np.load(os.path.join(ROOT, 'tests/golden/G8_wout_ncsx_op.npz')))
for name, ns, na, nt0, N in (("reference batch (5 surfaces x 24 alpha x 15 theta0, N=969)", 5, 24, 15, 969),
                             ("config 2 (16 x 8 x 8, N=513)", 16, 8, 8, 513),
                             ("config 3 (64 x 32 x 16, N=1025)", 64, 32, 16, 1025)):
    svals = np.linspace(0.5, 0.95, ns) if ns <= 16 else np.linspace(0.1, 0.95, ns)
    t = time.time(); tabs = ibs_amd.SurfaceTables.from_wout(wout, svals); t_spl = time.time() - t
    th = bo.theta_grid(N); alphas = np.linspace(0, np.pi, na); t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
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
