"""Sweeps per evaluation of the refinement, point by point (the reference batch refined one surface at a time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N = 969; th = ibs_amd.theta_grid(N)
for s in np.linspace(0.5, 0.95, 5):
    sv = np.array([s])
    tabs = ibs_amd.SurfaceTables.from_wout(wout, sv)
    scan = ibs_amd.BallooningScan(ctx, None, th, sv, tables=tabs, device=dev)
    st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
    xo, fo, ne = scan.refine_device(st)
    ev, sw, rounds, enq = ctx.refine_stats()
    print("s = %.4f: start %s -> x_opt %s gam %.6e  evaluations %d  sweeps %d (%.2f per evaluation)  rounds %d" % (s, np.round(st[0], 4), np.round(xo[0], 5), -fo[0], ev, sw, sw / max(ev, 1), rounds))
