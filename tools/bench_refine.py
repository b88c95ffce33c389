"""Refinement of the per-surface maxima (ball_scan.py:305-339): host-driven batched quasi-Newton vs the device state machine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
for ns, N in ((5, 969), (16, 513), (64, 1025)):
    svals = np.linspace(0.5, 0.95, ns) if ns <= 16 else np.linspace(0.1, 0.95, ns)
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = ibs_amd.theta_grid(N)
    scan = ibs_amd.BallooningScan(ctx, None, th, svals, tables=tabs, device=dev)
    tabs_c = scan.coarse()
    starts = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in tabs_c])
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time(); xh, fh, nev = scan.refine_batched(starts); t_h = time.time() - t
        torch.cuda.synchronize(); t = time.time(); xd, fd, ne = scan.refine_device(starts); t_d = time.time() - t
    print("%2d surfaces, N=%4d: host-driven %.2f ms (%d lockstep evaluations) | device state machine %.2f ms (max %d, mean %.1f evaluations per surface) | max |df| %.1e" % (
        ns, N, t_h * 1e3, nev, t_d * 1e3, ne.max(), ne.mean(), np.abs(fd - fh).max()), flush=True)
