"""where the wall time of BallooningScan.refine_device goes on the host side (reference batch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
import ctypes as C
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N = 969; svals = np.linspace(0.5, 0.95, 5); th = ibs_amd.theta_grid(N)
tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
scan = ibs_amd.BallooningScan(ctx, None, th, svals, tables=tabs, device=dev)
st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
for _ in range(3): scan.refine_device(st)
lib = ctx._lib
orig = lib.ibs_refine_f64
acc = {"c": 0.0}
class Wrap:
    def __call__(self, *a):
        t = time.perf_counter(); r = orig(*a); acc["c"] += time.perf_counter() - t; return r
lib.ibs_refine_f64 = Wrap()
n = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): scan.refine_device(st)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / n
print("refine_device %.1f us per call: C call %.1f us, Python around it %.1f us" % (tot * 1e6, acc["c"] / n * 1e6, (tot - acc["c"] / n) * 1e6))
