"""Batch size per round of the 365-point refinement (config-4 shape) and what the geometry kernel's quantisation costs: the main
form runs one line (116 us) per CU at a time, so 876 lines = 3.42 per CU take 4 line times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N, ns = 969, 5
svals = np.linspace(0.5, 0.95, ns); th = ibs_amd.theta_grid(N)
import bench
wouts, _, _ = bench.emulated_equilibria(wout)          # base + the 72 DOF-stepped equilibria of configs[3]
n_eq = len(wouts)
big = ibs_amd.SurfaceTables.from_wouts(wouts, svals)
scan = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), tables=big, device=dev, surf_index=np.arange(n_eq * ns))
st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
xo, fo, ne = scan.refine_device(st)
nc = [int((ne > r).sum()) for r in range(ne.max())]
print("n_c per round:", nc)
print("lines per round:", [3 * n for n in nc])
unit = 116.0
cur = sum(int(np.ceil(3 * n / 256)) * unit for n in nc)
ideal = sum(3 * n / 256 * unit for n in nc)
print("quantised model %.0f us, perfectly packed %.0f us" % (cur, ideal))
