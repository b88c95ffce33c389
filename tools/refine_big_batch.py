"""Robustness / scaling check of ibs_refine_f64 on a batch far larger than the reference ever forms: the 5 surfaces of n_eq perturbed
equilibria refined in ONE call (python tools/refine_big_batch.py [n_eq=400]: 2,000 points in 35 ms, 21,200 evaluations, 3 GB), and the
unperturbed equilibrium's five optima compared with the same five refined alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N, ns = 969, 5
svals = np.linspace(0.5, 0.95, ns); th = ibs_amd.theta_grid(N)
n_eq = int(sys.argv[1]) if len(sys.argv) > 1 else 400
t0 = time.time()
base = ibs_amd.SurfaceTables.from_wout(wout, svals)
tabs_all = []
for q in range(n_eq):
    w = dict(wout)
    if q:
        w["rmnc"] = wout["rmnc"].copy(); w["rmnc"][q % 200, :] *= (1 + 2e-3 * (1 + q / n_eq) * np.linspace(0, 1, wout["rmnc"].shape[1]) ** 2)
    tabs_all.append(ibs_amd.SurfaceTables.from_wout(w, svals))
big = ibs_amd.SurfaceTables.concat(tabs_all)
print("tables for %d equilibria: %.1f s" % (n_eq, time.time() - t0), flush=True)
scan = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), tables=big, device=dev, surf_index=np.arange(len(big.s)))
tabs_c = scan.coarse()
st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in tabs_c])
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    xo, fo, ne = scan.refine_device(st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%d points: refine %.2f ms, evaluations %d..%d (total %d), stats %s, finite %s, mem %.2f GB" % (len(st), 1e3 * dt, ne.min(), ne.max(), ne.sum(), ctx.refine_stats(), np.isfinite(fo).all(), torch.cuda.max_memory_allocated() / 1e9), flush=True)
# the first five points are the unperturbed equilibrium: same optimum as the small batch
small = ibs_amd.BallooningScan(ctx, None, th, svals, tables=base, device=dev)
st5 = np.array([ibs_amd.pick_start(t, small.alpha_scan, small.theta0_scan)[:2] for t in small.coarse()])
x5, f5, n5 = small.refine_device(st5)
print("first equilibrium in the big batch vs alone: max |df| %.2e" % np.abs(fo[:5] - f5).max())
