"""ibs_refine_f64 with the alpha-tangent staged in LDS (option refine_tangent = 1: one evaluation block per CU at N = 969) against read
from global memory in the sums (0: two blocks per CU) on 365 and 2,000 points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N, ns = 969, 5
svals = np.linspace(0.5, 0.95, ns); th = ibs_amd.theta_grid(N)
for n_eq in (73, 400):
    tabs_all = []
    for q in range(n_eq):
        w = dict(wout)
        if q:
            w["rmnc"] = wout["rmnc"].copy(); w["rmnc"][q % 200, :] *= (1 + 2e-3 * np.linspace(0, 1, wout["rmnc"].shape[1]) ** 2)
        tabs_all.append(ibs_amd.SurfaceTables.from_wout(w, svals))
    big = ibs_amd.SurfaceTables.concat(tabs_all)
    scan = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), tables=big, device=dev, surf_index=np.arange(len(big.s)))
    st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
    res = {}
    for opt in (1, 0, None):
        ctx.set_option("refine_tangent", opt)
        scan.refine_device(st)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            xo, fo, ne = scan.refine_device(st)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        res[opt] = fo
        print("%d points, refine_tangent=%s: %.3f ms (min %.3f), evaluations %d, stats %s" % (len(st), opt, 1e3 * np.median(ts), 1e3 * min(ts), ne.sum(), ctx.refine_stats()), flush=True)
    print("   max |f(1) - f(0)| %.2e" % np.abs(res[1] - res[0]).max())
