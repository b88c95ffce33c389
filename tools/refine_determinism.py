import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N = 969; th = ibs_amd.theta_grid(N)
for n_eq in (1, 73):
    svals = np.linspace(0.5, 0.95, 5)
    tabs_all = []
    for q in range(n_eq):
        w = dict(wout)
        if q:
            w["rmnc"] = wout["rmnc"].copy(); w["rmnc"][q % 200, :] *= (1 + 2e-3 * np.linspace(0, 1, wout["rmnc"].shape[1]) ** 2)
        tabs_all.append(ibs_amd.SurfaceTables.from_wout(w, svals))
    big = ibs_amd.SurfaceTables.concat(tabs_all)
    scan = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), tables=big, device=dev, surf_index=np.arange(len(big.s)))
    st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
    runs = [scan.refine_device(st) for _ in range(6)]
    same = all(np.array_equal(runs[0][0], r[0]) and np.array_equal(runs[0][1], r[1]) and np.array_equal(runs[0][2], r[2]) for r in runs[1:])
    print("%d points: 6 refinements bitwise identical (x_opt, f_opt, n_evals): %s" % (len(st), same))
    tabs2 = [scan.coarse() for _ in range(4)]
    print("   coarse tables bitwise identical over 4 scans: %s" % all(np.array_equal(tabs2[0], t) for t in tabs2[1:]))
