#!/usr/bin/env python3
"""Round 5: the remainder of a geometry batch beyond a multiple of the CU count in a finer form (option geo_split: 0 = on, -1 = off; needs tools/experiments/geo_split_tail.patch applied: measured and dropped, docs/EXPERIMENTS.md R5.4).
(1) the geometry call alone over line counts around the rounds of 256; (2) the refinement of the configs[3] batch (365 maxima)
and of the reference batch, both ways, with the optima compared.      python tools/geo_split_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench

ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N, na, nt0, ns = 969, 24, 15, 5
svals = np.linspace(0.5, 0.95, ns); th = ibs_amd.theta_grid(N)
tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
th_d = torch.from_numpy(th).to(dev)
print("geometry call alone, N = %d: us per call (HIP events, best of 5)   lines | split off | split on | ratio | arrays equal" % N)
for nl in (257, 300, 384, 385, 520, 640, 699, 780, 876, 1030, 1095, 1152, 1300, 2100):
    surf = torch.from_numpy((np.arange(nl) // 3 % ns).astype(np.int32)).to(dev)
    al = torch.from_numpy(np.linspace(0, np.pi, nl)).to(dev)
    res = {}
    for opt in (-1, 0):
        ctx.set_option("geo_split", opt)
        r = ctx.fieldline_geometry(tabs, surf, al, th_d, device=dev)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r = ctx.fieldline_geometry(tabs, surf, al, th_d, device=dev); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3)
        res[opt] = (best, r["geo"].clone())
    ctx.set_option("geo_split", None)
    print("   %5d | %8.1f | %8.1f | %.3f | %s" % (nl, res[-1][0], res[0][0], res[0][0] / res[-1][0],
                                                bool(torch.equal(res[-1][1], res[0][1]))), flush=True)

wouts, _, _ = bench.emulated_equilibria(wout)
n_eq = len(wouts)
big = ibs_amd.SurfaceTables.from_wouts(wouts, svals)
scan_big = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), nalpha=na, ntheta0=nt0, tables=big, device=dev, surf_index=np.arange(n_eq * ns))
scan_small = ibs_amd.BallooningScan(ctx, None, th, svals, tables=tabs, device=dev)
for name, scan in (("configs[3] batch (365 maxima)", scan_big), ("reference batch (5 maxima)", scan_small)):
    st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
    out = {}
    for opt in (-1, 0, -1, 0):
        ctx.set_option("geo_split", opt)
        scan.refine_device(st)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter()
            xo, fo, ne = scan.refine_device(st)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        out[opt] = (fo.copy(), ne.copy())
        print("%s, geo_split %2d: refine %.3f ms (min %.3f), evaluations %d, rounds %d" % (name, opt, 1e3 * float(np.median(ts)), 1e3 * min(ts), ne.sum(), ctx.refine_stats()[2]), flush=True)
    ctx.set_option("geo_split", None)
    print("   max |f_opt(split) - f_opt(no split)| = %.2e; evaluations equal: %s" % (np.abs(out[0][0] - out[-1][0]).max(), bool((out[0][1] == out[-1][1]).all())))
