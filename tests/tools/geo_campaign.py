"""Parity campaign of the device geometry (every form of k_geo_rows) against the numpy oracle (oracle/geometry_oracle.py =
utils.py:359-720 restated) on random field lines of the NCSX_op equilibrium, and against the reference's own arrays (G3):
max over lines and grid points of |device - oracle| / max|oracle| per array.   python tests/tools/geo_campaign.py [lines]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
from oracle import geometry_oracle as go
from oracle import ballooning_oracle as bo
G = os.path.join(ROOT, "tests", "golden")
n_lines = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
names = "bmag gradpar cvdrift cvdrift0 gds2 gds21 gds22 gbdrift".split()
rng = np.random.default_rng(4)
svals = np.sort(rng.uniform(0.05, 0.98, 12))
tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
otab = go.surface_tables_from_wout(wout, svals)
print("random lines: %d per grid, 12 surfaces s in [%.3f, %.3f], alpha in [0, 2 pi)" % (n_lines, svals[0], svals[-1]))
for N in (513, 969, 1025, 2049):
    th = ibs_amd.theta_grid(N)
    surf = rng.integers(0, len(svals), n_lines); al = rng.uniform(0.0, 2 * np.pi, n_lines)
    t0 = time.time()
    ref = np.stack([go.fieldline_geometry(otab, int(s), np.array([a]), th)[0] for s, a in zip(surf, al)])        # (lines, 8, N)
    t_or = time.time() - t0
    scale = np.abs(ref).max(axis=2, keepdims=True)
    for lpp, nm in ((None, "automatic"), (-2, "two points per lane"), (1, "one lane per point"), (2, "2 lanes per point"), (4, "4 lanes per point"), (8, "8 lanes per point")):
        ctx.set_option("geo_lpp", lpp)
        r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
        got = r["geo"].cpu().numpy().transpose(1, 0, 2)
        err = (np.abs(got - ref) / scale).max(axis=(0, 2))
        print("N = %4d  %-20s %-22s worst %.1e   %s" % (N, nm, ctx.last_launch()[0].replace("ibs::", ""), err.max(), "  ".join("%s %.0e" % (n_, e) for n_, e in zip(names, err))), flush=True)
    ctx.set_option("geo_lpp", None)
    print("          (oracle: %.1f s for %d lines)" % (t_or, n_lines), flush=True)
# the reference's own arrays (golden set G3, captured from utils.vmec_fieldlines)
g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz")); ref8 = dict(np.load(os.path.join(G, "G8_surface_tables.npz")))
t3 = ibs_amd.SurfaceTables.from_arrays(ref8)
for N in (513, 1025):
    th = bo.theta_grid(N); lines = g3["lines_%d" % N]
    surf = [int(np.argmin(np.abs(ref8["s"] - s))) for s, a in lines]
    geo_ref = g3["geo_%d" % N]
    for lpp in (None, -2, 1, 8):
        ctx.set_option("geo_lpp", lpp)
        r = ctx.fieldline_geometry(t3, surf, lines[:, 1], th)
        err = [(np.abs(r["geo"][q] - geo_ref[:, q]) / np.abs(geo_ref[:, q]).max(axis=1, keepdims=True)).max() for q in range(8)]
        print("G3 (reference's arrays) N = %4d geo_lpp %-4s worst %.1e   %s" % (N, lpp, max(err), "  ".join("%s %.0e" % (n_, e) for n_, e in zip(names, err))), flush=True)
    ctx.set_option("geo_lpp", None)
