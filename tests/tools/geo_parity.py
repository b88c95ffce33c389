"""max relative deviation of the device geometry (rows kernel) from the reference's arrays (G3) per quantity"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, ibs_amd
from oracle import ballooning_oracle as bo
G = os.path.join(ROOT, "tests", "golden")
ctx = ibs_amd.Context(0)
g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz")); ref = dict(np.load(os.path.join(G, "G8_surface_tables.npz")))
tabs = ibs_amd.SurfaceTables.from_arrays(ref)
names = "bmag gradpar cvdrift cvdrift0 gds2 gds21 gds22 gbdrift".split()
for N in (513, 1025):
    th = bo.theta_grid(N); lines = g3["lines_%d" % N]
    surf = [int(np.argmin(np.abs(ref["s"] - s))) for s, a in lines]
    geo_ref = g3["geo_%d" % N]
    r = ctx.fieldline_geometry(tabs, surf, lines[:, 1], th)
    out = []
    for q in range(8):
        scale = np.abs(geo_ref[:, q]).max(axis=1, keepdims=True)
        out.append("%s %.1e" % (names[q], (np.abs(r["geo"][q] - geo_ref[:, q]) / scale).max()))
    print("N=%d:" % N, "  ".join(out))
