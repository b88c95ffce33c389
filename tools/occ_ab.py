#!/usr/bin/env python3
"""A/B harness for occupancy experiments on the M = 16 kernels (IBS_LIB_PATH picks the build): configs[4] rows at N_zeta = 256 / 512 /
1024 (2^19 systems, FP64, both families), the configs[2]-shape scan (2,048 lines x 16 theta0, N = 1025) and the configs[3]-shape
scan (8,760 lines x 15 theta0, N = 969), the scans with theta0 chains of 4 (default), 3 and 2.   python tools/occ_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
print("library:", os.environ.get("IBS_LIB_PATH", "default"))


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


n = 1 << 19
for nz in (256, 512, 1024):
    for fam in ("smooth", "rough"):
        h, g, c, f = bench.c5_family(dev, fam, n, nz + 1, seed=20240 + nz)
        ms = timed(lambda: ctx.solve_gcf(h, g, c, f))
        r = ctx.solve_gcf(h, g, c, f, want_info=True)
        print("c5 N_zeta %4d %-6s %-46s %.3e solves/s  sweeps %.2f flagged %d" % (nz, fam, ctx.last_launch()[0], n / (ms * 1e-3),
              float((r["info"] & 0xffff).double().mean()), int(((r["info"] >> 16) != 0).sum())), flush=True)
        del g, c, f
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
for tag, ns, na, nt0, N, svals in (("c3 scan", 64, 32, 16, 1025, np.linspace(0.1, 0.95, 64)), ("c4 scan", 365, 24, 15, 969, np.tile(np.linspace(0.5, 0.95, 5), 73))):
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = ibs_amd.theta_grid(N)
    surf = torch.from_numpy(np.repeat(np.arange(ns), na).astype(np.int32)).to(dev)
    al = torch.from_numpy(np.tile(np.linspace(0, np.pi, na), ns)).to(dev)
    t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
    r = ctx.fieldline_geometry(tabs, surf, al, torch.from_numpy(th).to(dev), device=dev)
    geo = [r["geo"][k] for k in range(7)]
    ref = None
    for chain in (0, 3, 2):
        ctx.set_option("scan_chain", chain)
        ms = timed(lambda: ctx.gamma_scan(th[1] - th[0], *geo, r["dPdrho"], t0))
        sc = ctx.gamma_scan(th[1] - th[0], *geo, r["dPdrho"], t0, want_info=True)
        if ref is None:
            ref = sc["gam"].clone()
        print("%s chain %d %-40s %.3f ms  %.3e solves/s  sweeps %.2f  max|dgam vs default| %.1e" % (tag, chain, ctx.last_launch()[0], ms, ns * na * nt0 / (ms * 1e-3),
              float((sc["info"] & 0xffff).double().mean()), float((sc["gam"] - ref).abs().max())), flush=True)
    ctx.set_option("scan_chain", None)
