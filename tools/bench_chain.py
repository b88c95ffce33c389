"""Chained theta0 scan (k_gamma_scan_chain) against one wave per theta0 at the NCSX shapes: time and sweeps per solve
for several chain lengths and warm-start widths (IBS_SCAN_CHAIN / IBS_CHAIN_W1 / IBS_CHAIN_W2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
for ns, na, nt0, N in ((64, 32, 16, 1025), (73 * 5, 24, 15, 969)):
    svals = np.linspace(0.1, 0.95, ns) if ns == 64 else np.tile(np.linspace(0.5, 0.95, 5), 73)
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = ibs_amd.theta_grid(N)
    surf = np.repeat(np.arange(ns), na); al = np.tile(np.linspace(0, np.pi, na), ns)
    r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
    geo7 = [r["geo"][k] for k in range(7)]
    ref = None
    for chain, w1, w2 in ((1, 0.5, 1.0), (2, 0.5, 1.0), (4, 0.5, 1.0), (4, 0.25, 1.0), (4, 1.0, 1.0), (4, 0.5, 0.5), (4, 0.5, 2.0), (8, 0.5, 1.0)):
        os.environ["IBS_SCAN_CHAIN"] = str(chain); os.environ["IBS_CHAIN_W1"] = str(w1); os.environ["IBS_CHAIN_W2"] = str(w2)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = ctx.gamma_scan(th[1] - th[0], *geo7, r["dPdrho"], t0, want_info=True); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        if ref is None:
            ref = out
        print("%6d solves N=%4d chain %d w1 %.2f w2 %.2f: %.3f ms  %.2f sweeps  max|dgam| %.1e flagged %d" % (
            out["gam"].numel(), N, chain, w1, w2, best, float((out["info"] & 0xffff).double().mean()),
            float((out["gam"] - ref["gam"]).abs().max()), int(((out["info"] >> 16) != 0).sum())), flush=True)
