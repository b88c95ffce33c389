import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N, nt0 = 1025, 16
th = ibs_amd.theta_grid(N)
t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
for ns, na in ((2, 32), (4, 32), (8, 32), (16, 32), (32, 32)):
    tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.1, 0.95, ns))
    surf = np.repeat(np.arange(ns), na); al = np.tile(np.linspace(0, np.pi, na), ns)
    r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    geo7 = [r["geo"][k] for k in range(7)]
    line = "%6d solves:" % (ns * na * nt0)
    for chain in (1, 2, 4):
        os.environ["IBS_SCAN_CHAIN"] = str(chain)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = ctx.gamma_scan(th[1] - th[0], *geo7, r["dPdrho"], t0); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        line += "  chain %d: %.1f us" % (chain, best * 1e3)
    os.environ.pop("IBS_SCAN_CHAIN")
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = ctx.gamma_scan(th[1] - th[0], *geo7, r["dPdrho"], t0); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(line + "  auto: %.1f us" % (best * 1e3), flush=True)
