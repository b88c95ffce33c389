import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, ibs_amd
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load("/root/repo/tests/golden/G8_wout_ncsx_op.npz"))
tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.5, 0.95, 5))
N = 969
th = torch.from_numpy(ibs_amd.theta_grid(N)).to(dev)
surf = torch.from_numpy(np.repeat(np.arange(5), 3).astype(np.int32)).to(dev)
al = torch.from_numpy(np.tile(np.array([1.0, 1.002, 1.004]), 5)).to(dev)
for lpp in (4, 8, 4, 8, 0):
    ctx.set_option("geo_lpp", lpp)
    for _ in range(5): ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    torch.cuda.synchronize()
    print("15 lines x 969, geo_lpp=%d: %.1f us per call" % (lpp, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
