"""field-line geometry kernel (k_geo_rows): every form (geo_lpp = -2: two points per lane; 1 | 2 | 4 | 8 lanes per point; 0 = the
library's choice) on the configs[2] shape (64 surfaces x 32 alpha, N = 1025), the config-4 shape, the reference batch
(5 x 24, N = 969) and the refinement rounds' batches (3 lines per point)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ibs_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
shapes = ((64, 32, 1025, np.linspace(0.1, 0.95, 64)), (16, 32, 1025, np.linspace(0.1, 0.95, 16)),
          (73 * 5, 24, 969, np.linspace(0.5, 0.95, 73 * 5)), (73 * 5, 3, 969, np.linspace(0.5, 0.95, 73 * 5)),
          (73, 3, 969, np.linspace(0.5, 0.95, 73)),
          (16, 8, 513, np.linspace(0.5, 0.95, 16)), (5, 24, 969, np.linspace(0.5, 0.95, 5)), (5, 3, 969, np.linspace(0.5, 0.95, 5)),
          (1, 3, 969, np.linspace(0.5, 0.95, 1)))
only = sys.argv[1:] and [int(v) for v in sys.argv[1].split(",")]
for ns, na, N, svals in shapes:
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = torch.from_numpy(ibs_amd.theta_grid(N)).to(dev)
    surf = torch.from_numpy(np.repeat(np.arange(ns), na).astype(np.int32)).to(dev)
    al = torch.from_numpy(np.tile(np.linspace(0, np.pi, na), ns)).to(dev)
    ref = None
    for lpp in (only or (1, -2, 2, 4, 8, 0)):
        ctx.set_option("geo_lpp", lpp)
        for _ in range(3):
            r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        g = r["geo"]
        if ref is None:
            ref = g.clone()
        scale = ref.abs().amax(dim=2, keepdim=True)
        print("%5d lines x %d points  geo_lpp=%2d  %.3f ms  %.3e points/s  max rel |d| vs first %.2e" % (
            ns * na, N, lpp, dt * 1e3, ns * na * N / dt, float(((g - ref).abs() / scale).max().item())), flush=True)
    ctx.reset_options()
