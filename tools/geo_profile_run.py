"""workload for rocprofv3 passes over the field-line geometry kernels (configs[2] shape, one and two points per lane)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ibs_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
ns, na, N = 64, 32, 1025
tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.1, 0.95, ns))
th = torch.from_numpy(ibs_amd.theta_grid(N)).to(dev)
surf = torch.from_numpy(np.repeat(np.arange(ns), na).astype(np.int32)).to(dev)
al = torch.from_numpy(np.tile(np.linspace(0, np.pi, na), ns)).to(dev)
for lpp in (1, -2):
    ctx.set_option("geo_lpp", lpp)
    for _ in range(4):
        r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    torch.cuda.synchronize()
