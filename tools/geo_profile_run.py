"""Geometry kernels alone, for rocprofv3 (kernel trace or --pmc): configs[2] shape, config-4 refinement round, small refinement round.
python3 tools/geo_profile_run.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
for ns, na, N, svals in ((64, 32, 1025, np.linspace(0.1, 0.95, 64)), (365, 3, 969, np.linspace(0.5, 0.95, 365)),
                         (73, 3, 969, np.linspace(0.5, 0.95, 73)), (5, 3, 969, np.linspace(0.5, 0.95, 5))):
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = torch.from_numpy(ibs_amd.theta_grid(N)).to(dev)
    surf = torch.from_numpy(np.repeat(np.arange(ns), na).astype(np.int32)).to(dev)
    al = torch.from_numpy(np.tile(np.linspace(0, np.pi, na), ns)).to(dev)
    for _ in range(reps):
        r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    torch.cuda.synchronize()
