"""The configs[2]-shape geometry call alone (64 surfaces x 32 lines, N = 1025), a few times: workload for
rocprofv3 --kernel-trace --stats (what the call consists of besides the row kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.1, 0.95, 64))
th = torch.from_numpy(ibs_amd.theta_grid(1025)).to(dev)
surf = torch.from_numpy(np.repeat(np.arange(64), 32).astype(np.int32)).to(dev)
al = torch.from_numpy(np.tile(np.linspace(0, np.pi, 32), 64)).to(dev)
for _ in range(10):
    ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
torch.cuda.synchronize()
