import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
wout = dict(np.load(os.path.join(ROOT, 'tests/golden/G8_wout_ncsx_op.npz')))
for ns, na, N in ((16, 8, 513), (5, 24, 969), (64, 32, 1025)):
    tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.3, 0.95, ns)); th = ibs_amd.theta_grid(N)
    surf = np.repeat(np.arange(ns), na); al = np.tile(np.linspace(0, np.pi, na), ns)
    for lpp in (1, 2, 4):
        os.environ['IBS_GEO_LPP'] = str(lpp)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r = None
            import ctypes
            a.record(); r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print('lines %d N %d LPP %d: %.3f ms (incl. host glue)' % (ns * na, N, lpp, min(ts)))
