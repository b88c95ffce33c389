#!/usr/bin/env python3
"""Host side of AdjointStep.run() on the configs[3] shape: wall time of run() against its GPU phases, and a cProfile of five runs
(where the Python / ctypes time of the call goes).   python tools/c4_host_profile.py"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

ctx = ibs_amd.Context(0)
dev = torch.device("cuda", 0)
wout0 = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
wouts, steps, x0 = bench.emulated_equilibria(wout0)
wouts = [{k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in w.items()} for w in wouts]
n_eq, ns = len(wouts), 5
svals = np.linspace(0.5, 0.95, ns)
th = ibs_amd.theta_grid_for(11, 11)
f_other = 0.8 + 0.01 * np.arange(n_eq)
step = ibs_amd.AdjointStep(ctx, th, svals, dev, nalpha=24, ntheta0=15)
for _ in range(3):
    step.run(wouts, f_other, steps)
ts = []
for _ in range(9):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step.run(wouts, f_other, steps); ts.append((time.perf_counter() - t0) * 1e3)
ph = {}
step.run(wouts, f_other, steps, phases=ph)
print("run(): median %.3f ms, min %.3f; phases (separate pass) %s" % (np.median(ts), min(ts), {k: round(float(v), 3) for k, v in ph.items()}))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step.run(wouts, f_other, steps)
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22)
    print("\n".join(l[:170] for l in s.getvalue().split("\n")[4:40]))
