#!/usr/bin/env python3
"""Workload for rocprofv3 passes: 20 launches of the bench scan step + 2 stress launches.
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o tag -- python3 tools/profile_run.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
ctx = ibs_amd.Context(0)
h, geo7, dP, th0, *_ = bench.build_workload(0, dev)
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP, th0, bench.N_SURF)
for _ in range(20):
    plan()
torch.cuda.synchronize()
n = int(os.environ.get("IBS_STRESS_N", "262144"))
s = bench.stress(ctx, dev, n, "smooth", reps=2)
print("stress", s["solves_per_s"], flush=True)
w = bench.sturm_sweep(ctx, dev, n, reps=2)
print("sturm", w["sweeps_per_s"], flush=True)

# F1 geometry kernel: 2,048 lines x 1,025 points (3 launches)
import numpy as np  # noqa: E402
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.3, 0.95, 64))
surf = np.repeat(np.arange(64), 32); al = np.tile(np.linspace(0, np.pi, 32), 64)
for _ in range(3):
    ctx.fieldline_geometry(tabs, surf, al, ibs_amd.theta_grid(1025), device=dev)
torch.cuda.synchronize()
print("geometry done", flush=True)
