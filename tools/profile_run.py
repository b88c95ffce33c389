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

# F2 refinement: the reference batch (5 surfaces, N = 969) twice and the config-4 shape (365 maxima in one batch) once
th = ibs_amd.theta_grid(969)
sv5 = np.linspace(0.5, 0.95, 5)
tabs5 = ibs_amd.SurfaceTables.from_wout(wout, sv5)
scan5 = ibs_amd.BallooningScan(ctx, None, th, sv5, tables=tabs5, device=dev)
st5 = np.array([ibs_amd.pick_start(t, scan5.alpha_scan, scan5.theta0_scan)[:2] for t in scan5.coarse()])
for _ in range(2):
    scan5.refine_device(st5)
print("refine (reference batch) stats", ctx.refine_stats(), flush=True)
if os.environ.get("IBS_PROFILE_BIG", "1") != "0":
    big = ibs_amd.SurfaceTables.concat([tabs5] * 73)
    scanb = ibs_amd.BallooningScan(ctx, None, th, np.tile(sv5, 73), tables=big, device=dev)
    scanb.own = list(range(365))
    surf_b = np.arange(365)
    xo, fo, ne, rounds = ctx.refine(big, surf_b, np.tile(st5, (73, 1)), th, device=dev)
    print("refine (365 maxima) stats", ctx.refine_stats(), flush=True)
