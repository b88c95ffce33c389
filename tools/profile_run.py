#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: the legs of bench.py, each at the size the bench runs it, so that
tools/pmc_summary.py (entries keyed by kernel AND launch size) yields exactly the entries bench.py looks up.
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o fetch -- python3 tools/profile_run.py [leg ...]
legs (default: all but `refine`): headline stress sturm scan_large ncsx c5 refine"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

legs = sys.argv[1:] or ["headline", "stress", "sturm", "scan_large", "ncsx", "c5"]
dev = torch.device("cuda", 0)
ctx = ibs_amd.Context(0)
h, geo7, dP, th0, *_ = bench.build_workload(0, dev)
if "headline" in legs:
    plan = ibs_amd.ScanPlan(ctx, h, geo7, dP, th0, bench.N_SURF)
    for _ in range(20):
        plan.scan_argmax()
    torch.cuda.synchronize()
    print("headline", ctx.last_launch(), flush=True)
n = int(os.environ.get("IBS_STRESS_N", "262144"))
if "stress" in legs:
    s = bench.stress(ctx, dev, n, "smooth", reps=2)
    print("stress", s["solves_per_s"], ctx.last_launch(), flush=True)
    s = bench.stress(ctx, dev, max(n // 4, 1024), "rough", reps=2)
    print("stress_rough", s["solves_per_s"], ctx.last_launch(), flush=True)
if "sturm" in legs:
    w = bench.sturm_sweep(ctx, dev, n, reps=2)
    print("sturm", w["sweeps_per_s"], ctx.last_launch(), flush=True)
if "scan_large" in legs:
    w = bench.scan_large(ctx, dev, geo7, dP, reps=2)
    print("scan_large", w["solves_per_s"], ctx.last_launch(), flush=True)
if "ncsx" in legs:
    w = bench.ncsx_pipeline(ctx, dev)          # configs[2] shape (geometry + scan) and the reference batch (+ its refinement)
    print("ncsx", {k: (v["geometry_ms"], v["scan_ms"]) for k, v in w.items()}, flush=True)
if "c5" in legs:
    w = bench.c5_matrix(ctx, dev, budget_s=600.0)
    print("c5", [(r.get("n_zeta"), r.get("family"), r.get("mode"), r.get("solves_per_s")) for r in w["rows"]], flush=True)
if "refine" in legs:
    # F2 refinement: the config-4 shape (365 maxima in one batch) once
    wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
    th = ibs_amd.theta_grid(969)
    sv5 = np.linspace(0.5, 0.95, 5)
    big = ibs_amd.SurfaceTables.from_wouts([wout] * 73, sv5)
    scan = ibs_amd.BallooningScan(ctx, None, th, np.tile(sv5, 73), tables=big, device=dev, surf_index=np.arange(365))
    rows, bad = scan.device_rows(True)
    torch.cuda.synchronize()
    print("refine (365 maxima) stats", ctx.refine_stats(), flush=True)
