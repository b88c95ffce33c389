#!/usr/bin/env python3
"""Round 6: what a timed region of K bench steps costs beyond K kernels -- K = 1 ... 1000, with the HIP-event brackets of the
dominant kernel on every 8th step (bench.py's live kernel time), on every 64th, and without.   python tools/experiments/step_fixed_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF, n_pack=3)
for _ in range(6000):
    plan.scan_argmax(0)
torch.cuda.synchronize()


def region(K, every):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K // every + 1)] if every else None
    if every and PRE:
        for a, b in evs:
            a.record(); b.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        if every and k % every == 0:
            evs[k // every][0].record(); plan.scan_argmax(0); evs[k // every][1].record()
        else:
            plan.scan_argmax(0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


PRE = len(sys.argv) > 1            # any argument: the events are recorded once before the region (torch creates them lazily)
for K in (1, 5, 20, 100, 1000):
    row = []
    for every in (0, 64, 8):
        ts = [region(K, every) for _ in range(15)]
        row.append("%s: %8.1f us total, %6.2f us/step" % ({0: "no events", 64: "events every 64th", 8: "events every 8th"}[every], np.median(ts), np.median(ts) / K))
    print("K = %4d   " % K + "   |   ".join(row), flush=True)
