"""where do the microseconds of the overlapped gather go?  (1-rank RCCL group on one GPU)
host enqueue time per step vs stream time per step, for: scan only / in-stream gather / overlapped gather / overlapped
without the comm_wait dependency / with the wait folded into the start call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ibs_amd
import bench

dev = torch.device("cuda:0")
ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF, n_pack=4)
ctx.comm_init(None, 0, 1)
g2 = [torch.empty((bench.N_SURF, 2), dtype=torch.float64, device=dev) for _ in range(4)]


def run(mode, n=3000):
    def step(k):
        s = k & 1
        if mode == "scan":
            plan.scan_argmax(s)
        elif mode == "instream":
            plan.scan_argmax(s); ctx.allgather(plan.packs[s], g2[s])
        elif mode == "overlap":
            ctx.comm_wait(s); plan.scan_argmax(s); ctx.allgather_start(plan.packs[s], g2[s], s)
        elif mode == "overlap_nowait":
            plan.scan_argmax(s); ctx.allgather_start(plan.packs[s], g2[s], s)
        elif mode == "overlap_hostwait3":        # three slots, host-side wait on the slot two steps back
            s3 = k % 3
            plan.scan_argmax(s3); ctx.allgather_start(plan.packs[s3], g2[s3], s3, same_stream=True, host_wait=(k + 1) % 3)
        elif mode == "overlap_1call":
            plan.scan_argmax(s); ctx.allgather_start(plan.packs[s], g2[s], s, then_wait=1 - s, same_stream=True)
    for k in range(300):
        step(k)
    ctx.comm_wait(-1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        step(k)
    t1 = time.perf_counter()
    ctx.comm_wait(-1); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-16s host enqueue %.1f us/step   total %.1f us/step" % (mode, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)


for m in ("scan", "instream", "overlap", "overlap_nowait", "overlap_1call", "overlap_hostwait3", "scan"):
    run(m)
ctx.comm_destroy()
