#!/usr/bin/env python3
"""Host cost of ONE bench step per gather mode, with a one-rank RCCL group (what a one-GPU box can show of the N > 1 step):
time to ENQUEUE 2,000 steps (no synchronisation inside) against the time until they have run.  Where the two are equal the step is
host-bound, not GPU-bound.      python tools/step_host_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
NS = 16
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF, n_pack=NS)
gathered = [torch.empty((bench.N_SURF, 2), dtype=torch.float64, device=dev) for _ in range(NS)]
ctx.comm_init(dist, 0, 1)
issued = [0]


def step(mode):
    if mode.startswith("overlap"):
        ns = int(mode[7:])
        slot = issued[0] % ns
        plan.scan_argmax(slot)
        ctx.allgather_start(plan.packs[slot], gathered[slot], slot, same_stream=True, host_wait=(issued[0] + 1) % ns)
        issued[0] += 1
        return
    plan.scan_argmax(0)
    if mode == "native":
        ctx.allgather(plan.pack, gathered[0])
    elif mode == "torch":
        dist.all_gather_into_tensor(gathered[0], plan.pack)


K = 2000
print("%-28s %12s %12s" % ("mode", "enqueue us", "run us"))
for mode in ("none", "torch", "native", "overlap3", "overlap4", "overlap8", "overlap16"):
    for _ in range(200):
        step(mode)
    ctx.comm_wait(-1); torch.cuda.synchronize()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            step(mode)
        t1 = time.perf_counter()
        ctx.comm_wait(-1); torch.cuda.synchronize()
        t2 = time.perf_counter()
        cur = ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6)
        best = cur if best is None or cur[1] < best[1] else best
    print("%-28s %12.2f %12.2f" % ({"none": "scan only", "torch": "torch.distributed in-stream", "native": "library in-stream"}.get(mode, "library overlapped, %s slots" % mode[7:]), best[0], best[1]), flush=True)
ctx.comm_destroy()
dist.destroy_process_group()
