"""per-pass time of ONE rank's share of the sharded configs[2] leg (bench.C2Sharded.local_rows) for the shares that
2 / 4 / 8 ranks get, on one GPU: where strong scaling ends up host- or latency-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ibs_amd
import bench

dev = torch.device("cuda:0")
ctx = ibs_amd.Context(0)
job = bench.C2Sharded(ctx, dev)
for world in (1, 2, 4, 8):
    own = ibs_amd.shard_surfaces(job.NS, 0, world)
    for _ in range(3):
        job.local_rows(own)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        job.local_rows(own)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("share of %d ranks (%2d surfaces, %5d solves): host enqueue %.3f ms/pass, total %.3f ms/pass" % (
        world, len(own), len(own) * job.NA * job.NT0, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
