import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, ibs_amd, bench
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
for _ in range(2):
    s = bench.stress(ctx, dev, 262144, "smooth", reps=5); r = bench.stress(ctx, dev, 65536, "rough", reps=5)
    h, geo7, dP_d, th0_d, *_ = bench.build_workload(0, dev)
    l = bench.scan_large(ctx, dev, geo7, dP_d, reps=5)
    print("stress %.4g (%.2f sweeps)  rough %.4g  scan_large %.4g (%.2f sweeps)" % (s["solves_per_s"], s["mean_sweeps"], r["solves_per_s"], l["solves_per_s"], l["mean_sweeps"]), flush=True)
