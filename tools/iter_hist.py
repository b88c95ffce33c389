import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF)
plan.scan_argmax(); torch.cuda.synchronize()
it = (plan.info.cpu().numpy() & 0xffff).ravel()
print("mean %.2f  std %.2f  min %d  max %d" % (it.mean(), it.std(), it.min(), it.max()))
print(np.bincount(it))
