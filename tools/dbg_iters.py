import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
h, geo7, dP, th0, base, dPn, t0n = bench.build_workload(0, dev)
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP, th0, bench.N_SURF); plan(); torch.cuda.synchronize()
it = (plan.info.cpu().numpy() & 0xffff)
print('iters: mean %.2f max %d' % (it.mean(), it.max())); print(np.bincount(it.ravel()))
i, j = np.unravel_index(np.argmax(it), it.shape); print('worst line', i, 'theta0 idx', j)
# dump the worst system's g,c,f for offline analysis
from oracle import ballooning_oracle as bo
line = base[i]; cv, gd = bo.fold_theta0(t0n[j], line[2], line[3], line[4], line[5], line[6]); g, c, f = bo.gcf(dPn[i], line[0], line[1], cv, gd)
np.savez(os.path.join(ROOT, 'gpurun_out', 'worst_sys.npz'), g=g, c=c, f=f, h=h)
