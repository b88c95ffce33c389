import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
from oracle import ballooning_oracle as bo
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
h, geo7, dP, th0, base, dPn, t0n = bench.build_workload(0, dev)
r = ctx.gamma_scan(h, *[base[:, k, :] for k in range(7)], dPn, t0n, want_info=True)
it = r['info'] & 0xffff
print('scan path iters max', it.max(), 'argmax', np.unravel_index(np.argmax(it), it.shape))
i, j = np.unravel_index(np.argmax(it), it.shape)
# same systems through the raw path
G = np.empty((8, 513)); C = np.empty_like(G); F = np.empty_like(G)
for jj in range(8):
    line = base[i]; cv, gd = bo.fold_theta0(t0n[jj], line[2], line[3], line[4], line[5], line[6]); G[jj], C[jj], F[jj] = bo.gcf(dPn[i], line[0], line[1], cv, gd)
r2 = ctx.solve_gcf(h, G, C, F, want_info=True)
print('line', i, 'scan iters', it[i], 'raw iters', r2['info'] & 0xffff)
print('lam diff', r['lam'][i] - r2['lam'])
# only that line through the scan path
r3 = ctx.gamma_scan(h, *[base[i:i+1, k, :] for k in range(7)], dPn[i:i+1], t0n, want_info=True)
print('single-line scan iters', r3['info'] & 0xffff)
