import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, ibs_amd
from oracle import ballooning_oracle as bo
ctx = ibs_amd.Context(0)
G = np.load(os.path.join(ROOT, 'tests/golden/G1_salpha.npz')); P = G['params']; N = 1025; th = bo.theta_grid(N); h = th[1]-th[0]
par = P[P[:,0]==N][:,1:]
g = np.empty((len(par), N)); c = np.empty_like(g)
for k,(sh,al,t0) in enumerate(par): g[k], c[k] = bo.salpha_gc(th, sh, al, t0)
r = ctx.solve_gcf(h, g, c, g, want_info=True)
lo = np.array([bo.solve_gcf(th, g[k], c[k], g[k])[1] for k in range(len(par))])
err = r['lam'] - lo
normA = 4/h**2 + 4
print('normA', normA, 'eps*normA', 2.2e-16*normA, 'tol', 64*2.2e-16*normA)
print('err/ (eps normA):', np.round(err/(2.2e-16*normA), 1))
print('iters', r['info'] & 0xffff)
for m in (1, 4, 16, 64, 256):
    d = m*2.2e-16*normA
    ca = ctx.sturm_count(h, g, c, g, lo + d); cb = ctx.sturm_count(h, g, c, g, lo - d)
    print('m=%d  above!=0: %d  below!=1: %d' % (m, int((ca != 0).sum()), int((cb != 1).sum())))
