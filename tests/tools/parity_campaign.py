"""one-off parity campaign: many perturbed NCSX-like lines x theta0 values, GPU scan vs the C oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
from oracle import c_oracle as co
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
g3 = np.load(os.path.join(ROOT, "tests", "golden", "G3_ncsx_lines.npz"))
worst = 0.0
for N in (513, 969, 1025):
    geo = g3["geo_%d" % N]
    for seed in range(4):
        rng = np.random.default_rng(100 + seed)
        nl = 256
        base = geo[rng.integers(0, len(geo), nl)].copy()
        base[:, 4:7, :] *= (1 + rng.uniform(-0.2, 0.2, nl))[:, None, None]
        sc = (1 + rng.uniform(-0.5, 1.5, nl))
        base[:, 2:4, :] *= sc[:, None, None]; base[:, 7, :] *= sc[:, None]
        dP = -0.5 * np.mean((base[:, 2] - base[:, 7]) * base[:, 0] ** 2, axis=1) * (1 + rng.uniform(-0.5, 3.0, nl))
        th0 = np.sort(rng.uniform(-1.0, 2.5, 16))
        h = 8 * np.pi / (N - 1)
        arrs = [np.ascontiguousarray(base[:, k, :]) for k in range(7)]
        r = ctx.gamma_scan(h, *[torch.from_numpy(a).to(dev) for a in arrs], torch.from_numpy(dP).to(dev), torch.from_numpy(th0).to(dev), want_info=True)
        gam = r["gam"].cpu().numpy(); info = r["info"].cpu().numpy()
        gc, lc, _ = co.gamma_scan(h, *arrs, dP, th0, nthreads=16)
        d = np.abs(gam - gc).max()
        worst = max(worst, d)
        print("N=%d seed %d: %d solves  max|dgam| %.2e  max|dlam| %.2e  iters mean %.1f max %d  flagged %d  gam range [%.2e, %.2e]"
              % (N, seed, gam.size, d, np.abs(r["lam"].cpu().numpy() - lc).max(), (info & 0xffff).mean(), (info & 0xffff).max(), int(((info >> 16) != 0).sum()), gam.min(), gam.max()), flush=True)
print("worst", worst)
