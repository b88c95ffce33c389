"""geometry-fed scan at growing batch sizes (N = 513): time per launch and solves/s"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
g3 = np.load(os.path.join(ROOT, "tests", "golden", "G3_ncsx_lines.npz")); geo = g3["geo_513"]
th = ibs_amd.theta_grid(513)
for n_surf, n_alpha, n_t0 in ((16, 8, 8), (32, 8, 8), (64, 8, 8), (64, 16, 8), (64, 32, 16), (256, 32, 16)):
    nl = n_surf * n_alpha
    rng = np.random.default_rng(3)
    base = geo[np.arange(nl) % len(geo)].copy()
    base[:, 4:7, :] *= (1 + rng.uniform(-0.03, 0.03, nl))[:, None, None]
    dP = -0.5 * np.mean((base[:, 2] - base[:, 7]) * base[:, 0] ** 2, axis=1)
    geo7 = [torch.from_numpy(np.ascontiguousarray(base[:, k, :])).to(dev) for k in range(7)]
    plan = ibs_amd.ScanPlan(ctx, th[1] - th[0], geo7, torch.from_numpy(dP).to(dev),
                            torch.from_numpy(np.linspace(0, np.pi / 2, n_t0)).to(dev), n_surf=n_surf)
    plan(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); plan.scan(); b.record()
    torch.cuda.synchronize()
    ms = min(a.elapsed_time(b) for a, b in ev)
    print("%6d solves (%d x %d x %d): scan %.1f us  %.3e solves/s  sweeps %.2f" % (nl * n_t0, n_surf, n_alpha, n_t0, ms * 1e3, nl * n_t0 / (ms * 1e-3), float((plan.info & 0xffff).double().mean())), flush=True)
