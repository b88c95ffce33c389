"""Which form of the geometry kernel serves a batch of n lines best (N = 969, every line its own surface as in the refinement's
rounds): time per call for geo_lpp = 8, 4, 2, 1 (lanes per point) and -2 (two points per lane) against the automatic choice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
wouts, _, _ = bench.emulated_equilibria(wout)
many = ibs_amd.SurfaceTables.from_wouts(wouts, np.linspace(0.5, 0.95, 5))          # 365 surfaces
th = torch.from_numpy(ibs_amd.theta_grid(969)).to(dev)
rng = np.random.default_rng(2)


def timed(surf, al, lpp, reps=12):
    ctx.set_option("geo_lpp", lpp)
    d_s = torch.from_numpy(np.asarray(surf, dtype=np.int32)).to(dev); d_a = torch.from_numpy(np.asarray(al, dtype=np.float64)).to(dev)
    for _ in range(3):
        ctx.fieldline_geometry(many, d_s, d_a, th, device=dev)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for k in range(reps):
        ctx.fieldline_geometry(many, d_s, d_a, th, device=dev); e[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[k].elapsed_time(e[k + 1]) for k in range(reps)])) * 1e3, ctx.last_launch()[0]


print("%6s %6s | %8s %8s %8s %8s %8s | automatic" % ("points", "lines", "lpp 8", "lpp 4", "lpp 2", "lpp 1", "2 pts"))
for n_pts in (2, 5, 10, 18, 28, 40, 67, 100, 143, 182, 233, 292, 365):
    al3 = (rng.uniform(0.1, 3.0, n_pts)[:, None] + np.array([-0.002, 0.0, 0.002])[None]).reshape(-1)
    surf = np.repeat(np.arange(n_pts), 3)
    ts = [timed(surf, al3, lpp)[0] for lpp in (8, 4, 2, 1, -2)]
    ta, ka = timed(surf, al3, 0)
    print("%6d %6d | %8.1f %8.1f %8.1f %8.1f %8.1f | %8.1f %s" % (n_pts, 3 * n_pts, *ts, ta, ka.replace("ibs::", "")), flush=True)
