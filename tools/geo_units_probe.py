"""What a UNIT (one line = eight wave-items) of the main geometry form costs in the refinement's batches: every point brings its own
surface, so a block re-stages its table image every three lines.  Times the geometry call (HIP events) for batches of
n_pts x 3 lines with one surface per point against the same number of lines on 5 surfaces, at line counts that are and are
not multiples of the CU count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
wouts, _, _ = bench.emulated_equilibria(wout)
sv = np.linspace(0.5, 0.95, 5)
many = ibs_amd.SurfaceTables.from_wouts(wouts, sv)          # 365 surfaces
few = ibs_amd.SurfaceTables.from_wout(wout, sv)
th = torch.from_numpy(ibs_amd.theta_grid(969)).to(dev)


def timed(tabs, surf, al, lpp, reps=20):
    ctx.set_option("geo_lpp", lpp)
    d_s = torch.from_numpy(surf.astype(np.int32)).to(dev); d_a = torch.from_numpy(al).to(dev)
    for _ in range(3):
        ctx.fieldline_geometry(tabs, d_s, d_a, th, device=dev)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for k in range(reps):
        ctx.fieldline_geometry(tabs, d_s, d_a, th, device=dev); e[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[k].elapsed_time(e[k + 1]) for k in range(reps)])) * 1e3, ctx.last_launch()


rng = np.random.default_rng(0)
for n_pts in (85, 171, 256, 292, 341, 365):
    al3 = (rng.uniform(0.1, 3.0, n_pts)[:, None] + np.array([-0.002, 0.0, 0.002])[None]).reshape(-1)
    for lpp in (-2, 1):
        t_many, k = timed(many, np.repeat(np.arange(n_pts), 3), al3, lpp)
        t_few, _ = timed(few, np.repeat(np.arange(n_pts) % 5, 3), al3, lpp)
        t_sorted, _ = timed(few, np.sort(np.repeat(np.arange(n_pts) % 5, 3)), al3, lpp)
        print("%4d points = %4d lines (%.2f per CU), geo_lpp %2d %-26s: own surface per point %7.1f us | 5 surfaces, interleaved %7.1f | 5 surfaces, sorted %7.1f" % (
            n_pts, 3 * n_pts, 3 * n_pts / 256, lpp, k[0].replace("ibs::", ""), t_many, t_few, t_sorted), flush=True)
