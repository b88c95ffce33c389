"""Geometry call of ONE library build over the shapes that matter (IBS_LIB_PATH picks the build; tools/geo_ab.sh alternates two
builds inside one gpurun call and compares what they wrote).  python tools/geo_ab.py TAG -> gpurun_out/geo_ab/TAG.npz + timings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
tag = sys.argv[1]
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
wouts, _, _ = bench.emulated_equilibria(wout)
many = ibs_amd.SurfaceTables.from_wouts(wouts, np.linspace(0.5, 0.95, 5))          # 365 surfaces
rng = np.random.default_rng(1)


def timed(tabs, surf, al, N, lpp=0, reps=20):
    ctx.set_option("geo_lpp", lpp)
    th = torch.from_numpy(ibs_amd.theta_grid(N)).to(dev)
    d_s = torch.from_numpy(np.asarray(surf, dtype=np.int32)).to(dev); d_a = torch.from_numpy(np.asarray(al, dtype=np.float64)).to(dev)
    for _ in range(3):
        r = ctx.fieldline_geometry(tabs, d_s, d_a, th, device=dev)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for k in range(reps):
        ctx.fieldline_geometry(tabs, d_s, d_a, th, device=dev); e[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([e[k].elapsed_time(e[k + 1]) for k in range(reps)])) * 1e3, ctx.last_launch()[0], r


out = {}
cases = []
t64 = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.1, 0.95, 64))
cases.append(("c3 64 surfaces x 32 lines, N 1025", t64, np.repeat(np.arange(64), 32), np.tile(np.linspace(0, np.pi, 32), 64), 1025, 0))
cases.append(("c4 365 surfaces x 24 lines, N 969", many, np.repeat(np.arange(365), 24), np.tile(np.linspace(0, np.pi, 24), 365), 969, 0))
t5 = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.5, 0.95, 5))
cases.append(("reference 5 surfaces x 24 lines, N 969", t5, np.repeat(np.arange(5), 24), np.tile(np.linspace(0, np.pi, 24), 5), 969, 0))
for n_pts in (365, 171):
    al3 = (rng.uniform(0.1, 3.0, n_pts)[:, None] + np.array([-0.002, 0.0, 0.002])[None]).reshape(-1)
    cases.append(("refinement round: %d points x 3 lines, own surfaces, N 969" % n_pts, many, np.repeat(np.arange(n_pts), 3), al3, 969, 0))
cases.append(("one point per lane: 365 x 3 lines, N 969 (geo_lpp 1)", many, np.repeat(np.arange(365), 3), al3[:0].tolist() + list(
    (rng.uniform(0.1, 3.0, 365)[:, None] + np.array([-0.002, 0.0, 0.002])[None]).reshape(-1)), 969, 1))
for name, tabs, surf, al, N, lpp in cases:
    t, k, r = timed(tabs, surf, al, N, lpp)
    print("%-4s %-62s %-24s %8.1f us" % (tag, name, k.replace("ibs::", ""), t), flush=True)
    out[name] = r["geo"][:, :72].cpu().numpy()
    out[name + " dPdrho"] = r["dPdrho"].cpu().numpy()
os.makedirs(os.path.join(ROOT, "gpurun_out", "geo_ab"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "geo_ab", tag + ".npz"), **out)
if len(sys.argv) > 2:                                       # compare with an earlier run
    other = np.load(os.path.join(ROOT, "gpurun_out", "geo_ab", sys.argv[2] + ".npz"))
    for k in out:
        a, b = out[k], other[k]
        print("   %-70s max |new - old| / max |old| per array: %.2e" % (k, float(np.max(np.abs(a - b).reshape(a.shape[0], -1).max(-1) / np.abs(b).reshape(b.shape[0], -1).max(-1)))))
