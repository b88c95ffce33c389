"""Refinement-only workload for profiling (rocprofv3 --kernel-trace --stats -- python3 tools/refine_profile_run.py [small|big|both]):
small = the reference batch (5 surfaces, N = 969: bench.py's reference_batch leg), big = the config-4 shape (73 equilibria x 5
surfaces = 365 maxima in one batch: tools/bench_pipeline.py).  Prints wall time per ibs_refine_f64 call and evaluations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd

which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N, na, nt0, ns = 969, 24, 15, 5
svals = np.linspace(0.5, 0.95, ns); th = ibs_amd.theta_grid(N)


def starts_of(scan):
    tabs_c = scan.coarse()
    return np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in tabs_c])


def run(name, scan):
    st = starts_of(scan)
    scan.refine_device(st)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        xo, fo, ne = scan.refine_device(st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("%s: %d points, refine %.3f ms (min %.3f), evaluations %d..%d mean %.2f total %d, f_opt[:3] %s" % (
        name, len(st), 1e3 * float(np.median(ts)), 1e3 * min(ts), ne.min(), ne.max(), ne.mean(), ne.sum(), fo[:3]), flush=True)
    return xo, fo, ne


if which in ("small", "both"):
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    run("reference batch", ibs_amd.BallooningScan(ctx, None, th, svals, tables=tabs, device=dev))
if which in ("big", "both"):
    import bench
    wouts, _, _ = bench.emulated_equilibria(wout)          # base + the 72 DOF-stepped equilibria of configs[3]
    n_eq = len(wouts)
    big = ibs_amd.SurfaceTables.from_wouts(wouts, svals)
    # (surface k of the scan = table index k: tables.s repeats per equilibrium, the nearest-s default would map every
    #  equilibrium onto the first one -- which is what this tool measured until round 3)
    run("config-4 shape", ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, n_eq), nalpha=na, ntheta0=nt0, tables=big, device=dev,
                                                 surf_index=np.arange(n_eq * ns)))
print("refine stats of the last call (evaluations, sweeps, rounds, rounds enqueued):", ctx.refine_stats())
