#!/usr/bin/env python3
"""(Needs tools/experiments/refine_eval_512_threads.patch applied: the option refine_block is not in the library; EXPERIMENTS R5.9.)
k_refine_eval with 256 against 512 threads per block (option refine_block), alternating in one process: the reference batch
(5 maxima, N = 969) and the configs[3] shape (365 maxima).  Median / min wall time of ibs_refine_f64 per setting, evaluations, and
the largest difference of the refined maxima between the two settings (the partial sums are taken per wave: rounding differs).
   python tools/refine_block_ab.py [reps=9]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ctx = ibs_amd.Context(0)
dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
N, na, nt0, ns = 969, 24, 15, 5
svals = np.linspace(0.5, 0.95, ns)
th = ibs_amd.theta_grid(N)
small = ibs_amd.BallooningScan(ctx, None, th, svals, tables=ibs_amd.SurfaceTables.from_wout(wout, svals), device=dev)
wouts, _, _ = bench.emulated_equilibria(wout)
big = ibs_amd.BallooningScan(ctx, None, th, np.tile(svals, len(wouts)), nalpha=na, ntheta0=nt0, tables=ibs_amd.SurfaceTables.from_wouts(wouts, svals),
                             device=dev, surf_index=np.arange(len(wouts) * ns))
for name, scan in (("reference batch (5 maxima)", small), ("configs[3] shape (365 maxima)", big)):
    st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
    ts = {256: [], 512: []}
    out = {}
    for b in (256, 512):
        ctx.set_option("refine_block", b)
        scan.refine_device(st)
    for r in range(reps):
        for b in (256, 512) if r % 2 == 0 else (512, 256):
            ctx.set_option("refine_block", b)
            torch.cuda.synchronize(); t = time.perf_counter()
            out[b] = scan.refine_device(st)
            torch.cuda.synchronize(); ts[b].append(time.perf_counter() - t)
    ctx.set_option("refine_block", None)
    for b in (256, 512):
        xo, fo, ne = out[b]
        print("%-30s %d threads: refine %.3f ms median, %.3f min; evaluations %d (max %d per point)" % (name, b, 1e3 * np.median(ts[b]), 1e3 * min(ts[b]), ne.sum(), ne.max()))
    print("%-30s max |f_opt(256) - f_opt(512)| %.2e, max |x_opt difference| %.2e" % (name, np.abs(out[256][1] - out[512][1]).max(), np.abs(out[256][0] - out[512][0]).max()), flush=True)
