#!/usr/bin/env python3
"""Rough family of configs[4] on a long grid: is k_solve_gcf_direct as accurate as the staged kernel it replaces?  Both are solved for the
same systems; the systems on which they differ most (the ones whose floating-point Sturm counts are not monotone in the shift) and a
random sample are re-solved by the C oracle (oracle/ibs_oracle.c: Sturm bisection in division form), and each kernel's distance to it is
reported in units of ||A||.      python tests/tools/direct_vs_staged_accuracy.py [n_zeta] [n_sys]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
from oracle import c_oracle as co

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
N = nz + 1
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
for family in ("rough", "smooth"):
    h, g, c, f = bench.c5_family(dev, family, n, N, seed=20240 + nz)
    nA = bench.norm_a(h, g, c, f)
    out = {}
    for d in (0, 1):
        ctx.set_option("gcf_direct", d)
        out[d] = ctx.solve_gcf(h, g, c, f)["lam"]
        print(family, "gcf_direct", d, ctx.last_launch()[0])
    ctx.set_option("gcf_direct", None)
    rel = ((out[0] - out[1]).abs() / nA)
    worst = torch.argsort(rel, descending=True)[:192].cpu().numpy()
    rnd = np.random.default_rng(1).choice(n, size=192, replace=False)
    for name, pick in (("192 systems where the two kernels differ most", worst), ("192 random systems", rnd)):
        pk = torch.from_numpy(pick).to(dev)
        _, lam_c, _ = co.solve_gcf_batch(h, g[pk].cpu().numpy(), c[pk].cpu().numpy(), f[pk].cpu().numpy())
        na = nA[pk].cpu().numpy()
        e0 = np.abs(out[0][pk].cpu().numpy() - lam_c) / na; e1 = np.abs(out[1][pk].cpu().numpy() - lam_c) / na
        print("  N_zeta %d %-7s %-48s |lam - oracle| / ||A||: staged max %.2e median %.2e | direct max %.2e median %.2e | staged vs direct max %.2e" % (
            nz, family, name, e0.max(), np.median(e0), e1.max(), np.median(e1), float(rel[pk].max())), flush=True)
