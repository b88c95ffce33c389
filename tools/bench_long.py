#!/usr/bin/env python3
"""Round 6: the long-grid path (csrc/ibs_long.hip) -- s-alpha systems at N = 2,561 ... 65,537: latency of one system and rate of a
batch, growth rate wanted / eigenvalue only; the C oracle on one core beside it.      python tools/bench_long.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
for N in (2561, 4097, 8193, 16385, 65537):
    th = np.linspace(-4 * np.pi, 4 * np.pi, N); h = float(th[1] - th[0])
    rng = np.random.default_rng(N)
    for n in (1, 2048) + ((8192,) if N <= 8193 else ()):
        sh, al, t0 = rng.uniform(0.1, 2.0, (n, 1)), rng.uniform(0.0, 1.2, (n, 1)), rng.uniform(0.0, np.pi / 2, (n, 1))
        lam = sh * (th[None] - t0) - al * (np.sin(th)[None] - np.sin(t0))
        g = torch.from_numpy(1 + lam ** 2).to(dev); c = torch.from_numpy(al * (np.cos(th)[None] + np.sin(th)[None] * lam)).to(dev); f = g.clone()
        for want_gam in (True, False):
            ctx.solve_gcf(h, g, c, f, want_gam=want_gam); torch.cuda.synchronize()
            t0_ = time.perf_counter(); r = ctx.solve_gcf(h, g, c, f, want_gam=want_gam, want_info=True); torch.cuda.synchronize()
            dt = time.perf_counter() - t0_
            print("N %6d  %5d systems  gam=%d  %-32s %9.3f ms  = %9.1f systems/s   passes %.1f  flagged %d" % (
                N, n, want_gam, ctx.last_launch()[0], dt * 1e3, n / dt, float((r["info"] & 0xffff).double().mean()), int(((r["info"] >> 16) != 0).sum())), flush=True)
