#!/usr/bin/env python3
"""Round 6: the FP32 eigenvalue-only row of configs[4] at N_zeta = 256 (k_solve_gcf_g<double, 16, 16, float> without its growth-rate
stage) with the closing checks off / on, interleaved; FP64 and the growth-rate forms beside it.   python tools/experiments/f32lam_256_ab.py [nz:family ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
n = 1 << 20
cases = [(int(a.split(":")[0]), a.split(":")[1]) for a in sys.argv[1:]] or [(256, "smooth"), (256, "rough"), (512, "smooth")]
for nz, family in cases:
    h, g, c, f = bench.c5_family(dev, family, n, nz + 1, seed=20240 + nz)
    g32, c32, f32 = g.float(), c.float(), f.float()
    calls = {"f64 gam": lambda: ctx.solve_gcf(h, g, c, f), "f64 lam": lambda: ctx.solve_gcf(h, g, c, f, want_gam=False),
             "f32 gam": lambda: ctx.solve_gcf(h, g32, c32, f32, dtype=np.float32), "f32 lam": lambda: ctx.solve_gcf(h, g32, c32, f32, dtype=np.float32, want_gam=False)}
    for name, call in calls.items():
        best = {0: 1e9, 1: 1e9}
        for rep in range(6):
            for mode in (0, 1):
                ctx.set_option("reclose", mode)
                call(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); torch.cuda.synchronize()
                best[mode] = min(best[mode], a.elapsed_time(b))
        ctx.set_option("reclose", None)
        rc = ctx.solve_gcf(h, g, c, f, want_info=True)["info"] if name == "f64 gam" else None
        if rc is not None: print("   re-closed in division form: %d" % int((((rc >> 16) & 8) != 0).sum()))
        print("N_zeta %d %-6s %-8s %-46s checks off %.3e  on %.3e solves/s (%+.1f %%)" % (nz, family, name, ctx.last_launch()[0], n / best[0] * 1e3, n / best[1] * 1e3,
                                                                                 100 * (best[0] / best[1] - 1)), flush=True)
