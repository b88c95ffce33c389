"""Diagnostic (GPU box): error of the FP32 variant of the raw (g, c, f) solver against FP64 on the same systems,
config-5 families, N_zeta in {256, 512, 1024, 2048}: lam relative to ||A||, gam absolute."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import ibs_amd
from tests.test_gpu_configs import c5_family, norm_a

ctx = ibs_amd.Context(0)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for nz in (256, 512, 1024, 2048):
    N = nz + 1
    for fam in ("smooth", "rough"):
        h, g, c, f = c5_family(dev, fam, n, N, seed=20240 + nz)
        r64 = ctx.solve_gcf(h, g, c, f, want_info=True)
        r32 = ctx.solve_gcf(h, g.float(), c.float(), f.float(), want_info=True, dtype=np.float32)
        nA = norm_a(h, g, c, f)
        el = (r32["lam"].double() - r64["lam"]).abs() / nA
        eg = (r32["gam"].double() - r64["gam"]).abs()
        egl = (r32["gam"].double() - r64["gam"]).abs() / nA
        ok32 = ((r32["info"] >> 16) == 0)
        print("N_zeta %4d %-6s: flagged32 %d  |dlam|/||A|| max %.2e med %.2e (eps32 = 1.2e-7) | |dgam| max %.2e med %.2e  /||A|| max %.2e | ||A|| med %.1f  sweeps32 %.1f sweeps64 %.1f" % (
            nz, fam, int((~ok32).sum()), float(el.max()), float(el.median()), float(eg.max()), float(eg.median()), float(egl.max()),
            float(nA.median()), float((r32["info"] & 0xffff).double().mean()), float((r64["info"] & 0xffff).double().mean())))
        q = lambda t, pr: float(torch.quantile(t[:min(len(t), 1 << 20)], pr))
        print("      lam/||A|| p90 %.2e p99 %.2e p99.9 %.2e | gam abs p90 %.2e p99 %.2e p99.9 %.2e" % (q(el, .9), q(el, .99), q(el, .999), q(eg, .9), q(eg, .99), q(eg, .999)))
        if fam == "smooth":
            from scipy.linalg import eigh_tridiagonal
            worst = torch.argsort(el, descending=True)[:4].cpu().numpy()
            for k in worst:
                gg, cc, ff = (t[k].cpu().numpy() for t in (g, c, f))
                e = 0.5 * (gg[:-1] + gg[1:]) / h ** 2
                d = cc[1:-1] - (e[:-1] + e[1:])
                a = d / ff[1:-1]; b = e[1:-1] / np.sqrt(ff[1:-2] * ff[2:-1])
                w = eigh_tridiagonal(a, b, eigvals_only=True, select="i", select_range=(len(a) - 3, len(a) - 1))
                print("      worst sys %d: lam32 %.6f lam64 %.6f top3 LAPACK %s sweeps32 %d" % (k, float(r32["lam"][k]), float(r64["lam"][k]), w[::-1], int(r32["info"][k]) & 0xffff))
        del g, c, f, r64, r32
        torch.cuda.empty_cache()
