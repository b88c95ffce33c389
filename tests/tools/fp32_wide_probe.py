"""Diagnostic (GPU box): FP32 entry point of the raw solver, (a) all-FP32 kernel (lam only), (b) widened form (FP32 in HBM, FP64
solver: k_solve_gcf_wide) against FP64 on the same systems; config-5 families, N_zeta in {256, 512, 1024, 2048}."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import ibs_amd
from tests.test_gpu_configs import c5_family, norm_a
ctx = ibs_amd.Context(0); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
EPS32 = 1.1920929e-07
q = lambda t, pr: float(torch.quantile(t[:min(len(t), 1 << 20)], pr))
def timed(fn):
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) / 2
for nz in (256, 512, 1024, 2048):
    N = nz + 1
    for fam in ("smooth", "rough"):
        h, g, c, f = c5_family(dev, fam, n, N, seed=20240 + nz)
        g32, c32, f32 = g.float(), c.float(), f.float()
        r64 = ctx.solve_gcf(h, g, c, f)
        rw = ctx.solve_gcf(h, g32, c32, f32, want_info=True, dtype=np.float32)
        rl = ctx.solve_gcf(h, g32, c32, f32, want_info=True, dtype=np.float32, want_gam=False)
        r64r = ctx.solve_gcf(h, g32.double(), c32.double(), f32.double())          # FP64 path on the FP32-rounded systems
        nA = norm_a(h, g, c, f)
        t64 = timed(lambda: ctx.solve_gcf(h, g, c, f)); tw = timed(lambda: ctx.solve_gcf(h, g32, c32, f32, dtype=np.float32))
        tl = timed(lambda: ctx.solve_gcf(h, g32, c32, f32, dtype=np.float32, want_gam=False))
        el = (rl["lam"].double() - r64["lam"]).abs() / nA; ew = (rw["lam"].double() - r64["lam"]).abs() / nA
        eg = (rw["gam"].double() - r64["gam"]).abs(); egr = (rw["gam"].double() - r64r["gam"]).abs()
        print("N_zeta %4d %-6s: solves/s f64 %.2e | f32 wide %.2e | f32 lam-only %.2e ; flagged %d / %d" % (
            nz, fam, n / t64, n / tw, n / tl, int(((rw["info"] >> 16) != 0).sum()), int(((rl["info"] >> 16) != 0).sum())))
        print("      lam-only |dlam|/(eps32 ||A||): med %.2f p99.9 %.1f max %.1f | wide: med %.2f p99.9 %.2f max %.2f" % (
            float(el.median()) / EPS32, q(el, .999) / EPS32, float(el.max()) / EPS32, float(ew.median()) / EPS32, q(ew, .999) / EPS32, float(ew.max()) / EPS32))
        print("      wide |dgam| vs f64 on the f64 systems: med %.2e p99 %.2e max %.2e ; /(eps32 ||A||) med %.2f max %.2f | vs f64 on the ROUNDED systems: max %.2e (output rounding: gam * 6e-8)" % (
            float(eg.median()), q(eg, .99), float(eg.max()), float((eg / nA).median()) / EPS32, float((eg / nA).max()) / EPS32, float(egr.max())))
        del g, c, f, g32, c32, f32
        torch.cuda.empty_cache()
