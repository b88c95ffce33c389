#!/usr/bin/env python3
"""Every rows-per-lane instantiation M = 9 ... 32 of the read-from-global-memory kernels (k_solve_gcf_direct, _direct_w2,
k_solve_gcf_f32lam_direct, _w2), each on the shortest and the longest odd grid it serves (the last lane's chunk nearly empty / full):
direct against the LDS-staged kernels on the same systems, both families, ragged batch.
  FP64 and FP32-with-growth-rate: |dlam| / ||A||, |dgam| (smooth family: the growth rate is pinned there), status words equal;
  FP32 eigenvalues alone: |lam - lam64| in units of eps32 ||A|| against the stated N_zeta + 4.
   python tests/tools/direct_campaign.py [n_sys=6001]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

EPS32 = 1.1920929e-07
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6001
dev = torch.device("cuda", 0)
ctx = ibs_amd.Context(0)
bad = 0
print("%2s %5s %-7s | %-40s dlam/|A|   dgam     | %-40s dlam/|A|   dgam     | %-34s worst/eps32|A| (tolerance)" % (
    "M", "N", "family", "FP64 kernel", "FP32 rows, FP64 solver", "FP32 eigenvalues alone"))
for M in range(9, 33):
    for N in (64 * (M - 1) + 3, 64 * M + 1):
        for family in ("smooth", "rough"):
            h, g, c, f = bench.c5_family(dev, family, n, N, seed=31000 + N)
            nA = bench.norm_a(h, g, c, f)
            g32, c32, f32 = g.float(), c.float(), f.float()
            cells = []
            for args, kw in (((h, g, c, f), {}), ((h, g32, c32, f32), dict(dtype=np.float32))):
                res = {}
                for d in (0, 1):
                    ctx.set_option("force_p", 64); ctx.set_option("gcf_direct", d)
                    res[d] = ctx.solve_gcf(*args, want_info=True, **kw)
                    if d:
                        kern = ctx.last_launch()[0].replace("ibs::", "").split("(")[0]
                ctx.set_option("gcf_direct", None); ctx.set_option("force_p", None)
                assert "direct" in kern, kern
                dl = float(((res[0]["lam"].double() - res[1]["lam"].double()).abs() / nA).max())
                dg = float((res[0]["gam"].double() - res[1]["gam"].double()).abs().max())
                st_bad = int((((res[1]["info"] >> 16) & 3) != 0).sum())
                wide = bool(kw)
                ok = st_bad == 0 and dl <= (2e-13 * N if not wide else 3 * EPS32) and (family == "rough" or dg <= (1e-9 if not wide else 2e-6))
                bad += 0 if ok else 1
                cells.append("%-40s %.1e  %.1e%s" % (kern, dl, dg, "" if ok else " FAIL"))
                if not wide:
                    lam64 = res[1]["lam"]
            ctx.set_option("force_p", 64); ctx.set_option("gcf_direct", 1); ctx.set_option("f32_lam", 1)
            r = ctx.solve_gcf(h, g32, c32, f32, want_info=True, dtype=np.float32, want_gam=False)
            kern = ctx.last_launch()[0].replace("ibs::", "").split("(")[0]
            for o in ("force_p", "gcf_direct", "f32_lam"):
                ctx.set_option(o, None)
            assert "f32lam_direct" in kern, kern
            e = float(((r["lam"].double() - lam64).abs() / nA).max()) / EPS32
            ok = e <= N + 3 and int((((r["info"] >> 16) & ~4) != 0).sum()) == 0
            bad += 0 if ok else 1
            print("%2d %5d %-7s | %s | %s | %-34s %7.1f (%d)%s" % (M, N, family, cells[0], cells[1], kern, e, N + 3, "" if ok else " FAIL"), flush=True)
            del g, c, f, g32, c32, f32
    torch.cuda.empty_cache()
print("cells outside their tolerance: %d" % bad)
sys.exit(1 if bad else 0)
