#!/usr/bin/env python3
"""FP32 eigenvalue-only requests on the short grids, every grid length odd N = 131 ... 641 in steps of `step` (all rows-per-lane
instantiations M = 3 ... 10 of k_solve_gcf_f32lam_direct, grids that fill the last lane's chunk and grids that leave it almost
empty): library's choice and the forced all-FP32 direct form against the FP64 solve of the same systems, both families, ragged
batch.  Reports per M the worst |lam - lam64| in units of eps32 ||A|| (stated tolerance: N_zeta + 4), the share of systems the
certificate sent to FP64, and any status other than that informational bit.
   python tests/tools/f32lam_short_campaign.py [step=6] [n_sys=20003]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

EPS32 = 1.1920929e-07
step = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20003
dev = torch.device("cuda", 0)
ctx = ibs_amd.Context(0)
per_m = {}
bad = 0
for N in sorted(set(list(range(131, 642, step + (step & 1))) + [641])):      # (odd N only: the Simpson rule of the growth rate)
    M = (N - 2 + 63) // 64
    for family in ("smooth", "rough"):
        h, g, c, f = bench.c5_family(dev, family, n, N, seed=9000 + N)
        lam64 = ctx.solve_gcf(h, g, c, f)["lam"]
        nA = bench.norm_a(h, g, c, f)
        g32, c32, f32 = g.float(), c.float(), f.float()
        for forced in (0, 1):
            if forced:
                ctx.set_option("f32_lam", 1)
            r = ctx.solve_gcf(h, g32, c32, f32, want_info=True, dtype=np.float32, want_gam=False)
            kern = ctx.last_launch()[0].replace("ibs::", "")
            ctx.set_option("f32_lam", None)
            st = r["info"] >> 16
            other = int(((st & ~4) != 0).sum())
            e = float(((r["lam"].double() - lam64).abs() / nA).max()) / EPS32
            ok = other == 0 and e <= N + 3
            bad += 0 if ok else 1
            rec = per_m.setdefault((M, family, forced), dict(worst=0.0, worst_over_tol=0.0, resolved=0, systems=0, grids=0, kernels=set()))
            rec["worst"] = max(rec["worst"], e); rec["worst_over_tol"] = max(rec["worst_over_tol"], e / (N + 3))
            rec["resolved"] += int(((st & 4) != 0).sum()); rec["systems"] += n; rec["grids"] += 1
            rec["kernels"].add(kern.split("(")[0])
            if not ok:
                print("FAIL N %d %s forced %d %s: worst %.1f eps32 ||A|| (tolerance %d), other status on %d" % (N, family, forced, kern, e, N + 3, other), flush=True)
print("%2s %-7s %-7s %6s %9s | worst |dlam| / (eps32 ||A||)  worst / tolerance  re-solved in FP64 | kernels" % ("M", "family", "setting", "grids", "systems"))
for (M, family, forced), rec in sorted(per_m.items()):
    print("%2d %-7s %-7s %6d %9d | %26.1f  %17.3f  %17.2e | %s" % (M, family, "forced" if forced else "auto", rec["grids"], rec["systems"], rec["worst"],
                                                                     rec["worst_over_tol"], rec["resolved"] / rec["systems"], ", ".join(sorted(rec["kernels"]))))
print("grids x families x settings outside the stated tolerance or with a status other than the informational bit: %d" % bad)
sys.exit(1 if bad else 0)
