#!/usr/bin/env python3
"""configs[4] on the long grids, one wave per system: rows staged in LDS (k_solve_gcf / k_solve_gcf_rows / k_solve_gcf_wide) against
rows read straight from global memory (k_solve_gcf_direct, option gcf_direct), FP64 and FP32-with-growth-rate, both families.
   python tools/bench_direct.py [n_sys]     (default 2^19)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
dev = torch.device("cuda", 0)
ctx = ibs_amd.Context(0)
print("%6s %-7s %-7s | %-44s %9s | %-44s %9s | ratio  max|dlam| max|dgam|" % ("N_zeta", "family", "mode", "staged kernel", "solves/s", "direct kernel", "solves/s"))
for nz in [int(v) for v in os.environ.get("IBS_NZ", "768,1024,1536,2048").split(",")]:
    N = nz + 1
    for family in ("smooth", "rough"):
        h, g, c, f = bench.c5_family(dev, family, n, N, seed=20240 + nz)
        g32, c32, f32 = g.float(), c.float(), f.float()
        for mode in ("f64", "f32_gam"):
            args = (h, g, c, f) if mode == "f64" else (h, g32, c32, f32)
            kw = {} if mode == "f64" else dict(dtype=np.float32)
            res = {}
            for direct in (0, 1):
                ctx.set_option("gcf_direct", direct)
                r = ctx.solve_gcf(*args, want_info=True, **kw)
                name = ctx.last_launch()[0].replace("ibs::", "")
                torch.cuda.synchronize()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
                for a, b in evs:
                    a.record(); ctx.solve_gcf(*args, **kw); b.record()
                torch.cuda.synchronize()
                ms = min(a.elapsed_time(b) for a, b in evs)
                res[direct] = (name, n / (ms * 1e-3), r)
            ctx.set_option("gcf_direct", None)
            dl = float((res[0][2]["lam"].double() - res[1][2]["lam"].double()).abs().max())
            dg = float((res[0][2]["gam"].double() - res[1][2]["gam"].double()).abs().max())
            nb = int(((res[1][2]["info"] >> 16) & 3 != 0).sum())
            print("%6d %-7s %-7s | %-44s %9.3e | %-44s %9.3e | %5.2f  %.1e %.1e  flagged %d" % (
                nz, family, mode, res[0][0], res[0][1], res[1][0], res[1][1], res[1][1] / res[0][1], dl, dg, nb), flush=True)
        del g, c, f, g32, c32, f32
        torch.cuda.empty_cache()
