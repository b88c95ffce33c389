#!/usr/bin/env python3
"""configs[4], FP32 eigenvalues only (all-FP32 iteration + FP64 certificate): LDS-staged k_solve_gcf<float, M> (gcf_direct = 0) against
rows from global memory (k_solve_gcf_f32lam_direct), 2^19 systems, both families, results compared.   python tools/bench_f32lam.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
print("library:", os.environ.get("IBS_LIB_PATH", "default"))
for nz in [int(v) for v in os.environ.get("IBS_NZ", "768,1024,1536,1792,2048").split(",")]:
    for fam in ("smooth", "rough"):
        h, g, c, f = bench.c5_family(dev, fam, n, nz + 1, seed=20240 + nz)
        g32, c32, f32 = g.float(), c.float(), f.float()
        del g, c, f
        res = {}
        for d in (0, -1):
            ctx.set_option("gcf_direct", d); ctx.set_option("f32_lam", 1)
            r = ctx.solve_gcf(h, g32, c32, f32, dtype=np.float32, want_gam=False, want_info=True)
            name = ctx.last_launch()[0].replace("ibs::", "")
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ctx.solve_gcf(h, g32, c32, f32, dtype=np.float32, want_gam=False); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            res[d] = (name, n / (best * 1e-3), r)
        ctx.reset_options()
        same = bool(torch.equal(res[0][2]["lam"], res[-1][2]["lam"]))
        refl = int((((res[-1][2]["info"] >> 16) & 4) != 0).sum())
        print("%5d %-6s | %-28s %.3e | %-34s %.3e | %.2f | lam equal %s, re-solved in FP64 %d, flagged %d" % (nz, fam, res[0][0], res[0][1], res[-1][0], res[-1][1],
              res[-1][1] / res[0][1], same, refl, int((((res[-1][2]["info"] >> 16) & 3) != 0).sum())), flush=True)
