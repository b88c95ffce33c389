#!/usr/bin/env python3
"""configs[4] on the SHORT grids (N_zeta <= 640): the sub-wave form the dispatch picks (32 lanes per system, k_solve_gcf_g) against the
two one-wave-per-system forms -- rows staged in LDS (k_solve_gcf) and rows read straight from global memory (k_solve_gcf_direct,
needs a build with -DIBS_DIRECT_MIN_M <= rows per lane).  FP64, FP32 with growth rate, FP32 eigenvalue only; both families.
   IBS_NZ=384,512,576,640 IBS_MODES=f64,f32_gam,f32_lam python tools/bench_forms.py [n_sys]     (default 2^20)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ibs_amd  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda", 0)
ctx = ibs_amd.Context(0)
FORMS = (("picked", {}), ("sub-wave, FP64 solver", dict(f32_lam=2, gcf_direct=0)), ("staged", dict(force_p=64, gcf_direct=0)),
         ("direct", dict(force_p=64, gcf_direct=1)))
MODES = os.environ.get("IBS_MODES", "f64,f32_gam,f32_lam").split(",")
print("%6s %-7s %-7s | %s" % ("N_zeta", "family", "mode", " | ".join("%-46s %9s" % (k, "solves/s") for k, _ in FORMS)))
for nz in [int(v) for v in os.environ.get("IBS_NZ", "384,512,576,640").split(",")]:
    N = nz + 1
    for family in ("smooth", "rough"):
        h, g, c, f = bench.c5_family(dev, family, n, N, seed=20240 + nz)
        g32, c32, f32 = g.float(), c.float(), f.float()
        for mode in MODES:
            args = (h, g, c, f) if mode == "f64" else (h, g32, c32, f32)
            kw = {} if mode == "f64" else dict(dtype=np.float32)
            if mode == "f32_lam":
                kw["want_gam"] = False
            cells, ref = [], None
            for name, opts in FORMS:
                for k, v in opts.items():
                    ctx.set_option(k, v)
                r = ctx.solve_gcf(*args, want_info=True, **kw)
                kern = ctx.last_launch()[0].replace("ibs::", "")
                torch.cuda.synchronize()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
                for a, b in evs:
                    a.record(); ctx.solve_gcf(*args, **kw); b.record()
                torch.cuda.synchronize()
                ms = min(a.elapsed_time(b) for a, b in evs)
                for k in opts:
                    ctx.set_option(k, None)
                lam = r["lam"].double()
                if ref is None:
                    ref = lam
                cells.append("%-46s %9.3e" % (kern + " d%.0e" % float((lam - ref).abs().max()), n / (ms * 1e-3)))
            print("%6d %-7s %-7s | %s" % (nz, family, mode, " | ".join(cells)), flush=True)
        del g, c, f, g32, c32, f32
        torch.cuda.empty_cache()
