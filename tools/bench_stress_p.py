"""stress legs of bench.py for lanes-per-system P = 64 (3-row staging / row-streamed) and P = 32 (IBS_FORCE_P), smooth and rough"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, ibs_amd, bench
dev = torch.device("cuda", 0)
for P, rows in (("64", 0), ("64", 1), ("32", 1)):
    os.environ["IBS_FORCE_P"] = P
    ctx = ibs_amd.Context(0)
    ctx.set_option("gcf_rows", rows)
    for fam, n in (("smooth", 262144), ("rough", 65536)):
        s = bench.stress(ctx, dev, n, fam, reps=3)
        print("P=%s gcf_rows=%d %-6s: %.3e solves/s  %.3f ms  mean sweeps %.2f  nonconverged %d  %s" % (
            P, rows, fam, s["solves_per_s"], s["ms_per_launch"], s["mean_sweeps"], s["nonconverged"], ctx.last_launch()), flush=True)
