#!/usr/bin/env python3
"""A/B of library builds on ONE box, interleaved (round 6: what the re-close code costs the raw FP64 kernels' hot path).
    python tools/ab_reclose.py libA.so libB.so ...     -> solves/s per (N_zeta, family) and build, best of the rounds
Each build runs in its own child process (IBS_LIB_PATH), the builds alternate so that clock drift hits them alike."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(512, "smooth"), (512, "rough"), (1024, "rough"), (1536, "rough"), (2048, "smooth"), (2048, "rough")]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch, ibs_amd, bench
    dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
    out = {}
    for nz, fam in CASES:
        h, g, c, f = bench.c5_family(dev, fam, 1 << 20, nz + 1, seed=20240 + nz)
        ctx.solve_gcf(h, g, c, f)
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ctx.solve_gcf(h, g, c, f); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        out["%d_%s" % (nz, fam)] = (1 << 20) / (best * 1e-3)
        del g, c, f
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))
    sys.exit(0)

libs = sys.argv[1:]
best = {l: {} for l in libs}
for rnd in range(3):
    for l in libs:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, IBS_LIB_PATH=os.path.abspath(l)),
                           capture_output=True, text=True, timeout=600)
        line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
        if not line:
            print("child failed for", l, p.stderr[-500:]); continue
        for k, v in json.loads(line[0][7:]).items():
            best[l][k] = max(best[l].get(k, 0.0), v)
ref = libs[0]
print("%-14s" % "case" + "".join("%26s" % os.path.basename(l) for l in libs))
for nz, fam in CASES:
    k = "%d_%s" % (nz, fam)
    print("%-14s" % k + "".join("%16.3e (%+5.1f %%)" % (best[l].get(k, 0), 100 * (best[l].get(k, 0) / best[ref][k] - 1)) for l in libs))
