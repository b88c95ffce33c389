"""Diagnostic (GPU box): the host-driven scipy L-BFGS-B refinement (BallooningScan.refine) next to the reference's
stored trajectories (G5 tight, G9).  Prints per-case the number of evaluations, the first point where the
trajectories part by more than 1e-9, and the final (x, gam)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ibs_amd  # noqa: E402
from tests.test_gpu_refine import CASES, _scan  # noqa: E402

ctx = ibs_amd.Context(0)
np.set_printoptions(precision=12, linewidth=200)
for case in CASES:
    scan = _scan(ctx, case)
    tr = case["trace"]
    seen = []
    orig = scan.obj_w_grad

    def rec(x, s, orig=orig, seen=seen):
        v, j = orig(x, s)
        seen.append((x[0], x[1], v, j[0], j[1]))
        return v, j

    scan.obj_w_grad = rec
    t_opt, a_opt, gam_opt, res = scan.refine(case["s"], tr[0, 0], tr[0, 1])
    seen = np.array(seen)
    n = min(len(seen), len(tr))
    dx = np.abs(seen[:n, :2] - tr[:n, :2]).max(axis=1)
    first = int(np.argmax(dx > 1e-9)) if (dx > 1e-9).any() else -1
    print("%s: evals %d (ref %d) nit %d '%s' first split at %d; x_opt %s ref %s; gam_opt %.12e ref %.12e (d %.2e)" % (
        case["tag"], len(seen), len(tr), res.nit, res.message, first, np.array([a_opt, t_opt]), case["x_opt"], gam_opt,
        case["gam_opt"], gam_opt - case["gam_opt"]))
    if first >= 0 and os.environ.get("VERBOSE"):
        lo = max(first - 2, 0)
        print("  ours:\n", seen[lo:first + 3], "\n  ref:\n", tr[lo:first + 3])
    # the same maximisation driven by the library's own optimizer (csrc/ibs_lbfgsb2.hpp through the C ABI)
    scan.obj_w_grad = orig
    out = ibs_amd.minimize2(lambda x: orig(x, case["s"]), (tr[0, 0], tr[0, 1]), ((0.0, np.pi), (0.0, 0.5 * np.pi)))
    geo = np.asarray(scan.fieldlines(case["s"], np.array([out.x[0]])))[0]
    dP = -0.5 * np.mean((geo[2] - geo[7]) * geo[0] ** 2)
    r = ctx.gamma_scan(scan.h, *[geo[k][None] for k in range(7)], np.array([dP]), np.array([out.x[1]]))
    g2 = float(np.asarray(r["gam"])[0, 0])
    n2 = min(len(out.trace), len(tr))
    dx2 = np.abs(out.trace[:n2, :2] - tr[:n2, :2]).max(axis=1)
    f2 = int(np.argmax(dx2 > 1e-6)) if (dx2 > 1e-6).any() else -1
    print("   lbfgsb2: evals %d nit %d '%s' restarts %d skipped %d split(1e-6) at %d; x_opt %s; gam_opt %.12e (d vs ref %.2e, vs scipy-host %.2e)" % (
        len(out.trace), out.nit, out.message, out.restarts, out.skipped, f2, out.x, g2, g2 - case["gam_opt"], g2 - gam_opt))
