#!/usr/bin/env python3
"""Eigenvalue accuracy of the FP64 raw-system kernels on EVERY system of a config-5 batch (round 6).

The product-form (prefix-scan) Sturm counts that move the solver's bracket are exact for a matrix perturbed by up to ~N^2 eps ||A||
on iid-random coefficients; the kernels therefore check the closing bracket against the twisted factorisation at the last shift (its
Rayleigh polish must lie in the bracket, and it must count no OTHER eigenvalue above the shift) and close a suspect system again by
division-form multisection (csrc/ibs_wave.hpp: solve<true>, reclose_division; csrc/ibs_kernels.hip: k_fix_gcf).  This tool solves the
whole batch three ways -- checks off, suspects only marked, suspects re-closed -- and compares ALL results with the C oracle's
division-form bisection (oracle/ibs_oracle.c), in units of ||A||; it also times the three modes.

    python tests/tools/reclose_campaign.py [n_sys] [families] [nz,nz,...] [seed0]    (default 2^20, rough+smooth, 256 512 1024 1536 2048, 20240:
    the bench's batches; another seed0 = systems the detector rule was not chosen on; RECLOSE_OPTS=name=value,... sets library options first,
    e.g. gcf_direct=0,force_p=64 for the staged one-wave-per-system forms, which re-close in place)
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
from oracle import c_oracle as co

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
families = sys.argv[2].split(",") if len(sys.argv) > 2 else ["rough", "smooth"]
nzs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 512, 1024, 1536, 2048]
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 20240
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
for kv in filter(None, os.environ.get("RECLOSE_OPTS", "").split(",")):        # e.g. RECLOSE_OPTS=gcf_direct=0,force_p=64: the staged forms
    k, v = kv.split("="); ctx.set_option(k, float(v)); print("option %s = %s" % (k, v))
EPS = 2.220446049250313e-16


def timed(fn, reps=2):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return r, best


def oracle_lam(h, g, c, f, chunk=32768):
    out = np.empty(g.shape[0])
    for a in range(0, g.shape[0], chunk):
        out[a:a + chunk] = co.lam_batch(h, g[a:a + chunk].cpu().numpy(), c[a:a + chunk].cpu().numpy(), f[a:a + chunk].cpu().numpy())
    return out


for nz in nzs:
    N = nz + 1
    for family in families:
        h, g, c, f = bench.c5_family(dev, family, n, N, seed=seed0 + nz)
        nA = bench.norm_a(h, g, c, f).cpu().numpy()
        t0 = time.perf_counter(); lam_c = oracle_lam(h, g, c, f); t_or = time.perf_counter() - t0
        print("N_zeta %d %s (seed %d): %d systems, oracle %.0f s, bound 4 N eps = %.2e" % (nz, family, seed0 + nz, n, t_or, 4 * N * EPS), flush=True)
        for want_gam in (True, False):
            res = {}
            for mode in (0, 2, 1):
                ctx.set_option("reclose", mode)
                r, dt = timed(lambda: ctx.solve_gcf(h, g, c, f, want_info=True, want_gam=want_gam))
                res[mode] = (r["lam"].cpu().numpy(), r["info"].cpu().numpy(), dt)
                kern = ctx.last_launch()[0]
            ctx.set_option("reclose", None)
            e0 = np.abs(res[0][0] - lam_c) / nA
            st2 = (res[2][1].view(np.uint32) >> 16).astype(np.int64)
            marked = (st2 & 8) != 0
            bucket = (st2 >> 5) & 63           # floor(log2(distance of the polish from the bracket / tol)) + 8
            if want_gam and os.environ.get("RECLOSE_DUMP"):
                # raw material for choosing / re-checking the rule offline: every system whose polish fell outside its bracket
                # (mark-only mode), the distance bucket and the system's error with the checks off
                np.savez_compressed(os.path.join(os.environ["RECLOSE_DUMP"], "reclose_marks_%s_%d.npz" % (family, nz)),
                                    idx=np.nonzero(marked)[0].astype(np.int32), bucket=bucket[marked].astype(np.int8), e0=e0[marked],
                                    e0_unmarked_max=e0[~marked].max(), n=n, N=N)
            e1 = np.abs(res[1][0] - lam_c) / nA
            st1 = res[1][1] >> 16
            print("  %-44s gam=%d  solves/s off %.3e | mark %.3e | re-close %.3e (%+.1f %% vs off)" % (
                kern, want_gam, n / res[0][2], n / res[2][2], n / res[1][2], 100 * (res[0][2] / res[1][2] - 1)))
            print("    checks off      : max %.2e  99.99%% %.2e  median %.2e  | systems beyond 4 N eps: %d" % (
                e0.max(), np.quantile(e0, 0.9999), np.median(e0), int((e0 > 4 * N * EPS).sum())))
            hb = np.bincount(bucket[marked], minlength=64)
            extra = marked & (((st2 >> 11) & 1) != 0)        # the twisted factorisation counts another eigenvalue above the last shift
            print("    marked: %d systems; polish outside the bracket by distance bucket (2^(b-8) tol; 0 = inside): %s;  worst error per bucket: %s;  "
                  "another eigenvalue above the last shift (twisted count): %d systems, worst error %.1e" % (
                      int(marked.sum()), {int(b): int(hb[b]) for b in np.nonzero(hb)[0]},
                      {int(b): "%.1e" % e0[marked & (bucket == b)].max() for b in np.nonzero(hb)[0]},
                      int(extra.sum()), e0[extra].max() if extra.any() else 0.0))
            rc = (st1 & 8) != 0
            print("    re-closed       : %d systems (%.1e of the batch), status bits 0/1 set on %d;  max error ALL %.2e (= %.2f N eps; re-closed "
                  "ones %.2e), beyond 4 N eps: %d;  sweeps mean %.2f (re-closed: +%.1f passes)" % (
                      int(rc.sum()), rc.mean(), int(((st1 & 3) != 0).sum()), e1.max(), e1.max() / (N * EPS), e1[rc].max() if rc.any() else 0.0,
                      int((e1 > 4 * N * EPS).sum()), float((res[1][1] & 0xffff).mean()),
                      float((res[1][1][rc] & 0xffff).mean() - (res[2][1][rc] & 0xffff).mean()) if rc.any() else 0.0), flush=True)
            if want_gam and family == "smooth":
                pass
        del g, c, f
        torch.cuda.empty_cache()
