"""Phase timestamps inside k_gamma_scan<double,8> on the bench workload (debug build: make -C ideal-ballooning-solver_amd/csrc probe2 PM=8;
IBS_LIB_PATH=.../libibs_hip_probe2.so python tools/scan_probe.py).  One launch of the fused scan + argmax; per wave:
staging | set-up (incl. trial vector) | shift iteration (sweeps, decision code) | backward sweep | twisted | growth rate | epilogue."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
from ibs_amd import _lib
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF)
for _ in range(3): plan.scan_argmax()
torch.cuda.synchronize()
buf = np.zeros((1024 * 4, 16), dtype=np.int64)
_lib.lib().ibs_probe_read(C.c_void_p(buf.ctypes.data), buf.size)
b = buf.reshape(1024, 4, 16)
ok = b[:, :, 0] > 0
w = b[ok]                                        # (waves, 16)
t0 = w[:, 0].min()
us = lambda x: x * 0.01
print("waves stamped: %d; kernel span (first start -> last end) %.2f us; start skew max %.2f us" % (len(w), us(w[:, 4].max() - t0), us(w[:, 0].max() - t0)))
twice = bool((w[:, 15] > 0).any())                 # (-DIBS_PROBE_TWICE build: the set-up ran twice, stamp 15 after the first)
names = [("staging + barrier", 0, 1)] + ([("set-up, FIRST execution", 1, 15), ("set-up, second: rows", 15, 8), ("set-up, second: sums", 8, 9)] if twice else
         [("set-up: row loop", 1, 8), ("set-up: bounds, trial sums", 8, 9)]) + [
         ("set-up + trial vector", 15 if twice else 1, 2), ("shift iteration", 2, 10), ("backward sweep", 10, 11),
         ("twisted + polish", 11, 12), ("growth: assemble, halo", 3, 13), ("growth: Simpson rows", 13, 14), ("growth: sums, division", 14, 4),
         ("growth rate", 3, 4), ("whole wave", 0, 4)]
for nm, a, c in names:
    d = us(w[:, c] - w[:, a])
    print("   %-24s median %6.2f  min %6.2f  max %6.2f us" % (nm, np.median(d), d.min(), d.max()))
nsw = w[:, 7]
print("   sweeps per solve: mean %.2f max %d; per sweep %.3f us; decision code per iteration %.3f us" % (
    nsw.mean(), nsw.max(), us(w[:, 5].sum()) / max(1, (nsw + 1).sum()), us(w[:, 6].sum()) / max(1, nsw.sum())))
end = us(w[:, 4] - t0)
print("   wave end times: median %.2f  p90 %.2f  max %.2f us" % (np.median(end), np.quantile(end, 0.9), end.max()))
slow = np.argsort(-end)[:5]
print("   slowest waves: end %s sweeps %s" % (np.round(end[slow], 2), nsw[slow]))
