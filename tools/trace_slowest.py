"""dump the systems of the bench batch that need the most sweeps (and a few typical ones) as raw (g, c, f) files for
tools/trace_solve.hip:   python tools/trace_slowest.py && for f in tools/sys_513_*.bin; do /tmp/ts 513 $f; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, bench.N_SURF)
plan.scan_argmax(); torch.cuda.synchronize()
it = (plan.info.cpu().numpy() & 0xffff)
order = np.argsort(-it.ravel(), kind="stable")
picks = list(order[:4]) + list(order[len(order) // 2: len(order) // 2 + 2])
for k, flat in enumerate(picks):
    line, t = divmod(int(flat), it.shape[1])
    b = base[line]                       # bmag gradpar cvdrift cvdrift0 gds2 gds21 gds22 gbdrift
    th0 = theta0[t]
    gds2 = b[4] + 2 * th0 * b[5] + th0 ** 2 * b[6]
    cv = b[2] + th0 * b[3]
    gp = np.abs(b[1])
    g = gp * gds2 / b[0]; c = -dP[line] * cv / (gp * b[0]); f = gds2 / b[0] ** 2 / (gp * b[0])
    np.concatenate([g, c, f]).astype(np.float64).tofile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sys_513_%d.bin" % k))
    print("sys_513_%d.bin: line %d theta0 %d  sweeps %d" % (k, line, t, it[line, t]))
