"""The headline step with one wave per system (scan_pair = -1) against two waves per system on two shifts (scan_pair = 1):
iterations per solve, time per launch (HIP events around 400 back-to-back launches), results against each other.
Needs a library built with tools/experiments/pair_two_shifts.patch applied (the experiment was measured and dropped in round 4:
36.8 against 25.0 us per launch at 1,024 solves, 29.8 against 25.0 at <= 512); argument: surfaces of the step's batch to take."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)
n_surf = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_SURF          # surfaces of the step's batch taken (16 = all: 1,024 solves)
nl = n_surf * bench.N_ALPHA
plan = ibs_amd.ScanPlan(ctx, h, [g[:nl].contiguous() for g in geo7], dP_d[:nl].contiguous(), th0_d, n_surf)
print("%d solves per launch" % (nl * bench.N_THETA0))
res = {}
for mode in (-1, 1, -1, 1):
    ctx.set_option("scan_pair", mode)
    for _ in range(300):
        plan.scan_argmax()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(400):
        plan.scan_argmax()
    b.record(); torch.cuda.synchronize()
    it = (plan.info.cpu().numpy() & 0xffff).ravel()
    st = (plan.info.cpu().numpy() >> 16).ravel()
    res[mode] = (plan.gam.cpu().numpy().copy(), plan.lam.cpu().numpy().copy(), plan.pack.cpu().numpy().copy())
    print("scan_pair %2d: %-38s %7.2f us per launch | iterations mean %.2f max %d hist %s | flagged %d" % (
        mode, ctx.last_launch()[0], a.elapsed_time(b) / 400 * 1e3, it.mean(), it.max(), np.bincount(it)[3:].tolist(), int((st != 0).sum())), flush=True)
g0, l0, p0 = res[-1]; g1, l1, p1 = res[1]
print("max |gam pair - gam single| %.2e ; max |lam pair - lam single| %.2e ; argmax rows equal %s ; max |max gam diff| %.2e" % (
    np.abs(g0 - g1).max(), np.abs(l0 - l1).max(), np.array_equal(p0[:, 1], p1[:, 1]), np.abs(p0[:, 0] - p1[:, 0]).max()))
