"""solves/s of the geometry-fed scan + per-surface argmax against the number of solves per launch (N = 513, 8 theta0 per
line, 8 lines per surface: the bench step's batch replicated), and with two bench-sized batches in flight on two streams.
Shows where the latency-bound regime of the headline step (one wave per SIMD) ends."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ibs_amd
import bench

dev = torch.device("cuda:0")
ctx = ibs_amd.Context(0)
h, geo7, dP_d, th0_d, base, dP, theta0 = bench.build_workload(0, dev)


def timed(fn, n):
    for _ in range(max(20, n // 10)):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print("solves/launch   us/launch   solves/s    sweeps  kernel-choice")
for rep in (1, 2, 4, 8, 16, 64):
    g = [x.repeat(rep, 1) for x in geo7]
    plan = ibs_amd.ScanPlan(ctx, h, g, dP_d.repeat(rep), th0_d, bench.N_SURF * rep)
    dt = timed(plan.scan_argmax, 2000 // rep + 50)
    n = plan.n_lines * plan.n_t0
    info = plan.info.cpu().numpy()
    print("%10d   %9.1f   %.3e   %5.2f" % (n, dt * 1e6, n / dt, (info & 0xffff).mean()), flush=True)

# two bench-sized batches in flight: two contexts (the stream and the arrival counters are per context), two streams
ctx2 = ibs_amd.Context(0)
plans = [ibs_amd.ScanPlan(c, h, geo7, dP_d, th0_d, bench.N_SURF) for c in (ctx, ctx2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
k = [0]


def two():
    i = k[0] & 1
    k[0] += 1
    with torch.cuda.stream(streams[i]):
        plans[i].scan_argmax()


dt = timed(two, 4000)
print("two streams, 1024 solves per launch alternating: %.1f us per launch, %.3e solves/s" % (dt * 1e6, 1024 / dt), flush=True)

# dispatch alternatives for the mid-size batches (lanes per system x theta0 chain length)
if os.environ.get("IBS_SWEEP_OPTIONS", "1") == "1":
    print("solves/launch  force_p chain   us/launch   solves/s   sweeps")
    for rep in (8, 16, 32, 64):
        g = [x.repeat(rep, 1) for x in geo7]
        plan = ibs_amd.ScanPlan(ctx, h, g, dP_d.repeat(rep), th0_d, bench.N_SURF * rep)
        n = plan.n_lines * plan.n_t0
        for fp in (64, 32):
            for ch in (1, 2, 4):
                ctx.set_option("force_p", fp); ctx.set_option("scan_chain", ch)
                try:
                    dt = timed(plan.scan_argmax, 400 // rep + 20)
                    info = plan.info.cpu().numpy()
                    print("%10d   %5d %5d   %9.1f   %.3e   %5.2f" % (n, fp, ch, dt * 1e6, n / dt, (info & 0xffff).mean()), flush=True)
                except ibs_amd.IbsError as e:
                    print(n, fp, ch, "error", e)
        ctx.reset_options()
