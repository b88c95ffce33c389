"""like batch_sweep.py for the N = 1025 grid (16 rows per lane): golden NCSX lines replicated with a per-line perturbation,
16 theta0 per line, 32 lines per surface; solves per launch against chain length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ibs_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
ctx = ibs_amd.Context(0)
g3 = np.load(os.path.join(ROOT, "tests", "golden", "G3_ncsx_lines.npz"))
N = int(os.environ.get("IBS_SWEEP_N", "1025"))
geo = g3["geo_%d" % N]
NT0 = 16
th0 = torch.from_numpy(np.linspace(0, np.pi / 2, NT0)).to(dev)
h = 8 * np.pi / (N - 1)


def timed(fn, n):
    for _ in range(max(5, n // 10)):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print("N=%d  solves/launch  chain   us/launch   solves/s   sweeps" % N)
for n_surf in [int(x) for x in os.environ.get('IBS_SWEEP_SURFS', '1,2,4,8,16,32,64').split(',')]:
    nl = n_surf * 32
    rng = np.random.default_rng(7)
    base = geo[np.arange(nl) % len(geo)].copy()
    eps = rng.uniform(-0.03, 0.03, size=(nl, 2))
    base[:, 4:7, :] *= (1 + eps[:, 0])[:, None, None]
    base[:, 2:4, :] *= (1 + eps[:, 1])[:, None, None]
    base[:, 7, :] *= (1 + eps[:, 1])[:, None]
    dP = -0.5 * np.mean((base[:, 2] - base[:, 7]) * base[:, 0] ** 2, axis=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    plan = ibs_amd.ScanPlan(ctx, h, [t(base[:, k, :]) for k in range(7)], t(dP), th0, n_surf)
    n = nl * NT0
    for ch in (0, 1, 2, 4, 8, 0):
        ctx.set_option("scan_chain", ch)
        dt = timed(plan.scan_argmax, max(40, 40000 // n))
        info = plan.info.cpu().numpy()
        print("%10d   %5s   %9.1f   %.3e   %5.2f" % (n, "auto" if ch == 0 else ch, dt * 1e6, n / dt, (info & 0xffff).mean()), flush=True)
    ctx.reset_options()
