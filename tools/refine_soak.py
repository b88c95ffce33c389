"""Soak of ibs_refine_f64: thousands of random starts on random surfaces of the NCSX_op equilibrium, N = 513 and 969.  Every optimum must
be finite, inside the box (to rounding), and not below the value at its start (evaluated by the same fused objective with maxiter = 0 ...
which performs one iteration, so the comparison is against f after that iteration's accepted step: f_opt <= f_1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
wout = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(11)
for N in (513, 969):
    th = ibs_amd.theta_grid(N)
    svals = np.linspace(0.05, 0.98, 32)
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    ps = rng.integers(0, len(svals), n).astype(np.int32)
    st = np.stack([rng.uniform(0.0, np.pi, n), rng.uniform(0.0, 0.5 * np.pi, n)], axis=1)
    st[: n // 20, 0] = 0.0; st[n // 20: n // 10, 1] = 0.5 * np.pi            # some starts on the box
    t = time.perf_counter()
    x1, f1, n1, r1 = ctx.refine(tabs, ps, st, th, maxiter=0, device=dev)
    xo, fo, ne, ro = ctx.refine(tabs, ps, st, th, device=dev)
    dt = time.perf_counter() - t
    eb = 1e-15
    inbox = (xo[:, 0] >= -eb) & (xo[:, 0] <= np.pi + eb) & (xo[:, 1] >= -eb) & (xo[:, 1] <= 0.5 * np.pi + eb)
    print("N=%d: %d starts, %.1f ms for both calls; finite %s; in box %s; f_opt <= f_1 everywhere %s (worst excess %.2e); evaluations %d..%d mean %.1f, rounds %d; gam_opt range [%.3e, %.3e]" % (
        N, n, 1e3 * dt, np.isfinite(fo).all() and np.isfinite(xo).all(), inbox.all(), (fo <= f1 + 1e-12).all(), (fo - f1).max(), ne.min(), ne.max(), ne.mean(), ro, (-fo).min(), (-fo).max()), flush=True)
