"""configs[3] step (AdjointStep.run, 73 equilibria) against the plan of the coarse runs: number of runs and growth factor."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
wouts, steps, _ = bench.emulated_equilibria(wout)
f_other = 0.8 + 0.01 * np.arange(len(wouts))
th = ibs_amd.theta_grid(969); svals = np.linspace(0.5, 0.95, 5)
for nch, gr in ((4, 1.0), (4, 1.25), (5, 1.25), (6, 1.25), (6, 1.4), (8, 1.2), (3, 1.5), (2, 2.0), (1, 1.0)):
    st = ibs_amd.AdjointStep(ctx, th, svals, dev, nalpha=24, ntheta0=15, gamma_thresh=-2.0e-4, prefac=50.0, n_chunks=nch, chunk_growth=gr)
    for _ in range(2):
        r = st.run(wouts, f_other, steps)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = st.run(wouts, f_other, steps); ts.append((time.perf_counter() - t0) * 1e3)
    ph = {}
    st.run(wouts, f_other, steps, phases=ph)
    print("runs %d growth %.2f: %.2f ms (min of 5; %s)  host tables %.2f ms, geometry %.2f, scan %.2f, refine %.2f   fobj %.12f" % (
        nch, gr, min(ts), " ".join("%.2f" % t for t in ts), ph.get("host_tables_ms", 0), ph.get("geometry_ms", 0), ph.get("scan_argmax_ms", 0), ph.get("refine_ms", 0), r["fobj"]), flush=True)
