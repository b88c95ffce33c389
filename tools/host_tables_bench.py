"""Wall time of SurfaceTables.from_wouts (ibs_surface_tables_f64, the host's radial step) for the 73 equilibria of configs[3]
against the per-equilibrium numpy form, by thread count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, ibs_amd, bench
wout0 = dict(np.load(os.path.join(ROOT, "tests/golden/G8_wout_ncsx_op.npz")))
wouts, _, _ = bench.emulated_equilibria(wout0)
wouts = [{k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in w.items()} for w in wouts]
sv = np.linspace(0.5, 0.95, 5)
ibs_amd.SurfaceTables.from_wout(wouts[0], sv)
t = time.perf_counter(); a = [ibs_amd.SurfaceTables.from_wout(w, sv) for w in wouts]; print("per-equilibrium numpy: %.2f ms" % ((time.perf_counter() - t) * 1e3))
print("cores:", len(os.sched_getaffinity(0)))
for nt in (1, 2, 4, 8, 16, 32, 0):
    ts = []
    for _ in range(5):
        t = time.perf_counter(); ibs_amd.SurfaceTables.from_wouts(wouts, sv, n_threads=nt); ts.append((time.perf_counter() - t) * 1e3)
    print("from_wouts, %2d threads: min %.2f ms, median %.2f ms" % (nt, min(ts), float(np.median(ts))), flush=True)
