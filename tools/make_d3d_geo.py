"""Write the bench's D3D-shape batch (rank 0) as raw doubles for tools/phase_probe.hip -> tools/d3d_geo.bin"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 513; nl = int(sys.argv[2]) if len(sys.argv) > 2 else 128; nt = int(sys.argv[3]) if len(sys.argv) > 3 else 8
g3 = np.load(os.path.join(ROOT, "tests", "golden", "G3_ncsx_lines.npz")); geo = g3["geo_%d" % N]
rng = np.random.default_rng(1000); base = geo[np.arange(nl) % len(geo)].copy()
eps = rng.uniform(-0.03, 0.03, size=(nl, 2))
base[:, 4:7, :] *= (1 + eps[:, 0])[:, None, None]; base[:, 2:4, :] *= (1 + eps[:, 1])[:, None, None]; base[:, 7, :] *= (1 + eps[:, 1])[:, None]
dP = -0.5 * np.mean((base[:, 2] - base[:, 7]) * base[:, 0] ** 2, axis=1); th0 = np.linspace(0, 0.5 * np.pi, nt)
np.concatenate([np.ascontiguousarray(base[:, k, :]).ravel() for k in range(7)] + [dP, th0]).tofile(os.path.join(ROOT, "tools", "d3d_geo.bin" if N == 513 and nl == 128 else "geo_%d_%d_%d.bin" % (N, nl, nt)))
# raw (g, c, f) dumps of single systems of that batch for tools/trace_solve.hip:  python tools/make_d3d_geo.py 513 128 8 --sys 17 402
if "--sys" in sys.argv:
    for k in (int(v) for v in sys.argv[sys.argv.index("--sys") + 1:]):
        l, t = divmod(k, nt); b = base[l]; gp = np.abs(b[1]); gd = b[4] + 2 * th0[t] * b[5] + th0[t] ** 2 * b[6]
        np.concatenate([gp * gd / b[0], -dP[l] * (b[2] + th0[t] * b[3]) / (gp * b[0]), gd / b[0] ** 2 / (gp * b[0])]).tofile(os.path.join(ROOT, "tools", "sys_%d_%d.bin" % (N, k)))
