import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import ibs_amd, bench
from oracle import c_oracle as co
nz, n = 1024, 1000000
N = nz + 1
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
h, g, c, f = bench.c5_family(dev, "rough", n, N, seed=20240 + nz)
r = ctx.solve_gcf(h, g, c, f, want_info=True)
lam = r["lam"].cpu().numpy()
nA = bench.norm_a(h, g, c, f).cpu().numpy()
lam_c = np.empty(n)
for a in range(0, n, 32768):
    lam_c[a:a + 32768] = co.lam_batch(h, g[a:a + 32768].cpu().numpy(), c[a:a + 32768].cpu().numpy(), f[a:a + 32768].cpu().numpy())
err = np.abs(lam - lam_c) / nA
k = int(np.argmax(err))
print("worst system", k, "err", err[k], "lam", lam[k], "lam_c", lam_c[k], "normA", nA[k], "status", int(r["info"][k]) >> 16)
with open("gpurun_out/bad_1024.bin", "wb") as fh:
    fh.write(g[k].cpu().numpy().tobytes()); fh.write(c[k].cpu().numpy().tobytes()); fh.write(f[k].cpu().numpy().tobytes())
open("gpurun_out/bad_1024.txt", "w").write("%.17e %.17e\n" % (lam[k], lam_c[k]))
