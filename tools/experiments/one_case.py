import os, sys, torch
sys.path.insert(0, os.getcwd())
import ibs_amd, bench
nz = int(sys.argv[1]); fam = sys.argv[2]
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
h, g, c, f = bench.c5_family(dev, fam, 1 << 20, nz + 1, seed=20240 + nz)
for _ in range(4):
    ctx.solve_gcf(h, g, c, f)
torch.cuda.synchronize()
