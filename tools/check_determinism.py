import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, ibs_amd, bench
dev = torch.device("cuda", 0); ctx = ibs_amd.Context(0)
h, geo7, dP, th0, *_ = bench.build_workload(0, dev)
r = [ctx.gamma_scan(h, *geo7, dP, th0, want_info=True) for _ in range(4)]
torch.cuda.synchronize()
for k in range(1, 4):
    print("run", k, "bitwise equal gam:", bool((r[k]["gam"] == r[0]["gam"]).all().item()), "lam:", bool((r[k]["lam"] == r[0]["lam"]).all().item()), "info:", bool((r[k]["info"] == r[0]["info"]).all().item()), float((r[k]["gam"] - r[0]["gam"]).abs().max().item()))
