import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import ibs_amd
d = np.load("tests/golden/G10_rough_pair_1025.npz")
g, c, f = d["g"], d["c"], d["f"]; N = len(g); h = 8 * np.pi / (N - 1)
l2, l1 = float(d["lam_returned_round5"]), float(d["lam_max"])
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
def run(n, direct, reclose, pos=0, filler=None):
    G = np.tile(g, (n, 1)); C = np.tile(c, (n, 1)); F = np.tile(f, (n, 1))
    if filler is not None:
        rng = np.random.default_rng(5)
        G = np.exp(rng.uniform(np.log(0.01), np.log(50), (n, N))); C = rng.uniform(-2.5, 3.5, (n, N)); F = np.exp(rng.uniform(np.log(0.2), np.log(3e3), (n, N)))
        G[pos], C[pos], F[pos] = g, c, f
    ctx.set_option("gcf_direct", direct); ctx.set_option("reclose", reclose)
    r = ctx.solve_gcf(h, torch.from_numpy(G).to(dev), torch.from_numpy(C).to(dev), torch.from_numpy(F).to(dev), want_info=True)
    lam = r["lam"].cpu().numpy(); info = r["info"].cpu().numpy()
    k = ctx.last_launch()[0]
    sel = lam if filler is None else lam[pos:pos + 1]
    print("n=%d direct=%d reclose=%d %-45s lam-l1: %s  lam-l2: %s  info %s" % (n, direct, reclose, k, np.unique(np.round((sel - l1) / (l1 - l2), 6)), np.unique(np.round((sel - l2) / (l1 - l2), 6)), np.unique(info[pos:pos+1] >> 16) if filler is not None else np.unique(info >> 16)))
for n in (1, 4, 64, 4096):
    for direct in (0, 1):
        run(n, direct, 0)
run(4096, 1, 1)
for pos in (0, 1, 2, 3, 245944 % 4096):
    run(4096, 1, 0, pos, filler=True)
