"""crossover of lanes-per-system P (IBS_FORCE_P) against batch size, raw (g,c,f) smooth family"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
dev = torch.device("cuda", 0)
def fam(n_sys, N):
    th = torch.linspace(-4 * np.pi, 4 * np.pi, N, dtype=torch.float64, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)
    sh, al, t0 = u(0.1, 2.0, (n_sys, 1)), u(0.0, 1.2, (n_sys, 1)), u(0.0, np.pi / 2, (n_sys, 1))
    lam = sh * (th[None] - t0) - al * (torch.sin(th)[None] - torch.sin(t0))
    g = 1 + lam ** 2; c = al * (torch.cos(th)[None] + torch.sin(th)[None] * lam)
    return g, c, g.clone(), 8 * np.pi / (N - 1)
for N, Ps in ((513, ("64", "32")), (257, ("64", "32", "16")), (1025, ("64",))):
    for n in (1024, 2048, 4096, 8192, 16384, 65536):
        g, c, f, h = fam(n, N)
        row = []
        for P in Ps:
            os.environ["IBS_FORCE_P"] = P
            ctx = ibs_amd.Context(0)
            ctx.solve_gcf(h, g, c, f); torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in ev:
                a.record(); ctx.solve_gcf(h, g, c, f); b.record()
            torch.cuda.synchronize()
            row.append("P=%s %.1f us" % (P, 1e3 * min(a.elapsed_time(b) for a, b in ev)))
        print("N=%d n_sys=%d: %s" % (N, n, "  ".join(row)), flush=True)
