"""BASELINE config 5: 10^6 raw (g,c,f) systems, N_zeta in {256, 512, 1024, 2048}, FP64 and FP32,
smooth (s-alpha) and rough (iid) families, seeds numpy.random.default_rng(20240 + N_zeta) -> torch generator seed.
Prints one line per case: solves/s, algorithmic GB/s, mean sweeps, non-converged count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
dev = torch.device("cuda", 0)
ctx = ibs_amd.Context(0)
n_sys = int(os.environ.get("IBS_C5_N", "1000000"))
for nz in [int(v) for v in os.environ.get("IBS_C5_NZ", "256,512,1024,2048").split(",")]:
    N = nz + 1
    h = 8 * np.pi / nz
    th = torch.linspace(-4 * np.pi, 4 * np.pi, N, dtype=torch.float64, device=dev)
    for fam in ("smooth", "rough"):
        gen = torch.Generator(device=dev); gen.manual_seed(20240 + nz)
        u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)
        if fam == "smooth":
            sh, al, t0 = u(0.1, 2.0, (n_sys, 1)), u(0.0, 1.2, (n_sys, 1)), u(0.0, np.pi / 2, (n_sys, 1))
            lam = sh * (th[None] - t0) - al * (torch.sin(th)[None] - torch.sin(t0))
            g = 1 + lam ** 2
            c = al * (torch.cos(th)[None] + torch.sin(th)[None] * lam)
            f = g.clone()
            del lam
        else:
            g = torch.exp(u(np.log(0.01), np.log(50.0), (n_sys, N)))
            c = u(-2.5, 3.5, (n_sys, N))
            f = torch.exp(u(np.log(0.2), np.log(3e3), (n_sys, N)))
        for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
            if fam == "rough" and dt == np.float32:
                continue
            gg, cc, ff = (x.to(tdt) for x in (g, c, f))
            r = ctx.solve_gcf(h, gg, cc, ff, want_info=True, dtype=dt)
            torch.cuda.synchronize()
            nbad = int(((r["info"] >> 16) != 0).sum().item())
            sweeps = float((r["info"] & 0xffff).double().mean().item())
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
            for a, b in ev:
                a.record(); ctx.solve_gcf(h, gg, cc, ff, dtype=dt); b.record()
            torch.cuda.synchronize()
            ms = min(a.elapsed_time(b) for a, b in ev)
            w = 8 if dt == np.float64 else 4
            print("N_zeta=%4d %-6s %s  %8d systems: %.3e solves/s  %7.1f GB/s algorithmic (%.1f %% of 8 TB/s)  %.1f sweeps  flagged %d"
                  % (nz, fam, "f64" if w == 8 else "f32", n_sys, n_sys / (ms * 1e-3), n_sys * (3 * N + 1) * w / (ms * 1e-3) / 1e9,
                     100 * n_sys * (3 * N + 1) * w / (ms * 1e-3) / 8e12, sweeps, nbad), flush=True)
            del gg, cc, ff, r
        del g, c, f
        torch.cuda.empty_cache()
