"""32 lanes per system on the production grids (N = 969 / 1025: 31 / 32 rows per lane) against one wave per system:
the config-3 shape (64 x 32 x 16, N = 1025), the config-4 shape (73 x 5 x 24 x 15, N = 969) and the reference batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
wout = dict(np.load(os.path.join(ROOT, 'tests/golden/G8_wout_ncsx_op.npz')))
for name, neq, ns, na, nt0, N in (("reference batch", 1, 5, 24, 15, 969), ("config 3", 1, 64, 32, 16, 1025), ("config 4 shape", 73, 5, 24, 15, 969)):
    svals = np.linspace(0.5, 0.95, ns) if ns <= 16 else np.linspace(0.1, 0.95, ns)
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    if neq > 1:
        tabs = ibs_amd.SurfaceTables.concat([tabs] * neq)
    th = ibs_amd.theta_grid(N); t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(dev)
    surf = np.repeat(np.arange(neq * ns), na); al = np.tile(np.linspace(0, np.pi, na), neq * ns)
    r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    args = (th[1] - th[0], *[r['geo'][k] for k in range(7)], r['dPdrho'], t0)
    ref = None
    for P, chain in ((64, 0), (32, 0), (32, 1), (32, 2), (32, 4), (32, 5), (32, 8)):
        if P == 32 and nt0 % 2:
            continue
        ctx.set_option("force_p", P); ctx.set_option("scan_chain", chain)
        try:
            out = ctx.gamma_scan(*args, want_info=True)
        except Exception as e:
            print(name, P, chain, "failed:", e); continue
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.gamma_scan(*args); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        if ref is None:
            ref = out["gam"].clone()
        print("%-16s P=%d chain=%d: %.3f ms (min of 5), %.1f sweeps, flagged %d, max|dgam| vs first %.1e" % (
            name, P, chain, min(ts), float((out['info'] & 0xffff).double().mean()), int(((out['info'] >> 16) != 0).sum()),
            float((out["gam"] - ref).abs().max())))
