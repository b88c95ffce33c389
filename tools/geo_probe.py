"""Phase timestamps inside k_geo_rows (debug build of the library with -DGEO_PROBE):
   bash: make -C ideal-ballooning-solver_amd/csrc probe   ->  lib/libibs_hip_probe.so;   IBS_LIB_PATH=... python tools/geo_probe.py
phases of wave 0 of each block: block start | item start | after point set-up | after (P,Q) pass | after root solve |
after non-Nyquist synthesis | after Nyquist synthesis | after metric algebra + stores"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
from ibs_amd import _lib
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
names = ["start->item", "point set-up", "(P,Q) pass", "root solve", "synth mn", "synth nyq", "algebra+stores"]
for ns, na, N, lpp in ((5, 3, 969, 8), (5, 3, 969, 4), (73, 3, 969, 1), (64, 32, 1025, -2)):
    svals = np.linspace(0.5, 0.95, ns)
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = torch.from_numpy(ibs_amd.theta_grid(N)).to(dev)
    surf = torch.from_numpy(np.repeat(np.arange(ns), na).astype(np.int32)).to(dev)
    al = torch.from_numpy(np.tile(np.linspace(0, np.pi, na), ns)).to(dev)
    ctx.set_option("geo_lpp", lpp)
    for _ in range(3):
        r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    torch.cuda.synchronize()
    buf = np.zeros((256, 16), dtype=np.int64)
    _lib.lib().ibs_geo_probe_clear()
    r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev); torch.cuda.synchronize()
    rc = _lib.lib().ibs_geo_probe_read(C.c_void_p(buf.ctypes.data))
    d = np.diff(buf[:64, :8], axis=1) * 10.0 / 1e3        # us (100 MHz clock); last item of wave 0 of each block
    print("%d lines x %d, geo_lpp=%d: per-phase us (median over blocks | max):" % (ns * na, N, lpp))
    for k, nm in enumerate(names):
        print("   %-16s %7.2f | %7.2f" % (nm, np.median(d[:, k]), d[:, k].max()))
    nb = min(256, (ns * na * ((N + (8 if lpp == 8 else 64)) - 1) // 64 // 8) if lpp == 8 else 256)
    st = (buf[:, 0] - buf[buf[:, 0] > 0, 0].min()) * 0.01; en = (buf[:, 7] - buf[buf[:, 0] > 0, 0].min()) * 0.01
    ok = buf[:, 0] > 0
    print("   blocks probed %d: block start skew us min/median/max %.2f %.2f %.2f ; block end min/median/max %.2f %.2f %.2f" % (
        ok.sum(), st[ok].min(), np.median(st[ok]), st[ok].max(), en[ok].min(), np.median(en[ok]), en[ok].max()))
    ends = (buf[ok, 8:16] - buf[ok, 0:1]) * 0.01
    print("   waves 0..7 done at (us after block start, median over blocks): " + " ".join("%.1f" % np.median(ends[:, w]) for w in range(8)))
    print("   total item       %7.2f ; block span (start -> end of last item) median %.2f" % (np.median(d[:, 1:].sum(axis=1)), np.median((buf[:64, 7] - buf[:64, 0]) * 0.01)))
