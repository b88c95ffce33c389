"""Phase timestamps inside k_solve_gcf_rows<double,32> (N_zeta = 2048; debug build: make -C ideal-ballooning-solver_amd/csrc probe2 PM=32;
IBS_LIB_PATH=.../libibs_hip_probe2.so python tools/rows_probe.py): one wave per SIMD, 4,096 smooth systems (one wave-round of the chip x 4)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
from ibs_amd import _lib
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
nz, n_sys = 2048, 1024
N = nz + 1; h = 8 * np.pi / nz
th = torch.linspace(-4 * np.pi, 4 * np.pi, N, dtype=torch.float64, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(20240 + nz)
u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)
sh, al, t0 = u(0.1, 2.0, (n_sys, 1)), u(0.0, 1.2, (n_sys, 1)), u(0.0, np.pi / 2, (n_sys, 1))
lam = sh * (th[None] - t0) - al * (torch.sin(th)[None] - torch.sin(t0))
g = 1 + lam ** 2; c = al * (torch.cos(th)[None] + torch.sin(th)[None] * lam); f = g.clone()
def run(tag, call):
    for _ in range(2):
        r = call()
    torch.cuda.synchronize()
    buf = np.zeros((1024 * 4, 16), dtype=np.int64)
    _lib.lib().ibs_probe_read(C.c_void_p(buf.ctypes.data), buf.size)
    w = buf[buf[:, 0] > 0]
    us = lambda x: x * 0.01
    print("%s (%s): waves stamped %d; sweeps per solve %.2f" % (tag, ctx.last_launch()[0], len(w), float((r["info"] & 0xffff).double().mean())))
    for nm, a, b in (("sampled trial residual", 0, 1), ("set-up (3 staged rows)", 1, 2), ("shift iteration", 2, 10), ("backward sweep", 10, 11),
                     ("twisted + polish", 11, 12), ("growth rate (3 staged rows)", 3, 4), ("whole wave", 0, 4)):
        d = us(w[:, b] - w[:, a])
        print("   %-28s median %7.2f  min %7.2f  max %7.2f us" % (nm, np.median(d), d.min(), d.max()))
    print("   forward sweeps: %.3f us each; decision code %.3f us per main-branch iteration" % (us(w[:, 5].sum()) / max(1, (w[:, 7] + 3).sum()), us(w[:, 6].sum()) / max(1, w[:, 7].sum())))


run("FP64 rows", lambda: ctx.solve_gcf(h, g, c, f, want_info=True))
g32, c32, f32 = g.float(), c.float(), f.float()
run("FP32 rows, FP64 solver", lambda: ctx.solve_gcf(h, g32, c32, f32, want_info=True, dtype=np.float32))
