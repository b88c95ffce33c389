"""Phase timestamps inside k_refine_eval<double,16> (debug build: make -C ideal-ballooning-solver_amd/csrc probe2;
IBS_LIB_PATH=.../libibs_hip_probe2.so python tools/eval_probe.py): the LAST round of a refinement of the reference batch.
phases of wave 0 of each block: start | dPdrho + state in LDS | staging + tangent | set-up | solve | growth rate + gradient |
optimizer step | state written back"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
from ibs_amd import _lib
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
N = 969; svals = np.linspace(0.5, 0.95, 5); th = ibs_amd.theta_grid(N)
tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
scan = ibs_amd.BallooningScan(ctx, None, th, svals, tables=tabs, device=dev)
st = np.array([ibs_amd.pick_start(t, scan.alpha_scan, scan.theta0_scan)[:2] for t in scan.coarse()])
names = ["(unused)", "staging + tangent + dPdrho", "set-up", "solve", "eigenvector + sums", "optimizer step", "state write-back"]
for maxiter in (0, 30):        # maxiter 0: a single (cold) evaluation per point; 30: the last (warm) round of the slowest point
    scan.refine_device(st, maxiter=maxiter); xo, fo, ne = scan.refine_device(st, maxiter=maxiter)
    torch.cuda.synchronize()
    buf = np.zeros((1024 * 4, 16), dtype=np.int64)
    _lib.lib().ibs_probe_read(C.c_void_p(buf.ctypes.data), buf.size)
    b = buf.reshape(1024, 4, 16)[:5, 0, :8]
    b = b[b[:, 0] > 0]
    d = np.diff(b, axis=1) * 0.01
    print("maxiter %d: evaluations %s; blocks with stamps: %d; per-phase us of the stamped rounds (per block):" % (maxiter, ne, len(b)))
    for k, nm in enumerate(names):
        print("   %-20s %s" % (nm, np.round(d[:, k], 2)))
    print("   total                %s" % np.round(d.sum(axis=1), 2))
