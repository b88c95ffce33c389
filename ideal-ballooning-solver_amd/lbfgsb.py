"""Host driver of the bounded quasi-Newton state machine (csrc/ibs_lbfgsb2.hpp; C ABI ibs_lbfgsb2_*).

`minimize2(fun, x0, bounds, ...)` is the counterpart of the reference's
    scipy.optimize.minimize(obj_w_grad, x0, jac=True, bounds=..., options={"ftol", "gtol", "maxiter"})
(ball_scan.py:307-314) for the two unknowns (alpha, theta0).  No GPU is involved in the optimizer itself; `fun` is
whatever evaluates (val, jac) -- on the GPU path BallooningScan.obj_w_grad."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

TASKS = {10: "CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL", 11: "CONVERGENCE: RELATIVE REDUCTION OF F <= FACTR*EPSMCH",
         12: "ABNORMAL: ", 13: "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT", 14: "ERROR"}


class Result(dict):
    __getattr__ = dict.__getitem__


def minimize2(fun, x0, bounds, ftol=5.0e-11, gtol=2.0e-8, maxiter=30, maxls=20, args=()):
    lib = _lib.lib()
    st = C.create_string_buffer(lib.ibs_lbfgsb2_state_bytes())
    lo = np.array([bounds[0][0], bounds[1][0]], dtype=np.float64)
    hi = np.array([bounds[0][1], bounds[1][1]], dtype=np.float64)
    x = np.clip(np.asarray(x0, dtype=np.float64).reshape(2), lo, hi)
    p = lambda a: C.c_void_p(a.ctypes.data)
    check(lib.ibs_lbfgsb2_init(st, p(x), p(lo), p(hi), float(ftol), float(gtol), int(maxiter), int(maxls)), "ibs_lbfgsb2_init")
    xn = x.copy()
    trace = []
    while True:
        f, g = fun(xn.copy(), *args)
        g = np.ascontiguousarray(g, dtype=np.float64)
        trace.append((xn[0], xn[1], float(f), g[0], g[1]))
        more = check(lib.ibs_lbfgsb2_step(st, float(f), p(g), p(xn)), "ibs_lbfgsb2_step")
        if not more:
            break
    xo = np.empty(2); fo = C.c_double(0.0); cnt = np.zeros(5, dtype=np.int32)
    check(lib.ibs_lbfgsb2_result(st, p(xo), C.byref(fo), p(cnt)), "ibs_lbfgsb2_result")
    return Result(x=xo, fun=fo.value, nit=int(cnt[0]), nfev=len(trace), task=int(cnt[2]), message=TASKS.get(int(cnt[2]), "?"),
                  restarts=int(cnt[3]), skipped=int(cnt[4]), trace=np.array(trace))
