"""ctypes wrapper of the C oracle (oracle/ibs_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("IBS_ORACLE_SO") or os.path.join(_HERE, "_build", "libibs_oracle.so")   # (override: sanitizer build)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(_SO)
        P, D, I, L = C.c_void_p, C.c_double, C.c_int, C.c_long
        _lib.ibs_oracle_solve_gcf.argtypes = [I, D, P, P, P, P, P, P, P, P]
        _lib.ibs_oracle_solve_gcf_batch.argtypes = [L, I, D, P, P, P, L, P, P, I]
        _lib.ibs_oracle_gamma_scan.argtypes = [I, I, I, D, P, P, P, P, P, P, P, L, P, P, P, P, I]
        _lib.ibs_oracle_lam_batch.argtypes = [L, I, D, P, P, P, L, P, I]
        _lib.ibs_oracle_count_above_batch.argtypes = [L, I, D, P, P, P, L, P, P, I]
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


def solve_gcf(h, g, c, f):
    g, c, f = (np.ascontiguousarray(a, dtype=np.float64) for a in (g, c, f))
    N = len(g)
    lam = np.zeros(1); gam = np.zeros(1); X = np.zeros(N); dX = np.zeros(N); work = np.zeros(12 * N)
    rc = lib().ibs_oracle_solve_gcf(N, float(h), _p(g), _p(c), _p(f), _p(lam), _p(gam), _p(X), _p(dX), _p(work))
    assert rc == 0
    return gam[0], lam[0], X, dX


def solve_gcf_batch(h, g, c, f, nthreads=0):
    g, c, f = (np.ascontiguousarray(a, dtype=np.float64) for a in (g, c, f))
    n, N = g.shape
    lam = np.zeros(n); gam = np.zeros(n)
    used = lib().ibs_oracle_solve_gcf_batch(n, N, float(h), _p(g), _p(c), _p(f), N, _p(lam), _p(gam), int(nthreads))
    return gam, lam, used


def gamma_scan(h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, dPdrho, theta0, nthreads=0):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22)]
    dPdrho = np.ascontiguousarray(dPdrho, dtype=np.float64)
    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    nl, N = arrs[0].shape
    nt = len(theta0)
    gam = np.zeros((nl, nt)); lam = np.zeros((nl, nt))
    used = lib().ibs_oracle_gamma_scan(nl, nt, N, float(h), *[_p(a) for a in arrs], N, _p(dPdrho), _p(theta0),
                                       _p(gam), _p(lam), int(nthreads))
    return gam, lam, used


def lam_batch(h, g, c, f, nthreads=0):
    """lam_max alone of every system (division-form bisection, no eigenvector stage)."""
    g, c, f = (np.ascontiguousarray(a, dtype=np.float64) for a in (g, c, f))
    n, N = g.shape
    lam = np.zeros(n)
    lib().ibs_oracle_lam_batch(n, N, float(h), _p(g), _p(c), _p(f), N, _p(lam), int(nthreads))
    return lam


def count_above_batch(h, g, c, f, shift, nthreads=0):
    """division-form Sturm count of every system at its own shift: eigenvalues of (T, F) above shift[s]."""
    g, c, f = (np.ascontiguousarray(a, dtype=np.float64) for a in (g, c, f))
    shift = np.ascontiguousarray(shift, dtype=np.float64)
    n, N = g.shape
    cnt = np.zeros(n, dtype=np.int32)
    lib().ibs_oracle_count_above_batch(n, N, float(h), _p(g), _p(c), _p(f), N, _p(shift), _p(cnt), int(nthreads))
    return cnt
