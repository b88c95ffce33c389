"""Field-line geometry producer.

Host part: `SurfaceTables` = data-only counterpart of the reference's `vmec_splines` (utils.py:37-158)
plus the per-surface spline evaluation at the top of `vmec_fieldlines` (utils.py:311-357).  It takes the
plain wout tables (what `simsopt.mhd.vmec.Vmec.wout` exposes; SIMSOPT itself is not needed) and yields the
Fourier coefficient vectors of each requested surface.  Radial splines are 1-D host work (scipy FITPACK,
like the reference); everything per grid point runs on the GPU (`Context.fieldline_geometry`).
"""
import numpy as np

NAMES_MN = ("rmnc", "zmns", "lmns", "d_rmnc_d_s", "d_zmns_d_s", "d_lmns_d_s")
NAMES_NYQ = ("gmnc", "bmnc", "d_bmnc_d_s", "bsupvmnc", "bsubsmns", "bsubumnc", "bsubvmnc")
# wout arrays the geometry needs, in the order ibs_surface_tables_f64 takes them (stored (modes, ns) as in simsopt's Vmec.wout)
WOUT_ARRAYS = ("rmnc", "zmns", "lmns", "gmnc", "bmnc", "bsupvmnc", "bsubsmns", "bsubumnc", "bsubvmnc")


import functools


@functools.lru_cache(maxsize=16)
def _radial_weights(ns, svals_bytes):
    """(W_full, W_full', W_half, W_half'): matrices that evaluate, at svals, the not-a-knot cubic spline (and its
    derivative) through data given on VMEC's full mesh linspace(0, 1, ns) / half mesh (columns 1..ns-1)."""
    from scipy.interpolate import make_interp_spline
    svals = np.frombuffer(svals_bytes, dtype=np.float64)
    s_full = np.linspace(0, 1, ns)                           # vmec.s_full_grid
    s_half = s_full[1:] - 0.5 * (s_full[1] - s_full[0])      # vmec.s_half_grid
    out = []
    for grid in (s_full, s_half):
        sp = make_interp_spline(grid, np.eye(len(grid)), k=3)
        out += [np.ascontiguousarray(sp(svals)), np.ascontiguousarray(sp.derivative()(svals))]
    return tuple(out)


def mode_rows(xm, xn, dn=None, max_len=64):
    """runs of modes with equal m and n advancing by the common step dn (VMEC's ordering: dn = nfp) as int32
    (nrows, 2) = {first, count}, plus dn.  Any ordering is valid: runs simply get shorter (down to 1)."""
    xm = np.asarray(xm); xn = np.asarray(xn)
    if dn is None:
        d = np.diff(xn)[np.diff(xm) == 0]
        dn = float(np.median(d)) if len(d) else 1.0
    rows = []
    k, n = 0, len(xm)
    while k < n:
        e = k + 1
        while e < n and e - k < max_len and xm[e] == xm[k] and abs((xn[e] - xn[e - 1]) - dn) <= 1e-12 * max(1.0, abs(dn)):
            e += 1
        rows.append((k, e - k))
        k = e
    return np.ascontiguousarray(rows, dtype=np.int32), dn


@functools.lru_cache(maxsize=16)
def _mode_rows_cached(xm_bytes, xn_bytes):
    """mode_rows for the (unchanging) mode tables of an equilibrium family"""
    return mode_rows(np.frombuffer(xm_bytes, dtype=np.float64), np.frombuffer(xn_bytes, dtype=np.float64))


class SurfaceTables:
    """per-surface inputs of the geometry kernel, packed as the C ABI wants them (include/ibs.h)"""

    def __init__(self, s, xm, xn, xm_nyq, xn_nyq, tab_mn, tab_nyq, iota, d_iota_d_s, d_pressure_d_s, phiedge, Aminor_p):
        self.s = np.ascontiguousarray(s, dtype=np.float64)
        self.xm, self.xn = (np.ascontiguousarray(a, dtype=np.float64) for a in (xm, xn))
        self.xm_nyq, self.xn_nyq = (np.ascontiguousarray(a, dtype=np.float64) for a in (xm_nyq, xn_nyq))
        self.tab_mn = np.ascontiguousarray(tab_mn, dtype=np.float64)        # (n_surf, 6, mnmax)
        self.tab_nyq = np.ascontiguousarray(tab_nyq, dtype=np.float64)      # (n_surf, 7, mnmax_nyq)
        n = len(self.s)
        self.scal = np.ascontiguousarray(np.stack([self.s, iota, d_iota_d_s, d_pressure_d_s,
                                                   np.full(n, float(phiedge)), np.full(n, float(Aminor_p))], axis=1))
        assert self.tab_mn.shape == (n, 6, len(self.xm)) and self.tab_nyq.shape == (n, 7, len(self.xm_nyq))
        self.rows_mn, self.dn_mn = _mode_rows_cached(self.xm.tobytes(), self.xn.tobytes())
        self.rows_nyq, self.dn_nyq = _mode_rows_cached(self.xm_nyq.tobytes(), self.xn_nyq.tobytes())

    @classmethod
    def from_arrays(cls, d):
        """from a dict holding the per-surface vectors by name (e.g. tests/golden/G8_surface_tables.npz)"""
        return cls(d["s"], d["xm"], d["xn"], d["xm_nyq"], d["xn_nyq"],
                   np.stack([d[k] for k in NAMES_MN], axis=1), np.stack([d[k] for k in NAMES_NYQ], axis=1),
                   d["iota"], d["d_iota_d_s"], d["d_pressure_d_s"], float(d["phiedge"]), float(d["Aminor_p"]))

    @classmethod
    def concat(cls, tables):
        """one table set holding the surfaces of several equilibria (the base equilibrium and its DOF-perturbed
        copies, sims_runner_NCSX.py:151-276): surface index = i_equilibrium * n_surf + i_surface.  All members must share
        the mode tables; phiedge / Aminor_p are per-surface scalars in the packed layout, so they may differ."""
        t0 = tables[0]
        for t in tables[1:]:
            if not (np.array_equal(t.xm, t0.xm) and np.array_equal(t.xn, t0.xn) and np.array_equal(t.xm_nyq, t0.xm_nyq)
                    and np.array_equal(t.xn_nyq, t0.xn_nyq)):
                raise ValueError("SurfaceTables.concat: mode tables differ")
        out = cls.__new__(cls)
        out.s = np.concatenate([t.s for t in tables])
        out.xm, out.xn, out.xm_nyq, out.xn_nyq = t0.xm, t0.xn, t0.xm_nyq, t0.xn_nyq
        out.tab_mn = np.ascontiguousarray(np.concatenate([t.tab_mn for t in tables]))
        out.tab_nyq = np.ascontiguousarray(np.concatenate([t.tab_nyq for t in tables]))
        out.scal = np.ascontiguousarray(np.concatenate([t.scal for t in tables]))
        out.rows_mn, out.dn_mn, out.rows_nyq, out.dn_nyq = t0.rows_mn, t0.dn_mn, t0.rows_nyq, t0.dn_nyq
        return out

    @classmethod
    def frame(cls, wout0, svals, n_eq, pinned=False):
        """an EMPTY table set for the surfaces `svals` of n_eq equilibria that share wout0's radial mesh and mode tables
        (surface index = i_equilibrium * len(svals) + i_surface); fill(first, wouts) computes the tables of a run of
        equilibria in place.  pinned: the big arrays live in page-locked host memory (torch), so that their upload can run
        asynchronously while the host computes the next run (AdjointStep)."""
        ns = int(wout0["ns"])
        svals = np.atleast_1d(np.asarray(svals, dtype=np.float64))
        n_s = len(svals)
        out = cls.__new__(cls)
        out.s = np.tile(svals, n_eq)
        out.xm, out.xn = (np.ascontiguousarray(wout0[k], dtype=np.float64) for k in ("xm", "xn"))
        out.xm_nyq, out.xn_nyq = (np.ascontiguousarray(wout0[k], dtype=np.float64) for k in ("xm_nyq", "xn_nyq"))
        mnmax, mnq = len(out.xm), len(out.xm_nyq)

        out._pinned = {}                  # name -> page-locked torch tensor the numpy array of that name is a view of

        def alloc(name, shape):
            if pinned:
                import torch
                t = torch.empty(shape, dtype=torch.float64).pin_memory()
                out._pinned[name] = t
                return t.numpy()
            return np.empty(shape)
        out.tab_mn, out.tab_nyq, out.scal = alloc("tab_mn", (n_eq * n_s, 6, mnmax)), alloc("tab_nyq", (n_eq * n_s, 7, mnq)), alloc("scal", (n_eq * n_s, 6))
        out.scal[:, 0] = out.s
        out.rows_mn, out.dn_mn = _mode_rows_cached(out.xm.tobytes(), out.xn.tobytes())
        out.rows_nyq, out.dn_nyq = _mode_rows_cached(out.xm_nyq.tobytes(), out.xn_nyq.tobytes())
        out.n_equilibria, out.n_surf_per_equilibrium, out._ns, out._svals = n_eq, n_s, ns, svals
        Wf, Wfd, Wh, Whd = _radial_weights(ns, svals.tobytes())
        # half-mesh data sit in columns 1..ns-1 of the (mn, ns) tables: a zero column in front of the weights lets the
        # product run on the table as it lies in memory
        z = np.zeros((n_s, 1))
        out._weights = (Wf, Wfd, np.ascontiguousarray(np.hstack([z, Wh])), np.ascontiguousarray(np.hstack([z, Whd])), Wh, Whd)
        return out

    def fill(self, first, wouts, n_threads=0):
        """tables of the equilibria first .. first + len(wouts) - 1 of a frame(), in place: the radial splines of
        vmec_splines (utils.py:58-119) evaluated at the surfaces (utils.py:311-357) for all of them by the library's
        threaded host routine (ibs_surface_tables_f64: every wout table is read once where it lies)."""
        import ctypes as C
        from . import _lib
        wouts = list(wouts)
        ns, n_s, n = self._ns, self.n_surf_per_equilibrium, len(wouts)
        if first < 0 or first + n > self.n_equilibria:
            raise ValueError("SurfaceTables.fill: equilibria %d..%d outside the frame of %d" % (first, first + n - 1, self.n_equilibria))
        mnmax, mnq = len(self.xm), len(self.xm_nyq)
        Wf, Wfd, Wh0, Whd0, Wh, Whd = self._weights
        keep, ptrs = [], (C.c_void_p * (9 * n))()
        for q, w in enumerate(wouts):
            if int(w["ns"]) != ns or not all(w[k] is getattr(self, k) or np.array_equal(w[k], getattr(self, k)) for k in ("xm", "xn", "xm_nyq", "xn_nyq")):
                raise ValueError("SurfaceTables: the equilibria must share the radial mesh and the mode tables")
            for k, name in enumerate(WOUT_ARRAYS):
                a = np.ascontiguousarray(w[name], dtype=np.float64)          # (no copy for the arrays simsopt / numpy hand over)
                if a.shape != ((mnmax if k < 3 else mnq), ns):
                    raise ValueError("SurfaceTables: %s of equilibrium %d has shape %s" % (name, first + q, a.shape))
                keep.append(a)
                ptrs[9 * q + k] = a.ctypes.data
        p = lambda a: C.c_void_p(a.ctypes.data)
        r0, r1 = first * n_s, (first + n) * n_s
        _lib.check(_lib.lib().ibs_surface_tables_f64(n, ns, n_s, mnmax, mnq, ptrs, p(Wf), p(Wfd), p(Wh0), p(Whd0),
                                                     p(self.tab_mn[r0:r1]), p(self.tab_nyq[r0:r1]), int(n_threads)), "ibs_surface_tables_f64")
        pres = np.stack([np.asarray(w["pres"], dtype=np.float64)[1:] for w in wouts])        # utils.py:112
        iota = np.stack([np.asarray(w["iotas"], dtype=np.float64)[1:] for w in wouts])       # utils.py:118
        sc = self.scal[r0:r1]
        sc[:, 1] = (iota @ Wh.T).reshape(-1); sc[:, 2] = (iota @ Whd.T).reshape(-1); sc[:, 3] = (pres @ Whd.T).reshape(-1)
        sc[:, 4] = np.repeat([float(np.asarray(w["phi"])[-1]) for w in wouts], n_s)
        sc[:, 5] = np.repeat([float(w["Aminor_p"]) for w in wouts], n_s)
        return r0, r1

    @classmethod
    def from_wouts(cls, wouts, svals, n_threads=0):
        """the surfaces `svals` of SEVERAL equilibria in one table set (the base equilibrium and its DOF-perturbed copies
        of one optimizer step, sims_runner_NCSX.py:151-276; upstream every one of them runs its own vmec_splines,
        utils.py:37-158): surface index = i_equilibrium * len(svals) + i_surface, like concat([from_wout(w, svals) ...]).
        = frame() + fill() of all equilibria (73 equilibria = 115 MB of wout tables: 2.2 ms with 16 threads)."""
        wouts = list(wouts)
        out = cls.frame(wouts[0], svals, len(wouts))
        out.fill(0, wouts, n_threads)
        return out

    @classmethod
    def from_wout(cls, wout, svals):
        """wout: mapping with rmnc, zmns, lmns, gmnc, bmnc, bsupvmnc, bsubsmns, bsubumnc, bsubvmnc stored
        (mn, ns) as in simsopt's Vmec.wout, pres, iotas, phi (ns,), xm, xn, xm_nyq, xn_nyq, Aminor_p, ns."""
        ns = int(wout["ns"])
        svals = np.atleast_1d(np.asarray(svals, dtype=np.float64))
        # cubic interpolating splines with not-a-knot ends = what InterpolatedUnivariateSpline builds (utils.py:58-107).
        # Interpolation is linear in the data and the radial grids are the same for every mode, array and
        # equilibrium, so the splines are applied as four weight matrices (value / derivative on the full / half
        # mesh at svals), built once per (ns, svals) by splining the identity: one small matrix product per array.
        Wf, Wfd, Wh, Whd = _radial_weights(ns, svals.tobytes())

        def ev(tab, half, deriv=False):
            tab = np.asarray(tab, dtype=np.float64)
            if half:
                return (Whd if deriv else Wh) @ tab[:, 1:].T          # (n_s, modes)
            return (Wfd if deriv else Wf) @ tab.T

        mn = np.stack([ev(wout["rmnc"], False), ev(wout["zmns"], False), ev(wout["lmns"], True),
                       ev(wout["rmnc"], False, True), ev(wout["zmns"], False, True), ev(wout["lmns"], True, True)], axis=1)
        nyq = np.stack([ev(wout["gmnc"], True), ev(wout["bmnc"], True), ev(wout["bmnc"], True, True),
                        ev(wout["bsupvmnc"], True), ev(wout["bsubsmns"], False), ev(wout["bsubumnc"], True),
                        ev(wout["bsubvmnc"], True)], axis=1)
        pres_h = np.asarray(wout["pres"], dtype=np.float64)[1:]        # utils.py:112
        iota_h = np.asarray(wout["iotas"], dtype=np.float64)[1:]       # utils.py:118
        return cls(svals, wout["xm"], wout["xn"], wout["xm_nyq"], wout["xn_nyq"], mn, nyq, Wh @ iota_h,
                   Whd @ iota_h, Whd @ pres_h, float(np.asarray(wout["phi"])[-1]), float(wout["Aminor_p"]))
