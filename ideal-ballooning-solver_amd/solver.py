"""Host-side handle on the HIP solver: one Context per process/GPU.

Accepts numpy arrays (host memory: staged by the library) or torch CUDA tensors (device memory:
zero-copy, asynchronous on the torch current stream).  Mirrors the reference operators'
argument meaning; see operators.py for the exact drop-in signatures.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import MEM_DEVICE, MEM_HOST, IbsError, check


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class _Args:
    """marshals a homogeneous set of arrays (all numpy or all torch-cuda) to raw pointers"""

    def __init__(self, dtype=np.float64):
        self.mem = None
        self.keep = []
        self.dtype = np.dtype(dtype)

    def _torch_dtype(self):
        import torch
        return {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32,
                np.dtype(np.int32): torch.int32}

    def inp(self, x, dtype=None):
        dtype = np.dtype(dtype or self.dtype)
        if _is_torch(x):
            if not x.is_cuda:
                raise IbsError("torch tensors must live on the GPU (got %s)" % x.device)
            td = self._torch_dtype()[dtype]
            if x.dtype != td or not x.is_contiguous():
                x = x.to(td).contiguous()
            self._set(MEM_DEVICE)
            self.keep.append(x)
            return C.c_void_p(x.data_ptr())
        a = np.ascontiguousarray(x, dtype=dtype)
        self._set(MEM_HOST)
        self.keep.append(a)
        return C.c_void_p(a.ctypes.data)

    def out(self, shape, like_torch=None, dtype=None, want=True):
        dtype = np.dtype(dtype or self.dtype)
        if not want:
            return None, C.c_void_p(None)
        if self.mem == MEM_DEVICE:
            import torch
            t = torch.empty(shape, dtype=self._torch_dtype()[dtype], device=like_torch.device)
            self.keep.append(t)
            return t, C.c_void_p(t.data_ptr())
        a = np.empty(shape, dtype=dtype)
        self.keep.append(a)
        return a, C.c_void_p(a.ctypes.data)

    def _set(self, mem):
        if self.mem is None:
            self.mem = mem
        elif self.mem != mem:
            raise IbsError("mixing host (numpy) and device (torch.cuda) arrays in one call is not supported")



def _device_key(device):
    """'cuda:<index>' for every spelling of one device (torch.device('cuda'), 'cuda', 'cuda:0', a tensor's .device): the
    key of the resident copies a table set keeps per device, so that an upload under one spelling is found under another"""
    import torch
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


class Context:
    def __init__(self, device=0):
        self._lib = _lib.lib()
        self._h = C.c_void_p(None)
        check(self._lib.ibs_create(C.byref(self._h), int(device)), "ibs_create")
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._comm_world = 0
            self._lib.ibs_destroy(self._h)
            self._h = C.c_void_p(None)

    __del__ = close

    def _stream_from_torch(self, ref):
        import torch
        self._lib.ibs_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream))

    def synchronize(self):
        check(self._lib.ibs_synchronize(self._h), "ibs_synchronize")

    # ---- native RCCL all-gather (include/ibs.h: ibs_comm_*) -------------------------------------------------
    def comm_init(self, dist, rank, world):
        """create this rank's RCCL communicator inside the library; the unique id travels through `dist`
        (any initialised torch.distributed backend).  Collective over all ranks."""
        import torch
        # (IBS_RCCL_LIB: another library with RCCL's five entry points -- the shared-memory stand-in tests/cabi/fake_rccl.c lets the
        #  multi-rank code run on a one-GPU box, where RCCL refuses two ranks on one device)
        rccl = os.environ.get("IBS_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        # every rank takes part in the same collectives of `dist` whatever fails locally (a rank that raised before the
        # broadcast would leave the others waiting in it): load, id, broadcast, agreement, and only then ncclCommInitRank
        ident = C.create_string_buffer(128)
        problem = None
        try:
            check(self._lib.ibs_comm_load(rccl.encode() if os.path.exists(rccl) else None), "ibs_comm_load")
            if rank == 0:
                check(self._lib.ibs_comm_unique_id(ident), "ibs_comm_unique_id")
        except IbsError as e:
            problem = str(e)
        box = [ident.raw if (rank == 0 and problem is None) else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
            seen = [None] * world
            dist.all_gather_object(seen, problem)
        else:
            seen = [problem]
        if box[0] is None or any(p is not None for p in seen):
            raise IbsError("native RCCL communicator not available on every rank: %s" % ([p for p in seen if p] or ["no id from rank 0"])[0])
        ident = C.create_string_buffer(box[0], 128)
        check(self._lib.ibs_comm_init(self._h, ident, int(rank), int(world)), "ibs_comm_init")
        self._comm_world = int(world)

    def allgather(self, send, recv):
        """recv (world * n,) <- every rank's send (n,): float64 CUDA tensors, asynchronous on the current torch stream"""
        self._stream_from_torch(send)
        check(self._lib.ibs_comm_allgather_f64(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                               send.numel()), "ibs_comm_allgather_f64")

    def allgather_start(self, send, recv, slot=0, then_wait=-1, same_stream=False, host_wait=None):
        """the same gather on the communicator's own stream, ordered after the work enqueued so far on the current torch
        stream: later launches do not wait for it.  comm_wait(slot) before `send` / `recv` are reused or read;
        then_wait >= 0 does comm_wait(then_wait) in the same call; host_wait = s instead waits for slot s on the host
        (no stream operation; for callers that run several slots ahead).  same_stream: the context's stream is already
        the current torch stream (the call follows a launch of this context)."""
        if host_wait is not None:
            then_wait = -2 - int(host_wait)
        if not same_stream:
            self._stream_from_torch(send)
        check(self._lib.ibs_comm_allgather_start_f64(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                                     send.numel(), int(slot), int(then_wait)), "ibs_comm_allgather_start_f64")

    def comm_wait(self, slot=-1):
        """order the context's stream after the overlapped gather of `slot` (default: all pending); the host does not block"""
        check(self._lib.ibs_comm_wait(self._h, int(slot)), "ibs_comm_wait")

    def comm_destroy(self):
        self._comm_world = 0            # gathers go back to torch.distributed
        check(self._lib.ibs_comm_destroy(self._h), "ibs_comm_destroy")

    def last_launch(self):
        """(kernel name as rocprofv3 prints it, waves of the launch) of the solver / geometry kernel this thread launched last"""
        buf = C.create_string_buffer(96)
        nb, nt = _lib._I64(0), _lib._I32(0)
        self._lib.ibs_last_launch(buf, 96, C.byref(nb), C.byref(nt))
        return buf.value.decode(), int(nb.value) * int(nt.value) // 64

    def set_option(self, name, value):
        """diagnostic override of a dispatch heuristic of this context (include/ibs.h: ibs_set_option);
        value None = back to the context's default"""
        check(self._lib.ibs_set_option(self._h, name.encode(), float("nan") if value is None else float(value)),
              "ibs_set_option")

    def reset_options(self):
        self.set_option("all", None)

    # ---- raw (g, c, f) systems --------------------------------------------------------------
    def solve_gcf(self, h, g, c, f, want_X=False, want_info=False, dtype=np.float64, gh=None, want_gam=True):
        """g, c, f: (n_sys, N).  Returns dict(lam, gam[, X, dX][, info]).  gh (n_sys, N) optional half-grid g
        (first N-1 columns used): see ibs_solve_gcfh_f64.  float32: with want_gam / want_X the systems are widened to
        FP64 inside the solver (FP32 in HBM only); want_gam=False and no X = the all-FP32 kernel, eigenvalues only."""
        ar = _Args(dtype)
        n_sys, N = g.shape
        pg, pc, pf = ar.inp(g), ar.inp(c), ar.inp(f)
        ref = g if ar.mem == MEM_DEVICE else None
        if ref is not None:
            self._stream_from_torch(ref)
        lam, plam = ar.out((n_sys,), ref)
        gam, pgam = ar.out((n_sys,), ref, want=want_gam)
        X, pX = ar.out((n_sys, N), ref, want=want_X)
        dX, pdX = ar.out((n_sys, N), ref, want=want_X)
        info, pinfo = ar.out((n_sys,), ref, dtype=np.int32, want=want_info)
        hh = float(h)
        if gh is not None:
            pgh = ar.inp(gh)
            rc = check(self._lib.ibs_solve_gcfh_f64(self._h, n_sys, N, hh, pg, pgh, pc, pf, N, plam, pgam, pX, pdX, pinfo,
                                                    ar.mem), "ibs_solve_gcfh_f64")
        else:
            fn = self._lib.ibs_solve_gcf_f64 if np.dtype(dtype) == np.float64 else self._lib.ibs_solve_gcf_f32
            rc = check(fn(self._h, n_sys, N, hh, pg, pc, pf, N, plam, pgam, pX, pdX, pinfo, ar.mem), "ibs_solve_gcf")
        out = dict(lam=lam, gam=gam, nbad=rc)
        if want_X:
            out.update(X=X, dX=dX)
        if want_info:
            out.update(info=info)
        return out

    # ---- geometry x theta0 scan ---------------------------------------------------------------
    def gamma_scan(self, h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, dPdrho, theta0,
                   want_X=False, want_dtheta0=False, want_info=False, lam_guess=None, guess_width=None):
        """geometry arrays: (n_lines, N); dPdrho: (n_lines,); theta0: (n_theta0,).
        Returns dict(gam, lam[, X, dX][, dgam_dtheta0]) shaped (n_lines, n_theta0[, N])."""
        ar = _Args()
        n_lines, N = bmag.shape
        n_t0 = int(theta0.shape[0])
        ptrs = [ar.inp(a) for a in (bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22)]
        pdP, pt0 = ar.inp(dPdrho), ar.inp(theta0)
        ref = bmag if ar.mem == MEM_DEVICE else None
        if ref is not None:
            self._stream_from_torch(ref)
        gam, pgam = ar.out((n_lines, n_t0), ref)
        lam, plam = ar.out((n_lines, n_t0), ref)
        X, pX = ar.out((n_lines, n_t0, N), ref, want=want_X)
        dX, pdX = ar.out((n_lines, n_t0, N), ref, want=want_X)
        dth, pdth = ar.out((n_lines, n_t0), ref, want=want_dtheta0)
        info, pinfo = ar.out((n_lines, n_t0), ref, dtype=np.int32, want=want_info)
        if lam_guess is not None:
            pg = ar.inp(lam_guess)
            rc = check(self._lib.ibs_gamma_scan_warm_f64(self._h, n_lines, n_t0, N, float(h), *ptrs, N, pdP, pt0, pg,
                                                         float(guess_width), pgam, plam, pX, pdX, pdth, pinfo, ar.mem),
                       "ibs_gamma_scan_warm_f64")
        else:
            rc = check(self._lib.ibs_gamma_scan_f64(self._h, n_lines, n_t0, N, float(h), *ptrs, N, pdP, pt0,
                                                    pgam, plam, pX, pdX, pdth, pinfo, ar.mem), "ibs_gamma_scan_f64")
        out = dict(gam=gam, lam=lam, nbad=rc)
        if want_X:
            out.update(X=X, dX=dX)
        if want_dtheta0:
            out.update(dgam_dtheta0=dth)
        if want_info:
            out.update(info=info)
        return out

    def gamma_scan_argmax(self, h, geo7, dPdrho, theta0, n_surf):
        """coarse scan + per-surface first maximum in ONE C call on device tensors (ibs_gamma_scan_argmax_f64; replaces
        ball_scan.py:248-295 for all surfaces at once).  geo7: seven (n_lines, N) tensors, lines surface-major.
        Returns dict(gam, lam (n_lines, n_theta0), pack (n_surf, 2) = (max, first row-major index), info)."""
        import torch
        ar = _Args()
        n_lines, N = geo7[0].shape
        n_t0 = int(theta0.shape[0])
        ptrs = [ar.inp(a) for a in geo7]
        pdP, pt0 = ar.inp(dPdrho), ar.inp(theta0)
        if ar.mem != MEM_DEVICE:
            raise IbsError("gamma_scan_argmax takes device tensors")
        ref = geo7[0]
        self._stream_from_torch(ref)
        gam, pgam = ar.out((n_lines, n_t0), ref)
        lam, plam = ar.out((n_lines, n_t0), ref)
        pack, ppack = ar.out((n_surf, 2), ref)
        info, pinfo = ar.out((n_lines, n_t0), ref, dtype=np.int32)
        check(self._lib.ibs_gamma_scan_argmax_f64(self._h, n_lines, n_t0, N, float(h), *ptrs, N, pdP, pt0, int(n_surf),
                                                  pgam, plam, ppack, pinfo), "ibs_gamma_scan_argmax_f64")
        return dict(gam=gam, lam=lam, pack=pack, info=info)

    def gamma_points(self, h, bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, dPdrho, theta0,
                     want_X=False, want_dtheta0=False, want_info=False):
        """one (line, theta0) pair per point (ibs_gamma_points_f64: the final solve of ball_scan.py:322-339 for many
        surfaces at once).  geometry arrays (n_pts, N); dPdrho, theta0 (n_pts,).  Returns dict(gam, lam[, X, dX][, ...])."""
        ar = _Args()
        n_pts, N = bmag.shape
        ptrs = [ar.inp(a) for a in (bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22)]
        pdP, pt0 = ar.inp(dPdrho), ar.inp(theta0)
        ref = bmag if ar.mem == MEM_DEVICE else None
        if ref is not None:
            self._stream_from_torch(ref)
        gam, pgam = ar.out((n_pts,), ref)
        lam, plam = ar.out((n_pts,), ref)
        X, pX = ar.out((n_pts, N), ref, want=want_X)
        dX, pdX = ar.out((n_pts, N), ref, want=want_X)
        dth, pdth = ar.out((n_pts,), ref, want=want_dtheta0)
        info, pinfo = ar.out((n_pts,), ref, dtype=np.int32, want=want_info)
        rc = check(self._lib.ibs_gamma_points_f64(self._h, n_pts, N, float(h), *ptrs, N, pdP, pt0, pgam, plam, pX, pdX,
                                                  pdth, pinfo, ar.mem), "ibs_gamma_points_f64")
        out = dict(gam=gam, lam=lam, nbad=rc)
        if want_X:
            out.update(X=X, dX=dX)
        if want_dtheta0:
            out.update(dgam_dtheta0=dth)
        if want_info:
            out.update(info=info)
        return out

    def scan_starts(self, alpha_scan, theta0_scan, pack, n_bad):
        """start points (n_surf, 2) = (alpha, theta0) of the refinement from the per-surface maxima `pack`, on the device
        (ibs_scan_starts_f64; ball_scan.py:279-295).  All arguments device tensors; n_bad: int32 tensor of one element that
        accumulates the number of surfaces whose maximum is not finite."""
        import torch
        n_surf = pack.shape[0]
        start = torch.empty((n_surf, 2), dtype=torch.float64, device=pack.device)
        self._stream_from_torch(pack)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self._lib.ibs_scan_starts_f64(self._h, n_surf, alpha_scan.shape[0], theta0_scan.shape[0], p(alpha_scan),
                                            p(theta0_scan), p(pack), p(start), C.c_void_p(None), p(n_bad)), "ibs_scan_starts_f64")
        return start

    def obj_w_grad(self, h, geo, theta0, del_alpha=0.004, want_info=False):
        """geo: (n_pts, 3, 8, N) -- lines (alpha-d/2, alpha, alpha+d/2) x (bmag, gradpar, cvdrift, cvdrift0,
        gds2, gds21, gds22, gbdrift); theta0: (n_pts,).  Returns (val (n_pts,), jac (n_pts, 2)) with the
        sign convention of utils.py:1728: val = -gam, jac = (-dgam/dalpha, -dgam/dtheta0)."""
        ar = _Args()
        n_pts, three, eight, N = geo.shape
        if three != 3 or eight != 8:
            raise IbsError("geo must be (n_pts, 3, 8, N)")
        pg, pt = ar.inp(geo), ar.inp(theta0)
        ref = geo if ar.mem == MEM_DEVICE else None
        if ref is not None:
            self._stream_from_torch(ref)
        val, pval = ar.out((n_pts,), ref)
        jac, pjac = ar.out((n_pts, 2), ref)
        info, pinfo = ar.out((n_pts,), ref, dtype=np.int32, want=want_info)
        check(self._lib.ibs_obj_w_grad_f64(self._h, n_pts, N, float(h), pg, N, pt, float(del_alpha), pval, pjac,
                                           pinfo, ar.mem), "ibs_obj_w_grad_f64")
        return (val, jac, info) if want_info else (val, jac)

    def fieldline_geometry(self, tables, line_surf, line_alpha, theta, device=None, use_rows=True):
        """geometry of the field lines (tables.s[line_surf[i]], line_alpha[i]) on the grid theta (row F1).
        Returns dict(geo=(8, n_lines, N), dPdrho=(n_lines,)); geo[0..6] + dPdrho feed gamma_scan directly.
        device=None: numpy in / numpy out (staged);  device=torch.device(...): results stay in HBM."""
        n_lines = len(line_surf)
        N = len(theta)
        resident = device is not None and all(_is_torch(a) for a in (line_surf, line_alpha, theta))
        if not resident:
            ls = np.ascontiguousarray(line_surf, dtype=np.int32)
            la = np.ascontiguousarray(line_alpha, dtype=np.float64)
            th = np.ascontiguousarray(theta, dtype=np.float64)
            if ls.size and (ls.min() < 0 or ls.max() >= len(tables.s)):
                raise IbsError("line_surf out of range")
        host = [tables.xm, tables.xn, tables.xm_nyq, tables.xn_nyq, tables.tab_mn, tables.tab_nyq, tables.scal]
        rows = [tables.rows_mn, tables.rows_nyq] if use_rows else [np.zeros((0, 2), np.int32)] * 2
        if device is None:
            geo = np.empty((8, n_lines, N)); dP = np.empty(n_lines)
            p = lambda a: C.c_void_p(a.ctypes.data)
            check(self._lib.ibs_fieldline_geometry_f64(self._h, len(tables.s), len(tables.xm), len(tables.xm_nyq),
                                                       *[p(a) for a in host], n_lines, p(ls), p(la), N, p(th), N, p(geo),
                                                       p(dP), len(rows[0]), p(rows[0]), len(rows[1]), p(rows[1]),
                                                       float(tables.dn_mn), float(tables.dn_nyq), MEM_HOST), "ibs_fieldline_geometry_f64")
            return dict(geo=geo, dPdrho=dP)
        import torch
        # tables are uploaded once and stay resident; the device copies live ON the tables object (a cache keyed by
        # id(tables) would hand a later object that re-uses the id the previous object's tables)
        allc = self._device_tables(tables, device)
        dev = allc[:7]
        d_rows = allc[7:]
        if resident:      # index / angle / grid tensors already in HBM (int32, float64, float64): no upload, no host check
            #               (the geometry kernel clamps the surface index itself)
            d_ls, d_la, d_th = line_surf.to(torch.int32).contiguous(), line_alpha.to(torch.float64).contiguous(), theta.to(torch.float64).contiguous()
        else:
            d_ls, d_la, d_th = (torch.from_numpy(a).to(device) for a in (ls, la, th))
        geo = torch.empty((8, n_lines, N), dtype=torch.float64, device=device)
        dP = torch.empty((n_lines,), dtype=torch.float64, device=device)
        self._stream_from_torch(geo)
        p = lambda t: C.c_void_p(t.data_ptr())
        nr = (len(tables.rows_mn), len(tables.rows_nyq)) if use_rows else (0, 0)
        check(self._lib.ibs_fieldline_geometry_f64(self._h, len(tables.s), len(tables.xm), len(tables.xm_nyq),
                                                   *[p(t) for t in dev], n_lines, p(d_ls), p(d_la), N, p(d_th), N, p(geo),
                                                   p(dP), nr[0], p(d_rows[0]), nr[1], p(d_rows[1]), float(tables.dn_mn), float(tables.dn_nyq),
                                                   MEM_DEVICE),
              "ibs_fieldline_geometry_f64")
        self._keep = (d_ls, d_la, d_th)
        return dict(geo=geo, dPdrho=dP)

    def _device_tables(self, tables, device):
        """the surface/mode/row tables of `tables` resident on `device` (uploaded once, cached on the object)"""
        import torch
        cache = tables.__dict__.setdefault("_device_copies", {})
        key = _device_key(device)
        if key not in cache:
            host = [tables.xm, tables.xn, tables.xm_nyq, tables.xn_nyq, tables.tab_mn, tables.tab_nyq, tables.scal]
            cache[key] = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in host + [tables.rows_mn, tables.rows_nyq]]
            # (a caching allocator may hand these rows the addresses of a freed table set: the library's verdict on device-resident
            #  rows is keyed by address and size)
            self.set_option("forget_rows", 1)
        return cache[key]

    def upload_tables_rows(self, tables, device, r0, r1):
        """(re)upload the surfaces r0 .. r1 - 1 of `tables` into its resident device copies (allocated on first use);
        asynchronous on the current torch stream when the host arrays are page-locked (SurfaceTables.frame(pinned=True)).
        For callers that fill a frame piece by piece while earlier pieces are already being worked on."""
        import torch
        cache = tables.__dict__.setdefault("_device_copies", {})
        key = _device_key(device)
        if key not in cache:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            e = lambda a: torch.empty(a.shape, dtype=torch.float64, device=device)
            cache[key] = [t(tables.xm), t(tables.xn), t(tables.xm_nyq), t(tables.xn_nyq), e(tables.tab_mn), e(tables.tab_nyq),
                          e(tables.scal), t(tables.rows_mn), t(tables.rows_nyq)]
            self.set_option("forget_rows", 1)
        dev = cache[key]
        pinned = getattr(tables, "_pinned", {})
        for k, name in ((4, "tab_mn"), (5, "tab_nyq"), (6, "scal")):
            src = pinned[name] if name in pinned else torch.from_numpy(getattr(tables, name))
            dev[k][r0:r1].copy_(src[r0:r1], non_blocking=True)

    def refine(self, tables, pt_surf, starts, theta, del_alpha=0.004, maxiter=30, ftol=5.0e-11, gtol=2.0e-8, device=None):
        """maximise gam over (alpha, theta0) from starts (n, 2) on surfaces tables.s[pt_surf] -- the whole
        quasi-Newton loop runs on the device (ibs_refine_f64; replaces ball_scan.py:305-314).
        Returns (x_opt (n, 2), f_opt (n,) = -gam, n_evals (n,), rounds) as numpy."""
        ps = np.ascontiguousarray(pt_surf, dtype=np.int32)
        st = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 2)
        th = np.ascontiguousarray(theta, dtype=np.float64)
        n = len(ps)
        if st.shape[0] != n:
            raise IbsError("starts must be (len(pt_surf), 2)")
        if ps.size and (ps.min() < 0 or ps.max() >= len(tables.s)):
            raise IbsError("pt_surf out of range")
        nr = (len(tables.rows_mn), len(tables.rows_nyq))
        head = (self._h, len(tables.s), len(tables.xm), len(tables.xm_nyq))
        tail = (float(del_alpha), int(maxiter), float(ftol), float(gtol))
        if device is None:
            xo = np.empty((n, 2)); fo = np.empty(n); ne = np.zeros(n, dtype=np.int32)
            p = lambda a: C.c_void_p(a.ctypes.data)
            host = [tables.xm, tables.xn, tables.xm_nyq, tables.xn_nyq, tables.tab_mn, tables.tab_nyq, tables.scal]
            rounds = check(self._lib.ibs_refine_f64(*head, *[p(a) for a in host], nr[0], p(tables.rows_mn), nr[1],
                                                    p(tables.rows_nyq), float(tables.dn_mn), float(tables.dn_nyq), n, p(ps),
                                                    p(st), len(th), p(th), *tail, p(xo), p(fo), p(ne), MEM_HOST),
                           "ibs_refine_f64")
            return xo, fo, ne, rounds
        import torch
        dev = self._device_tables(tables, device)
        # the grid and the point -> surface map rarely change between calls: their device copies are kept on the tables
        # object, keyed by content; the start points go up in one copy, the results come back in one
        cache = tables.__dict__.setdefault("_refine_inputs", {})
        key = (_device_key(device), th.tobytes(), ps.tobytes())
        if key not in cache:
            cache.clear()
            cache[key] = (torch.from_numpy(ps).to(device), torch.from_numpy(th).to(device))
        d_ps, d_th = cache[key]
        d_st = torch.from_numpy(st).to(device)
        out = torch.empty((4 * n,), dtype=torch.float64, device=device)     # x_opt (2n) | f_opt (n) | n_evals (n int32 in n/2.. words)
        xo, fo = out[:2 * n].view(n, 2), out[2 * n:3 * n]
        ne = out[3 * n:].view(torch.int32)[:n]
        self._stream_from_torch(out)
        p = lambda t: C.c_void_p(t.data_ptr())
        rounds = check(self._lib.ibs_refine_f64(*head, *[p(t) for t in dev[:7]], nr[0], p(dev[7]), nr[1], p(dev[8]),
                                                float(tables.dn_mn), float(tables.dn_nyq), n, p(d_ps), p(d_st), len(th),
                                                p(d_th), *tail, p(xo), p(fo), p(ne), MEM_DEVICE), "ibs_refine_f64")
        h = out.cpu()
        return (h[:2 * n].view(n, 2).numpy(), h[2 * n:3 * n].numpy(), h[3 * n:].view(torch.int32)[:n].numpy().copy(), rounds)

    def refine_device(self, tables, d_pt_surf, d_start, d_theta, del_alpha=0.004, maxiter=30, ftol=5.0e-11, gtol=2.0e-8):
        """refine() on device tensors, results left in HBM: d_pt_surf (n,) int32, d_start (n, 2), d_theta (N,) float64.
        Returns (x_opt (n, 2), f_opt (n,) = -gam, n_evals (n,) int32, rounds); the tensors are stream-ordered on the
        current torch stream (include/ibs.h: ibs_refine_f64 with device pointers)."""
        import torch
        device = d_start.device
        n = int(d_pt_surf.shape[0])
        dev = self._device_tables(tables, device)
        nr = (len(tables.rows_mn), len(tables.rows_nyq))
        out = torch.empty((4 * n,), dtype=torch.float64, device=device)     # x_opt (2n) | f_opt (n) | n_evals (n int32 in n/2.. words)
        xo, fo = out[:2 * n].view(n, 2), out[2 * n:3 * n]
        ne = out[3 * n:].view(torch.int32)[:n]
        self._stream_from_torch(out)
        p = lambda t: C.c_void_p(t.data_ptr())
        rounds = check(self._lib.ibs_refine_f64(self._h, len(tables.s), len(tables.xm), len(tables.xm_nyq),
                                                *[p(t) for t in dev[:7]], nr[0], p(dev[7]), nr[1], p(dev[8]),
                                                float(tables.dn_mn), float(tables.dn_nyq), n, p(d_pt_surf), p(d_start),
                                                int(d_theta.shape[0]), p(d_theta), float(del_alpha), int(maxiter), float(ftol),
                                                float(gtol), p(xo), p(fo), p(ne), MEM_DEVICE), "ibs_refine_f64")
        self._keep_refine = (d_pt_surf, d_start, d_theta, out)
        return xo, fo, ne, rounds

    def refine_stats(self):
        """(evaluations, forward sweeps, rounds needed, rounds enqueued) of the last refine() call of this context"""
        out = np.zeros(4, dtype=np.int64)
        check(self._lib.ibs_refine_stats(self._h, C.c_void_p(out.ctypes.data)), "ibs_refine_stats")
        return tuple(int(v) for v in out)

    def hf_grad(self, X, dX, f, g_p, c_p, f_p, gam):
        """Hellmann-Feynman d(gam)/dp for caller-built tangents (utils.py:1676-1680, 1721-1725).
        X, dX, f, g_p, c_p, f_p: (n_sys, N); gam: (n_sys,) -> jac (n_sys,)"""
        ar = _Args()
        n_sys, N = X.shape
        ptrs = [ar.inp(a) for a in (X, dX, f, g_p, c_p, f_p)]
        pgam = ar.inp(gam)
        ref = X if ar.mem == MEM_DEVICE else None
        if ref is not None:
            self._stream_from_torch(ref)
        jac, pjac = ar.out((n_sys,), ref)
        check(self._lib.ibs_hf_grad_f64(self._h, n_sys, N, *ptrs, N, pgam, pjac, ar.mem), "ibs_hf_grad_f64")
        return jac

    def sturm_count(self, h, g, c, f, shift, exact=False):
        """eigenvalues of (T, F) above shift[i] per system (ibs_sturm_count_f64).  exact=True: the division-form kernel (lanes as
        systems: a few eps ||A||, any N, ~5 TB/s in big batches) instead of the prefix-product sweep (N <= 2050: ~6 TB/s, exact for
        ~N eps ||A|| on smooth and up to ~N^2 eps ||A|| on iid-random coefficients: within that distance of an eigenvalue it can be
        off by one).  Without it the library picks by size (include/ibs.h: ibs_sturm_count_f64)."""
        if exact:
            self.set_option("sturm_form", 2)
            try:
                return self.sturm_count(h, g, c, f, shift)
            finally:
                self.set_option("sturm_form", None)
        ar = _Args()
        n_sys, N = g.shape
        pg, pc, pf, ps = ar.inp(g), ar.inp(c), ar.inp(f), ar.inp(shift)
        ref = g if ar.mem == MEM_DEVICE else None
        if ref is not None:
            self._stream_from_torch(ref)
        cnt, pcnt = ar.out((n_sys,), ref, dtype=np.int32)
        check(self._lib.ibs_sturm_count_f64(self._h, n_sys, N, float(h), pg, pc, pf, N, ps, pcnt, ar.mem),
              "ibs_sturm_count_f64")
        return cnt

    def surface_argmax(self, gam):
        """gam: (n_surf, n_per_surf) -> (idx int32 (n_surf,), val (n_surf,)); first index on ties (ball_scan.py:283-288)."""
        ar = _Args()
        n_surf, n_per = gam.shape
        pg = ar.inp(gam)
        ref = gam if ar.mem == MEM_DEVICE else None
        if ref is not None:
            self._stream_from_torch(ref)
        idx, pidx = ar.out((n_surf,), ref, dtype=np.int32)
        val, pval = ar.out((n_surf,), ref)
        check(self._lib.ibs_surface_argmax_f64(self._h, n_surf, n_per, pg, pidx, pval, ar.mem), "ibs_surface_argmax_f64")
        return idx, val


_DEFAULT = {}


def default_context(device=0):
    if device not in _DEFAULT:
        _DEFAULT[device] = Context(device)
    return _DEFAULT[device]


class ScanPlan:
    """Pre-marshalled geometry x theta0 scan + per-surface argmax on device-resident tensors: one
    optimizer iteration's worth of work is two kernel launches and no allocation.
    (Counterpart of the per-surface body of ball_scan.py:248-295.)

    geometry tensors: (n_lines, N) torch.cuda float64, lines ordered surface-major
    (n_lines = n_surf * n_alpha).  Results live in .gam (n_lines, n_theta0), .lam, .best_val (n_surf,),
    .best_idx (n_surf,) -- index into the flattened (alpha, theta0) table of the surface; both are views of
    .pack (n_surf, 2) float64, the buffer the per-surface all-gather sends."""

    def __init__(self, ctx, h, geo7, dPdrho, theta0, n_surf, want_dtheta0=False, n_pack=2):
        import torch
        self.ctx = ctx
        self.lib = ctx._lib
        t64 = torch.float64
        self.geo = [g.to(t64).contiguous() for g in geo7]
        dev = self.geo[0].device
        self.dP = dPdrho.to(t64).contiguous()
        self.t0 = theta0.to(t64).contiguous()
        n_lines, N = self.geo[0].shape
        n_t0 = self.t0.shape[0]
        if n_lines % n_surf:
            raise IbsError("n_lines=%d is not a multiple of n_surf=%d" % (n_lines, n_surf))
        self.n_lines, self.N, self.n_t0, self.n_surf = n_lines, N, n_t0, n_surf
        self.gam = torch.empty((n_lines, n_t0), dtype=t64, device=dev)
        self.lam = torch.empty((n_lines, n_t0), dtype=t64, device=dev)
        self.dth0 = torch.empty((n_lines, n_t0), dtype=t64, device=dev) if want_dtheta0 else None
        self.info = torch.empty((n_lines, n_t0), dtype=torch.int32, device=dev)
        # (lam_max, flat index) per surface; n_pack buffers so that the all-gather of one step can still be reading its
        # buffer while later steps' argmax write the others (argmax(slot))
        self.packs = [torch.empty((n_surf, 2), dtype=t64, device=dev) for _ in range(max(1, int(n_pack)))]
        self.pack = self.packs[0]
        self.best_val = self.pack[:, 0]
        self.best_idx = self.pack[:, 1]
        ctx._stream_from_torch(self.geo[0])
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)
        self._scan_args = (ctx._h, n_lines, n_t0, N, float(h), *[p(g) for g in self.geo], N, p(self.dP), p(self.t0),
                           p(self.gam), p(self.lam), C.c_void_p(None), C.c_void_p(None), p(self.dth0), p(self.info),
                           MEM_DEVICE)
        self._amax_args = [(ctx._h, n_surf, (n_lines // n_surf) * n_t0, p(self.gam), p(pk)) for pk in self.packs]
        self._fused_args = [(ctx._h, n_lines, n_t0, N, float(h), *[p(g) for g in self.geo], N, p(self.dP), p(self.t0), n_surf,
                             p(self.gam), p(self.lam), p(pk), p(self.info)) for pk in self.packs]

    def _use_current_stream(self):
        # the Context's stream is shared mutable state: launch on the caller's CURRENT torch stream, like every other
        # Context call (a plan used under torch.cuda.stream(s) must be ordered against that stream's tensors)
        import torch
        self.lib.ibs_set_stream(self.ctx._h, C.c_void_p(torch.cuda.current_stream(self.gam.device).cuda_stream))

    def scan(self):
        self._use_current_stream()
        rc = self.lib.ibs_gamma_scan_f64(*self._scan_args)
        if rc < 0:
            check(rc, "ibs_gamma_scan_f64")

    def argmax(self, slot=0):
        self._use_current_stream()
        rc = self.lib.ibs_surface_argmax_pack_f64(*self._amax_args[slot])
        if rc < 0:
            check(rc, "ibs_surface_argmax_pack_f64")

    def scan_argmax(self, slot=0):
        """scan + per-surface first maximum in one C call (ibs_gamma_scan_argmax_f64: one kernel launch for small batches)"""
        self._use_current_stream()
        rc = self.lib.ibs_gamma_scan_argmax_f64(*self._fused_args[slot])
        if rc < 0:
            check(rc, "ibs_gamma_scan_argmax_f64")

    def __call__(self):
        self.scan_argmax()
