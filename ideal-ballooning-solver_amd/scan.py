"""Per-equilibrium scan driver: the build's counterpart of ball_scan.py:190-386.

One process per GPU.  Surfaces are sharded over ranks (surface r goes to rank r % world, the
`mpi.group` <-> surface mapping of ball_scan.py:172, 251-252); every rank scans its surfaces on
its GPU, refines the per-surface maximum, and one all-gather (RCCL on GPUs, gloo in CPU tests)
replaces the three comm_lead.Gather calls of ball_scan.py:345-347.  No data-path collective
exists besides that gather.

Geometry is produced on the host by the caller (SIMSOPT/VMEC stay untouched, BASELINE north_star):
`fieldlines(s, alphas)` must return an array (len(alphas), 8, N) in the order
bmag, gradpar_theta_pest, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift -- e.g. a thin wrapper
around the reference's own utils.vmec_fieldlines (ball_scan.py:251-261).
"""
import numpy as np

from ._lib import IbsError

GEO_ORDER = ("bmag", "gradpar_theta_pest", "cvdrift", "cvdrift0", "gds2", "gds21", "gds22", "gbdrift")


def shard_surfaces(n_surf, rank, world):
    """indices of the surfaces owned by `rank` (round-robin)"""
    return list(range(rank, n_surf, world))


def pick_start(gam_table, alpha_scan, theta0_scan):
    """ball_scan.py:279-295: start point of the refinement from the coarse table (first maximum on ties;
    an all-zero table starts from (0, 0) with sigma0 = 0.05)."""
    if not np.all(np.isfinite(gam_table)):
        raise IbsError("coarse table holds %d non-finite growth rates (invalid geometry?)"
                       % int(np.sum(~np.isfinite(gam_table))))
    m = np.max(gam_table)
    if m == 0.0:
        return 0.0, 0.0, 0.05, None
    idx = np.where(gam_table == m)
    i, j = int(idx[0][0]), int(idx[1][0])
    return float(alpha_scan[i]), float(theta0_scan[j]), 1.3 * abs(float(gam_table[i, j])) + 0.05, (i, j)


def gather_rows_tensor(local, n_surf, rank, world, dist, ctx=None):
    """ONE all-gather of per-surface rows held as a torch tensor (device tensor: RCCL, in-stream; CPU tensor: gloo).
    ctx: a Context that holds a native communicator (Context.comm_init): the collective is then issued by the library
    itself (ibs_comm_allgather_f64) instead of torch.distributed.
    local: (n_local, k) rows of the surfaces shard_surfaces(n_surf, rank, world) lists.  Returns (n_surf, k) in surface
    order on every rank (replaces the three comm_lead.Gather of ball_scan.py:345-347)."""
    import torch
    k = local.shape[1]
    if world == 1:
        return local.clone()
    n_max = (n_surf + world - 1) // world
    pad = torch.full((n_max, k), float("nan"), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * n_max, k), dtype=local.dtype, device=local.device)
    if ctx is not None and getattr(ctx, "_comm_world", 0) == world and local.is_cuda:
        ctx.allgather(pad, out)
    else:
        dist.all_gather_into_tensor(out, pad)
    # row of surface j: rank j % world, slot j // world
    j = torch.arange(n_surf, device=local.device)
    return out[(j % world) * n_max + j // world]


def gather_surfaces(local, n_surf, rank, world, dist=None, device=None, ctx=None):
    """all-gather of per-surface rows.  local: (n_local, k) rows of the surfaces shard_surfaces() lists.
    Returns (n_surf, k) on every rank (replaces ball_scan.py:345-347)."""
    local = np.asarray(local, dtype=np.float64)
    if world == 1:
        return local.copy()
    import torch
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    return gather_rows_tensor(t, n_surf, rank, world, dist, ctx).cpu().numpy()


class BallooningScan:
    """Coarse (alpha, theta0) scan -> argmax -> L-BFGS-B refinement -> final solve, per surface."""

    def __init__(self, ctx, fieldlines, theta, rho_arr, nalpha=24, ntheta0=15, del_alpha=0.004,
                 rank=0, world=1, dist=None, gather_device=None, tables=None, device=None, surf_index=None):
        """fieldlines: host geometry callable (see module docstring), or None together with
        tables=SurfaceTables (row F1): then the geometry is produced on `device` by the HIP geometry kernel and consumed
        there -- coarse scan, per-surface maximum, start points, refinement and final solve all stay in HBM and ONE small
        copy returns the rows.  surf_index[k] = index of surface k (of rho_arr) in `tables`; default: the surface of
        tables.s nearest to rho_arr[k].  Table sets that hold several equilibria (SurfaceTables.from_wouts: s repeats
        per equilibrium) need the explicit index."""
        self.ctx = ctx
        self.tables = tables
        self.device = device
        self._resident = {}
        if tables is not None:
            fieldlines = self._device_fieldlines_host
        self.fieldlines = fieldlines
        self.theta = np.asarray(theta, dtype=np.float64)
        self.h = float((self.theta[-1] - self.theta[0]) / (len(self.theta) - 1))
        self.rho_arr = np.asarray(rho_arr, dtype=np.float64)               # ball_scan.py:197
        self.alpha_scan = np.linspace(0, np.pi, nalpha)                    # ball_scan.py:226
        self.theta0_scan = np.linspace(0.0, 0.5 * np.pi, ntheta0)          # ball_scan.py:225
        self.del_alpha = del_alpha
        self.rank, self.world, self.dist, self.gather_device = rank, world, dist, gather_device
        self._native_gather = world > 1 and getattr(ctx, "_comm_world", 0) == world
        self.own = shard_surfaces(len(self.rho_arr), rank, world)
        if tables is not None:
            if surf_index is None:
                surf_index = [int(np.argmin(np.abs(tables.s - r))) for r in self.rho_arr]
            self.surf_index = np.asarray(surf_index, dtype=np.int32)
            if self.surf_index.shape != self.rho_arr.shape or (len(self.surf_index) and (
                    self.surf_index.min() < 0 or self.surf_index.max() >= len(tables.s))):
                raise IbsError("surf_index must give one table index in [0, %d) per surface" % len(tables.s))

    def _own_surf(self):
        """table indices of the surfaces this rank owns"""
        return self.surf_index[np.asarray(self.own, dtype=np.int64)] if len(self.own) else np.zeros(0, dtype=np.int32)

    def _device_fieldlines_host(self, s, alphas):
        """geometry kernel behind the host-callable interface (used by the final solve / tests)"""
        js = int(np.argmin(np.abs(self.tables.s - s)))
        alphas = np.atleast_1d(np.asarray(alphas, dtype=np.float64))
        r = self.ctx.fieldline_geometry(self.tables, [js] * len(alphas), alphas, self.theta)
        return np.ascontiguousarray(np.transpose(r["geo"], (1, 0, 2)))

    # -- A5: coarse scan of the surfaces this rank owns, one launch
    def coarse(self):
        if self.tables is not None and self.device is not None and self.own:
            import torch
            na = len(self.alpha_scan)
            surf = np.repeat(self._own_surf(), na)
            r = self.ctx.fieldline_geometry(self.tables, surf, np.tile(self.alpha_scan, len(self.own)), self.theta,
                                            device=self.device)
            t0 = torch.from_numpy(self.theta0_scan).to(self.device)
            out = self.ctx.gamma_scan(self.h, *[r["geo"][k] for k in range(7)], r["dPdrho"], t0, want_info=True)
            # device-pointer calls are asynchronous and return no count of flagged systems: read the info words
            nbad = int(((out["info"] >> 16) != 0).sum().item())
            if nbad:
                raise IbsError("%d of %d coarse-scan solves were flagged (status word != 0: invalid data or iteration cap)"
                               % (nbad, out["info"].numel()))
            return out["gam"].cpu().numpy().reshape(len(self.own), na, len(self.theta0_scan))
        geos = [np.asarray(self.fieldlines(self.rho_arr[k], self.alpha_scan)) for k in self.own]
        if not geos:
            return np.zeros((0, len(self.alpha_scan), len(self.theta0_scan)))
        geo = np.concatenate(geos, axis=0)                                 # (n_own*nalpha, 8, N)
        dP = -0.5 * np.mean((geo[:, 2] - geo[:, 7]) * geo[:, 0] ** 2, axis=1)   # ball_scan.py:262
        r = self.ctx.gamma_scan(self.h, *[np.ascontiguousarray(geo[:, k]) for k in range(7)], dP, self.theta0_scan)
        if r.get("nbad", 0):
            raise IbsError("%d coarse-scan solves were flagged (status word != 0: invalid data or iteration cap)" % r["nbad"])
        return np.asarray(r["gam"]).reshape(len(self.own), len(self.alpha_scan), len(self.theta0_scan))

    # -- A6: objective with gradient at one point of one surface (utils.py:1632-1728)
    def obj_w_grad(self, x, s):
        a, t0 = float(x[0]), float(x[1])
        d = self.del_alpha
        geo = np.asarray(self.fieldlines(s, np.array([a - 0.5 * d, a, a + 0.5 * d])))
        val, jac = self.ctx.obj_w_grad(self.h, geo[None], np.array([t0]), d)
        return float(val[0]), np.asarray(jac[0], dtype=np.float64)

    # -- A7: refinement + final solve (ball_scan.py:305-339)
    def refine(self, s, a0, t0):
        from scipy.optimize import minimize
        res = minimize(self.obj_w_grad, x0=(a0, t0), args=(s,), jac=True,
                       bounds=((0.0, np.pi), (0.0, 0.5 * np.pi)),
                       options={"ftol": 5.0e-11, "gtol": 2.0e-08, "maxiter": 30})
        a, t = float(res.x[0]), float(res.x[1])
        geo = np.asarray(self.fieldlines(s, np.array([a])))[0]
        dP = -0.5 * np.mean((geo[2] - geo[7]) * geo[0] ** 2)
        r = self.ctx.gamma_scan(self.h, *[geo[k][None] for k in range(7)], np.array([dP]), np.array([t]))
        return t, a, float(np.asarray(r["gam"])[0, 0]), res

    # -- F2: all owned surfaces refined in lockstep; every evaluation of every surface is ONE batched launch
    def batched_obj_w_grad(self, surf_idx, X):
        """objective and gradient at X[k] = (alpha, theta0) of surface surf_idx[k] for all k at once
        (device geometry for the 3 n lines, then the fused obj_w_grad kernel).  Returns (val (n,), jac (n, 2))."""
        n = len(surf_idx)
        d = self.del_alpha
        al = np.stack([X[:, 0] - 0.5 * d, X[:, 0], X[:, 0] + 0.5 * d], axis=1).reshape(-1)
        r = self.ctx.fieldline_geometry(self.tables, np.repeat(surf_idx, 3), al, self.theta, device=self.device)
        N = len(self.theta)
        geo = r["geo"].view(8, n, 3, N).permute(1, 2, 0, 3).contiguous()
        import torch
        t0 = torch.from_numpy(np.ascontiguousarray(X[:, 1])).to(self.device)
        val, jac = self.ctx.obj_w_grad(self.h, geo, t0, d)
        return val.cpu().numpy(), jac.cpu().numpy()

    def refine_batched(self, starts, maxiter=30, ftol=5.0e-11, gtol=2.0e-8):
        """the per-surface L-BFGS-B of ball_scan.py:307-314 (same bounds, tolerances and iteration cap) for every owned
        surface at once, driven from the host: one optimizer state per surface (csrc/ibs_lbfgsb2.hpp through the C ABI
        ibs_lbfgsb2_*), and every round ONE batched geometry + objective launch for the surfaces still running.
        The host-driven form of refine_device(), which it is tested against.
        starts: (n, 2).  Returns (x_opt (n, 2), f_opt (n,) = -gam, rounds)."""
        import ctypes as C
        from . import _lib
        lib = _lib.lib()
        lo = np.array([0.0, 0.0]); hi = np.array([np.pi, 0.5 * np.pi])
        surf = self._own_surf()
        n = len(surf)
        p = lambda a: C.c_void_p(a.ctypes.data)
        x = np.clip(np.asarray(starts, dtype=np.float64).reshape(n, 2), lo, hi)
        states = [C.create_string_buffer(lib.ibs_lbfgsb2_state_bytes()) for _ in range(n)]
        for k in range(n):
            lib.ibs_lbfgsb2_init(states[k], p(x[k]), p(lo), p(hi), float(ftol), float(gtol), int(maxiter), 20)
        active = np.ones(n, dtype=bool)
        rounds = 0
        while active.any():
            idx = np.nonzero(active)[0]
            f, g = self.batched_obj_w_grad(surf[idx], x[idx])
            rounds += 1
            for q, k in enumerate(idx):
                gk = np.ascontiguousarray(g[q], dtype=np.float64)
                if not lib.ibs_lbfgsb2_step(states[k], float(f[q]), p(gk), p(x[k])):
                    active[k] = False
        fo = np.empty(n)
        for k in range(n):
            fk = C.c_double(0.0)
            lib.ibs_lbfgsb2_result(states[k], p(x[k]), C.byref(fk), None)
            fo[k] = fk.value
        return x, fo, rounds

    def refine_device(self, starts, maxiter=30, ftol=5.0e-11, gtol=2.0e-8):
        """the same maximisation with the L-BFGS-B state machines on the device as well (ibs_refine_f64): no host
        round trip per evaluation.  Returns (x_opt (n, 2), f_opt (n,) = -gam, evaluations per surface (n,)).
        f_opt is the objective at the optimizer's last accepted iterate (scipy's res.fun); run() re-evaluates gam at
        x_opt like ball_scan.py:322-339 does."""
        surf = self._own_surf()
        xo, fo, ne, _ = self.ctx.refine(self.tables, surf, np.asarray(starts, dtype=np.float64).reshape(len(surf), 2),
                                        self.theta, self.del_alpha, maxiter, ftol, gtol, device=self.device)
        return xo, fo, ne

    def final_solve_device(self, xo):
        """gam at the refined (alpha, theta0) of every owned surface: the final geometry + solve of ball_scan.py:322-339
        (the value the reference stores; the optimizer's own f is the value at its last ACCEPTED iterate, which after a
        collapsed line search is the same point, after a maxiter stop as well).  ONE field line per point
        (ibs_gamma_points_f64): no tangent lines, no gradient sums."""
        import torch
        xo = np.asarray(xo, dtype=np.float64).reshape(-1, 2)
        r = self.ctx.fieldline_geometry(self.tables, self._own_surf(), np.ascontiguousarray(xo[:, 0]), self.theta, device=self.device)
        t0 = torch.from_numpy(np.ascontiguousarray(xo[:, 1])).to(self.device)
        out = self.ctx.gamma_points(self.h, *[r["geo"][k] for k in range(7)], r["dPdrho"], t0)
        return out["gam"].cpu().numpy()

    # -- the whole per-surface worker of ball_scan.py:248-339 for the owned surfaces, resident in HBM
    def _resident_inputs(self):
        """device copies of what does not change between optimizer iterations: line -> (surface, alpha) tables of the coarse
        scan, the grids, the point -> surface map of the refinement"""
        import torch
        key = (tuple(self.own), str(self.device))
        if self._resident.get("key") != key:
            dev, na = self.device, len(self.alpha_scan)
            own = self._own_surf()
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            self._resident = dict(key=key, surf=t(np.repeat(own, na).astype(np.int32)), al=t(np.tile(self.alpha_scan, len(own))),
                                  th=t(self.theta), t0=t(self.theta0_scan), alpha=t(self.alpha_scan), pt_surf=t(own.astype(np.int32)),
                                  n_bad=torch.zeros(1, dtype=torch.int32, device=dev))
        return self._resident

    def device_rows(self, refine=True, phases=None, chunks=None, fill=None):
        """(theta0*, alpha*, gam) of the owned surfaces as an (n_own, 3) DEVICE tensor + a device scalar counting what went
        wrong (flagged solves, non-finite maxima): geometry -> coarse scan with the fused per-surface first maximum
        (ibs_gamma_scan_argmax_f64) -> start points on the device (ibs_scan_starts_f64) -> L-BFGS-B per surface on the device
        (ibs_refine_f64 on device pointers) -> final geometry + solve, one line per point (ibs_gamma_points_f64).  Nothing
        returns to the host in between.
        chunks: optional list of (c0, c1) ranges of owned surfaces: the coarse part (geometry, scan, starts) runs chunk by chunk,
        and fill(c0, c1) -- if given -- is called on the host before a chunk's launches (AdjointStep: the tables of the next
        equilibria are computed and uploaded while the GPU works on the previous ones).
        phases: optional dict filled with per-phase milliseconds (HIP events; adds one synchronisation at the end)."""
        import time
        import torch
        ctx, dev = self.ctx, self.device
        n = len(self.own)
        if n == 0:
            return torch.empty((0, 3), dtype=torch.float64, device=dev), torch.zeros((), dtype=torch.float64, device=dev)
        res = self._resident_inputs()
        na = len(self.alpha_scan)
        ev = {}

        def mark(name):
            if phases is not None:
                e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(name, []).append(e)
        res["n_bad"].zero_()
        chunks = chunks or [(0, n)]
        start = torch.empty((n, 2), dtype=torch.float64, device=dev)
        gmax = torch.empty((n,), dtype=torch.float64, device=dev)
        bad = res["n_bad"][0] * 0
        t_fill = 0.0
        for c0, c1 in chunks:
            if fill is not None:
                t0 = time.perf_counter(); fill(c0, c1); t_fill += time.perf_counter() - t0
            mark("g0")
            geo = ctx.fieldline_geometry(self.tables, res["surf"][c0 * na:c1 * na], res["al"][c0 * na:c1 * na], res["th"], device=dev)
            mark("g1")
            sc = ctx.gamma_scan_argmax(self.h, [geo["geo"][k] for k in range(7)], geo["dPdrho"], res["t0"], c1 - c0)
            st = ctx.scan_starts(res["alpha"], res["t0"], sc["pack"], res["n_bad"])
            mark("s1")
            if len(chunks) == 1:
                start, gmax = st, sc["pack"][:, 0]
            else:
                start[c0:c1] = st; gmax[c0:c1] = sc["pack"][:, 0]
            bad = bad + ((sc["info"] >> 16) != 0).sum()
        mark("r0")
        if refine:
            if len(self.theta) > 2050:
                # ibs_refine_f64 holds the register-resident evaluation kernel (N <= 2050): beyond, the same L-BFGS-B state machines
                # run on the host (ibs_lbfgsb2_*) and every round is ONE batched geometry + ibs_obj_w_grad_f64 launch for the
                # surfaces still running (refine_batched: the form refine_device is tested against)
                xh, fh, rounds = self.refine_batched(start.cpu().numpy())
                xo = torch.from_numpy(xh).to(dev); ne = None
            else:
                xo, fo, ne, rounds = ctx.refine_device(self.tables, res["pt_surf"], start, res["th"], self.del_alpha)
            mark("r1")
            xa, xt = xo[:, 0].contiguous(), xo[:, 1].contiguous()
            gf = ctx.fieldline_geometry(self.tables, res["pt_surf"], xa, res["th"], device=dev)
            fin = ctx.gamma_points(self.h, *[gf["geo"][k] for k in range(7)], gf["dPdrho"], xt, want_info=True)
            rows = torch.stack([xt, xa, fin["gam"]], dim=1)
            bad = bad + ((fin["info"] >> 16) != 0).sum()
            self.last_refine = dict(n_evals=ne, rounds=rounds)
        else:
            mark("r1")
            rows = torch.stack([start[:, 1], start[:, 0], gmax], dim=1)
        bad = bad + res["n_bad"][0]
        mark("f1")
        if phases is not None:
            torch.cuda.synchronize()
            span = lambda a, b: sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[b]))
            phases["geometry_ms"] = span("g0", "g1"); phases["scan_argmax_ms"] = span("g1", "s1")
            phases["refine_ms"] = span("r0", "r1"); phases["final_solve_ms"] = span("r1", "f1")
            if fill is not None:
                phases["host_tables_ms"] = t_fill * 1e3
                phases["coarse_chunks"] = len(chunks)
        return rows, bad.to(torch.float64)

    def local_rows(self, refine=True):
        """(theta0*, alpha*, gam) of the surfaces this rank owns, (n_own, 3): coarse scan -> argmax -> refinement -> final
        solve (ball_scan.py:248-339), no collective"""
        if self.tables is not None and self.device is not None:
            import torch
            rows, bad = self.device_rows(refine)
            host = torch.cat([rows.reshape(-1), bad.reshape(1)]).cpu().numpy()          # the one copy (and synchronisation)
            if host[-1] != 0 or not np.all(np.isfinite(host[:-1])):
                raise IbsError("%d solves of this rank's scan were flagged or produced non-finite growth rates (status word != 0: "
                               "invalid data or iteration cap)" % int(host[-1]))
            return host[:-1].reshape(len(self.own), 3)
        tabs = self.coarse()
        rows = []
        for k, tab in zip(self.own, tabs):
            a0, t0, sigma0, ij = pick_start(tab, self.alpha_scan, self.theta0_scan)
            if refine:
                t, a, gam, _ = self.refine(self.rho_arr[k], a0, t0)
            else:
                t, a, gam = t0, a0, float(np.max(tab))
            rows.append((t, a, gam))
        return np.array(rows, dtype=np.float64).reshape(len(self.own), 3)

    def run(self, refine=True):
        """returns (theta0_arr, alpha_arr, gam_arr), each (nsurfs,), identical on every rank.
        Every rank takes part in the ONE gather whatever happens on its own shard: a rank-local failure of ANY kind (flagged
        solves, non-finite tables, a geometry producer that raises, an out-of-memory error of the framework) travels through
        the collective as NaN rows and is raised on EVERY rank afterwards -- a rank that raised before the gather would
        leave the others waiting in it."""
        err = None
        try:
            local = self.local_rows(refine)
        except Exception as e:             # (re-raised after the gather, whatever it is)
            err = e
            local = np.full((len(self.own), 3), np.nan)
        # one decision for every gather of this object, the same on every rank: the library's own communicator when the
        # context holds one for this world (Context.comm_init is all-or-none over the ranks), else torch.distributed
        full = gather_surfaces(local, len(self.rho_arr), self.rank, self.world, self.dist, self.gather_device,
                               self.ctx if self._native_gather else None)
        if err is not None:
            raise err
        if self.world > 1 and not np.all(np.isfinite(full)):
            bad = sorted(set(int(j) % self.world for j in np.nonzero(~np.isfinite(full).all(axis=1))[0]))
            raise IbsError("surface rows of rank(s) %s are not finite: the scan failed there (see that rank's error)" % bad)
        return full[:, 0], full[:, 1], full[:, 2]


# -- A8 / F4: on-disk history contract of ball_scan.py:359-384 (consumed by sims_runner_*.py:198-199, 300-306)
def append_history(path, dof_idx, iter0, gam_arr, theta0_arr, alpha_arr):
    import os
    out = {}
    for name, row in (("ball_gam", gam_arr), ("ball_theta0", theta0_arr), ("ball_alpha", alpha_arr)):
        fn = os.path.join(path, "%s%d.npy" % (name, int(dof_idx)))
        old = np.load(fn, allow_pickle=True)
        if iter0 == 0:
            new = np.delete(np.append(old, row), 0)        # ball_scan.py:369-375: replace the placeholder
        else:
            new = np.vstack((old, row))                    # ball_scan.py:376-379
        np.save(fn, new)
        out[name] = new
    return out
