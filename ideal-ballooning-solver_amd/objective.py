"""Consumer contract of the scan results.

The optimizer drivers combine the per-surface growth rates of the base equilibrium (dof 0) and of the
DOF-perturbed equilibria (dof 1..n) into the objective and its forward-difference gradient
(sims_runner_NCSX.py:249-261, 300-313).  With one process per GPU the DOF-perturbed equilibria are sharded
over ranks; the only exchange is one gather of the per-surface rows (AdjointStep) or one all-reduce (sum) of a
(n_dof+1)-vector (allreduce_dof_vector).
"""
import numpy as np

from ._lib import IbsError


def ballooning_objective(f_other, gam, gamma_thresh=-2.0e-4, prefac=50.0):
    """f0 = f_other + prefac * sum_s max(gam_s - gamma_thresh, 0)   (sims_runner_NCSX.py:254-257, 311-313;
    thresholds/weights: sims_runner_NCSX.py:56-57, sims_runner_D3D.py:57-58, sims_runner_HBERG.py:55-56).
    gam: (..., nsurfs).  fobj returns sqrt(f0) (sims_runner_NCSX.py:318)."""
    gam = np.asarray(gam, dtype=np.float64)
    return np.asarray(f_other, dtype=np.float64) + prefac * np.sum(np.maximum(gam - gamma_thresh, 0.0), axis=-1)


def dof_fd_gradient(f0_arr, step_arr):
    """forward-difference gradient of sqrt(f0) over the DOFs (sims_runner_NCSX.py:258-261):
    df[i-1] = (f0_arr[i] - f0_arr[0]) / step_arr[i] * 0.5 / sqrt(f0_arr[0]),  i = 1..n_dof."""
    f0_arr = np.asarray(f0_arr, dtype=np.float64)
    step_arr = np.asarray(step_arr, dtype=np.float64)
    return (f0_arr[1:] - f0_arr[0]) / step_arr[1:] * 0.5 * 1 / np.sqrt(f0_arr[0])


def dof_steps(x0, isabs, abs_step=1.0e-3, rel_step=2.0e-3):
    """finite-difference step of every DOF (create_dict.py:67, 70; sims_runner_NCSX.py:190-196):
    abs_step where flagged absolute, else rel_step * x0.  Entry 0 (base equilibrium) is unused."""
    x0 = np.asarray(x0, dtype=np.float64)
    isabs = np.asarray(isabs)
    return np.concatenate([[1.0], np.where(isabs[1:] == 1, abs_step, rel_step * x0)])


def shard_dofs(n_dof_plus_1, rank, world):
    return list(range(rank, n_dof_plus_1, world))


def allreduce_dof_vector(local_values, owned, n_dof_plus_1, world, dist=None, device=None):
    """every rank contributes the entries of the DOF-perturbed equilibria it scanned; one all-reduce(sum)
    yields the full (n_dof+1,) vector everywhere (RCCL on GPUs, gloo in CPU tests)."""
    vec = np.zeros(n_dof_plus_1)
    vec[list(owned)] = np.asarray(local_values, dtype=np.float64)
    if world == 1:
        return vec
    import torch
    t = torch.from_numpy(vec)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def chunk_cuts(n, n_chunks, growth=1.0):
    """boundaries of the coarse runs of AdjointStep: n equilibria in at most n_chunks runs, run k + 1 holding `growth` times the
    equilibria of run k (rounded; empty runs dropped).  Returns the increasing list [0, ..., n]."""
    nch = max(1, min(int(n_chunks), int(n)))
    wts = np.cumsum([0.0] + [float(growth) ** k for k in range(nch)])
    return sorted(set(int(round(n * w / wts[-1])) for w in wts))


class AdjointStep:
    """The ballooning work of ONE optimizer iteration for all equilibria of the step at once: the base equilibrium and one
    per boundary DOF (BASELINE configs[3]: 73 x 5 surfaces x 24 alpha x 15 theta0, N = 969).

    Upstream this is ball_submit.py:64-95 (one `srun ball_scan.py iter dof ngroups` per DOF-perturbed equilibrium, each
    running vmec_splines -> coarse scan -> argmax -> L-BFGS-B -> final solve, ball_scan.py:190-347) followed by
    sims_runner_NCSX.py:249-261 (objective of every equilibrium, forward-difference gradient over the DOFs).  Here the
    equilibria of a rank form ONE batch: the radial spline step for all of them (SurfaceTables.from_wouts), one geometry
    launch, one scan + argmax call, one refinement call, one final solve, one copy of the rows (BallooningScan.device_rows).
    With world > 1 the equilibria are dealt round-robin over the ranks (the reference's DP-1 level) and ONE
    all-gather of [n_eq_local][n_surf * 3] doubles assembles the table everywhere.

    SIMSOPT / VMEC stay outside: `wouts` are the wout tables of the (already converged) equilibria, `f_other` the
    non-ballooning part of each equilibrium's objective (Simsopt_runner.py -> f{dof}.npy), `steps` the DOF steps
    (dof_steps / ScanConfig.dof_step)."""

    def __init__(self, ctx, theta, svals, device, nalpha=24, ntheta0=15, del_alpha=0.004, gamma_thresh=-2.0e-4, prefac=50.0,
                 rank=0, world=1, dist=None, n_threads=0, n_chunks=4, gather_device=None, chunk_growth=1.0):
        self.ctx, self.device = ctx, device
        self.theta = np.asarray(theta, dtype=np.float64)
        self.svals = np.atleast_1d(np.asarray(svals, dtype=np.float64))       # ball_scan.py:197
        self.nalpha, self.ntheta0, self.del_alpha = int(nalpha), int(ntheta0), float(del_alpha)
        self.gamma_thresh, self.prefac = float(gamma_thresh), float(prefac)
        self.rank, self.world, self.dist, self.n_threads = int(rank), int(world), dist, int(n_threads)
        # the coarse part runs in n_chunks runs of equilibria: the host computes (ibs_surface_tables_f64) and uploads the tables
        # of run k + 1 while the GPU scans run k -- of the 2.2 ms the radial step of 73 equilibria takes, the first run's share
        # is all that stays exposed
        self.n_chunks = max(1, int(n_chunks))
        # (chunk_growth > 1: run k + 1 holds that many times the equilibria of run k -- a short first run exposes less host time;
        # the host must still finish run k + 1's tables while the GPU works on run k)
        self.chunk_growth = float(chunk_growth)
        self.gather_device = gather_device        # None: the rows are gathered where they are (RCCL); "cpu": through host copies (gloo)
        self._scan = None
        self._frame = None

    def _scan_for(self, tables, n_eq_local):
        from .scan import BallooningScan
        n = n_eq_local * len(self.svals)
        if self._scan is None or len(self._scan.rho_arr) != n:
            self._scan = BallooningScan(self.ctx, None, self.theta, np.tile(self.svals, n_eq_local), nalpha=self.nalpha,
                                        ntheta0=self.ntheta0, del_alpha=self.del_alpha, tables=tables, device=self.device,
                                        surf_index=np.arange(n))
        self._scan.tables = tables
        return self._scan

    def run(self, wouts, f_other, steps, refine=True, phases=None):
        """wouts: the n_eq = n_dof + 1 equilibria (entry 0 = base); f_other (n_eq,); steps (n_eq,) with entry 0 unused.
        Returns dict(gam, theta0, alpha: (n_eq, n_surf); f0: (n_eq,); fobj = sqrt(f0[0]) (sims_runner_NCSX.py:318);
        dfobj: (n_dof,) (sims_runner_NCSX.py:258-261)), identical on every rank."""
        import time
        import torch
        from .geometry import SurfaceTables
        from .scan import gather_rows_tensor
        n_eq, ns = len(wouts), len(self.svals)
        own = shard_dofs(n_eq, self.rank, self.world)
        err = None
        try:                               # row of an equilibrium: n_surf x (theta0*, alpha*, gam) + the count of what went wrong
            if own:
                mine = [wouts[q] for q in own]
                fr = self._frame
                if fr is None or fr.n_equilibria != len(own) or not np.array_equal(fr._svals, self.svals) or \
                        fr._ns != int(mine[0]["ns"]) or len(fr.xm) != len(mine[0]["xm"]) or len(fr.xm_nyq) != len(mine[0]["xm_nyq"]):
                    pinned = str(self.device).startswith("cuda")
                    fr = self._frame = SurfaceTables.frame(mine[0], self.svals, len(own), pinned=pinned)
                    self._scan = None
                scan = self._scan_for(fr, len(own))
                nch = min(self.n_chunks, len(own))
                cuts = chunk_cuts(len(own), nch, self.chunk_growth); nch = len(cuts) - 1

                def fill(c0, c1):          # (owned-surface range -> equilibria range: chunks are cut at equilibrium boundaries)
                    q0, q1 = c0 // ns, c1 // ns
                    r0, r1 = fr.fill(q0, mine[q0:q1], self.n_threads)
                    if pinned_upload:
                        self.ctx.upload_tables_rows(fr, self.device, r0, r1)
                pinned_upload = str(self.device).startswith("cuda")
                if not pinned_upload:      # (CPU stand-ins in the tests: no device copies to manage)
                    fr.fill(0, mine, self.n_threads)
                r, bad = scan.device_rows(refine, phases, chunks=[(cuts[k] * ns, cuts[k + 1] * ns) for k in range(nch)],
                                          fill=fill if pinned_upload else None)
                rows = torch.cat([r.reshape(len(own), 3 * ns), bad.reshape(1, 1).expand(len(own), 1)], dim=1)
            else:
                rows = torch.zeros((0, 3 * ns + 1), dtype=torch.float64, device=self.device)
        except Exception as e:             # (carried through the gather as NaN rows and raised on every rank afterwards)
            err = e
            rows = torch.full((len(own), 3 * ns + 1), float("nan"), dtype=torch.float64, device=self.device)
        t0 = time.perf_counter()
        if self.gather_device is not None and self.world > 1:
            rows = rows.to(self.gather_device)
        full = gather_rows_tensor(rows, n_eq, self.rank, self.world, self.dist, self.ctx if getattr(self.ctx, "_comm_world", 0) == self.world else None)
        host = full.cpu().numpy()                                          # the one copy (and synchronisation)
        if err is not None:
            raise err
        if not np.all(np.isfinite(host)) or np.any(host[:, -1] != 0):
            raise IbsError("the scan of %d equilibria was flagged or produced non-finite growth rates" %
                           int(np.sum(~np.isfinite(host).all(axis=1) | (host[:, -1] != 0))))
        tab = host[:, :-1].reshape(n_eq, ns, 3)
        gam = tab[:, :, 2]
        f0 = ballooning_objective(f_other, gam, self.gamma_thresh, self.prefac)        # sims_runner_NCSX.py:254-257
        out = dict(theta0=tab[:, :, 0], alpha=tab[:, :, 1], gam=gam, f0=f0, fobj=float(np.sqrt(f0[0])),
                   dfobj=dof_fd_gradient(f0, steps))                                   # sims_runner_NCSX.py:258-261
        if phases is not None:
            phases["gather_copy_objective_ms"] = (time.perf_counter() - t0) * 1e3
        return out
