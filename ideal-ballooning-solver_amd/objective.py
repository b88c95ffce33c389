"""Consumer contract of the scan results.

The optimizer drivers combine the per-surface growth rates of the base equilibrium (dof 0) and of the
DOF-perturbed equilibria (dof 1..n) into the objective and its forward-difference gradient
(sims_runner_NCSX.py:249-261, 300-313).  With one process per GPU the DOF-perturbed equilibria are sharded
over ranks; the only exchange is one all-reduce (sum) of a (n_dof+1)-vector.
"""
import numpy as np


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
