"""On-disk configuration contract of the scan.

The reference configures a run through `params_dict.pkl` (written by create_dict.py:143-165, read by every worker:
ball_scan.py:26-28, sims_runner_NCSX.py:33-52) and hard-codes the rest (grid rule ball_scan.py:201-208, surfaces
ball_scan.py:197, thresholds sims_runner_*.py).  This module reads that file unchanged and derives from it what the
GPU scan and the objective need, so that a `sims_runner_*.py` set-up can be pointed at this package.
"""
import pickle

import numpy as np

from ._lib import IbsError

# keys create_dict.py:143-160 writes
PARAMS_KEYS = ("maxf", "eqbm_option", "pol_idxs", "tor_idxs", "iotaidxs", "isphifree", "totalndofs", "nsurfs",
               "abs_step", "rel_step", "username", "nprocspernode", "nodesperball", "totalnexecball", "njobsball",
               "nodespersimsopt")
# (gamma_thresh, prefac) of the ballooning term per eqbm_option: sims_runner_D3D.py:57-58, sims_runner_NCSX.py:56-57,
# sims_runner_HBERG.py:55-56
OBJECTIVE_CONSTANTS = {0: (-2.0e-4, 2.0), 1: (-2.0e-4, 50.0), 2: (-3.0e-4, 50.0)}


def theta_grid_for(mpol, ntor, theta_fac=4):
    """the theta_PEST grid of ball_scan.py:201-209 for an equilibrium with (mpol, ntor):
    ntheta = 2 mpol theta_fac + 1 (ntor = 0) or 2 mpol ntor theta_fac + 1, on [-theta_fac pi, theta_fac pi]
    -> 641 points for D3D (mpol = 80), 969 for NCSX / HBERG (mpol = ntor = 11)."""
    mpol, ntor, theta_fac = int(mpol), int(ntor), int(theta_fac)
    ntheta = 2 * mpol * theta_fac + 1 if ntor == 0 else 2 * mpol * ntor * theta_fac + 1
    return np.linspace(-theta_fac * np.pi, theta_fac * np.pi, ntheta)


def count_boundary_dofs(eqbm_option, pol_idxs, tor_idxs):
    """ndofsb of create_dict.py:47-55"""
    pol_idxs, tor_idxs = np.asarray(pol_idxs), np.asarray(tor_idxs)
    if int(eqbm_option) == 0:
        return int(len(pol_idxs) + len(tor_idxs))
    n = 0
    for i in range(len(pol_idxs)):
        n += 2 * int(tor_idxs[0]) if pol_idxs[i] == 0 else 2 * (2 * int(tor_idxs[i]) + 1)
    return n


class ScanConfig:
    """what one optimizer iteration's ballooning scan needs, derived from the reference's params dict"""

    def __init__(self, params):
        missing = [k for k in ("eqbm_option", "totalndofs", "nsurfs", "abs_step", "rel_step") if k not in params]
        if missing:
            raise IbsError("params dict lacks %s (create_dict.py:143-160)" % missing)
        self.params = dict(params)
        self.eqbm_option = int(params["eqbm_option"])
        self.totalndofs = int(params["totalndofs"])
        self.n_equilibria = self.totalndofs + 1                       # base + one per DOF (create_dict.py:86)
        self.nsurfs = int(params["nsurfs"])
        self.rho_arr = np.linspace(0.5, 0.95, self.nsurfs)            # ball_scan.py:197
        self.abs_step = float(params["abs_step"])                    # create_dict.py:67
        self.rel_step = float(params["rel_step"])                    # create_dict.py:70
        self.nalpha, self.ntheta0 = 24, 15                            # ball_scan.py:223-224
        self.gamma_thresh, self.prefac = OBJECTIVE_CONSTANTS.get(self.eqbm_option, OBJECTIVE_CONSTANTS[1])
        if "pol_idxs" in params and "tor_idxs" in params:
            nb = count_boundary_dofs(self.eqbm_option, params["pol_idxs"], params["tor_idxs"])
            ni = len(np.atleast_1d(params.get("iotaidxs", [])))
            if nb + ni + int(params.get("isphifree", 0)) != self.totalndofs:
                raise IbsError("totalndofs=%d is inconsistent with the DOF index sets (%d boundary + %d iota + %d)"
                               % (self.totalndofs, nb, ni, int(params.get("isphifree", 0))))

    def dof_step(self, x):
        """finite-difference step of a DOF of value x (ball_scan.py:129-139): absolute for |x| <= 1e-2, else relative"""
        x = np.asarray(x, dtype=np.float64)
        return np.where(np.abs(x) <= 1.0e-2, self.abs_step, self.rel_step * x)


def load_params_dict(path="params_dict.pkl"):
    """read the pickle create_dict.py:162-163 writes; returns ScanConfig"""
    with open(path, "rb") as f:
        d = pickle.load(f)
    if not isinstance(d, dict):
        raise IbsError("%s does not hold a dict" % path)
    return ScanConfig(d)


def create_history_placeholders(path, totalndofs):
    """the 0-d placeholder files arr_create2.py:87-97 creates for every equilibrium (base + DOFs);
    append_history() replaces the placeholder at iteration 0 (ball_scan.py:369-375)."""
    import os
    for i in range(int(totalndofs) + 1):
        for name in ("ball_gam", "ball_theta0", "ball_alpha"):
            np.save(os.path.join(path, "%s%d.npy" % (name, i)), np.empty([]))
