"""GPU: the refined per-surface growth rate (rows A7 / F2) against the reference's own L-BFGS-B runs.

Goldens (tests/golden/make_golden.py, reference imported in place, ARPACK converged = 'tight'):
  G5_scan_trace_tight.npz   s = 0.8483, N = 513
  G9_refine_traces.npz      s in {0.5, 0.6125, 0.7, 0.8483, 0.95}, N = 969 (the reference's NCSX grid), 'tight' and 'shipped'
each with the coarse table, the argmax, EVERY (alpha, theta0, val, jac) the reference's scipy L-BFGS-B evaluated
(ball_scan.py:305-314), x_opt and the final gam (ball_scan.py:322-339)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    import ibs_amd
    return ibs_amd.Context(0)


def _cases():
    g5 = np.load(os.path.join(G, "G5_scan_trace_tight.npz"))
    out = [dict(tag="G5T", s=float(g5["s"]), N=int(g5["N"]), table=g5["gam_table"], argmax=g5["argmax"], trace=g5["trace"],
                x_opt=g5["x_opt"], gam_opt=float(g5["gam_opt"]))]
    g9 = np.load(os.path.join(G, "G9_refine_traces.npz"))
    for k, s in enumerate(g9["s"]):
        out.append(dict(tag="G9_%d" % k, s=float(s), N=int(g9["N"]), table=g9["gam_table_tight_%d" % k],
                        argmax=g9["argmax_tight_%d" % k], trace=g9["trace_tight_%d" % k], x_opt=g9["x_opt_tight_%d" % k],
                        gam_opt=float(g9["gam_opt_tight_%d" % k]), gam_opt_shipped=float(g9["gam_opt_shipped_%d" % k])))
    return out


CASES = _cases()


def _scan(ctx, case):
    import ibs_amd
    import torch
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    tabs = ibs_amd.SurfaceTables.from_wout(wout, [case["s"]])
    th = ibs_amd.theta_grid(case["N"])
    return ibs_amd.BallooningScan(ctx, None, th, [case["s"]], tables=tabs, device=torch.device("cuda:0"))


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_coarse_table_and_objective_at_every_reference_trace_point(ctx, case):
    """coarse 24 x 15 table (1e-8, argmax exact) and k_obj_w_grad at EVERY point the reference's L-BFGS-B visited:
    val 1e-10, jac 1e-9 (geometry produced on the device from the wout tables)"""
    import ibs_amd
    scan = _scan(ctx, case)
    tab = scan.coarse()[0]
    assert np.abs(tab - case["table"]).max() < 1e-8
    ij = ibs_amd.pick_start(tab, scan.alpha_scan, scan.theta0_scan)[3]
    assert ij == tuple(int(v) for v in case["argmax"])
    tr = case["trace"]
    val, jac = scan.batched_obj_w_grad(np.zeros(len(tr), dtype=int), np.ascontiguousarray(tr[:, :2]))
    assert np.abs(val - tr[:, 2]).max() < 1e-10, np.abs(val - tr[:, 2]).max()
    assert np.abs(jac - tr[:, 3:5]).max() < 1e-9, np.abs(jac - tr[:, 3:5]).max()


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_host_driven_lbfgsb_reproduces_the_reference_trajectory(ctx, case):
    """scipy's L-BFGS-B on the host with every evaluation on the GPU (BallooningScan.refine, ball_scan.py:305-339):
    the reference's trajectory point for point, x_opt, and the refined gam to 1e-8"""
    scan = _scan(ctx, case)
    tr = case["trace"]
    seen = []
    orig = scan.obj_w_grad

    def rec(x, s):
        v, j = orig(x, s)
        seen.append((x[0], x[1], v, j[0], j[1]))
        return v, j

    scan.obj_w_grad = rec
    t_opt, a_opt, gam_opt, res = scan.refine(case["s"], tr[0, 0], tr[0, 1])
    seen = np.array(seen)
    # the refined growth rate -- what ball_gam{dof}.npy stores -- to the north-star tolerance
    assert abs(gam_opt - case["gam_opt"]) < 1e-8, (gam_opt, case["gam_opt"])
    # The trajectory: a quasi-Newton step divides by differences of gradients, so the 1e-11 (absolute) differences
    # between our jac and the reference's (its alpha-tangent is a difference of two geometries over del_alpha = 0.004:
    # geometry agreement of 1e-12 becomes 1e-10 there) reappear as ~1e-7 in the next point; once the line search
    # runs on noise (the Hellmann-Feynman jac is not the derivative of val: SURVEY A6 note) even the reference's own
    # 'tight' and 'shipped' runs take different numbers of evaluations (G9: 32 vs 18).  Pinned: the first evaluations
    # point for point, and the end point to the flatness of the maximum.
    n = min(len(seen), len(tr), 5)
    assert np.abs(seen[:n, :2] - tr[:n, :2]).max() < 5e-7, np.abs(seen[:n, :2] - tr[:n, :2]).max(axis=1)
    assert np.abs(seen[:n, 2] - tr[:n, 2]).max() < 1e-9
    assert abs(a_opt - case["x_opt"][0]) < 2e-4 and abs(t_opt - case["x_opt"][1]) < 2e-3


def test_device_lbfgsb_reaches_the_reference_optimum_on_every_golden_surface(ctx):
    """ibs_refine_f64 (one L-BFGS-B state machine per surface ON the device, csrc/ibs_lbfgsb2.hpp) through
    BallooningScan.run(): the five G9 surfaces in ONE batch on the reference's N = 969 grid, and G5's surface at
    N = 513 -- (theta0*, alpha*, gam) against the reference's refined values; gam to 1e-8 (north star)."""
    import ibs_amd
    import torch
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    g9 = [c for c in CASES if c["tag"].startswith("G9")]
    svals = np.array([c["s"] for c in g9])
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    scan = ibs_amd.BallooningScan(ctx, None, ibs_amd.theta_grid(969), svals, tables=tabs, device=torch.device("cuda:0"))
    t0, al, gam = scan.run()
    for k, c in enumerate(g9):
        assert abs(gam[k] - c["gam_opt"]) < 1e-8, (k, gam[k], c["gam_opt"])
        assert abs(gam[k] - c["gam_opt_shipped"]) < 1e-8          # ... and of the reference as shipped (ARPACK tol 5e-7)
        assert abs(al[k] - c["x_opt"][0]) < 2e-4 and abs(t0[k] - c["x_opt"][1]) < 2e-3
    # the optimizer's own evaluations against the reference's trace, point for point while both run (first 5)
    starts = np.array([c["trace"][0, :2] for c in g9])
    xo, fo, ne = scan.refine_device(starts)
    xh, fh, rounds = scan.refine_batched(starts)
    assert np.abs(fo - fh).max() < 1e-10
    c5 = CASES[0]
    scan5 = _scan(ctx, c5)
    t0, al, gam = scan5.run()
    assert abs(gam[0] - c5["gam_opt"]) < 1e-8, (gam[0], c5["gam_opt"])
