"""CPU: the oracle (oracle/ballooning_oracle.py) against the golden vectors captured from the
reference (tests/golden/make_golden.py).  This is what pins the oracle (task section 3)."""
import os

import numpy as np
import pytest

from oracle import ballooning_oracle as bo

G = os.path.join(os.path.dirname(__file__), "golden")


def test_G1_salpha_gam_and_eigenfunction():
    g1 = np.load(os.path.join(G, "G1_salpha.npz"))
    for (N, sh, al, t0), gref in zip(g1["params"], g1["gam"]):
        N = int(N)
        th = bo.theta_grid(N)
        g, c = bo.salpha_gc(th, sh, al, t0)
        gam, X, dX, gg, cc, ff = bo.gamma_ball_full(-1.0, th, np.ones(N), np.ones(N), c, g)
        assert abs(gam - gref) < 1e-10          # reference ARPACK noise on smooth systems ~1e-13
        key = "X_%d_%g_%g_%g" % (N, sh, al, t0)
        if key in g1:
            assert np.abs(X - g1[key]).max() < 1e-9
            assert np.abs(dX - g1["d" + key]).max() < 1e-8
    # values quoted in SURVEY.md section 8c
    th = bo.theta_grid(257)
    g, c = bo.salpha_gc(th, 1.0, 0.8, 0.0)
    assert abs(bo.gamma_ball_full(-1.0, th, np.ones(257), np.ones(257), c, g)[0] - 0.109331381940) < 1e-9


def test_G2_stability_booleans_exact():
    tab = np.load(os.path.join(G, "G2_salpha_stability.npz"))["table"]
    assert 0.2 < tab[:, 3].mean() < 0.8          # both outcomes are represented
    for sh, al, t0, a, b in tab:
        assert bo.salpha_unstable(sh, al, t0, 1601, 61) == int(a)
        assert bo.salpha_unstable(sh, al, t0, 401, 20) == int(b)


@pytest.mark.parametrize("N", [513, 969, 1025])
def test_G3_ncsx_lines(N):
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    th = bo.theta_grid(N)
    for i, geo in enumerate(g3["geo_%d" % N]):
        bmag, gp, cv, cv0, gd2, gd21, gd22, gb = geo
        dP = bo.dPdrho_of(cv, gb, bmag)
        assert abs(dP - g3["dPdrho_%d" % N][i]) < 1e-15
        for j, t0 in enumerate(g3["theta0"]):
            cvf, gdf = bo.fold_theta0(t0, cv, cv0, gd2, gd21, gd22)
            gam = bo.gamma_ball_full(dP, th, bmag, gp, cvf, gdf)[0]
            assert abs(gam - g3["gam_tight_%d" % N][i, j]) < 1e-13   # reference, ARPACK converged
            assert abs(gam - g3["gam_%d" % N][i, j]) < 1e-8          # reference as shipped (tol 5e-7)


def test_G4_obj_w_grad():
    g4 = np.load(os.path.join(G, "G4_obj_w_grad.npz"))
    th = bo.theta_grid(513)
    for p, geo, v, j, vt, jt in zip(g4["pts"], g4["geo"], g4["val"], g4["jac"], g4["val_tight"], g4["jac_tight"]):
        val, jac = bo.obj_w_grad_lines(th, p[2], geo[0], geo[1], geo[2], float(g4["del_alpha"]))
        assert abs(val - vt) < 1e-13 and np.abs(jac - jt).max() < 1e-11
        assert abs(val - v) < 1e-8 and np.abs(jac - j).max() < 1e-7  # shipped ARPACK tolerance


def test_G5_coarse_scan_and_argmax():
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    th = bo.theta_grid(513)
    rows = [0, 7, 15, 23]
    tab = bo.coarse_scan(th, g5["geo"][rows], g5["dPdrho"][rows], g5["theta0_scan"])
    assert np.abs(tab - g5["gam_table"][rows]).max() < 1e-8
    assert bo.argmax_first(g5["gam_table"]) == tuple(int(v) for v in g5["argmax"])
    assert bo.argmax_first(np.zeros((3, 3))) is None
    # refined optimum re-solved with the stored geometry (ball_scan.py:322-339)
    bmag, gp, cv, cv0, gd2, gd21, gd22, gb = g5["geo_opt"]
    cvf, gdf = bo.fold_theta0(g5["x_opt"][1], cv, cv0, gd2, gd21, gd22)
    gam = bo.gamma_ball_full(float(g5["dPdrho_opt"]), th, bmag, gp, cvf, gdf)[0]
    assert abs(gam - float(g5["gam_opt"])) < 1e-8


@pytest.mark.parametrize("N", [257, 513])
def test_G6_rough_random(N):
    g6 = np.load(os.path.join(G, "G6_random_rough.npz"))
    th = bo.theta_grid(N)
    for k in range(len(g6["gcf_%d" % N])):
        gam, lam, X, dX = bo.solve_gcf(th, *g6["gcf_%d" % N][k])
        assert abs(gam - g6["gam_tight_%d" % N][k]) < 1e-10
        d, e, fd, h, _, _, _ = bo.assemble(th, *g6["gcf_%d" % N][k])
        assert bo.sturm_count_above(d, e, fd, lam + 1e-9) == 0
        assert bo.sturm_count_above(d, e, fd, lam - 1e-9) == 1


def test_G7_sign_pins():
    g7 = np.load(os.path.join(G, "G7_cobra_pins.npz"))
    assert (g7["gamma_max_op"] < 0).all() and g7["gamma_max_og"].max() > 0.02
