"""CPU: the C oracle (cpu_baseline port) against the numpy oracle and the goldens."""
import os

import numpy as np

from oracle import ballooning_oracle as bo
from oracle import c_oracle as co

G = os.path.join(os.path.dirname(__file__), "golden")


def test_c_oracle_salpha_matches_numpy_oracle():
    for N in (257, 513):
        th = bo.theta_grid(N)
        for sh, al, t0 in ((1.0, 0.8, 0.0), (0.3, 1.1, 0.3), (1.5, 0.3, 0.1)):
            g, c = bo.salpha_gc(th, sh, al, t0)
            gam, lam, X, dX = co.solve_gcf(th[1] - th[0], g, c, g)
            go, lo, Xo, dXo = bo.solve_gcf(th, g, c, g)
            assert abs(gam - go) < 1e-11 and abs(lam - lo) < 1e-11
            assert np.abs(X - Xo).max() < 1e-9 and np.abs(dX - dXo).max() < 1e-8


def test_c_oracle_scan_matches_goldens():
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    geo = g3["geo_513"]
    th = bo.theta_grid(513)
    gam, lam, used = co.gamma_scan(th[1] - th[0], *[geo[:, k, :] for k in range(7)], g3["dPdrho_513"], g3["theta0"])
    assert used >= 1
    assert np.abs(gam - g3["gam_tight_513"]).max() < 1e-11
    assert np.abs(gam - g3["gam_513"]).max() < 1e-8


def test_c_oracle_rough_eigenvalue():
    g6 = np.load(os.path.join(G, "G6_random_rough.npz"))
    th = bo.theta_grid(513)
    gcf = g6["gcf_513"]
    gam, lam, _ = co.solve_gcf_batch(th[1] - th[0], gcf[:, 0], gcf[:, 1], gcf[:, 2])
    for k in range(len(gcf)):
        assert abs(lam[k] - bo.solve_gcf(th, *gcf[k])[1]) < 1e-10
