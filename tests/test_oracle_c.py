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


def test_c_oracle_lam_batch_and_division_form_counts():
    """the arbiters of the 10^6-system eigenvalue tests (round 6): lam_batch = the bisection of solve_gcf without the eigenvector
    stage; count_above_batch = the division-form Sturm count, consistent with it to a few ulp of ||A||"""
    g6 = np.load(os.path.join(G, "G6_random_rough.npz"))
    th = bo.theta_grid(513)
    h = th[1] - th[0]
    gcf = g6["gcf_513"]
    g, c, f = gcf[:, 0], gcf[:, 1], gcf[:, 2]
    _, lam, _ = co.solve_gcf_batch(h, g, c, f)
    assert np.array_equal(co.lam_batch(h, g, c, f), lam)
    e = 0.5 * (g[:, :-1] + g[:, 1:]) / h ** 2
    nA = ((np.abs(c[:, 1:-1] - (e[:, :-1] + e[:, 1:])) + e[:, :-1] + e[:, 1:]) / f[:, 1:-1]).max(axis=1)
    d = 8 * 2.220446049250313e-16 * nA
    assert (co.count_above_batch(h, g, c, f, lam + d) == 0).all() and (co.count_above_batch(h, g, c, f, lam - d) >= 1).all()
    assert (co.count_above_batch(h, g, c, f, np.full(len(g), 1e9)) == 0).all()
    assert (co.count_above_batch(h, g, c, f, np.full(len(g), -1e9)) == 511).all()


def test_G10_fixture_holds_a_top_pair_the_oracle_resolves():
    """tests/golden/G10_rough_pair_1025.npz (captured from the GPU's own 10^6-system batch, tools/experiments/find_bad_system.py): the
    system on which the round-5 kernels closed on lam_2.  LAPACK (scipy's eigh_tridiagonal on the symmetrised pencil) and the C
    oracle agree on lam_max to 1e-13 ||A||; lam_2 is the value round 5 returned."""
    from scipy.linalg import eigh_tridiagonal
    d = np.load(os.path.join(G, "G10_rough_pair_1025.npz"))
    g, c, f = d["g"], d["c"], d["f"]
    N = len(g); h = 8 * np.pi / (N - 1)
    e = 0.5 * (g[:-1] + g[1:]) / h ** 2
    dd = c[1:-1] - (e[:-1] + e[1:]); fd = f[1:-1]; s = 1 / np.sqrt(fd)
    w = eigh_tridiagonal(dd * s * s, e[1:-1] * s[:-1] * s[1:], eigvals_only=True, select="i", select_range=(N - 4, N - 3))
    nA = ((np.abs(dd) + e[:-1] + e[1:]) / fd).max()
    lam_c = co.lam_batch(h, g[None], c[None], f[None])[0]
    assert abs(lam_c - w[1]) < 1e-13 * nA and abs(float(d["lam_max"]) - lam_c) < 1e-15
    assert abs(float(d["lam_returned_round5"]) - w[0]) < 1e-13 * nA
    assert 1.4e-10 < (w[1] - w[0]) / nA < 1.6e-10
    assert co.count_above_batch(h, g[None], c[None], f[None], np.array([0.5 * (w[0] + w[1])]))[0] == 1
