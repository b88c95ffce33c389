"""CPU oracle for the ideal-ballooning hot path.  TEST INFRASTRUCTURE ONLY.

This is a numpy/scipy restatement of the reference algorithm, used as the checker in
tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  It is NOT part of
the product path: nothing under ideal-ballooning-solver_amd/ may import it.

Parity status: PINNED against golden vectors generated in the build container by importing
the reference itself (tests/golden/make_golden.py -> tests/golden/G*.npz) and against the
reference's own s-alpha test functions (G2).  One link is formally unpinned upstream: the
reference takes its eigenvector from ARPACK inside an unpinned scipy (utils.py:1597,
tol=5e-7); this oracle takes the *exact* dominant eigenvector (LAPACK dstebz/dstein on the
symmetrised tridiagonal) of the same matrix, so reference and oracle agree to the
reference's own ARPACK noise floor (<= 1e-8 on gam; measured in tests/test_oracle_golden.py).

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import numpy as np
from scipy.linalg import eigh_tridiagonal


# ------------------------------------------------------------------------------------------
# A0  grid and start vector (ball_scan.py:201-209, 233)
def theta_grid(N, theta_fac=4):
    return np.linspace(-theta_fac * np.pi, theta_fac * np.pi, N)


def vguess(theta, theta_fac=4):
    return (1 - np.tanh(theta[1:-1] / np.pi) ** 2) * np.cos(theta[1:-1] / (2 * theta_fac))


# A1  per-line pressure-gradient scalar (ball_scan.py:262; utils.py:1657)
def dPdrho_of(cvdrift, gbdrift, bmag):
    return -1.0 * 0.5 * np.mean((cvdrift - gbdrift) * bmag ** 2)


# A2  theta0 fold (ball_scan.py:267-268; utils.py:1659-1660)
def fold_theta0(theta0, cvdrift, cvdrift0, gds2, gds21, gds22):
    return cvdrift + theta0 * cvdrift0, gds2 + 2 * theta0 * gds21 + theta0 ** 2 * gds22


# A3  coefficients (utils.py:1560-1562)
def gcf(dPdrho, B, gradpar, cvdrift, gds2):
    gp = np.abs(gradpar)
    g = gp * gds2 / B
    c = -1 * dPdrho * cvdrift * 1 / (gp * B)
    f = gds2 / B ** 2 * 1 / (gp * B)
    return g, c, f


def regrid_uniform(theta, g, c, f):
    """utils.py:1567-1571: np.interp onto linspace(theta[0], theta[-1], N) (identity if uniform)."""
    tu = np.linspace(theta[0], theta[-1], len(theta))
    return tu, np.interp(tu, theta, g), np.interp(tu, theta, c), np.interp(tu, theta, f)


def assemble(theta, g, c, f):
    """Tridiagonal pencil (T, F) of utils.py:1574-1592 in band form.

    rows r = 0..n-1 (n = N-2) sit on grid points j = r+1.
    d[r]   = -(gh[r] + gh[r+1])/h^2 + c[j]
    e[k]   = gh[k]/h^2, k = 0..N-2  (e[r] couples rows r-1 and r; e[0], e[n] touch the zero ends)
    fd[r]  = f[j]
    Note g_u_half interpolates the ORIGINAL (theta, g) at the uniform half points (utils.py:1576).
    """
    tu, gu, cu, fu = regrid_uniform(theta, g, c, f)
    th_half = 0.5 * (tu[:-1] + tu[1:])
    h = np.diff(th_half)[2]
    gh = np.interp(th_half, theta, g)
    e = gh / h ** 2
    d = -(gh[1:] + gh[:-1]) / h ** 2 + cu[1:-1]
    return d, e, fu[1:-1].copy(), h, gu, cu, fu


def sturm_count_above(d, e, fd, lam):
    """number of eigenvalues of (T, F) strictly greater than lam (division-form Sturm sequence;
    SURVEY.md Appendix A).  n - this = number below."""
    n = len(d)
    cnt = 0
    q = d[0] - lam * fd[0]
    tiny = 1e-300
    if q == 0:
        q = -tiny
    if q > 0:
        cnt += 1
    for r in range(1, n):
        q = (d[r] - lam * fd[r]) - e[r] ** 2 / q
        if q == 0:
            q = -tiny
        if q > 0:
            cnt += 1
    return cnt


def top_eigenpair(d, e, fd):
    """largest eigenvalue/eigenvector of T x = lam F x via the similar symmetric matrix
    F^-1/2 T F^-1/2 (A = F^-1 T of utils.py:1584-1592 has the same spectrum)."""
    n = len(d)
    a = d / fd
    b = e[1:n] / np.sqrt(fd[:-1] * fd[1:])
    w, v = eigh_tridiagonal(a, b, select="i", select_range=(n - 1, n - 1))
    x = v[:, 0] / np.sqrt(fd)
    return w[0], x


def rayleigh_growth(x, h, g, c, f):
    """utils.py:1601-1621: normalise, FD derivative (2nd order at the ends, 4th inside),
    Simpson ratio with unit spacing (N odd: classical composite rule)."""
    N = len(x) + 2
    X = np.zeros(N)
    dX = np.zeros(N)
    X[1:-1] = x / np.max(np.abs(x))
    dX[0] = (-1.5 * X[0] + 2 * X[1] - 0.5 * X[2]) / h
    dX[1] = (X[2] - X[0]) / (2 * h)
    dX[-2] = (X[-1] - X[-3]) / (2 * h)
    dX[-1] = (0.5 * X[-3] - 2 * X[-2] + 1.5 * 0.0) / h
    dX[2:-2] = 2 / (3 * h) * (X[3:-1] - X[1:-3]) - (X[4:] - X[0:-4]) / (12 * h)
    Y0 = -g * dX ** 2 + c * X ** 2
    Y1 = f * X ** 2
    return simpson_unit(Y0) / simpson_unit(Y1), X, dX


def simpson_unit(y):
    """scipy.integrate.simps(y) with dx=1 for an odd number of samples."""
    N = len(y)
    if N % 2 == 0:
        raise ValueError("oracle restates the odd-N composite Simpson rule only (reference grids are odd)")
    return (y[0] + y[-1] + 4.0 * np.sum(y[1:-1:2]) + 2.0 * np.sum(y[2:-1:2])) / 3.0


def solve_gcf(theta, g, c, f):
    """raw (g, c, f) entry: returns (gam, lam_matrix, X, dX)."""
    d, e, fd, h, gu, cu, fu = assemble(theta, g, c, f)
    lam, x = top_eigenpair(d, e, fd)
    if x[np.argmax(np.abs(x))] < 0:
        x = -x
    gam, X, dX = rayleigh_growth(x, h, gu, cu, fu)
    return gam, lam, X, dX


# A4  the operator (utils.py:1550-1624).  vguess/sigma0 only steer ARPACK upstream: accepted, ignored.
def gamma_ball_full(dPdrho, theta_PEST, B, gradpar, cvdrift, gds2, vguess=None, sigma0=0.42):
    g, c, f = gcf(dPdrho, B, gradpar, cvdrift, gds2)
    d, e, fd, h, gu, cu, fu = assemble(theta_PEST, g, c, f)
    lam, x = top_eigenpair(d, e, fd)
    if x[np.argmax(np.abs(x))] < 0:
        x = -x
    gam, X, dX = rayleigh_growth(x, h, gu, cu, fu)
    return gam, X, dX, gu, cu, fu


def lambda_matrix(dPdrho, theta_PEST, B, gradpar, cvdrift, gds2):
    g, c, f = gcf(dPdrho, B, gradpar, cvdrift, gds2)
    d, e, fd, h, gu, cu, fu = assemble(theta_PEST, g, c, f)
    return top_eigenpair(d, e, fd)[0]


# A6  Hellmann-Feynman derivative (utils.py:1666-1680, 1721-1725)
def hf_derivative(gam, X, dX, f, g_p, c_p, f_p):
    Y1 = simpson_unit(f * X ** 2)
    return (simpson_unit(c_p * X ** 2) / Y1 - simpson_unit(g_p * dX ** 2) / Y1
            - gam * simpson_unit(f_p * X ** 2) / Y1)


def obj_w_grad_lines(theta, theta0, line_l, line_c, line_r, del_alpha=0.004):
    """utils.py:1632-1728 given the three field lines (alpha-d/2, alpha, alpha+d/2) as tuples
    (bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift).  Returns (-gam, [-dg/dalpha, -dg/dtheta0])."""
    def prep(line):
        bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift = line
        dP = dPdrho_of(cvdrift, gbdrift, bmag)
        cv, gd = fold_theta0(theta0, cvdrift, cvdrift0, gds2, gds21, gds22)
        return dP, cv, gd

    bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22, gbdrift = line_c
    dP, cv, gd = prep(line_c)
    gam, X, dX, g, c, f = gamma_ball_full(dP, theta, bmag, gradpar, cv, gd)
    gp = np.abs(gradpar)
    g_t = gp * (2 * gds21 + 2 * theta0 * gds22) / bmag          # utils.py:1669
    c_t = -1 * dP * cvdrift0 * 1 / (gp * bmag)                  # utils.py:1670
    f_t = (2 * gds21 + 2 * theta0 * gds22) / bmag ** 2 * 1 / (gp * bmag)   # utils.py:1671-1673
    jac_t = hf_derivative(gam, X, dX, f, g_t, c_t, f_t)
    dPr, cvr, gdr = prep(line_r)
    dPl, cvl, gdl = prep(line_l)
    g_r, c_r, f_r = gcf(dPr, line_r[0], line_r[1], cvr, gdr)     # utils.py:1705-1707
    g_l, c_l, f_l = gcf(dPl, line_l[0], line_l[1], cvl, gdl)     # utils.py:1709-1711
    jac_a = hf_derivative(gam, X, dX, f, (g_r - g_l) / del_alpha, (c_r - c_l) / del_alpha, (f_r - f_l) / del_alpha)
    return -1 * gam, np.array([-1 * jac_a, -1 * jac_t])


# A5  coarse scan + argmax (ball_scan.py:248-295) on pre-computed lines
def coarse_scan(theta, lines7, dPdrho, theta0_scan):
    """lines7: (nalpha, 7, N) in order bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22."""
    na = len(lines7)
    tab = np.zeros((na, len(theta0_scan)))
    for i in range(na):
        bmag, gradpar, cvdrift, cvdrift0, gds2, gds21, gds22 = lines7[i]
        for j, t0 in enumerate(theta0_scan):
            cv, gd = fold_theta0(t0, cvdrift, cvdrift0, gds2, gds21, gds22)
            tab[i, j] = gamma_ball_full(dPdrho[i], theta, bmag, gradpar, cv, gd)[0]
    return tab


def argmax_first(tab):
    """ball_scan.py:279-295: first (row-major) index of the maximum; all-zero table -> None."""
    m = np.max(tab)
    if m == 0.0:
        return None
    idx = np.where(tab == m)
    return int(idx[0][0]), int(idx[1][0])


# s-alpha analytic coefficients (tests/shifted-circle-s-alpha/bishop_ball_s-alpha.py:30-45)
def salpha_gc(theta, shat, alpha, theta0):
    lam = shat * (theta - theta0) - alpha * (np.sin(theta) - np.sin(theta0))
    return 1 + lam ** 2, alpha * (np.cos(theta) + np.sin(theta) * lam)


def salpha_unstable(shat, alpha, theta0, N=1601, extent=61):
    """'isunstable' of check_ball (bishop_ball_s-alpha.py:20-115) restated as: the gamma=0
    three-term recurrence changes sign  <=>  T (f-independent at lambda=0) has a positive
    eigenvalue  <=>  Sturm count above 0 is > 0."""
    theta = np.linspace(-extent * np.pi, extent * np.pi, N)
    g, c = salpha_gc(theta, shat, alpha, theta0)
    d, e, fd, h, gu, cu, fu = assemble(theta, g, c, np.ones(N))
    return int(sturm_count_above(d, e, fd, 0.0) > 0)


def gamma_ball_full_dense_arpack(dPdrho, theta_PEST, B, gradpar, cvdrift, gds2, vguess=None, sigma0=0.42):
    """Faithful-COST restatement of utils.py:1582-1624: materialise the dense (N-2)^2 matrix
    A = F^-1 T and take the eigenpair nearest sigma0 by ARPACK shift-invert (dense LU inside), as the
    reference does.  Used only to time what the reference's formulation costs on the GPU box's CPU."""
    from scipy.sparse.linalg import eigs
    g, c, f = gcf(dPdrho, B, gradpar, cvdrift, gds2)
    d, e, fd, h, gu, cu, fu = assemble(theta_PEST, g, c, f)
    n = len(d)
    A = np.zeros((n, n))
    idx = np.arange(n)
    A[idx, idx] = d / fd
    A[idx[1:], idx[:-1]] = e[1:n] / fd[1:]
    A[idx[:-1], idx[1:]] = e[1:n] / fd[:-1]
    w, v = eigs(A, 1, sigma=sigma0, v0=vguess, tol=5.0e-7, OPpart="r")
    x = v[:, 0].real
    gam, X, dX = rayleigh_growth(x, h, gu, cu, fu)
    return gam, X, dX, gu, cu, fu
