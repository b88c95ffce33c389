"""CPU: the geometry oracle (row F1) against the arrays the reference itself produced (G3) from G8 inputs."""
import os

import numpy as np
import pytest

from oracle import ballooning_oracle as bo
from oracle import geometry_oracle as go

G = os.path.join(os.path.dirname(__file__), "golden")


def test_surface_tables_from_wout_match_reference_splines():
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    ref = np.load(os.path.join(G, "G8_surface_tables.npz"))
    tab = go.surface_tables_from_wout(wout, ref["s"])
    for k in go.NAMES_MN + go.NAMES_NYQ + ("iota", "d_iota_d_s", "d_pressure_d_s"):
        assert np.abs(tab[k] - ref[k]).max() <= 1e-12 * max(1.0, np.abs(ref[k]).max()), k
    assert abs(tab["phiedge"] - float(ref["phiedge"])) < 1e-15


@pytest.mark.parametrize("N", [513, 1025])
def test_fieldline_geometry_matches_reference(N):
    g3 = np.load(os.path.join(G, "G3_ncsx_lines.npz"))
    ref = dict(np.load(os.path.join(G, "G8_surface_tables.npz")))
    ref["phiedge"] = float(ref["phiedge"]); ref["Aminor_p"] = float(ref["Aminor_p"])
    th = bo.theta_grid(N)
    lines = g3["lines_%d" % N]
    geo = g3["geo_%d" % N]
    for k in range(0, len(lines), 3):
        s, a = lines[k]
        js = int(np.argmin(np.abs(ref["s"] - s)))
        mine = go.fieldline_geometry(ref, js, [a], th)[0]
        for q in range(8):
            scale = np.abs(geo[k, q]).max()
            assert np.abs(mine[q] - geo[k, q]).max() < 1e-9 * scale, (k, q)
