"""The C ABI from a host that is not Python: tests/cabi/host_salpha.c is compiled against include/ibs.h with gcc (C99) and linked with
the library and a HIP runtime only; on the GPU it solves the reference's s-alpha systems and must reproduce the golden growth rates
captured from the reference (G1).  The CPU part checks that the header is valid C and that the program links."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "host_salpha.c")
LIBDIR = os.path.join(ROOT, "ideal-ballooning-solver_amd", "lib")


def _hip_runtime_dir():
    for d in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib"), "/opt/rocm/lib"):
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    return None


def _build(tmp_path):
    exe = str(tmp_path / "host_salpha")
    hip = _hip_runtime_dir()
    assert hip is not None, "no libamdhip64.so under /opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", SRC, "-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-libs_hip",
           "-L", hip, "-lamdhip64", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + hip, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_header_is_valid_c_and_a_c_host_links(tmp_path):
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), SRC],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if not os.path.exists(os.path.join(LIBDIR, "libibs_hip.so")):
        pytest.skip("library not built")
    _build(tmp_path)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_host_reproduces_the_reference_goldens(tmp_path):
    exe = _build(tmp_path)
    g1 = np.load(os.path.join(ROOT, "tests", "golden", "G1_salpha.npz"))
    for N in (257, 513):
        sel = g1["params"][:, 0] == N
        params = g1["params"][sel][:, 1:]
        args = [exe, str(N)] + [repr(float(v)) for row in params for v in row]
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr[-500:])
        assert "expected error" in r.stderr and "N" in r.stderr
        rows = [ln.split() for ln in r.stdout.strip().splitlines()]
        assert len(rows) == len(params)
        gam = np.array([float(x[4]) for x in rows]); info = np.array([int(x[5]) for x in rows])
        assert (((info >> 16) & 3) == 0).all()
        assert np.abs(gam - g1["gam"][sel]).max() < 1e-8            # the stated FP64 tolerance
