"""CPU: the C-ABI library builds, loads without a GPU and exports every symbol include/ibs.h declares."""
import ctypes
import os
import re

import pytest

import ibs_amd
from ibs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ibs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ibs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.lib()
    names = header_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)


def test_version_and_error_string():
    lib = _lib.lib()
    assert lib.ibs_version() >= 100
    assert isinstance(lib.ibs_last_error(), bytes)


def test_no_silent_cpu_fallback():
    """without a GPU the product path must fail loudly"""
    lib = _lib.lib()
    if lib.ibs_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ibs_amd.IbsError):
        ibs_amd.Context(0)


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.lib()
    assert lib.ibs_create(None, 0) < 0
    assert lib.ibs_solve_gcf_f64(None, 1, 513, 0.1, None, None, None, 513, None, None, None, None, None, 0) < 0
    assert b"null" in lib.ibs_last_error()


def test_rccl_stand_in_compiles_as_strict_c():
    """tests/cabi/fake_rccl.c (test infrastructure of the multi-rank GPU tests): builds with -Wall -Wextra -Werror against the HIP
    headers and exports RCCL's five entry points the library binds (ibs_api.hip: ibs_comm_load)"""
    import shutil
    import subprocess
    import tempfile
    import pytest
    inc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(inc, "hip", "hip_runtime_api.h")):
        pytest.skip("no gcc / HIP headers")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "libfake_rccl.so")
        r = subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I", inc, os.path.join(root, "tests", "cabi", "fake_rccl.c"),
                            "-o", so, "-lrt", "-lpthread"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
        for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString"):
            assert " T " + name in syms, name
