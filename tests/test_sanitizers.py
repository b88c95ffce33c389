"""CPU: the host-side native code (C oracle through its ctypes marshalling, the optimizer state machine) under
AddressSanitizer + UBSan (tools/run_sanitizers.sh).  GPU sanitizers are not available on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_ubsan():
    if not shutil.which("gcc") or not os.path.exists(subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True,
                                                                    text=True).stdout.strip()):
        pytest.skip("no gcc / libasan here")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "run_sanitizers.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitizers: clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
