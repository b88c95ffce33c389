"""Checks on the generated gfx950 code (hipcc cross-compiles: no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ideal-ballooning-solver_amd", "csrc")


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_dpp_sources_of_the_geometry_kernels_are_not_fresh_valu_results(tmp_path):
    """The geometry kernels' table broadcasts are inline-asm `v_fmac_f64_dpp ... row_newbcast` (csrc/ibs_geometry.hip, fmac_bc):
    the hazard recognizer does not look inside inline asm, so the rule it would enforce -- no VALU write of the DPP source in the
    two instructions before the DPP read -- is checked here on the generated code; and the kernels that use the broadcasts must
    not spill (their register budget is what the layout was sized for)."""
    out = tmp_path / "geo.s"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                        "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, "ibs_geometry.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    name, scratch = None, {}
    for line in r.stderr.split("\n"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    rows = {k: v for k, v in scratch.items() if "k_geo_rows" in k}
    assert len(rows) >= 6 and all(v == 0 for v in rows.values()), rows
    prev, n_dpp = [], 0
    for line in out.read_text().split("\n"):
        t = line.strip()
        if not line.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        if op == "v_fmac_f64_dpp":
            n_dpp += 1
            src = _regs(t.split(",")[1].strip())
            for pop, pl in prev[-2:]:
                if pop.startswith("v_"):
                    dst = _regs(pl.split(None, 1)[1].split(",")[0].strip())
                    assert not (dst & src), "VALU result read by a DPP instruction too early:\n  %s\n  %s" % (pl, t)
        prev.append((op, t))
    assert n_dpp >= 300, n_dpp                                  # (the three one-lane-per-point instantiations)
    assert "row_newbcast" in out.read_text()
