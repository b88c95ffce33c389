#!/usr/bin/env python3
"""VGPR / AGPR / scratch / occupancy of every kernel of one translation unit, from hipcc -Rpass-analysis=kernel-resource-usage.
   python tools/resource_usage.py ibs_kernels.hip -DIBS_M=16        (run from anywhere; cross-compiles, no GPU needed)"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ideal-ballooning-solver_amd", "csrc")


def usage(src, defs):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DIBS_WITH_F32", "-Wno-unused-value", *defs,
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in out.split("\n"):
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]|VGPRs Spill): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "Function Name":
            if cur:
                rows.append(cur)
            cur = {"name": v}
        else:
            cur[k] = v
    if cur:
        rows.append(cur)
    return rows


if __name__ == "__main__":
    for r in usage(sys.argv[1], sys.argv[2:]):
        nm = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        print("%-52s VGPR %-4s AGPR %-4s scratch %-5s occupancy %-2s SGPR %-4s" % (
            nm, r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("TotalSGPRs")))
