#!/usr/bin/env python3
"""Condense rocprofv3 --pmc CSVs (separate FETCH_SIZE / WRITE_SIZE / SQ passes of tools/profile_run.py)
into profiles/<tag>_pmc.json.   usage: tools/pmc_summary.py gpurun_out/pmc_r01 profiles/r01_pmc.json [set-name]

Launches are grouped by (kernel name, grid size, workgroup size): two legs that run the same kernel at different sizes get
two entries, keyed "<kernel> @<grid>x<wg>" (grid = threads of the launch).  Every entry carries the spread of its
launches' VALU instruction counts (`valu_spread` = max / min): a persistent kernel whose grid does not depend on the
batch (k_geo_rows) shows mixed batch sizes there, and bench.py refuses to quote such an entry.
(the set name is stored under "_set": bench.py quotes it next to every counter it replays from this file)"""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = collections.defaultdict(dict)
for tag in ("fetch", "write", "sq"):
    fn = os.path.join(src, "%s_counter_collection.csv" % tag)
    if not os.path.exists(fn):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "ibs::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "").strip()
        key = "%s @%sx%s" % (short, r["Grid_Size"], r["Workgroup_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, vals in v.items():
            out[k][c] = dict(mean=sum(vals) / len(vals), n=len(vals))
            if c == "SQ_INSTS_VALU":
                out[k]["valu_spread"] = max(vals) / max(min(vals), 1.0)
for k, v in list(out.items()):
    name, _, dims = k.partition(" @")
    v["kernel"] = name
    v["grid_threads"], v["workgroup"] = (int(x) for x in dims.split("x"))
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of
        # the bytes of a coalesced streaming read -> double it; WRITE_SIZE is exact.
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024
    if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
        v["valu_insts_per_wave"] = v["SQ_INSTS_VALU"]["mean"] / v["SQ_WAVES"]["mean"]
        v["valu_busy_frac_of_wave_lifetime"] = v["SQ_ACTIVE_INST_VALU"]["mean"] / v["SQ_WAVE_CYCLES"]["mean"]
out["_set"] = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(dst).replace("_pmc.json", "")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: ({kk: vv for kk, vv in v.items() if not isinstance(vv, dict)} if isinstance(v, dict) else v)
                  for k, v in out.items()}, indent=1, sort_keys=True))
