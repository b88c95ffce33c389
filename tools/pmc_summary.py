#!/usr/bin/env python3
"""Condense rocprofv3 --pmc CSVs (separate FETCH_SIZE / WRITE_SIZE / SQ passes of tools/profile_run.py)
into profiles/<tag>_pmc.json.   usage: tools/pmc_summary.py gpurun_out/pmc_r01 profiles/r01_pmc.json [set-name]

One entry per (kernel name, grid size, workgroup size, WORK CLASS), keyed "<kernel> @<grid>x<wg> #<k>": launches of one kernel at
one launch size are split into classes of equal work -- VALU instruction counts within 2 % of the class's first launch, in order of
first appearance -- so that two legs which run the same kernel at the same size on different data (the smooth and the rough
family of configs[4]) or a persistent kernel whose grid does not show its batch size (k_geo_rows) never share an entry.  The
classes come from the SQ pass; the FETCH / WRITE passes run the same program, so their launches are matched by their ordinal among
the library's launches.  bench.py looks an entry up by exact kernel name, waves per launch and class number.
(the set name is stored under "_set": bench.py quotes it next to every counter it replays from this file)"""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]


def launches(tag):
    """the library's launches of one pass in dispatch order: [(short name, grid, wg, {counter: value})]"""
    fn = os.path.join(src, "%s_counter_collection.csv" % tag)
    if not os.path.exists(fn):
        return None
    by_id = collections.OrderedDict()
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "ibs::" not in k:
            continue
        did = int(r["Dispatch_Id"])
        if did not in by_id:
            by_id[did] = [k.split("(")[0].replace("void ", "").strip(), int(r["Grid_Size"]), int(r["Workgroup_Size"]), {}]
        by_id[did][3][r["Counter_Name"]] = float(r["Counter_Value"])
    return [by_id[d] for d in sorted(by_id)]


sq = launches("sq")
if sq is None:
    sys.exit("pmc_summary: %s holds no sq_counter_collection.csv" % src)
classes = collections.defaultdict(list)          # (name, grid, wg) -> [reference VALU count of class k]
keys = []
for name, grid, wg, c in sq:
    v = c.get("SQ_INSTS_VALU", 0.0)
    refs = classes[(name, grid, wg)]
    for k, ref in enumerate(refs):
        if abs(v - ref) <= 0.02 * max(ref, 1.0):
            break
    else:
        refs.append(v); k = len(refs) - 1
    keys.append("%s @%dx%d #%d" % (name, grid, wg, k))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for key, (name, grid, wg, c) in zip(keys, sq):
    for cn, val in c.items():
        acc[key][cn].append(val)
mismatch = 0
for tag in ("fetch", "write"):
    ls = launches(tag)
    if ls is None:
        continue
    if len(ls) != len(sq):
        print("pmc_summary: the %s pass saw %d launches, the sq pass %d: matched by kernel and grid only" % (tag, len(ls), len(sq)), file=sys.stderr)
    for i, (name, grid, wg, c) in enumerate(ls):
        key = keys[i] if i < len(keys) and keys[i].startswith("%s @%dx%d #" % (name, grid, wg)) and len(ls) == len(sq) else None
        if key is None:
            mismatch += 1
            key = "%s @%dx%d #0" % (name, grid, wg)
        for cn, val in c.items():
            acc[key][cn].append(val)
out = {}
for key, v in acc.items():
    e = {c: dict(mean=sum(vals) / len(vals), n=len(vals)) for c, vals in v.items()}
    name, _, rest = key.partition(" @")
    dims, _, cls = rest.partition(" #")
    e["kernel"] = name
    e["grid_threads"], e["workgroup"] = (int(x) for x in dims.split("x"))
    e["work_class"] = int(cls)
    if "SQ_INSTS_VALU" in v:
        e["valu_spread"] = max(v["SQ_INSTS_VALU"]) / max(min(v["SQ_INSTS_VALU"]), 1.0)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of
        # the bytes of a coalesced streaming read -> double it; WRITE_SIZE is exact.
        e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024
    if "SQ_INSTS_VALU" in e and "SQ_WAVES" in e:
        e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"]["mean"] / e["SQ_WAVES"]["mean"]
        e["valu_busy_frac_of_wave_lifetime"] = e["SQ_ACTIVE_INST_VALU"]["mean"] / e["SQ_WAVE_CYCLES"]["mean"]
    out[key] = e
out["_set"] = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(dst).replace("_pmc.json", "")
out["_unmatched_launches_of_the_byte_passes"] = mismatch
# the sources the counted kernels were compiled from (bench.src_sha): bench.py refuses to replay an entry once its kernel
# group's sources differ from this ("stale"); IBS_PMC_SRC_ROOT = the csrc directory of the tree the passes ran on
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
out["_src_sha"] = bench.src_sha(os.environ.get("IBS_PMC_SRC_ROOT"))
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k in sorted(out):
    v = out[k]
    if isinstance(v, dict) and not k.startswith("_"):
        print("%-72s n %3d waves %9.0f valu/wave %8.0f traffic %s" % (k, v.get("SQ_WAVES", {}).get("n", 0), v.get("SQ_WAVES", {}).get("mean", 0),
                                                                      v.get("valu_insts_per_wave", 0), v.get("hbm_bytes_per_launch")))
if mismatch:
    sys.exit("pmc_summary: %d launches of the byte passes could not be matched to the SQ pass's work classes; bench.py will "
             "withhold `traffic` for this set" % mismatch)
