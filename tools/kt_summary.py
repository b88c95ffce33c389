"""Per-kernel duration summary of a rocprofv3 --kernel-trace csv: python tools/kt_summary.py DIR [substr]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
        rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in rows.items():
    print(k, "n=%d min %.1f avg %.1f us" % (len(v), min(v), sum(v) / len(v)))
