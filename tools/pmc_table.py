"""Per-kernel sums of a rocprofv3 --pmc csv (counter_collection): python tools/pmc_table.py DIR [kernel-substring]"""
import csv, glob, sys, collections
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if sub not in k: continue
        key = (k, r.get("Grid_Size", ""))
        d = agg.setdefault(key, collections.OrderedDict())
        c = r["Counter_Name"]; v = float(r["Counter_Value"])
        d.setdefault(c, [0, 0.0]); d[c][0] += 1; d[c][1] += v
for (k, g), d in agg.items():
    print(k, "grid", g)
    n = None
    for c, (cnt, tot) in d.items():
        print("   %-28s launches %4d  per launch %.4g" % (c, cnt, tot / cnt))
    w = d.get("SQ_WAVES"); v = d.get("SQ_INSTS_VALU"); a = d.get("SQ_ACTIVE_INST_VALU"); wc = d.get("SQ_WAVE_CYCLES")
    if w and v: print("   -> VALU insts per wave %.0f" % (v[1] / w[1]))
    if a and wc: print("   -> SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES %.3f" % (a[1] / wc[1]))
