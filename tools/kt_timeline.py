"""Timeline of a rocprofv3 --kernel-trace csv: busy time, gaps between consecutive dispatches and per-kernel totals.
python tools/kt_timeline.py DIR [--last N] [--dump]      (--dump: one line per dispatch of the last N)"""
import csv, glob, sys, collections
d = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", "")))
rows.sort()
if last:
    rows = rows[-last:]
tot = collections.OrderedDict()
busy = 0; gap = 0; prev_end = None
for s, e, n, gx, gy in rows:
    a = tot.setdefault(n, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    if prev_end is not None:
        g = max(0, s - prev_end) / 1e3
        if g < 200: a[2] += g; gap += g     # (longer pauses are host pauses between calls)
    busy += (e - s) / 1e3
    if "--dump" in sys.argv:
        print("%10.1f us  dur %8.1f  gap %6.1f  %s [%s,%s]" % ((s - rows[0][0]) / 1e3, (e - s) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3, n, gx, gy))
    prev_end = max(prev_end or 0, e)
print("span %.1f us, kernels busy %.1f us, gaps<200us %.1f us" % ((rows[-1][1] - rows[0][0]) / 1e3, busy, gap))
for n, (c, t, g) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-50s n=%5d total %9.1f us avg %7.1f us  gap-before avg %5.1f us" % (n, c, t, t / c, g / c))
