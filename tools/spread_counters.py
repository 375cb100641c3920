"""Scratch: per-dispatch counters of k_walk_first_order next to the dispatch's duration (rocprofv3 --kernel-trace --pmc ... CSVs),
to see which counter moves with the 64 / 74 ms levels.  usage: spread_counters.py <rocprof output dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    if "k_walk_first_order" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    if "k_walk_first_order" in r["Kernel_Name"]:
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({n for v in vals.values() for n in v})
print("dispatch  ms      " + "  ".join(names))
for k in sorted(vals, key=lambda x: int(x)):
    print("%-8s %6.1f  " % (k, dur.get(k, -1)) + "  ".join("%.4g" % vals[k].get(n, float("nan")) for n in names))
