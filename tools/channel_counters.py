"""Scratch: per-dispatch, per-instance (memory channel / XCD) values of raw TCC counters of k_walk_first_order next to the dispatch's
duration — does a slow placement show a channel imbalance?  usage: channel_counters.py <rocprof output dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    if "k_walk_first_order" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
rows = [r for r in csv.DictReader(open(cc[0])) if "k_walk_first_order" in r["Kernel_Name"]]
if rows:
    print("columns:", list(rows[0].keys()))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    vals[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals, key=lambda x: int(x)):
    for n, v in sorted(vals[k].items()):
        s = sum(v)
        print("dispatch %-6s %6.1f ms  %-34s n=%-4d sum=%.4g min=%.4g max=%.4g max/mean=%.3f" % (
            k, dur.get(k, -1), n, len(v), s, min(v), max(v), max(v) / (s / len(v)) if s else 0))
