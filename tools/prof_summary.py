"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) into a short text file for profiles/."""
import csv, sys, collections, os, re

def short(name):
    m = re.search(r'(k_[a-z_0-9]+)', name)
    if m: return m.group(1)
    if 'radix_sort' in name: return 'rocprim::radix_sort_' + ('onesweep_iteration' if 'onesweep_iteration' in name else 'other')
    return name[:60]

def kernel_stats(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in rows:
        n = short(r['Kernel_Name']); d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        a = agg[n]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = ["%-34s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%")]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append("%-34s %6d %12.3f %12.3f %12.3f %12.3f %6.1f" % (n, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    return "\n".join(out)

def counters(path, want):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    out = []
    for k, cs in agg.items():
        if want and want not in k: continue
        for c, v in cs.items():
            out.append("%-34s %-16s n=%d avg=%.6g min=%.6g max=%.6g" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
    return "\n".join(out)

if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    print(kernel_stats(path) if mode == "stats" else counters(path, sys.argv[3] if len(sys.argv) > 3 else ""))
