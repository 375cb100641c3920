cd $GRAFT_REPO_ROOT
O=gpurun_out/s30; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/tools/cluster_timing.py 24 1 > $GRAFT_REPO_ROOT/$O/out.txt 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
tail -3 $O/out.txt
f=$(find $O/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" | head -14
t=$(find $O/kt -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# gaps between consecutive kernels of the last 400 launches
last = rows[-400:]
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(last, last[1:])]
durs = {}
for r in last:
    n = r["Kernel_Name"].split("(")[0][-30:]
    durs.setdefault(n, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("last 400 launches: mean gap us", sum(gaps) / len(gaps) / 1e3, "max", max(gaps) / 1e3)
for n, d in durs.items(): print(n, len(d), "mean us", sum(d) / len(d) / 1e3)
PY
find $O/kt -name '*.csv' -size +20M -delete
