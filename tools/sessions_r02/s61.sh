cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s61; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 < /dev/null; tail -1 $O/smoke.txt
( time timeout 1800 python bench.py > $O/bench.txt 2> $O/bench.err < /dev/null ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -1 $O/bench.txt > $O/bench.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/s61/bench.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"].get("request_rate_frac"))
for c in d.get("configs", []): print(c.get("name"), c.get("value"), c.get("error"), c.get("fraction_of_replicated"))
print("e2e", d.get("end_to_end", {}).get("walk_steps_per_s"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s61/kt -- python $GRAFT_REPO_ROOT/bench.py --configs 0 --end-to-end 0 --cpu-baseline 0 > $GRAFT_REPO_ROOT/gpurun_out/s61/bench_profiled.txt 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/s61/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200 > gpurun_out/s61/kernel_stats_head.txt; cat gpurun_out/s61/kernel_stats_head.txt | cut -c1-160
tail -1 gpurun_out/s61/bench_profiled.txt | cut -c1-400
find gpurun_out/s61/kt -name '*.csv' -size +20M -delete
