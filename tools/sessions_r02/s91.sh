cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s91; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 < /dev/null; tail -1 $O/smoke.txt
( time timeout 1800 python bench.py > $O/bench.txt 2> $O/bench.err < /dev/null ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -1 $O/bench.txt > $O/bench.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/s91/bench.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"].get("request_rate_frac"))
for c in d.get("configs", []): print(c.get("name"), c.get("value"), c.get("error"), c.get("fraction_of_replicated"), (c.get("roofline") or {}).get("physical_traffic_frac"))
print("e2e", d.get("end_to_end", {}).get("walk_steps_per_s"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
