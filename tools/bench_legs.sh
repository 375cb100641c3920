# Scratch: the default bench run with its wall time and the wall time of every optional leg (run from the repo root on the GPU box).
cd $GRAFT_REPO_ROOT
t0=$(date +%s.%N)
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err
t1=$(date +%s.%N)
echo "bench.py wall: $(echo "$t1 - $t0" | bc) s, line $(wc -c < gpurun_out/bench_line.json) bytes"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail.json"))
print("before cpu baseline", d.get("wall_s_before_cpu_baseline"))
for c in d.get("configs", []): print(c["name"], round(c.get("leg_wall_s", 0), 1), c.get("error", ""))
PY
