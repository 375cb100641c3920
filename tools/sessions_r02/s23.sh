set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s23; mkdir -p $O
SRW_TIMING=1 SRW_EB_RESERVE_GB=14 SRW_HUB_BUDGET_GB=4 SRW_EB_BUDGET_GB=200 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5_more.txt 2>&1 < /dev/null; grep -E "^iter|edge tables" $O/c5_more.txt
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err < /dev/null; tail -4 $O/bench.err; python - <<'PY'
import json
s=open('gpurun_out/s23/bench.json').read(); j=json.loads(s[s.index('{"metric'):].splitlines()[0])
print(j['value'], j['end_to_end'].get('seconds'))
for c in j['configs']: print(c['name'], c.get('value'), c.get('kernel_ms'), c.get('setup_s'), c.get('roofline',{}).get('kernel'), c.get('strategy_steps'), c.get('error'))
PY
