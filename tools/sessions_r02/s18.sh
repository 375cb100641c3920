set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s18
mkdir -p $O
summ() { f=$(find "$1" -name "*$2*.csv" 2>/dev/null | head -1); if [ -n "$f" ]; then python $R/tools/prof_summary.py $3 "$f" $4; else echo "no $2 csv under $1"; fi; }
cd $R; timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED" $O/pytest.txt | head; cd /tmp
G="python $R/tools/one_walk.py 24w 0.25 4 reference 2"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3r_trace -- $G > $O/c3r_trace.txt 2>&1 < /dev/null; grep "^iter" $O/c3r_trace.txt
summ $O/c3r_trace kernel_trace stats > $O/c3r_kernel_stats.txt; head -14 $O/c3r_kernel_stats.txt
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/c3r_$n -- $G > $O/c3r_$n.txt 2>&1 < /dev/null
  summ $O/c3r_$n counter_collection counters k_walk_tables >> $O/c3r_counters.txt
done
cat $O/c3r_counters.txt
cd $R; ( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err < /dev/null; tail -4 $O/bench.err; python - <<'PY'
import json
s=open('gpurun_out/s18/bench.json').read(); j=json.loads(s[s.index('{"metric'):].splitlines()[0])
print(j['value'], j['end_to_end'])
for c in j['configs']: print(c['name'], c.get('value'), c.get('kernel_ms'), c.get('setup_s'), c.get('strategy_steps'), c.get('error'))
PY
