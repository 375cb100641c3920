set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s19; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED" $O/pytest.txt | head
for pq in "0.5 1" "4 1"; do
  timeout 300 python tools/one_walk.py 20 $pq reference 3 > $O/q1_20_${pq// /_}.txt 2>&1 < /dev/null; grep "^iter" $O/q1_20_${pq// /_}.txt
  SRW_NO_Q1_KERNEL=1 timeout 300 python tools/one_walk.py 20 $pq reference 2 > $O/noq1_20_${pq// /_}.txt 2>&1 < /dev/null; grep "^iter 1" $O/noq1_20_${pq// /_}.txt
done
timeout 600 python tools/one_walk.py 24w 0.25 1 reference 3 > $O/q1_24w.txt 2>&1 < /dev/null; grep "^iter" $O/q1_24w.txt
timeout 600 python tools/one_walk.py 26 0.5 1 reference 3 > $O/q1_26.txt 2>&1 < /dev/null; grep "^iter" $O/q1_26.txt
