set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s15; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED" $O/pytest.txt | head
SRW_TIMING=1 timeout 400 python tools/cluster_timing.py 24 1,2,8 > $O/cluster.txt 2>&1 < /dev/null; grep -E "cluster world|overflow|replicated" $O/cluster.txt
timeout 600 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/c3.txt 2>&1 < /dev/null; grep "^iter" $O/c3.txt
timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5.txt 2>&1 < /dev/null; grep "^iter" $O/c5.txt
timeout 300 python tools/one_walk.py 20 0.25 4 reference 3 > $O/c20.txt 2>&1 < /dev/null; grep "^iter" $O/c20.txt
