cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s57; mkdir -p $O
timeout 1800 python -m pytest tests/ -q -m gpu -x > $O/pytest.txt 2>&1 < /dev/null; tail -5 $O/pytest.txt | cut -c1-300
SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/c3.txt 2>&1 < /dev/null; grep -E "^iter|edge tables\]" $O/c3.txt | cut -c1-200
timeout 900 python tools/one_walk.py 24w 0.25 1 reference 3 > $O/q1.txt 2>&1 < /dev/null; grep -E "^iter" $O/q1.txt | cut -c1-120
SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5.txt 2>&1 < /dev/null; grep -E "^iter|edge tables\]" $O/c5.txt | cut -c1-200
