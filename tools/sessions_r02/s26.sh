set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s26; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -k "q1 or scale or weighted" > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED" $O/pytest.txt | head
timeout 600 python tools/one_walk.py 24w 0.25 1 reference 3 > $O/q1_24w.txt 2>&1 < /dev/null; grep "^iter" $O/q1_24w.txt
timeout 600 python tools/one_walk.py 26 0.5 1 reference 3 > $O/q1_26.txt 2>&1 < /dev/null; grep "^iter" $O/q1_26.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/tools/one_walk.py 24w 0.25 1 reference 3 > /dev/null 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-160
