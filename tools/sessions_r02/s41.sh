cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s41; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "q1 or scale or weighted" > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed|^FAILED|Error" $O/pytest.txt | tail -5
timeout 600 python tools/one_walk.py 24w 0.25 1 reference 3 > $O/q1_24w.txt 2>&1 < /dev/null; grep -E "^iter" $O/q1_24w.txt
timeout 600 python tools/one_walk.py 26 0.5 1 reference 3 > $O/q1_26.txt 2>&1 < /dev/null; grep -E "^iter" $O/q1_26.txt
