cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s39; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed|^FAILED|Error" $O/pytest.txt | tail -5
export SRW_DEBUG_HANDOVER=1
timeout 600 python tools/one_walk.py 24w 0.25 1 reference 3 > $O/q1_24w.txt 2>&1 < /dev/null; grep -E "^iter" $O/q1_24w.txt
timeout 600 python tools/one_walk.py 26 0.5 1 reference 3 > $O/q1_26.txt 2>&1 < /dev/null; grep -E "^iter" $O/q1_26.txt
timeout 600 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/r_24w.txt 2>&1 < /dev/null; grep -E "^iter" $O/r_24w.txt
