cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s42; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed|^FAILED|Error" $O/pytest.txt | tail -5
timeout 600 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/r_24w.txt 2>&1 < /dev/null; grep -E "^iter" $O/r_24w.txt
timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/r_c5.txt 2>&1 < /dev/null; grep -E "^iter" $O/r_c5.txt
