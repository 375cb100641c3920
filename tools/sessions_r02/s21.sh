set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s21; mkdir -p $O
timeout 300 python tools/dbg_q1.py > $O/dbg.txt 2>&1 < /dev/null; cat $O/dbg.txt | grep -v amdgpu
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -k "q1 or scale or weighted" > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED" $O/pytest.txt | head
timeout 600 python tools/one_walk.py 24w 0.25 1 reference 2 > $O/q1_24w.txt 2>&1 < /dev/null; grep "^iter 1" $O/q1_24w.txt
timeout 600 python tools/one_walk.py 26 0.5 1 reference 2 > $O/q1_26.txt 2>&1 < /dev/null; grep "^iter 1" $O/q1_26.txt
