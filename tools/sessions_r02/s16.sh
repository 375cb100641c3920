set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s16; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -k "scale or rmat or edge_tables or binned or giant or karate or fuzz or weighted or directed" > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt; grep -E "^FAILED|Error" $O/pytest.txt | head
SRW_TIMING=1 timeout 600 python tools/one_walk.py 24w 0.25 4 reference 2 > $O/c3.txt 2>&1 < /dev/null; grep -E "^iter|edge tables" $O/c3.txt
SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5.txt 2>&1 < /dev/null; grep -E "^iter|edge tables" $O/c5.txt
