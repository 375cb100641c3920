cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s86; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_properties.py tests/test_sparse_ids.py -q -m gpu > $O/parity.txt 2>&1 < /dev/null; echo "parity: $(grep -E 'passed|failed' $O/parity.txt | tail -1)"; grep -E "^FAILED" $O/parity.txt | head -5
SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5.txt 2>&1 < /dev/null; grep -E "^iter|edge tables\]|edge hash vs" $O/c5.txt | cut -c1-220
SRW_NO_IDS32=1 SRW_TIMING=1 timeout 900 python tools/one_walk.py 20 0.25 4 reference 3 > $O/r20_no.txt 2>&1 < /dev/null; grep -E "^iter [12]" $O/r20_no.txt | cut -c1-120
SRW_TIMING=1 timeout 900 python tools/one_walk.py 20 0.25 4 reference 3 > $O/r20.txt 2>&1 < /dev/null; grep -E "^iter [12]" $O/r20.txt | cut -c1-120
