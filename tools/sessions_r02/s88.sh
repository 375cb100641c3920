cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s88; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_properties.py tests/test_sparse_ids.py -q -m gpu > $O/parity.txt 2>&1 < /dev/null; echo "parity: $(grep -E 'passed|failed' $O/parity.txt | tail -1)"; grep -E "^FAILED" $O/parity.txt | head -5
SRW_EB_DROP_EHASH=1 timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -k "biased or directed or binned or edge_table or rmat_vs_oracle or giant" > $O/parity_drop.txt 2>&1 < /dev/null; echo "parity, hash dropped: $(grep -E 'passed|failed' $O/parity_drop.txt | tail -1)"
SRW_TIMING=1 timeout 900 python tools/one_walk.py 26d 4 0.5 reference 2 27 > $O/c5.txt 2>&1 < /dev/null; grep -E "^iter|row filters|edge hash vs" $O/c5.txt | cut -c1-220
SRW_TIMING=1 timeout 900 python tools/one_walk.py 24w 0.25 4 reference 3 > $O/c3.txt 2>&1 < /dev/null; grep -E "^iter [12]|row filters" $O/c3.txt | cut -c1-200
