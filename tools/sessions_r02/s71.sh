cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s71; mkdir -p $O
SRW_EB_DROP_EHASH=1 timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_properties.py tests/test_sparse_ids.py -q -m gpu > $O/drop.txt 2>&1 < /dev/null; echo "SRW_EB_DROP_EHASH=1: $(tail -1 $O/drop.txt)"; grep -E "^FAILED" $O/drop.txt | head
timeout 1800 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; echo "default: $(tail -1 $O/pytest.txt)"; grep -E "^FAILED" $O/pytest.txt | head
