cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s52; mkdir -p $O
timeout 600 python -m pytest tests/test_sparse_ids.py -q -m gpu -x > $O/sparse.txt 2>&1 < /dev/null; tail -30 $O/sparse.txt | cut -c1-300
timeout 1800 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -8
