set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s14; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cluster or sharded" > $O/pytest.txt 2>&1 < /dev/null; grep -E "passed|failed" $O/pytest.txt
SRW_TIMING=1 timeout 400 python tools/cluster_timing.py 24 1,2,8 > $O/cluster.txt 2>&1 < /dev/null; grep -E "cluster world|overflow|replicated" $O/cluster.txt
