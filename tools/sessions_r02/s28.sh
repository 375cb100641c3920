set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s28; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "shard or cluster or q1" > $O/pytest.txt 2>&1 < /dev/null; tail -5 $O/pytest.txt
timeout 900 python tools/cluster_timing.py 24 1,2,8 > $O/timing.txt 2>&1 < /dev/null; cat $O/timing.txt | tail -12
SRW_SHARD_PROFILE=1 timeout 600 python tools/cluster_timing.py 24 1 > $O/prof.txt 2>&1 < /dev/null; grep "shard profile" $O/prof.txt | tail -3
SRW_SHARD_NO_LINKS=1 timeout 600 python tools/cluster_timing.py 24 1 > $O/nolinks.txt 2>&1 < /dev/null; tail -3 $O/nolinks.txt
