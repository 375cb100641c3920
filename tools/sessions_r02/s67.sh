cd $GRAFT_REPO_ROOT
ulimit -c 0
O=gpurun_out/s67; mkdir -p $O
timeout 1200 python -m pytest tests/test_bench_contract.py -q -m gpu -x > $O/bc.txt 2>&1 < /dev/null; tail -25 $O/bc.txt | cut -c1-300
